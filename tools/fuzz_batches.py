"""Randomised batch / chunk / knob invariance screen on the GPU: random ragged batches (1..8 clips of 1..998 frames),
random frames_per_chunk and contexts_per_chunk, arithmetic mode and conv_variant, and a random setting of every knob that is bit-identical
by contract (epilogue_wide, consumer_interleave), both models -- every clip's
logits and waveform must equal, bit for bit, the same clip run alone with the default knobs (same arithmetic mode
and conv_variant), and be finite.  Exercises the tile-boundary handling of every conv kernel at many M that no
fixed test hits.
    python tools/fuzz_batches.py [iterations] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import engine, hip, synth  # noqa: E402
from nhans_amd.apply import normalise, trim_to_frames  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    knobs = ("epilogue_wide", "consumer_interleave")     # bit-identical by contract
    defaults = {"epilogue_wide": 1, "consumer_interleave": 1}
    for kind in ("denoiser", "separator"):
        eng = engine.Engine(kind, precision="f16x3")
        variants = [-1, -1, 0, 1, 2]
        pool = []
        for i in range(14):
            secs = float(rng.choice([0.025, 0.035, 0.1, 0.33, 0.8, 1.7, 2.5, 5.0, 10.0], p=[.15, .1, .15, .15, .15, .1, .1, .05, .05]))
            mix = trim_to_frames(normalise(synth.mixture(500 + i, secs)))
            ca = normalise(synth.silent()) if kind == "denoiser" else normalise(synth.speaker_context(500 + i, low=True))
            cb = normalise(synth.noise_context(500 + i)) if kind == "denoiser" else normalise(synth.speaker_context(500 + i, low=False))
            pool.append((mix, ca, cb))
        alone = {}                                      # (precision, conv_variant, winograd, clip) -> results with default knobs

        def reference(prec, variant, wino, i):
            key = (prec, variant, wino, i)
            if key not in alone:
                eng.set_precision(prec)
                eng.set_option("conv_variant", variant)
                eng.set_option("winograd", wino)
                eng.set_option("frames_per_chunk", 3776)
                for k, v in defaults.items():
                    eng.set_option(k, v)
                eng.set_option("contexts_per_chunk", 64)
                r = eng.enhance([pool[i][0]], [pool[i][1]], [pool[i][2]], want_mixed=False, taps=True)
                alone[key] = (r["logits"].copy(), r["denoised_wav"][0].copy(), r["emb"].copy(), r["logmag"].copy())
            return alone[key]

        for it in range(iters):
            n = int(rng.integers(1, 9))
            ids = [int(x) for x in rng.integers(0, len(pool), n)]
            prec = "f16x3" if rng.random() < 0.8 else "f32"
            variant = int(rng.choice(variants))
            wino = int(rng.random() < 0.7)               # Winograd form of the 4x4 stack convs (another summation: part of the key)
            refs = [reference(prec, variant, wino, i) for i in ids]
            fpc = int(rng.choice([1, 7, 33, 100, 257, 1024, 3776]))
            cfg = {k: int(rng.integers(0, 3 if k == "consumer_interleave" else 2)) for k in knobs}
            cfg["contexts_per_chunk"] = int(rng.choice([1, 3, 5, 64]))      # tower passes with a remainder chunk
            eng.set_precision(prec)
            eng.set_option("conv_variant", variant)
            eng.set_option("winograd", wino)
            eng.set_option("frames_per_chunk", fpc)
            for k, v in cfg.items():
                eng.set_option(k, v)
            r = eng.enhance([pool[i][0] for i in ids], [pool[i][1] for i in ids], [pool[i][2] for i in ids],
                            want_mixed=False, taps=True)
            f0 = 0
            for k, i in enumerate(ids):
                lg, wav, emb, lm = refs[k]
                got_lg = r["logits"][f0:f0 + len(lg)]
                got_lm = r["logmag"][f0:f0 + len(lg)]
                got_emb = r["emb"][[k, n + k]]
                f0 += len(lg)
                ok = np.array_equal(got_lg, lg) and np.array_equal(r["denoised_wav"][k], wav) and np.isfinite(wav).all()
                if not ok:
                    bad += 1
                    fr = np.abs(got_lg - lg).max(axis=1)
                    print("MISMATCH", kind, "iter", it, "clip", i, "pos", k, "of", ids, prec, "variant", variant, "winograd", wino, "fpc", fpc, cfg,
                          "logits %.3g" % float(fr.max()), "frames differing", int((fr > 0).sum()), "of", len(fr),
                          "first", int(np.argmax(fr > 0)), "| emb %.3g" % float(np.abs(got_emb - emb).max()),
                          "| logmag %.3g" % float(np.abs(got_lm - lm).max()), "| frames before", f0 - len(lg))
        eng.close()
        print(kind, "done:", iters, "random batches")
    print("fuzz: %d mismatches" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
