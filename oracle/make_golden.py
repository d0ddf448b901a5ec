"""Generates the committed golden vectors under tests/golden/ from the float64 oracle.

Run in the build container (needs /root/reference only for the shipped example wavs and .index
files, which are data):   python oracle/make_golden.py
The GPU box never runs this; it reads the .npz/.json/.wav fixtures.

Weights are the seeded synthetic recipe (nhans_amd.weights.synthetic_weights, seed 7) because the
reference's trained weights are git-LFS pointers; inputs are the reference's exp2_noisy.wav and
seeded synthetic material (nhans_amd.synth).
"""
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import spec, synth, tfbundle, weights  # noqa: E402
from oracle import nhans_oracle as O  # noqa: E402

REF = "/root/reference"
SN = REF + "/N_HANS___Selective_Noise"
SS = REF + "/N_HANS___Source_Separation"
OUT = os.path.join(ROOT, "tests", "golden")
SUB = 97          # block activations are stored subsampled: flat[::SUB]


def f32(a):
    return np.asarray(a, dtype=np.float32)


def run_case(W, kind, mix, ca, cb, frames, tap_frames):
    """Oracle stages on normalised inputs; network only on `frames` (None = all)."""
    t0 = time.time()
    res = O.enhance(mix, ca, cb, W, kind, batch=8, frames=frames)
    out = dict(logmag=f32(res["logmag"]), phase=f32(res["phase"]), emb_a=f32(res["emb_a"]), emb_b=f32(res["emb_b"]))
    sel = np.arange(res["logmag"].shape[0]) if frames is None else np.asarray(frames)
    out["frames"] = sel.astype(np.int32)
    out["logits"] = f32(res["logits"][sel])
    if frames is None:
        out["denoised_wav"] = f32(res["denoised_wav"])
        out["mixed_wav"] = f32(res["mixed_wav"])
    if tap_frames:
        win = O.strided_crop(res["logmag"], O.MIX_WIN)
        taps = {}
        n = len(tap_frames)
        O.mask_net(win[tap_frames], np.repeat(res["emb_a"][None], n, 0), np.repeat(res["emb_b"][None], n, 0), W, kind, taps)
        out["tap_frames"] = np.asarray(tap_frames, dtype=np.int32)
        for i, (name, _) in enumerate(O.STACK):
            out["block%d" % i] = f32(taps[name].reshape(-1)[::SUB])
        out["block8"] = f32(taps["last_conv"].reshape(-1)[::SUB])
    print("  case done in %.1f s" % (time.time() - t0), flush=True)
    return out


DEMO_MIXED = REF + "/DEMO_N-HANS/denoising/example2/82057_1_10_652-129742-0017_Silent7_0O-gZoirpRA_30.000_5_8_mixed.wav"


def copy_data_fixtures():
    """Data files of the reference tree that the tests read (inputs / format samples, no source)."""
    for src, dst in ((SN + "/audio_examples/exp2_noisy.wav", "exp2_noisy.wav"),
                     # the README's denoising example (README.md:51): its --neg recording and the output
                     # the reference produced with its trained weights -- the end-to-end golden of
                     # tests/test_real_weights.py, usable once the LFS weight blob is supplied
                     (SN + "/audio_examples/exp2_noise.wav", "exp2_noise.wav"),
                     (SN + "/audio_examples/exp2_denoised.wav", "exp2_denoised.wav"),
                     (SN + "/trained_model/81448_0-1000000.index", "denoiser.index"),
                     (SS + "/trained_model/81457_2-545000.index", "separator.index"),
                     # a TensorFlow-written waveform: `*_mixed.wav` of the demo material is
                     # inverse_stft(stft(x)) dumped by SN/main.py:296-306 (22,480 float32 samples)
                     (DEMO_MIXED, "demo_tf_istft_mixed.wav")):
        shutil.copyfile(src, os.path.join(OUT, dst))
        os.chmod(os.path.join(OUT, dst), 0o644)


DEMO = REF + "/DEMO_N-HANS/"
# TensorFlow-written demo material (dumped by SN/main.py:296-306 / SS/main.py): first 8,400 samples (52 frames) of
# `_mixed.wav` of six more examples of both models, and two (mixed, target, negNoise) triples of the denoiser
DEMO_MIXED_MORE = [
    "selective_noise_suppression/example5/82076_1_10_121-121726-0005_Imx5o81QWk0_120.000_3za2WvNjiBk_0.000_8_5_mixed.wav",
    "Selective_Noise_Suppression_samples/snsExample3_4446-2275-0032_-ocADGlyaHc_30.000_6rNmM0Mt3zI_30.000_-3_0_mixed.wav",
    "denoising/example3/82057_1_10_1462-170142-0022_Silent17_2ATRc7EonvI_30.000_5_8_mixed.wav",
    "source_separation/example9/82024_1_0_00166_00221_0_mixed.wav",
    "source_separation/example12/82024_1_0_00298_00095_0_mixed.wav",
    "source_separation/example4/82024_1_0_00083_00062_5_mixed.wav",
]
DEMO_TRIPLES = [
    "selective_noise_suppression/example5/82076_1_10_121-121726-0005_Imx5o81QWk0_120.000_3za2WvNjiBk_0.000_8_5_",
    "Selective_Noise_Suppression_samples/snsExample2_4446-2273-0015_-NRx0SBMjo0_24.000_2yrL0F_0UGc_0.000_5_3_",
]


def demo_tf_fixtures(nsamp=8400):
    """-> tests/golden/demo_tf_segments.npz: float32 segments of reference-shipped, TensorFlow-written wavs (data
    only).  `mixed_<i>`: a `_mixed.wav` segment (STFT -> iSTFT fixed point in its interior); `triple_<i>_{mixed,
    target,negNoise}`: the same samples of the three dumps of one utterance, which pin the reference's mixing
    rule (SN/apply.py:96-102): target and noise are divided by the peak of the ALREADY normalised mixture (~1),
    so  peak_raw * mixed == target + negNoise  with peak_raw != 1."""
    from scipy.io import wavfile
    out = {}
    for i, rel in enumerate(DEMO_MIXED_MORE):
        r, y = wavfile.read(DEMO + rel)
        assert r == 16000 and y.dtype == np.float32 and len(y) >= nsamp
        out["mixed_%d" % i] = y[:nsamp].copy()
        out["mixed_%d_source" % i] = np.array(rel)
    for i, base in enumerate(DEMO_TRIPLES):
        for k in ("mixed", "target", "negNoise"):
            r, y = wavfile.read(DEMO + base + k + ".wav")
            assert r == 16000 and y.dtype == np.float32
            out["triple_%d_%s" % (i, k)] = y[:nsamp].copy()
        out["triple_%d_source" % i] = np.array(base)
    np.savez_compressed(os.path.join(OUT, "demo_tf_segments.npz"), **out)
    print("demo_tf_segments.npz:", os.path.getsize(os.path.join(OUT, "demo_tf_segments.npz")), "bytes")


def new_cases_r2():
    """Cases added in round 2 (kept separate so they can be regenerated without the long exp2 run)."""
    Wd = weights.synthetic_weights("denoiser", 7)
    Ws = weights.synthetic_weights("separator", 7)
    print("case separator10s (separator, 10 s clip, 12 frames)", flush=True)
    mix = O.trim_to_frames(O.normalise(synth.mixture(5, 10.0)))
    fr = [0, 1, 16, 17, 18, 250, 499, 700, 979, 980, 996, 997]
    res = run_case(Ws, "separator", mix, O.normalise(synth.speaker_context(5, low=True)),
                   O.normalise(synth.speaker_context(5, low=False)), fr, None)
    np.savez_compressed(OUT + "/case_separator10s.npz", **res)

    print("case postproc (denoiser, 0.6 s clip, all frames: removed / snr_est / compensated)", flush=True)
    mix = O.trim_to_frames(O.normalise(synth.mixture(40, 0.6)))
    ca, cb = O.normalise(synth.silent()), O.normalise(synth.noise_context(40))
    out = {}
    for tag, kw in (("fixed", dict(compensate=0.3, ac=False)), ("ac", dict(compensate=0.0, ac=True))):
        r = O.enhance(mix, ca, cb, Wd, "denoiser", batch=8, **kw)
        for k in ("denoised_wav", "mixed_wav", "removed_wav", "compensated_wav"):
            out["%s_%s" % (k, tag)] = f32(r[k])
        out["snr_est_%s" % tag] = np.float64(r["snr_est"])
    np.savez_compressed(OUT + "/case_postproc.npz", **out)


def full_10s_cases():
    """The metric's own clip (BASELINE.json `metric`: RMS vs reference on 10 s / 16 kHz clips; SN/apply.py:189-204,453-458):
    ALL 998 frames of the 10 s denoiser clip of case_synth10s and the 10 s separator clip of case_separator10s through
    the float64 path, logits and reconstructed waveform -> tests/golden/case_full10s_<kind>.npz.  Computed with the
    torch float64 restatement (oracle/torch_ref.py: minutes instead of an hour) and pinned to the numpy oracle on the
    12 frames the small cases already hold."""
    import torch
    from oracle.torch_ref import TorchRef
    torch.set_num_threads(os.cpu_count() or 1)
    for kind, seed, small in (("denoiser", 0, "case_synth10s"), ("separator", 5, "case_separator10s")):
        t0 = time.time()
        W = weights.synthetic_weights(kind, 7)
        mix = O.trim_to_frames(O.normalise(synth.mixture(seed, 10.0)))
        if kind == "denoiser":
            ca, cb = O.normalise(synth.silent()), O.normalise(synth.noise_context(seed))
        else:
            ca, cb = O.normalise(synth.speaker_context(seed, low=True)), O.normalise(synth.speaker_context(seed, low=False))
        r = TorchRef(W, kind, torch.float64).enhance(mix, ca, cb, faithful=False, mb=50)
        logits, wav = r["logits"].numpy(), r["denoised_wav"].numpy()
        g = dict(np.load(os.path.join(OUT, small + ".npz")))
        fr = g["frames"].astype(np.int64)
        pin = float(np.abs(logits[fr] - g["logits"]).max())
        pin_lm = float(np.abs(r["logmag"].numpy() - g["logmag"]).max())
        print("  %s: %d frames, %d samples in %.0f s; |torch f64 - numpy oracle| on the %d frames of %s: logits %.2e, logmag %.2e"
              % (kind, logits.shape[0], len(wav), time.time() - t0, len(fr), small, pin, pin_lm), flush=True)
        assert pin < 1e-6 and pin_lm < 1e-6
        np.savez_compressed(os.path.join(OUT, "case_full10s_%s.npz" % kind), logits=f32(logits), denoised_wav=f32(wav),
                            features_case=np.array(small))


def main():
    os.makedirs(OUT, exist_ok=True)
    # ---- data fixtures from the reference tree
    copy_data_fixtures()
    if len(sys.argv) > 1 and sys.argv[1] == "r2":
        new_cases_r2()
        return
    geo = {"cases": [], "examples": {}}
    for n in (49600, 63520, 160000, 400, 559, 560, 399, 32240):
        kept, t = spec.frames_for_samples(n)
        geo["cases"].append({"n": n, "kept": kept, "frames": t, "out_len": (t - 1) * 160 + 400 if t else 0})
    from scipy.io import wavfile
    for name in ("exp1", "exp2"):
        _, a = wavfile.read("%s/audio_examples/%s_noisy.wav" % (SN, name))
        r, b = wavfile.read("%s/audio_examples/%s_denoised.wav" % (SN, name))
        geo["examples"][name] = {"in_len": int(len(a)), "out_len": int(len(b)), "out_dtype": str(b.dtype), "rate": int(r)}
    ws = O.istft_window()
    geo["wsyn"] = {"0": float(ws[0]), "1": float(ws[1]), "200": float(ws[200]), "399": float(ws[399])}
    json.dump(geo, open(OUT + "/geometry.json", "w"), indent=1)
    inv = {}
    for kind, prefix in (("denoiser", SN + "/trained_model/81448_0-1000000"), ("separator", SS + "/trained_model/81457_2-545000")):
        ent = tfbundle.read_index(prefix + ".index")
        inv[kind] = [[k, list(v["shape"]), v["dtype"], v["offset"], v["size"]] for k, v in ent.items()]
    json.dump(inv, open(OUT + "/checkpoint_index.json", "w"))

    Wd = weights.synthetic_weights("denoiser", 7)
    Ws = weights.synthetic_weights("separator", 7)

    print("case exp2 (denoiser, all 308 frames)", flush=True)
    mix = O.trim_to_frames(O.normalise(O.read_wav(OUT + "/exp2_noisy.wav")))
    res = run_case(Wd, "denoiser", mix, O.normalise(synth.silent()), O.normalise(synth.noise_context(0)), None, [0, 154])
    np.savez_compressed(OUT + "/case_exp2.npz", **res)

    print("case synth10s (denoiser, 10 s clip, 12 frames)", flush=True)
    mix = O.trim_to_frames(O.normalise(synth.mixture(0, 10.0)))
    fr = [0, 1, 16, 17, 18, 250, 499, 700, 979, 980, 996, 997]
    res = run_case(Wd, "denoiser", mix, O.normalise(synth.silent()), O.normalise(synth.noise_context(0)), fr, None)
    np.savez_compressed(OUT + "/case_synth10s.npz", **res)

    print("case separator (2 s, 8 frames)", flush=True)
    mix = O.trim_to_frames(O.normalise(synth.mixture(3, 2.0)))
    res = run_case(Ws, "separator", mix, O.normalise(synth.speaker_context(3, low=False)),
                   O.normalise(synth.speaker_context(3, low=True)), [0, 5, 40, 99, 100, 150, 196, 197], [99])
    np.savez_compressed(OUT + "/case_separator.npz", **res)

    print("case ragged (3 short clips, all frames)", flush=True)
    lens = [400, 560, 16000 + 80]          # 1, 2 and 99 frames
    allres = {}
    for i, n in enumerate(lens):
        mix = O.trim_to_frames(O.normalise(synth.mixture(10 + i, n / 16000.0)))
        r = run_case(Wd, "denoiser", mix, O.normalise(synth.silent()), O.normalise(synth.noise_context(10 + i)), None, None)
        for k in ("logmag", "logits", "denoised_wav", "emb_b"):
            allres["%s_%d" % (k, i)] = r[k]
    allres["lens"] = np.asarray(lens, dtype=np.int64)
    np.savez_compressed(OUT + "/case_ragged.npz", **allres)
    new_cases_r2()
    demo_tf_fixtures()
    full_10s_cases()
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "demo":      # only the TensorFlow-written demo segments (seconds)
        demo_tf_fixtures()
    elif len(sys.argv) > 1 and sys.argv[1] == "full10s":  # only the two full-length 10 s cases (minutes)
        full_10s_cases()
    else:
        main()
