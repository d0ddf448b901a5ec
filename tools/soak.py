"""Race screen: run the whole enhance path N times on the same inputs and demand bit-identical output
(the LDS-DMA pipelines order their data with counted vmcnt + raw barriers only; a hazard shows up as a
run-to-run difference long before it breaks a tolerance).   python tools/soak.py [runs] [clips]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import engine, hip, synth  # noqa: E402
from nhans_amd.apply import normalise, trim_to_frames  # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    clips = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for prec, opts in (("f16x3", {}), ("f16x3", {"winograd": 0}), ("f32", {})):
        eng = engine.Engine("denoiser", precision=prec)
        for k, v in opts.items():
            eng.set_option(k, v)
        mixes = [trim_to_frames(normalise(synth.mixture(i, 10.0 if i % 2 == 0 else 3.7))) for i in range(clips)]
        ca = [normalise(synth.silent()) for _ in range(clips)]
        cb = [normalise(synth.noise_context(i)) for i in range(clips)]
        ref = None
        for r in range(runs):
            out = eng.enhance(mixes, ca, cb, want_mixed=False, taps=True)
            cur = [out["logits"]] + list(out["denoised_wav"])
            if ref is None:
                ref = cur
            else:
                for a, b in zip(ref, cur):
                    if not np.array_equal(a, b):
                        print("MISMATCH at run", r, prec, float(np.abs(a - b).max()))
                        sys.exit(1)
        print(prec, opts or "", "ok:", runs, "runs x", clips, "clips bit-identical")
        eng.close()


if __name__ == "__main__":
    main()
