"""Development probe (under tests/ because it uses the oracle; not collected by pytest): end-to-end logit error of
the HIP path in both arithmetic modes and of the float32 torch-CPU restatement, all against the float64 oracle, on the
random cases of test_gpu_fuzz.py -- separates what float32 itself costs from what the f16x3 split adds.
    python tests/accuracy_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nhans_amd  # noqa: E402,F401
from nhans_amd import engine, weights  # noqa: E402
from oracle import nhans_oracle as O  # noqa: E402
from oracle.torch_ref import TorchRef  # noqa: E402
from test_gpu_fuzz import _clip  # noqa: E402


def main():
    for kind, seed in (("denoiser", 11), ("denoiser", 12), ("separator", 13), ("separator", 14), ("separator", 15)):
        W = weights.synthetic_weights(kind, 7)
        eng = engine.Engine(kind, W)
        ref32 = TorchRef(W, kind, torch.float32)
        rng = np.random.default_rng(seed)
        clips = [_clip(rng, int(f), kind) for f in rng.integers(1, 6, size=3)]
        for i, (mix, ca, cb) in enumerate(clips):
            o = O.enhance(mix, ca, cb, W, kind)
            row = []
            for prec in ("f16x3", "f32"):
                eng.set_precision(prec)
                g = eng.enhance([mix], [ca], [cb], want_mixed=False, taps=True)
                row.append("HIP %s: logits %.2e emb %.2e" % (prec, np.abs(g["logits"] - o["logits"]).max(),
                                                             max(np.abs(g["emb"][0] - o["emb_a"]).max(), np.abs(g["emb"][1] - o["emb_b"]).max())))
            with torch.no_grad():
                t = ref32.enhance(mix, ca, cb, faithful=False)
            row.append("torch-CPU f32: logits %.2e" % np.abs(t["logits"].numpy() - o["logits"]).max())
            print(kind, seed, "clip", i, "frames", o["logits"].shape[0], "|", " | ".join(row))
        eng.close()


if __name__ == "__main__":
    main()
