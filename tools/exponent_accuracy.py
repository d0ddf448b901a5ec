"""Logit error of the f16x3 path against the float64 golden (identical features: tests/golden/case_exp2.npz) as a
function of the activation exponents: calibrated (nhans_create), all zero (round 2's storage), calibrated +-k.
    python tools/exponent_accuracy.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import engine, hip, weights as weights_mod  # noqa: E402


def main():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "case_exp2.npz")))
    eng = engine.Engine("denoiser", weights_mod.synthetic_weights("denoiser", 7), precision="f16x3")
    lm = torch.from_numpy(g["logmag"]).cuda()
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    n = lm.shape[0]
    base = eng.activation_exponents()
    rows = {"calibrated_exponents": base, "calibration_amax": [float("%.4g" % a) for a in eng.activation_amax()]}
    for wino in (1, 0):
        eng.set_option("winograd", wino)
        for name, exps in [("zero", [0] * hip.NUM_ACTIVATIONS)] + [("calibrated%+d" % k, [e + k for e in base]) for k in (-6, -4, -2, 0, 2, 4)]:
            eng.set_activation_exponents(exps)
            lg, _ = eng.mask_net(lm, [0, n], ea, eb)
            st = eng.take_status()
            err = np.abs(lg.cpu().numpy().astype(np.float64) - g["logits"])
            rows["winograd=%d %s" % (wino, name)] = {"max_abs_err": float(err.max()), "rms_err": float(np.sqrt((err ** 2).mean())), "status": st}
    eng.set_precision("f32")
    lg, _ = eng.mask_net(lm, [0, n], ea, eb)
    err = np.abs(lg.cpu().numpy().astype(np.float64) - g["logits"])
    rows["f32 matrix-core path"] = {"max_abs_err": float(err.max()), "rms_err": float(np.sqrt((err ** 2).mean()))}
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
