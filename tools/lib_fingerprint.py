"""Dev tool: a fingerprint of what the library under $NHANS_LIB computes -- sha256 of the logits of the golden exp2 features
(f16x3 with the Winograd form on and off, f32) and of a three-clip ragged end-to-end batch.  Two builds that are meant to differ in
speed only (tools/ab_variant_libs.sh) must print the same lines.
    NHANS_LIB=build_ab/variants/libnhans_X.so python tools/lib_fingerprint.py"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import apply, engine, synth  # noqa: E402


def main():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "case_exp2.npz")))
    eng = engine.Engine("denoiser", precision="f16x3")
    lm = torch.from_numpy(g["logmag"]).cuda()
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    out = []
    for prec, wino in (("f16x3", 1), ("f16x3", 0), ("f32", 1)):
        eng.set_precision(prec)
        eng.set_option("winograd", wino)
        lg = eng.mask_net(lm, [0, 308], ea, eb)[0].cpu().numpy()
        out.append("%s winograd=%d logits %s (vs golden %.2e)" % (prec, wino, hashlib.sha256(lg.tobytes()).hexdigest()[:16], np.abs(lg - g["logits"]).max()))
    eng.set_precision("f16x3")
    eng.set_option("winograd", 1)
    mixes = [apply.trim_to_frames(apply.normalise(synth.mixture(900 + i, s))) for i, s in enumerate((1.3, 0.4, 10.0))]
    ca = [apply.normalise(synth.silent()) for _ in mixes]
    cb = [apply.normalise(synth.noise_context(900 + i)) for i in range(3)]
    res = eng.enhance(mixes, ca, cb, want_mixed=True, taps=True)
    h = hashlib.sha256()
    for k in ("denoised_wav", "mixed_wav"):
        for w in res[k]:
            h.update(w.tobytes())
    h.update(res["logits"].tobytes())
    out.append("3-clip batch end to end %s" % h.hexdigest()[:16])
    print(os.environ.get("NHANS_LIB", "default library"))
    for l in out:
        print("  " + l)


if __name__ == "__main__":
    main()
