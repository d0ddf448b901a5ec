"""Dev check: the Winograd kernels (option winograd=1: conv_wino.hip, 2: conv_wino128.hip) against the direct kernels (0) on small frame counts / several clips."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa
from nhans_amd import engine
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "case_exp2.npz")))
eng = engine.Engine("denoiser", precision="f16x3")
eng.set_option("profile", 1)
lm_all = torch.from_numpy(g["logmag"]).cuda()
rng = np.random.default_rng(0)
for lens, scale in (([40], 0.0), ([40], -1.0), ([40], 0.05), ([40], 0.5), ([308], -1.0), ([308], 0.5), ([3, 1, 5], 0.1)):
    n = sum(lens)
    lm = lm_all[:n].contiguous()
    off = [0] + list(np.cumsum(lens))
    if scale < 0:
        ea = torch.from_numpy(np.repeat(g["emb_a"][None], len(lens), 0)).cuda()
        eb = torch.from_numpy(np.repeat(g["emb_b"][None], len(lens), 0)).cuda()
    else:
        ea = torch.from_numpy(rng.standard_normal((len(lens), 512)).astype(np.float32) * scale).cuda()
        eb = torch.from_numpy(rng.standard_normal((len(lens), 512)).astype(np.float32) * scale).cuda()
    out = {}
    for w in (0, 1, 2):
        eng.set_option("winograd", w)
        out[w] = eng.mask_net(lm, off, ea, eb)[0].cpu().numpy()
        st = eng.take_status()
        prof_names = sorted(k for k in eng.profile() if "wino" in k) if w else []
        print("   winograd", w, prof_names, "status", st, "max |logit| %.3g" % np.abs(out[w]).max(), "emb |max| %.3g" % float(ea.abs().max()))
    for w in (1, 2):
        d = np.abs(out[0] - out[w]).max(axis=1)
        print(lens, scale, "max |wino%d - direct| = %.3e" % (w, d.max()), "per frame:", np.array2string(d[:12], precision=1))
