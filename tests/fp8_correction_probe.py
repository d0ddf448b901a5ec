"""Development probe (under tests/ because it uses the oracle; not collected by pytest): could the two CORRECTION
products of the split-f16 arithmetic run on the 8-bit matrix path?

Today every MAC of the f16x3 mode is three f16 MFMA products with f32 accumulation: x_hi w_hi + x_hi w_lo + x_lo w_hi.
The step sits at the socket's power limit and 60 % of its energy is those MFMAs: the only large
lever left is fewer matrix-core cycles per MAC.  The two correction products are 2^-11 of the main one, so they need
~9 bits of relative accuracy in total, not 22 -- and gfx950's block-scaled 8-bit MFMA (v_mfma_scale_f32_32x32x64_f8f6f4,
one E8M0 scale per 32 elements along K) runs at twice the f16 rate: 1 + 1/2 + 1/2 = 2 instead of 3 matrix-core units
per MAC if x_hi, x_lo, w_hi, w_lo of the correction products may be rounded to e4m3 (3 mantissa bits).

This restates the network on the CPU with that arithmetic (the Probe of tests/winograd_probe.py with another product
rule) and measures the end-to-end logit error against the committed float64 goldens, Winograd form as in the product
(F(5,4) along W for the 4x4 convs).  Scales are ONE power of two per pixel (all channels) for activations and per
output column for weights -- coarser than the hardware's one per 32 elements, i.e. pessimistic.

    python tests/fp8_correction_probe.py [frames]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nhans_amd  # noqa: E402,F401
from nhans_amd import weights  # noqa: E402
import winograd_probe as WP  # noqa: E402

F8 = {"e4m3": (torch.float8_e4m3fn, 448.0), "e5m2": (torch.float8_e5m2, 57344.0)}


def q8(t, dim, fmt):
    """round to an 8-bit float with one power-of-two scale per slice along `dim` (kept in f32)."""
    dt, mx = F8[fmt]
    amax = t.abs().amax(dim=dim, keepdim=True).clamp_min(1e-30)
    s = torch.exp2(torch.floor(torch.log2(mx / amax)))
    return (t * s).to(dt).to(torch.float32) / s


class Probe8(WP.Probe):
    def __init__(self, W, kind, plan, mode):
        super().__init__(W, kind, plan, products=3)
        self.mode = mode

    def x3(self, a, b, fn):
        ah, al = WP.split(a)
        bh, bl = WP.split(b)
        out = fn(ah, bh)
        if self.mode == "f16x3":
            return out + fn(ah, bl) + fn(al, bh)
        if self.mode == "f16x1":
            return out
        fmt = self.mode
        # activations: a is [B, C, H, W] (direct) or [B, C, u, v, p, q] (Winograd V): channels are dim 1
        # weights: b is [O, I, kh, kw] (direct, scale per O) or [p, q, I, O] (Winograd U, scale per O)
        if b.dim() == 4 and a.dim() == 4:
            qb = lambda t: q8(t, (1, 2, 3), fmt)
        else:
            qb = lambda t: q8(t, (0, 1, 2), fmt)
        qa = lambda t: q8(t, 1, fmt)
        return out + fn(qa(ah), qb(bl)) + fn(qa(al), qb(bh))


def main():
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.set_num_threads(os.cpu_count() or 1)
    golden = os.path.join(ROOT, "tests", "golden")
    plan = {(4, 4): (1, 5, (), WP.P7)}
    print("%-44s %12s %12s %12s" % ("correction products", "exp2 logits", "10s logits", "sep logits"))
    for mode in ("f16x3", "e4m3", "e5m2", "f16x1"):
        row = []
        for case, kind in (("case_exp2", "denoiser"), ("case_synth10s", "denoiser"), ("case_separator10s", "separator")):
            g = dict(np.load(os.path.join(golden, case + ".npz")))
            W = weights.synthetic_weights(kind, 7)
            ref = Probe8(W, kind, plan, mode)
            lm = torch.from_numpy(g["logmag"])
            frames = g["frames"][:: max(1, len(g["frames"]) // nfr)][:nfr].astype(np.int64)
            pos = {int(f): i for i, f in enumerate(g["frames"])}
            win = ref.windows(lm)[frames]
            ea = torch.from_numpy(g["emb_a"])[None].expand(len(frames), -1)
            eb = torch.from_numpy(g["emb_b"])[None].expand(len(frames), -1)
            with torch.no_grad():
                out, _ = ref.mask_net(win, ea, eb)
            want = g["logits"][[pos[int(f)] for f in frames]]
            row.append(float(np.abs(out.numpy() - want).max()))
        name = {"f16x3": "f16 x f16 (today)", "e4m3": "e4m3 x e4m3, power-of-two scales",
                "e5m2": "e5m2 x e5m2, power-of-two scales", "f16x1": "none (x_hi w_hi only)"}[mode]
        print("%-44s %12.2e %12.2e %12.2e" % (name, row[0], row[1], row[2]), flush=True)


if __name__ == "__main__":
    main()
