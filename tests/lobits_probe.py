"""Development probe (under tests/ because it uses the oracle; not collected by pytest): how many mantissa bits do the
`lo` halves of the f16x3 arithmetic need?

VERDICT r05 item 1: the step is bound by the socket's power cap and operand bit activity is power
(tools/ubench/mfma_lobits.hip measures that side).  Two of the three products of every MAC carry a `lo` half
(x = hi + lo, a*b ~ ah*bh + ah*bl + al*bh).  This probe rounds `lo` (to nearest) to k mantissa bits -- activations and
weights separately -- in the CPU emulation of the conv kernels (tests/winograd_probe.py: direct and the shipped 1-D
F(5,4) Winograd form of the 4x4 convs, whose transformed input is re-split and so rounded as well) and reports
max |logit - float64| on the three goldens and two more weight recipes.  Arithmetic being emulated: tf.nn.conv2d in
float32 (SN/blocks.py:44); bar 1e-4 (BASELINE.json), gate for shipping a masked split 3e-5 on the goldens.

    python tests/lobits_probe.py [frames]          (CPU, several minutes)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nhans_amd  # noqa: E402,F401
import fp8_layer_probe as FP  # noqa: E402
import winograd_probe as WP  # noqa: E402


def round_bits(lo, k):
    """f16-valued f32 tensor rounded to nearest at k mantissa bits (k = 10: unchanged); subnormal halves keep their grid."""
    if k >= 10:
        return lo
    m, e = torch.frexp(lo)                          # lo = m * 2^e, |m| in [0.5, 1): k mantissa bits = k + 1 significant
    e = torch.clamp(e, min=-13)                     # f16 subnormals: fixed grid of 2^-24
    q = torch.exp2((e - (k + 1)).float())
    return torch.round(lo / q) * q


class ProbeLo(WP.Probe):
    def __init__(self, W, kind, plan, ka, kb):
        super().__init__(W, kind, plan, products=3)
        self.ka, self.kb = ka, kb

    def x3(self, a, b, fn):
        ah, al = WP.split(a)
        bh, bl = WP.split(b)
        al = torch.zeros_like(al) if self.ka < 0 else round_bits(al, self.ka)
        bl = torch.zeros_like(bl) if self.kb < 0 else round_bits(bl, self.kb)
        return fn(ah, bh) + fn(ah, bl) + fn(al, bh)


WINO = {(4, 4): (1, 5, (), WP.P7)}


def main():
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.set_num_threads(os.cpu_count() or 1)
    cs = FP.cases(nfr)
    print("max |logit - float64| over %d frames per case; bar 1e-4, gate for a masked split 3e-5 on the goldens" % nfr)
    print("%-9s %-4s %-4s " % ("form", "kA", "kB") + " ".join("%12s" % c[0][-12:] for c in cs))
    rows = [(10, 10)] + [(k, 10) for k in (8, 6, 4, 2)] + [(10, k) for k in (8, 6, 4, 2)] + [(k, k) for k in (8, 7, 6, 5, 4, 2, 0)] + [(-1, -1)]
    for form, plan in (("direct", {}), ("winograd", WINO)):
        for ka, kb in rows:
            row = []
            for case, kind, W, lm, ea, eb, frames, want in cs:
                ref = ProbeLo(W, kind, plan, ka, kb)
                win = ref.windows(torch.from_numpy(lm))[frames]
                a = torch.from_numpy(ea)[None].expand(len(frames), -1)
                b = torch.from_numpy(eb)[None].expand(len(frames), -1)
                with torch.no_grad():
                    out, _ = ref.mask_net(win, a, b)
                row.append(float(np.abs(out.numpy() - want).max()))
            print("%-9s %-4d %-4d " % (form, ka, kb) + " ".join("%12.2e" % r for r in row), flush=True)


if __name__ == "__main__":
    main()
