"""Development probe (under tests/ because it uses the oracle; not collected by pytest): the 8-bit correction products of
tests/fp8_correction_probe.py decided PER LAYER ("decide with the oracle, per layer").

Round 4 rejected e4m3 correction products (x_hi w_lo + x_lo w_hi on the block-scaled 8-bit matrix path, 2 instead of 3
matrix-core units per MAC) applied to ALL 17 stack convs: 1.4-1.6e-4 on the logits against a 1e-4 bar.  The direct
kernels of the stride-1 3x3 convs are matrix-bound (MFMA-busy 0.70 at the socket's power limit), so a subset that
holds the bar would still be worth a third of those launches' matrix work.  This probe applies the 8-bit rule to
subsets only -- everything else keeps f16 corrections -- with two scale granularities:
  coarse   one power of two per pixel (activations) / per output column (weights): what round 4 probed
  block32  one power of two per 32 consecutive channels of a pixel / of a weight column's taps: the hardware's E8M0 scale
           per 32 K-elements of v_mfma_scale_f32_32x32x64_f8f6f4
and reports max |logit - float64 golden| on the three golden cases and on two more weight recipes (heavy tails,
trained-BN statistics: tests/weight_recipes.py; their reference is the float64 oracle run here).

    python tests/fp8_layer_probe.py [frames]          (CPU, a few minutes)
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
from oracle.torch_ref import TorchRef  # noqa: E402
import weight_recipes  # noqa: E402
import winograd_probe as WP  # noqa: E402

E4M3_MAX = 448.0


def q8(t, dims, block_dim=None):
    """e4m3 with power-of-two scales: one per slice over `dims`, or -- block_dim given -- one per 32 consecutive
    elements along block_dim (and per element of every other dimension not in dims)."""
    if block_dim is not None and t.shape[block_dim] % 32 == 0:
        shp = list(t.shape)
        shp[block_dim:block_dim + 1] = [shp[block_dim] // 32, 32]
        tb = t.reshape(shp)
        red = [block_dim + 1] + [d + (1 if d > block_dim else 0) for d in dims if d != block_dim]
        amax = tb.abs().amax(dim=red, keepdim=True).clamp_min(1e-30)
        s = torch.exp2(torch.floor(torch.log2(E4M3_MAX / amax)))
        return ((tb * s).to(torch.float8_e4m3fn).to(torch.float32) / s).reshape(t.shape)
    amax = t.abs().amax(dim=dims, keepdim=True).clamp_min(1e-30)
    s = torch.exp2(torch.floor(torch.log2(E4M3_MAX / amax)))
    return (t * s).to(torch.float8_e4m3fn).to(torch.float32) / s


class ProbeLayers(WP.Probe):
    """Direct emulation of every stack conv (no Winograd: the question is about the direct kernels' layers); the convs
    whose scope is in `layers` take e4m3 correction products."""

    def __init__(self, W, kind, layers, block32):
        super().__init__(W, kind, {}, products=3)
        self.layers, self.block32, self.scope = set(layers), block32, None

    def conv(self, x, scope, stride, padding, bias):
        self.scope = scope
        return super().conv(x, scope, stride, padding, bias)

    def x3(self, a, b, fn):
        ah, al = WP.split(a)
        bh, bl = WP.split(b)
        out = fn(ah, bh)
        if self.scope not in self.layers:
            return out + fn(ah, bl) + fn(al, bh)
        # a: [B, C, H, W] activations (K runs over C and the taps: one scale per pixel, or per 32 channels of a pixel);
        # b: [O, I, kh, kw] weights (one scale per output column, or per 32 input channels of a tap of a column)
        if self.block32:
            qa = lambda t: q8(t, (1,), 1)
            qb = lambda t: q8(t, (1,), 1)
        else:
            qa = lambda t: q8(t, (1,))
            qb = lambda t: q8(t, (1, 2, 3))
        return out + fn(qa(ah), qb(bl)) + fn(qa(al), qb(bh))


S3 = ["resblock3_1_conv2", "resblock3_2_conv1", "resblock3_2_conv2"]
S4 = ["resblock4_1_conv2", "resblock4_2_conv1", "resblock4_2_conv2"]
SUBSETS = [
    ("none (f16 corrections everywhere)", []),
    ("the six 3x3 stride-1 convs", S3 + S4),
    ("resblock3's three", S3),
    ("resblock4's three", S4),
    ("resblock4_2_conv2 alone", S4[2:]),
    ("resblock3_2_conv1 alone", S3[1:2]),
    ("all 16 stack convs with Cin >= 64", None),
]


def cases(nfr):
    golden = os.path.join(ROOT, "tests", "golden")
    out = []
    for case, kind in (("case_exp2", "denoiser"), ("case_synth10s", "denoiser"), ("case_separator10s", "separator")):
        g = dict(np.load(os.path.join(golden, case + ".npz")))
        W = weights.synthetic_weights(kind, 7)
        frames = g["frames"][:: max(1, len(g["frames"]) // nfr)][:nfr].astype(np.int64)
        pos = {int(f): i for i, f in enumerate(g["frames"])}
        want = g["logits"][[pos[int(f)] for f in frames]]
        out.append((case, kind, W, g["logmag"], g["emb_a"], g["emb_b"], frames, want))
    # two more weight sets on the exp2 features: the reference is the float64 restatement computed here
    g = dict(np.load(os.path.join(golden, "case_exp2.npz")))
    frames = g["frames"][:: max(1, len(g["frames"]) // nfr)][:nfr].astype(np.int64)
    for name, W in (("heavy", weight_recipes.heavy("denoiser")), ("trained_bn", weight_recipes.trained_bn("denoiser"))):
        r64 = TorchRef(W, "denoiser", torch.float64)
        win = r64.windows(torch.from_numpy(g["logmag"]).double())[frames]
        ea = torch.from_numpy(g["emb_a"]).double()[None].expand(len(frames), -1)
        eb = torch.from_numpy(g["emb_b"]).double()[None].expand(len(frames), -1)
        with torch.no_grad():
            want = r64.mask_net(win, ea, eb)[0].numpy()
        out.append((name, "denoiser", W, g["logmag"], g["emb_a"], g["emb_b"], frames, want))
    return out


def main():
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    torch.set_num_threads(os.cpu_count() or 1)
    cs = cases(nfr)
    print("max |logit - float64| over %d frames per case; bar 1e-4 (golden logits are O(5); trained_bn's are O(40): its bar scales)" % nfr)
    print("%-40s %-8s " % ("8-bit corrections on", "scales") + " ".join("%12s" % c[0][-12:] for c in cs))
    for name, layers in SUBSETS:
        for block32 in ((False, True) if layers != [] else (False,)):
            row = []
            for case, kind, W, lm, ea, eb, frames, want in cs:
                ls = layers
                if ls is None:
                    ls = [k[:-2] for k, v in W.items() if k.endswith("/w") and v.ndim == 4 and k.startswith("resblock") and v.shape[2] >= 64
                          and "transform" not in k]
                ref = ProbeLayers(W, kind, ls, block32)
                win = ref.windows(torch.from_numpy(lm))[frames]
                a = torch.from_numpy(ea)[None].expand(len(frames), -1)
                b = torch.from_numpy(eb)[None].expand(len(frames), -1)
                with torch.no_grad():
                    out, _ = ref.mask_net(win, a, b)
                row.append(float(np.abs(out.numpy() - want).max()))
            print("%-40s %-8s " % (name, "block32" if block32 else "coarse") + " ".join("%12.2e" % r for r in row), flush=True)


if __name__ == "__main__":
    main()
