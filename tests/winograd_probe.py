"""Development probe (under tests/ because it uses the oracle; not collected by pytest): would a
MAC-reducing Winograd form of the stride-1 convs of the residual stack hold the 1e-4 logit bar
when it runs on the split-f16 (hi + lo, three products, f32 accumulate) matrix path?

VERDICT r02 item 2: the 3x3 stride-1 convs (33.5 % of the FLOPs) need 16 instead of 36 products
per 2x2 outputs as F(2x2, 3x3), the 4x4 stride-1 convs (54.5 %) 25 instead of 64 as F(2x2, 4x4).
Before any kernel is written this restates such a layer on the CPU with the arithmetic the kernel
would perform and measures the end-to-end logit error against the committed float64 goldens:

  * U = G g G^T folded on the host in float64, per output channel scaled by a power of two into
    [32, 64), split into hi/lo f16 (what fold.py does with the direct weights);
  * the stored activation is its split form (hi + lo); V = B^T d B in float32 (VALU), re-split;
  * per transform position: M = V_hi U_hi + V_hi U_lo + V_lo U_hi, float32 accumulation;
  * Y = A^T M A in float32, then the layer's epilogue as today.

"direct" is the same emulation of today's kernels (no transform), so that the probe's own floor
is visible.  Gate (set by the review): logits <= 3e-5 on identical features.

    python tests/winograd_probe.py [frames]
"""
import os
import sys
from fractions import Fraction

import numpy as np
import sympy
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import weights  # noqa: E402
from oracle.torch_ref import TorchRef  # noqa: E402


# ---------------------------------------------------------------------------------- transforms
def cook_toom(m, r, points):
    """Matrices of F(m, r) (m outputs of an r-tap correlation) from len(points) = m + r - 2 finite
    interpolation points plus infinity: returns (AT [m, n], G [n, r], BT [n, n]) as exact rationals,
    Y = AT ((G g) * (BT d)).  AT and G are the evaluation matrices; BT is solved from the identity."""
    n = m + r - 1
    assert len(points) == n - 1
    pts = [sympy.Rational(p) for p in points]
    AT = sympy.Matrix(m, n, lambda i, j: (pts[j] ** i if j < n - 1 else (1 if i == m - 1 else 0)))
    norm = [sympy.prod([pts[j] - pts[k] for k in range(n - 1) if k != j]) for j in range(n - 1)]
    G = sympy.Matrix(n, r, lambda j, i: (pts[j] ** i / norm[j] if j < n - 1 else (1 if i == r - 1 else 0)))
    BT = sympy.zeros(n, n)
    for col in range(n):
        rows, rhs = [], []
        for i in range(m):
            for k in range(r):
                rows.append([AT[i, j] * G[j, k] for j in range(n)])
                rhs.append(1 if col == i + k else 0)
        sol = sympy.Matrix(rows).gauss_jordan_solve(sympy.Matrix(rhs))[0]
        assert not sol.free_symbols
        BT[:, col] = sol
    f = lambda M: np.array(M.tolist(), dtype=np.float64)
    return f(AT), f(G), f(BT)


def check_transform(m, r, points):
    AT, G, BT = cook_toom(m, r, points)
    rng = np.random.default_rng(0)
    g, d = rng.standard_normal(r), rng.standard_normal(m + r - 1)
    ref = np.array([np.dot(d[i:i + r], g) for i in range(m)])
    got = AT @ ((G @ g) * (BT @ d))
    assert np.abs(got - ref).max() < 1e-12, (m, r, points)
    return AT, G, BT


# ---------------------------------------------------------------------------------- split-f16 arithmetic
def split(x):
    """f32 tensor -> (hi, lo) f16 values held in f32; x ~ hi + lo."""
    hi = x.to(torch.float16).to(torch.float32)
    lo = (x - hi).to(torch.float16).to(torch.float32)
    return hi, lo


def stored(x):
    hi, lo = split(x)
    return hi + lo


def col_scale(w2d):
    """per output column power of two that brings max|w| into [32, 64) (fold.py: so that `lo` stays a normal f16)."""
    mx = w2d.abs().amax(dim=0).clamp_min(1e-30)
    return torch.exp2(5 - torch.floor(torch.log2(mx)))


class Probe(TorchRef):
    """TorchRef (float32) whose stride-1 stack convs run as emulated split-f16 kernels: direct or Winograd."""

    def __init__(self, W, kind, plan, products=3):
        super().__init__(W, kind, torch.float32)
        self.W64 = {k: torch.from_numpy(np.ascontiguousarray(v)).double().permute(3, 2, 0, 1).contiguous()
                    for k, v in W.items() if k.endswith("/w") and v.ndim == 4}
        self.plan = plan            # kernel size -> None (direct) or (mh, mw, points_h, points_w)
        self.products = products
        self.cache = {}
        self.macs = {"direct": 0, "done": 0}

    def x3(self, a, b, fn):
        """fn(a, b) bilinear, evaluated as the three (or `products`) f16 products with f32 accumulation."""
        ah, al = split(a)
        bh, bl = split(b)
        out = fn(ah, bh)
        if self.products >= 2:
            out = out + fn(ah, bl)
        if self.products >= 3:
            out = out + fn(al, bh)
        return out

    def conv(self, x, scope, stride, padding, bias):
        w = self.W[scope + "/w"]
        kh, kw = w.shape[2], w.shape[3]
        emulate = (not scope.startswith("embedding") and w.shape[1] >= 64 and padding == "SAME")
        if not emulate:
            return super().conv(x, scope, stride, padding, bias)
        pads = []
        for n, k, s in ((x.shape[3], kw, stride[1]), (x.shape[2], kh, stride[0])):
            tot = max((-(-n // s) - 1) * s + k - n, 0)
            pads += [tot // 2, tot - tot // 2]
        ho, wo = -(-x.shape[2] // stride[0]), -(-x.shape[3] // stride[1])
        direct_macs = x.shape[0] * ho * wo * kh * kw * w.shape[1] * w.shape[0]
        self.macs["direct"] += direct_macs
        x = stored(x)
        plan = self.plan.get((kh, kw)) if stride == (1, 1) else None
        if plan is None:
            sc = col_scale(w.reshape(w.shape[0], -1).t())
            out = self.x3(F.pad(x, pads), w * sc[:, None, None, None], lambda a, b: F.conv2d(a, b, stride=stride))
            out = out / sc[None, :, None, None]
            self.macs["done"] += direct_macs
        else:
            out = self.winograd(x, scope, pads, plan, ho, wo)
        return out + self.W[scope + "/b"] if bias else out

    def winograd(self, x, scope, pads, plan, ho, wo):
        mh, mw, ph, pw = plan
        w64 = self.W64[scope + "/w"]                      # [O, I, kh, kw]
        kh, kw = w64.shape[2], w64.shape[3]
        key = (mh, kh, tuple(ph)), (mw, kw, tuple(pw))
        if key not in self.cache:
            self.cache[key] = (check_transform(mh, kh, ph) if mh > 1 else None,
                               check_transform(mw, kw, pw) if mw > 1 else None)
        th, tw = self.cache[key]
        ident = lambda n, k: (np.eye(n), np.eye(k), np.eye(n + k - 1))
        ATh, Gh, BTh = th if th else ident(1, kh)
        ATw, Gw, BTw = tw if tw else ident(1, kw)
        if th is None:       # 1-D along W: each filter row is its own position (nh = kh, reduced with the channels)
            ATh, Gh, BTh = np.ones((1, kh)), np.eye(kh), np.eye(kh)
        nh, nw = BTh.shape[0], BTw.shape[0]
        # U[p, q, I, O] in float64, scaled per output channel, split
        U = torch.einsum("pa,oiab,qb->pqio", torch.from_numpy(Gh), w64, torch.from_numpy(Gw))
        sc = col_scale(U.permute(0, 1, 2, 3).reshape(-1, U.shape[3]).float()).double()
        U = (U * sc).float()
        # tiles: pad so that the tile grid covers the output
        nth, ntw = -(-ho // mh), -(-wo // mw)
        need_h, need_w = (nth - 1) * mh + nh, (ntw - 1) * mw + nw
        xp = F.pad(x, pads)
        xp = F.pad(xp, (0, max(0, need_w - xp.shape[3]), 0, max(0, need_h - xp.shape[2])))
        t = xp.unfold(2, nh, mh).unfold(3, nw, mw)        # [B, C, nth, ntw, nh, nw]
        t = t[:, :, :nth, :ntw]
        V = torch.einsum("pa,bcuvax,qx->bcuvpq", torch.from_numpy(BTh).float(), t, torch.from_numpy(BTw).float())
        M = self.x3(V, U, lambda a, b: torch.einsum("bcuvpq,pqco->bouvpq", a, b))
        Y = torch.einsum("ip,bouvpq,jq->bouivj", torch.from_numpy(ATh).float(), M, torch.from_numpy(ATw).float())
        Y = Y.reshape(Y.shape[0], Y.shape[1], nth * mh, ntw * mw)[:, :, :ho, :wo]
        self.macs["done"] += x.shape[0] * nth * ntw * nh * nw * U.shape[2] * U.shape[3]
        return Y / sc.float()[None, :, None, None]


Fr = Fraction
P5 = (0, 1, -1, 2, -2)
P5m = (0, 1, -1, Fr(1, 2), -2)
P7 = (0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2))
PLANS = {
    "direct (today's kernels, emulated)": {},
    "2-D F(2x2,3x3) on the 3x3 convs": {(3, 3): (2, 2, (0, 1, -1), (0, 1, -1))},
    "2-D F(2x2,4x4) pts(0,1,-1,2) on the 4x4 convs": {(4, 4): (2, 2, (0, 1, -1, 2), (0, 1, -1, 2))},
    "2-D F(4x4,3x3) + F(3x3,4x4) pts(0,+-1,+-2)": {(3, 3): (4, 4, P5, P5), (4, 4): (3, 3, P5, P5)},
    "1-D along W: F(2,3) / F(2,4)": {(3, 3): (1, 2, (), (0, 1, -1)), (4, 4): (1, 2, (), (0, 1, -1, 2))},
    "1-D along W: F(4,3) / F(3,4) pts(0,+-1,+-2)": {(3, 3): (1, 4, (), P5), (4, 4): (1, 3, (), P5)},
    "1-D along W: F(4,3) / F(3,4) pts(0,+-1,1/2,-2)": {(3, 3): (1, 4, (), P5m), (4, 4): (1, 3, (), P5m)},
    "1-D along W: F(6,3) / F(5,4) pts(0,+-1,+-2,+-1/2)": {(3, 3): (1, 6, (), P7), (4, 4): (1, 5, (), P7)},
    "1-D along W: F(2,3) / F(5,4)": {(3, 3): (1, 2, (), (0, 1, -1)), (4, 4): (1, 5, (), P7)},
}


def main():
    nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    torch.set_num_threads(os.cpu_count() or 1)
    golden = os.path.join(ROOT, "tests", "golden")
    print("%-48s %12s %12s %12s %s" % ("variant", "exp2 logits", "10s logits", "sep logits", "MACs vs direct"))
    for name, plan in PLANS.items():
        row, ratio = [], None
        for case, kind in (("case_exp2", "denoiser"), ("case_synth10s", "denoiser"), ("case_separator10s", "separator")):
            g = dict(np.load(os.path.join(golden, case + ".npz")))
            W = weights.synthetic_weights(kind, 7)
            ref = Probe(W, kind, plan)
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
            ratio = ref.macs["done"] / ref.macs["direct"]
        print("%-48s %12.2e %12.2e %12.2e %.3f" % (name, row[0], row[1], row[2], ratio), flush=True)


if __name__ == "__main__":
    main()
