"""CPU ORACLE, second implementation (test infrastructure, NOT product code).

Independent restatement of the same reference path from torch CPU ops (torch.stft, F.conv2d
with explicit asymmetric padding, torch.fft.irfft + F.fold).  Two uses only:
  * cross-checking oracle/nhans_oracle.py (tests/test_oracle.py), in float64;
  * bench.py's `cpu_baseline` leg: the reference-faithful float32 CPU path (minibatch of 100
    frames, both 200-frame contexts tiled per frame and the embedding tower re-run for every
    minibatch exactly as SN/apply.py:381-387,440-446 feeds the TF graph), on oneDNN kernels --
    the same class of library TF-CPU uses.  "PARITY UNPINNED" applies as in nhans_oracle.py.

Citations relative to /root/reference (SN = N_HANS___Selective_Noise).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

WIN, HOP, BINS, MIX_WIN, NOISE_WIN = 400, 160, 201, 35, 200
TOWER = [("noise_resblock1_1", (3, 2)), ("noise_resblock2_1", (3, 2)),
         ("noise_resblock3_1", (1, 1)), ("noise_resblock4_1", (1, 2))]
STACK = [("resblock1_1", 1), ("resblock1_2", 1), ("resblock2_1", 2), ("resblock2_2", 1),
         ("resblock3_1", 2), ("resblock3_2", 1), ("resblock4_1", 2), ("resblock4_2", 1)]


class TorchRef:
    """Weights are converted once to NCHW/OIHW tensors of `dtype`."""

    def __init__(self, weights, kind="denoiser", dtype=torch.float32):
        self.kind, self.dtype = kind, dtype
        self.W = {}
        for k, v in weights.items():
            t = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
            if k.endswith("/w") and t.ndim == 4:
                t = t.permute(3, 2, 0, 1).contiguous()          # HWIO -> OIHW
            elif t.ndim == 4:
                t = t.reshape(1, -1, 1, 1)                       # [1,1,1,C] -> [1,C,1,1]
            self.W[k] = t

    # -- features (SN/apply.py:368-389)
    def features(self, wav):
        x = torch.as_tensor(np.asarray(wav), dtype=self.dtype)
        win = torch.hann_window(WIN, periodic=True, dtype=self.dtype)
        s = torch.stft(x, n_fft=WIN, hop_length=HOP, win_length=WIN, window=win, center=False,
                       return_complex=True).transpose(0, 1)
        return torch.log(s.abs() + 1e-5), torch.angle(s)

    @staticmethod
    def windows(lm):
        p = F.pad(lm, (0, 0, (MIX_WIN + 1) // 2 - 1, MIX_WIN // 2))
        return p.unfold(0, MIX_WIN, 1).permute(0, 2, 1)         # [T,35,201]

    # -- primitives (SN/blocks.py)
    def conv(self, x, scope, stride, padding, bias):
        w = self.W[scope + "/w"]
        if padding == "SAME":
            pads = []
            for n, k, s in ((x.shape[3], w.shape[3], stride[1]), (x.shape[2], w.shape[2], stride[0])):
                tot = max((-(-n // s) - 1) * s + k - n, 0)
                pads += [tot // 2, tot - tot // 2]
            x = F.pad(x, pads)
        out = F.conv2d(x, w, stride=stride)
        return out + self.W[scope + "/b"] if bias else out

    def bn(self, x, scope):
        g, b = self.W[scope + "/gamma"], self.W[scope + "/beta"]
        m, v = self.W[scope + "/pop_mean"], self.W[scope + "/pop_variance"]
        return (x - m) * (g * torch.rsqrt(v + 1e-3)) + b

    # -- embedding tower (SN/main.py:102-124,190-216)
    def tower(self, ctx):
        x = ctx[:, None]
        for name, st in TOWER:
            s = "embedding/" + name
            p1 = torch.relu(self.bn(self.conv(x, s + "_conv1", st, "SAME", False), s + "_conv1"))
            p1 = self.conv(p1, s + "_conv2", (1, 1), "SAME", True)
            p2 = self.conv(x, s + "_transform", st, "SAME", True)
            x = torch.relu(self.bn(p1 + p2, s + "_addition"))
        return x.mean(dim=(2, 3))

    def cont_embed(self, n, scope):
        z = torch.arange(n, dtype=self.dtype).reshape(n, 1)
        z = torch.relu(self.bn(z @ self.W[scope + "_dense1/w"], scope + scope + "_dense1"))
        z = torch.relu(self.bn(z @ self.W[scope + "_dense2/w"], scope + scope + "_dense2"))
        return z @ self.W[scope + "_dense3/w"]

    # -- conditioned stack + head (SN/main.py:126-187,219-242)
    def mask_net(self, mixed, ea, eb):
        sa, sb = (("_noise_pos_emb", "_noise_neg_emb") if self.kind == "denoiser"
                  else ("_noise_emb", "_clean_emb"))
        x = mixed[:, None]
        for name, st in STACK:
            def cond(t, s):
                pa = ea @ self.W[s + sa + "/w"] + self.W[s + sa + "/b"]
                pb = eb @ self.W[s + sb + "/w"] + self.W[s + sb + "/b"]
                te = self.cont_embed(t.shape[2], s + "_temb").t()[None, :, :, None]
                fe = self.cont_embed(t.shape[3], s + "_femb").t()[None, :, None, :]
                return t + pa[:, :, None, None] + pb[:, :, None, None] + te + fe
            p1 = self.conv(x, name + "_conv1", (st, st), "SAME", False)
            p1 = torch.relu(self.bn(cond(p1, name + "_conv1"), name + "_conv1"))
            p1 = cond(self.conv(p1, name + "_conv2", (1, 1), "SAME", True), name + "_conv2")
            p2 = x if x.shape[1] == p1.shape[1] else self.conv(x, name + "_transform", (st, st), "SAME", True)
            x = torch.relu(self.bn(p1 + p2, name + "_addition"))
        x = torch.relu(self.bn(self.conv(x, "last_conv", (1, 1), "VALID", False), "last_conv"))
        x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)       # NHWC flatten: idx = w*512 + c
        out = x @ self.W["last_dense/w"] + self.W["last_dense/b"]
        return out, mixed[:, MIX_WIN // 2, :] + out

    # -- reconstruction (SN/apply.py:189-204)
    def istft(self, logspec, phase):
        spec = torch.polar(torch.exp(logspec), phase)
        w = torch.hann_window(WIN, periodic=True, dtype=self.dtype)
        den = torch.zeros(HOP, dtype=self.dtype)
        for q in range(math.ceil(WIN / HOP)):
            seg = w[q * HOP:(q + 1) * HOP] ** 2
            den[:len(seg)] += seg
        wsyn = w / den.repeat(math.ceil(WIN / HOP))[:WIN]
        fr = torch.fft.irfft(spec, n=WIN, dim=1) * wsyn
        t = fr.shape[0]
        n = (t - 1) * HOP + WIN
        return F.fold(fr.t()[None], (1, n), (1, WIN), stride=(1, HOP)).reshape(n)

    # -- whole path
    @torch.no_grad()
    def enhance(self, mixed_wav, ctx_a_wav, ctx_b_wav, faithful=False, mb=100, max_batches=None):
        """faithful=True: tile the contexts per frame and run the tower inside every minibatch
        (reference behaviour).  faithful=False: embeddings once per clip (equal results)."""
        lm, ph = self.features(mixed_wav)
        ca = self.features(ctx_a_wav)[0][:NOISE_WIN]
        cb = self.features(ctx_b_wav)[0][:NOISE_WIN]
        win = self.windows(lm)
        t = lm.shape[0]
        outs = []
        if not faithful:
            ea, eb = self.tower(ca[None]), self.tower(cb[None])
        nb = math.ceil(t / mb)
        if max_batches is not None:
            nb = min(nb, max_batches)
        for i in range(nb):
            b = win[i * mb:(i + 1) * mb]
            if faithful:
                ea = self.tower(ca[None].expand(len(b), -1, -1))
                eb = self.tower(cb[None].expand(len(b), -1, -1))
                o, _ = self.mask_net(b, ea, eb)
            else:
                o, _ = self.mask_net(b, ea.expand(len(b), -1), eb.expand(len(b), -1))
            outs.append(o)
        logits = torch.cat(outs, 0)
        nfr = logits.shape[0]
        den = lm[:nfr] + logits
        return dict(logmag=lm, phase=ph, logits=logits, denoised=den,
                    denoised_wav=self.istft(den, ph[:nfr]), frames_done=nfr)
