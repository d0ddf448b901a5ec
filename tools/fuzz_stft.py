"""Randomised screen of the STFT / iSTFT kernels alone: ragged batches of random lengths (one frame ... 12 s) against
torch.stft in float64 (linear-magnitude bar 1e-5 * max|X|, phase where the bin is loud), the context mode (first 200
frames, log-magnitude only), and the inverse against torch's overlap-add of irfft frames.
    python tools/fuzz_stft.py [iterations] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nhans_amd  # noqa: E402,F401
from nhans_amd import engine, spec  # noqa: E402


def ref_stft(x):
    w = torch.hann_window(400, periodic=True, dtype=torch.float64)
    return torch.stft(torch.from_numpy(x).double(), 400, 160, 400, w, center=False, return_complex=True).T.numpy()


def ref_istft(lm, ph):
    """tf.signal.inverse_stft with inverse_stft_window_fn(160, periodic hann): irfft, synthesis window, overlap-add."""
    t = lm.shape[0]
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(400) / 400)
    # the synthesis window of TF: w / sum over hops of w^2 (periodic in 160)
    wsq = np.zeros(160)
    for i in range(400):
        wsq[i % 160] += w[i] ** 2
    wsyn = w / wsq[np.arange(400) % 160]
    frames = np.fft.irfft(np.exp(lm.astype(np.float64)) * np.exp(1j * ph.astype(np.float64)), 400, axis=1) * wsyn
    out = np.zeros((t - 1) * 160 + 400)
    for i in range(t):
        out[i * 160:i * 160 + 400] += frames[i]
    return out


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    eng = engine.Engine("denoiser", precision="f16x3")
    bad = 0
    for it in range(iters):
        n = int(rng.integers(1, 9))
        frames = [int(rng.choice([1, 2, 3, 22, 23, 24, 45, 46, 47, 200, 201, int(rng.integers(1, 1200))])) for _ in range(n)]
        wavs = []
        for f in frames:
            x = rng.standard_normal(400 + 160 * (f - 1)).astype(np.float32)
            x *= np.float32(10.0 ** rng.uniform(-3, 0))                       # loud and quiet clips in one batch
            if rng.random() < 0.3:
                x[:len(x) // 2] = 0                                           # a silent half
            wavs.append(x)
        off = np.concatenate([[0], np.cumsum([len(w) for w in wavs])]).tolist()
        wav_t = torch.from_numpy(np.concatenate(wavs)).cuda()
        lm, ph = eng.stft_features(wav_t, off)
        lm, ph = lm.cpu().numpy(), ph.cpu().numpy()
        foff = np.concatenate([[0], np.cumsum(frames)]).tolist()
        for i, w in enumerate(wavs):
            X = ref_stft(w)
            mag = np.abs(X)
            got = np.exp(lm[foff[i]:foff[i + 1]].astype(np.float64)) - 1e-5
            e_mag = np.abs(got - mag).max() / max(mag.max(), 1e-3)
            loud = mag > 1e-2 * mag.max() if mag.max() > 0 else np.zeros_like(mag, bool)
            d = np.angle(np.exp(1j * (ph[foff[i]:foff[i + 1]] - np.angle(X))))
            e_ph = np.abs(d[loud]).max() if loud.any() else 0.0
            if not (e_mag < 1e-5 and e_ph < 1e-4):
                bad += 1
                print("STFT MISMATCH iter", it, "clip", i, "frames", frames[i], "mag %.2e phase %.2e" % (e_mag, e_ph))
        # context mode: the first 200 frames of the clips that have them, log-magnitude only
        long_ids = [i for i, f in enumerate(frames) if f >= 200]
        if long_ids:
            cw = [wavs[i] for i in long_ids]
            coff = np.concatenate([[0], np.cumsum([len(w) for w in cw])]).tolist()
            clm, _ = eng.stft_features(torch.from_numpy(np.concatenate(cw)).cuda(), coff, max_frames=200, want_phase=False)
            clm = clm.cpu().numpy()
            for j, i in enumerate(long_ids):
                if not np.array_equal(clm[200 * j:200 * (j + 1)], lm[foff[i]:foff[i] + 200]):
                    bad += 1
                    print("CONTEXT MISMATCH iter", it, "clip", i)
        out, ooff = eng.istft(torch.from_numpy(lm).cuda(), torch.from_numpy(ph).cuda(), foff)
        out = out.cpu().numpy()
        for i in range(n):
            ref = ref_istft(lm[foff[i]:foff[i + 1]], ph[foff[i]:foff[i + 1]])
            e = np.abs(out[ooff[i]:ooff[i + 1]] - ref).max() / max(np.abs(ref).max(), 1e-3)
            if not e < 2e-5:
                bad += 1
                print("ISTFT MISMATCH iter", it, "clip", i, "frames", frames[i], "%.2e" % e)
    print("fuzz_stft: %d iterations, %d mismatches" % (iters, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
