"""Deterministic synthetic 16 kHz material for benchmarks and parity tests (no external data).

Recipe: speech-like = 8 harmonics of an f0 gliding 110-220 Hz under a
4 Hz syllabic envelope, plus coloured Gaussian noise at 5 dB SNR; peak-normalised and quantised
to int16 so the reference's read/normalise path is exercised.  Clip i uses seed 20240 + i, its
noise context seed 90210 + i.
"""
import numpy as np

from . import spec


def coloured_noise(n, rng, pole=0.9):
    """First-order low-passed white noise, unit variance."""
    w = rng.standard_normal(n)
    y = np.empty(n)
    acc = 0.0
    # y[k] = pole*y[k-1] + w[k] via a stable vectorised recursion (lfilter without scipy import cost)
    from scipy.signal import lfilter
    y = lfilter([1.0], [1.0, -pole], w)
    return y / (np.std(y) + 1e-12)


def speechlike(n, rng, f0_lo=110.0, f0_hi=220.0):
    t = np.arange(n) / spec.FS
    glide = f0_lo + (f0_hi - f0_lo) * 0.5 * (1 + np.sin(2 * np.pi * 0.35 * t + rng.uniform(0, 2 * np.pi)))
    phase = 2 * np.pi * np.cumsum(glide) / spec.FS
    amps = rng.uniform(0.3, 1.0, size=8) / np.arange(1, 9)
    sig = sum(a * np.sin(h * phase) for h, a in zip(range(1, 9), amps))
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi))
    sig = sig * env
    return sig / (np.std(sig) + 1e-12)


def to_int16(x):
    x = x / (np.max(np.abs(x)) + 1e-12)
    return np.round(x * 32767.0).astype(np.int16)


def mixture(i, seconds=10.0, snr_db=5.0):
    """Clip i: int16 speech-like signal + coloured noise at `snr_db`."""
    rng = np.random.default_rng(20240 + i)
    n = int(round(seconds * spec.FS))
    s = speechlike(n, rng)
    nz = coloured_noise(n, rng) * 10 ** (-snr_db / 20.0)
    return to_int16(s + nz)


def noise_context(i, seconds=3.0):
    """Conditioning recording for clip i (>= 32,240 samples so it yields 200 frames)."""
    rng = np.random.default_rng(90210 + i)
    return to_int16(coloured_noise(int(round(seconds * spec.FS)), rng))


def speaker_context(i, seconds=3.0, low=True):
    """Separator conditioning: a synthetic 'speaker' with its own f0 range."""
    rng = np.random.default_rng((31337 if low else 42424) + i)
    lo, hi = (90.0, 160.0) if low else (180.0, 300.0)
    return to_int16(speechlike(int(round(seconds * spec.FS)), rng, lo, hi))


def silent(seconds=3.0):
    return np.zeros(int(round(seconds * spec.FS)), dtype=np.int16)
