"""Mixture synthesis used by the reference's demo and evaluation paths (host-side numpy, no GPU):
`domixing` / `combine_signals` of the denoiser (SN/apply.py:56-139 == SN/reader.py:128-223) and of
the separator (SS/apply.py:54-108).  Restated with the reference's arithmetic, including its
quirks, so that the device path is fed bit-identical waveforms:

  * powers are accumulated with Python's builtin sum() over float32 samples (sequential float32);
  * after `mixed` has been normalised, the *already normalised* mixed is used again to "normalise"
    target and the two noise signals (SN/apply.py:98-102), i.e. they are divided by ~1, not by the
    mixture's original peak;
  * the separator's combine_signals trims with an unconditional slice, so a clean recording whose
    length already fits an exact number of frames becomes EMPTY (`x[:-0]`, SS/apply.py:98).
"""
import hashlib

import numpy as np

from . import spec

SNRS_DENOISER = [-3, 0, 3, 5, 8]            # SN/apply.py:129, SN/reader.py:199
SNRS_SEPARATOR = [-5, -3, -1, 0, 1, 3, 5]   # SS/apply.py:101


def _fit_length(noise, n):
    """Repeat the noise if it is shorter than the speech, cut it if longer (SN/apply.py:58-72)."""
    nse = noise
    while n - len(nse) > 0:
        diff = n - len(nse)
        nse = np.concatenate([nse, noise[:diff]], axis=0)
    if n - len(noise) < 0:
        nse = noise[:n]
    return nse


def _power(x):
    return sum(abs(x) * abs(x)) / x.shape[0]


def domixing(cleansamples, noisepossamples, noisenegsamples, snr_pos, snr_neg):
    """Denoiser mixing (SN/apply.py:56-104).  Returns (mixed, target, K_pos, K_neg,
    noise_pos_signal, noise_neg_signal)."""
    nse_pos = _fit_length(noisepossamples, len(cleansamples))
    nse_neg = _fit_length(noisenegsamples, len(cleansamples))
    sig = cleansamples
    psignal, pnoise_pos, pnoise_neg = _power(sig), _power(nse_pos), _power(nse_neg)
    if pnoise_pos == 0:
        K_pos = 1
    else:
        K_pos = np.sqrt((psignal / pnoise_pos) * pow(10, -snr_pos / 10.0))
    if pnoise_neg == 0:
        K_neg = 1
    else:
        K_neg = np.sqrt((psignal / pnoise_neg) * pow(10, -snr_neg / 10.0))
    noise_pos_scaled = K_pos * nse_pos
    noise_neg_scaled = K_neg * nse_neg
    mixed = sig + noise_pos_scaled + noise_neg_scaled
    mixed = mixed / (max(abs(mixed)) + 0.000001)
    target = sig + noise_pos_scaled
    target = target / (max(abs(mixed)) + 0.000001)          # sic: the normalised `mixed`
    noise_pos_signal = noise_pos_scaled / (max(abs(mixed)) + 0.000001)
    noise_neg_signal = noise_neg_scaled / (max(abs(mixed)) + 0.000001)
    return mixed, target, K_pos, K_neg, noise_pos_signal, noise_neg_signal


def domixing_separator(cleansamples, noisesamples, snr):
    """Separator mixing (SS/apply.py:54-79).  Returns (mixed, K)."""
    nse = _fit_length(noisesamples, len(cleansamples))
    sig = cleansamples
    psignal, pnoise = _power(sig), _power(nse)
    if pnoise == 0:
        K = 1
    else:
        K = (psignal / pnoise) * pow(10, -snr / 10.0)
    K = np.sqrt(K)
    mixed = sig + K * nse
    mixed = mixed / (max(abs(mixed)) + 0.000001)
    return mixed, K


def eval_snrs(cleanpath):
    """Deterministic per-file SNRs of the evaluation reader (SN/reader.py:211-216).  The
    reference hashes a Python-2 str; here the path is UTF-8 encoded first."""
    h = hashlib.md5(cleanpath.encode("utf-8") if isinstance(cleanpath, str) else cleanpath).hexdigest()
    return (SNRS_DENOISER[int(h[:8], 16) % len(SNRS_DENOISER)],
            SNRS_DENOISER[int(h[:6], 16) % len(SNRS_DENOISER)])


def _normalise(x):
    with np.errstate(over="ignore"):
        return (x / (max(abs(x)) + 0.000001)).astype(np.float32)


def combine_signals(read_wav, cleanpath, noisepospath, noisenegpath, snrs=None):
    """Denoiser demo/eval front end.  snrs=None: the demo's fixed (0, 0) dB (SN/apply.py:128-134);
    otherwise a (snr_pos, snr_neg) pair, e.g. eval_snrs(cleanpath) for the evaluation reader.
    Returns (target, noise_pos_signal, noise_neg_signal, mixed, snr_pos, snr_neg)."""
    clean = _normalise(read_wav(cleanpath))
    pos = _normalise(read_wav(noisepospath))
    neg = _normalise(read_wav(noisenegpath))
    if (len(clean) - spec.WIN) % spec.HOP != 0:
        clean = clean[:-((len(clean) - spec.WIN) % spec.HOP)]
    snr_pos, snr_neg = (SNRS_DENOISER[1], SNRS_DENOISER[1]) if snrs is None else snrs
    mixed, target, _, _, pos_sig, neg_sig = domixing(clean, pos, neg, snr_pos, snr_neg)
    return target, pos_sig, neg_sig, mixed, np.array(snr_pos, dtype=np.int32), np.array(snr_neg, dtype=np.int32)


def combine_signals_separator(read_wav, cleanpath, noisepath):
    """Separator demo front end (SS/apply.py:82-105).  Returns (clean, noise*K, mixed, snr)."""
    clean = _normalise(read_wav(cleanpath))
    noise = _normalise(read_wav(noisepath))
    clean = clean[:-((len(clean) - spec.WIN) % spec.HOP)]     # sic: unconditional (see module doc)
    snr = 0
    mixed, K = domixing_separator(clean, noise, snr)
    return clean, noise * K, mixed, np.array(snr, dtype=np.int32)
