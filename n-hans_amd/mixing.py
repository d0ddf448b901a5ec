"""Mixture synthesis used by the reference's demo and evaluation paths (host-side numpy, no GPU):
`domixing` / `combine_signals` of the denoiser (SN/apply.py:56-139 == SN/reader.py:128-223) and of
the separator (SS/apply.py:54-108).  Restated with the reference's arithmetic, including its
quirks, so that the device path is fed bit-identical waveforms:

  * powers: `sum(abs(x) * abs(x)) / n` with Python's builtin sum().  The reference is pinned to
    TensorFlow 1.14 / NumPy 1.x, where `0 + np.float32` is a float64 (value-based promotion of the
    int start value), so the float32 squares are added one after the other IN FLOAT64 and the gain
    is a float64 scalar; a float64 scalar times a float32 array stays float32 there (the scalar is
    cast down).  NumPy 2 (this image) would do both differently, so both are spelled out here;
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
    """A noise recording as long as the speech: whole repetitions plus a head piece when it is
    shorter, its first n samples when it is longer (what SN/apply.py:58-72 arrives at by repeated
    concatenation; an empty recording stays empty there too)."""
    m = len(noise)
    if m >= n or m == 0:
        return noise[:n]
    reps, rest = divmod(n, m)
    return np.concatenate([noise] * reps + [noise[:rest]], axis=0)


def _mean_power(x):
    """Mean square of a recording the way the reference's stack computes it: squares in the
    recording's own precision, then added one after the other in float64 (builtin sum() under
    NumPy 1.x, see the module text; np.add.accumulate performs the same left-to-right additions
    without the per-element Python cost)."""
    sq = np.abs(x) * np.abs(x)
    if sq.ndim != 1 or len(sq) == 0:
        return sum(sq) / x.shape[0]
    return np.add.accumulate(sq.astype(np.float64))[-1] / x.shape[0]


def _scaled(gain, x):
    """gain * x with NumPy 1.x casting: a float scalar never widens a floating-point array."""
    if isinstance(x, np.ndarray) and x.dtype.kind == 'f':
        return x.dtype.type(gain) * x
    return gain * x


def _gain_for_snr(p_speech, p_noise, snr_db):
    """Factor on the noise that puts it snr_db below the speech; silence is left alone (gain 1)."""
    if p_noise == 0:
        return 1
    return np.sqrt((p_speech / p_noise) * pow(10, -snr_db / 10.0))


def _peak(x):
    return np.max(np.abs(x)) + 0.000001


def domixing(speech, noise_keep, noise_drop, snr_keep_db, snr_drop_db):
    """Three-way mixture of the denoiser (behaviour of SN/apply.py:56-104): speech, the noise the
    model is told to KEEP (positive conditioning) and the one it is told to DROP (negative), each
    noise brought to its SNR against the speech.  Returns (mixture, target, gain_keep, gain_drop,
    keep_signal, drop_signal) where target = speech + kept noise.

    Quirk preserved: only the mixture is peak-normalised by its own peak; the other three signals
    are divided by the peak of the ALREADY normalised mixture (~1), not by the original peak."""
    n = len(speech)
    keep = _fit_length(noise_keep, n)
    drop = _fit_length(noise_drop, n)
    p_speech = _mean_power(speech)
    gain_keep = _gain_for_snr(p_speech, _mean_power(keep), snr_keep_db)
    gain_drop = _gain_for_snr(p_speech, _mean_power(drop), snr_drop_db)
    keep_scaled = _scaled(gain_keep, keep)
    drop_scaled = _scaled(gain_drop, drop)
    raw = speech + keep_scaled + drop_scaled
    mixture = raw / _peak(raw)
    unit = _peak(mixture)
    return mixture, (speech + keep_scaled) / unit, gain_keep, gain_drop, keep_scaled / unit, drop_scaled / unit


def domixing_separator(target_speech, interferer, snr_db):
    """Two-speaker mixture of the separator (behaviour of SS/apply.py:54-79).  Returns
    (mixture, gain) with the mixture peak-normalised."""
    other = _fit_length(interferer, len(target_speech))
    gain = _gain_for_snr(_mean_power(target_speech), _mean_power(other), snr_db)
    raw = target_speech + _scaled(gain, other)
    return raw / _peak(raw), gain


def eval_snrs(cleanpath):
    """Deterministic per-file SNRs of the evaluation reader (SN/reader.py:211-216).  The
    reference hashes a Python-2 str; here the path is UTF-8 encoded first."""
    h = hashlib.md5(cleanpath.encode("utf-8") if isinstance(cleanpath, str) else cleanpath).hexdigest()
    return (SNRS_DENOISER[int(h[:8], 16) % len(SNRS_DENOISER)],
            SNRS_DENOISER[int(h[:6], 16) % len(SNRS_DENOISER)])


def _normalise(x):
    with np.errstate(over="ignore"):       # (int16 abs(-32768) wraps, as in the reference)
        return (x / (np.max(np.abs(x)) + 0.000001)).astype(np.float32)


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
    return clean, _scaled(K, noise), mixed, np.array(snr, dtype=np.int32)
