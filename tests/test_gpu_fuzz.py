"""Seeded random cases through the C ABI against the float64 oracle and against themselves (edge
cases -- ragged inputs, odd sizes).  tools/fuzz_batches.py is the long-running form of the invariance half."""
import numpy as np
import pytest
import torch

import nhans_amd  # noqa: F401
from nhans_amd import apply, engine, synth
from oracle import nhans_oracle as O
from oracle.torch_ref import TorchRef

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def _eng_d(lib_built, weights_denoiser):
    e = engine.Engine("denoiser", weights_denoiser)
    yield e
    e.close()


@pytest.fixture(scope="module")
def _eng_s(lib_built, weights_separator):
    e = engine.Engine("separator", weights_separator)
    yield e
    e.close()

# End to end (wav in, float32 features, float32 embeddings) the float32 arithmetic of ANY implementation sits
# 0.5-1.4e-4 from the float64 oracle in the logits: log(|X| + 1e-5) amplifies float32 FFT rounding at silent bins and
# the error rides through the embeddings.  tests/accuracy_probe.py: HIP f16x3, HIP exact-f32 and the float32 torch-CPU
# (oneDNN) restatement all land there, the CPU one slightly worse.  The 1e-4 bar of north_star is between float32
# implementations on identical features (test_gpu_parity.py holds it at 1.4e-5); here the bars are "within 2.5e-4 of
# float64" and "no further from float64 than the float32 CPU restatement is" (x1.5 + 2e-5).
E2E_LOGIT_TOL = 2.5e-4
WAV_RMS_TOL = 1e-3      # north_star: reconstructed waveform within 1e-3 RMS


def _clip(rng, frames, kind):
    n = 400 + 160 * (frames - 1) + int(rng.integers(0, 160))          # untrimmed tail included
    seed = int(rng.integers(0, 10000))
    mix = apply.trim_to_frames(apply.normalise(synth.mixture(seed, n / 16000.0 + 1e-4)[:n]))
    assert len(mix) == 400 + 160 * (frames - 1)
    if kind == "denoiser":
        ca = apply.normalise(synth.noise_context(seed + 1, float(rng.uniform(2.1, 3.0))))
        cb = apply.normalise(synth.speaker_context(seed + 2, float(rng.uniform(2.1, 3.0))))
    else:
        ca = apply.normalise(synth.speaker_context(seed + 1, float(rng.uniform(2.1, 3.0)), low=True))
        cb = apply.normalise(synth.speaker_context(seed + 2, float(rng.uniform(2.1, 3.0)), low=False))
    return mix, ca, cb


@pytest.mark.parametrize("kind, seed", [("denoiser", 11), ("denoiser", 12), ("separator", 13)])
def test_random_ragged_batches_match_the_oracle(_eng_d, _eng_s, weights_denoiser, weights_separator, kind, seed):
    eng, W = (_eng_d, weights_denoiser) if kind == "denoiser" else (_eng_s, weights_separator)
    eng.set_precision("f16x3")
    rng = np.random.default_rng(seed)
    clips = [_clip(rng, int(f), kind) for f in rng.integers(1, 6, size=3)]
    fpc = int(rng.choice([1, 2, 5, 3776]))
    try:
        eng.set_option("frames_per_chunk", fpc)
        got = eng.enhance([c[0] for c in clips], [c[1] for c in clips], [c[2] for c in clips], want_mixed=True, taps=True)
    finally:
        eng.set_option("frames_per_chunk", 3776)
    f0 = 0
    cpu32 = TorchRef(W, kind, torch.float32)
    for i, (mix, ca, cb) in enumerate(clips):
        ref = O.enhance(mix, ca, cb, W, kind)
        t = ref["logits"].shape[0]
        err = np.abs(got["logits"][f0:f0 + t] - ref["logits"]).max()
        with torch.no_grad():
            err_cpu32 = np.abs(cpu32.enhance(mix, ca, cb, faithful=False)["logits"].numpy() - ref["logits"]).max()
        assert err < E2E_LOGIT_TOL and err < 1.5 * err_cpu32 + 2e-5, (kind, seed, i, fpc, err, err_cpu32)
        assert np.sqrt(np.mean((got["denoised_wav"][i] - ref["denoised_wav"]) ** 2)) < WAV_RMS_TOL
        assert np.sqrt(np.mean((got["mixed_wav"][i] - ref["mixed_wav"]) ** 2)) < WAV_RMS_TOL
        f0 += t


def test_random_lengths_stft_istft_match_the_oracle(_eng_d):
    """STFT features and the inverse at 12 random lengths incl. the one-frame clip, one ragged launch."""
    rng = np.random.default_rng(21)
    lens = [400] + [int(x) for x in rng.integers(400, 9000, size=11)]
    wavs = [apply.trim_to_frames(apply.normalise(synth.mixture(100 + i, 1.0)[:n])) for i, n in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum([len(w) for w in wavs])]).tolist()
    lm, ph = _eng_d.stft_features(torch.from_numpy(np.concatenate(wavs)).cuda(), off)
    lm, ph = lm.cpu().numpy(), ph.cpu().numpy()
    foff = [0]
    for w in wavs:
        foff.append(foff[-1] + 1 + (len(w) - 400) // 160)
    for i, w in enumerate(wavs):
        X = O.stft(w)
        mag = np.abs(X)
        got_mag = np.exp(lm[foff[i]:foff[i + 1]].astype(np.float64)) - 1e-5
        assert np.abs(got_mag - mag).max() <= 1e-5 * max(mag.max(), 1.0), i        # 1e-5 * max|X|
        loud = mag > 1e-2 * mag.max()
        d = np.angle(np.exp(1j * (ph[foff[i]:foff[i + 1]] - np.angle(X))))
        assert np.abs(d[loud]).max() < 1e-4, i
    out, _ = _eng_d.istft(torch.from_numpy(lm).cuda(), torch.from_numpy(ph).cuda(), foff)
    out = out.cpu().numpy()
    for i, w in enumerate(wavs):
        ref = O.recover_samples(lm[foff[i]:foff[i + 1]].astype(np.float64), ph[foff[i]:foff[i + 1]].astype(np.float64))
        assert np.abs(out[off[i]:off[i + 1]] - ref).max() < 2e-5, i
