"""CPU tests of the oracle itself: pinned against everything the reference tree can pin
(geometry of the shipped wav pairs, the checkpoint inventory, documented TF op semantics), against
the independent torch restatement, and against the committed golden vectors."""
import json
import os

import numpy as np
import pytest
import torch

import nhans_amd  # noqa: F401
from nhans_amd import spec, synth
from oracle import nhans_oracle as O
from oracle.torch_ref import TorchRef
from conftest import GOLDEN, load_case


@pytest.fixture(scope="module")
def geometry():
    return json.load(open(os.path.join(GOLDEN, "geometry.json")))


def test_trim_and_frame_geometry_matches_shipped_examples(geometry):
    # exp1/exp2 noisy -> denoised lengths shipped with the reference (SN/audio_examples)
    for name, ex in geometry["examples"].items():
        x = np.zeros(ex["in_len"], dtype=np.float32)
        kept = len(O.trim_to_frames(x))
        t = O.stft(np.zeros(kept)).shape[0]
        assert (t - 1) * O.HOP + O.WIN == kept == ex["out_len"], name
        assert ex["out_dtype"] == "float32" and ex["rate"] == 16000
    for c in geometry["cases"]:
        kept, t = spec.frames_for_samples(c["n"])
        assert (kept, t) == (c["kept"], c["frames"])
        if c["n"] >= O.WIN:
            assert len(O.trim_to_frames(np.zeros(c["n"]))) == c["kept"]
            assert O.stft(np.zeros(c["kept"])).shape == (c["frames"], 201)
    assert spec.frames_for_samples(160000) == (159920, 998)
    assert spec.frames_for_samples(49600) == (49520, 308)


def test_synthesis_window_known_values(geometry):
    w = O.istft_window()
    assert w[0] == 0.0
    assert abs(w[200] - 0.9820893862) < 1e-9
    assert abs(w[1] - 6.5573e-5) < 1e-8
    for k, v in geometry["wsyn"].items():
        assert abs(w[int(k)] - v) < 1e-15


def test_same_padding_splits():
    # asymmetric for even kernels, (0,1) for 18 -> 9 with k=3,s=2
    assert O.same_pad(35, 4, 1) == (35, 1, 2)
    assert O.same_pad(201, 4, 2) == (101, 1, 2)
    assert O.same_pad(18, 3, 2) == (9, 0, 1)
    assert O.same_pad(101, 3, 2) == (51, 1, 1)
    assert O.same_pad(200, 8, 3) == (67, 3, 3)
    assert O.same_pad(67, 8, 1) == (67, 3, 4)
    assert O.same_pad(51, 4, 2) == (26, 1, 2)


def test_normalise_quirks():
    x = np.array([0, 100, -200], dtype=np.int16)
    np.testing.assert_allclose(O.normalise(x), (x / (200 + 1e-6)).astype(np.float32))
    assert O.normalise(np.zeros(10, dtype=np.int16)).max() == 0.0
    # int16 abs(-32768) wraps to -32768 in the reference's `abs(samples)`
    y = np.array([-32768, 5], dtype=np.int16)
    assert O.normalise(y)[1] == np.float32(5 / (5 + 1e-6))


def test_silent_context_is_log_floor():
    lm, _ = O.logmag_phase(O.stft(O.normalise(synth.silent())))
    assert lm.shape[0] >= 200
    np.testing.assert_allclose(O.context(lm), np.log(1e-5), rtol=0, atol=0)


def test_window_padding_is_zero_not_floor():
    lm = np.full((5, 201), -3.0)
    w = O.strided_crop(lm, 35)
    assert w.shape == (5, 35, 201)
    assert (w[0, :17] == 0.0).all() and (w[0, 17] == -3.0).all()
    assert (w[4, 17 + 1:] == 0.0).all()
    assert (w[2, 15:20] == -3.0).all() and (w[2, 14] == 0.0).all() and (w[2, 20] == 0.0).all()


def test_stft_istft_against_torch_and_interior_identity():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(400 + 160 * 40)
    s = O.stft(x)
    ref = torch.stft(torch.from_numpy(x), 400, 160, 400, torch.hann_window(400, periodic=True, dtype=torch.float64),
                     center=False, return_complex=True).T.numpy()
    assert np.abs(s - ref).max() < 1e-11
    y = O.inverse_stft(s)
    assert len(y) == len(x)
    assert np.abs(y[240:-240] - x[240:-240]).max() < 1e-12      # exact only in the interior
    assert abs(y[0]) < 1e-12 and np.abs(y[:240] - x[:240]).max() > 1e-3


def test_oracle_matches_torch_restatement(weights_denoiser, weights_separator):
    for kind, W in (("denoiser", weights_denoiser), ("separator", weights_separator)):
        mix = O.trim_to_frames(O.normalise(synth.mixture(5, 0.6)))
        ca = O.normalise(synth.speaker_context(5, low=True))
        cb = O.normalise(synth.noise_context(5))
        frames = [0, 30]
        r = O.enhance(mix, ca, cb, W, kind, frames=frames)
        R = TorchRef(W, kind, torch.float64)
        lm, ph = R.features(mix)
        assert np.abs(lm.numpy() - r["logmag"]).max() < 1e-9
        ea = R.tower(R.features(ca)[0][:200][None])
        eb = R.tower(R.features(cb)[0][:200][None])
        assert np.abs(ea.numpy()[0] - r["emb_a"]).max() < 1e-10
        assert np.abs(eb.numpy()[0] - r["emb_b"]).max() < 1e-10
        out, _ = R.mask_net(R.windows(lm)[frames], ea.expand(2, -1), eb.expand(2, -1))
        assert np.abs(out.numpy() - r["logits"][frames]).max() < 1e-9
        w = R.istft(torch.from_numpy(r["denoised"]), torch.from_numpy(r["phase"])).numpy()
        assert np.abs(w - r["denoised_wav"]).max() < 1e-12


def test_reference_faithful_mode_equals_deduplicated(weights_denoiser):
    """Tiling the contexts and re-running the tower per frame (reference) == embeddings once (F7)."""
    W = weights_denoiser
    mix = O.trim_to_frames(O.normalise(synth.mixture(6, 0.0475)))     # 3 frames
    ca, cb = O.normalise(synth.silent()), O.normalise(synth.noise_context(6))
    R = TorchRef(W, "denoiser", torch.float64)
    a = R.enhance(mix, ca, cb, faithful=True, mb=2)
    b = R.enhance(mix, ca, cb, faithful=False, mb=2)
    assert a["logits"].shape[0] == 3
    assert np.abs(a["logits"].numpy() - b["logits"].numpy()).max() < 1e-10
    lm, _ = O.logmag_phase(O.stft(mix))
    la, _ = O.logmag_phase(O.stft(ca))
    lb, _ = O.logmag_phase(O.stft(cb))
    out, _ = O.model(O.strided_crop(lm, 35), np.repeat(O.context(la)[None], 3, 0),
                     np.repeat(O.context(lb)[None], 3, 0), W)
    assert np.abs(out - a["logits"].numpy()).max() < 1e-9


def test_golden_vectors_reproduce(weights_denoiser, weights_separator):
    """The committed fixtures are what the oracle computes today (one frame per case keeps it fast)."""
    g = load_case("case_exp2")
    mix = O.trim_to_frames(O.normalise(O.read_wav(os.path.join(GOLDEN, "exp2_noisy.wav"))))
    r = O.enhance(mix, O.normalise(synth.silent()), O.normalise(synth.noise_context(0)), weights_denoiser,
                  "denoiser", frames=[154])
    assert r["logmag"].shape == (308, 201) and g["denoised_wav"].shape == (49520,)
    assert np.abs(r["logmag"] - g["logmag"]).max() < 1e-5
    assert np.abs(r["emb_b"] - g["emb_b"]).max() < 1e-5
    assert np.abs(r["logits"][154] - g["logits"][154]).max() < 1e-5
    s = load_case("case_separator")
    mix = O.trim_to_frames(O.normalise(synth.mixture(3, 2.0)))
    r = O.enhance(mix, O.normalise(synth.speaker_context(3, low=False)), O.normalise(synth.speaker_context(3, low=True)),
                  weights_separator, "separator", frames=[99])
    i = list(s["frames"]).index(99)
    assert np.abs(r["logits"][99] - s["logits"][i]).max() < 1e-5


def test_oracle_round_trips_a_tensorflow_written_waveform():
    """tests/golden/demo_tf_istft_mixed.wav is inverse_stft(stft(x)) written by TensorFlow
    (SN/main.py:296-306 dump of the demo material).  A signal of that form is a fixed point of
    STFT -> iSTFT in its interior (full overlap), so the oracle pair must return TF's own samples."""
    from scipy.io import wavfile
    r, y = wavfile.read(os.path.join(GOLDEN, "demo_tf_istft_mixed.wav"))
    assert r == 16000 and y.dtype == np.float32 and len(y) == 22480 and (len(y) - 400) % 160 == 0
    z = O.recover_samples(*O.logmag_phase(O.stft(y)))
    assert z.shape == y.shape
    # the 1e-5 magnitude floor of log(|X| + 1e-5) is the only thing that is not exactly inverted
    assert np.abs(z[240:-240] - y[240:-240]).max() < 5e-5
    # TF's synthesis window leaves the partially covered edges attenuated: a second pass attenuates again
    assert np.abs(z[:100]).sum() < np.abs(y[:100]).sum() or np.abs(y[:100]).sum() == 0


def _demo_segments():
    return dict(np.load(os.path.join(GOLDEN, "demo_tf_segments.npz")))


def test_oracle_round_trips_more_tensorflow_written_segments():
    """Six more `_mixed.wav` dumps of the demo material (both models; first 52 frames each): the oracle's STFT ->
    log-magnitude / phase -> iSTFT returns TensorFlow's samples in the interior."""
    d = _demo_segments()
    n = 0
    while "mixed_%d" % n in d:
        y = d["mixed_%d" % n]
        assert y.dtype == np.float32 and (len(y) - 400) % 160 == 0
        z = O.recover_samples(*O.logmag_phase(O.stft(y)))
        assert np.abs(z[240:-240] - y[240:-240]).max() < 5e-5, str(d["mixed_%d_source" % n])
        n += 1
    assert n == 6


def test_tensorflow_written_triples_pin_the_mixing_rule():
    """SN/apply.py:96-102 divides target and noise by the peak of the ALREADY normalised mixture (~1), not by the raw
    peak the mixture itself was divided by.  The reference's own dumps show it: peak_raw * mixed == target + negNoise
    sample by sample (interior of the overlap-add), with peak_raw clearly not 1 -- and the oracle's domixing
    reproduces exactly that relation (same rule, checked on synthetic signals)."""
    d = _demo_segments()
    for i in range(2):
        m, t, n = (d["triple_%d_%s" % (i, k)].astype(np.float64) for k in ("mixed", "target", "negNoise"))
        s, sl = t + n, slice(240, -240)
        pk = float(s[sl] @ m[sl]) / float(m[sl] @ m[sl])
        assert abs(pk - 1.0) > 1e-3, pk                                   # i.e. NOT normalised like the mixture
        assert np.abs(pk * m[sl] - s[sl]).max() < 2e-5 * max(1.0, np.abs(s[sl]).max())
    # the oracle's rule gives the same relation: target + neg signal == mixture * (raw peak / (max|mixed| + 1e-6))
    clean = O.trim_to_frames(O.normalise(synth.mixture(21, 1.0)))
    pos, neg = O.normalise(synth.noise_context(21, 1.0)), O.normalise(synth.speaker_context(21, 1.0))
    mixed, target, kp, kn, pos_s, neg_s = O.domixing(clean, pos, neg, 5, 3)
    raw = clean + np.float32(kp) * O._fit(pos, len(clean)) + np.float32(kn) * O._fit(neg, len(clean))
    pk = (np.abs(raw).max() + 1e-6) / (np.abs(mixed).max() + 1e-6)
    assert abs(pk - 1.0) > 1e-3
    assert np.abs(pk * mixed.astype(np.float64) - (target.astype(np.float64) + neg_s)).max() < 1e-5


def test_full_ten_second_goldens_are_pinned_to_the_numpy_oracle():
    """tests/golden/case_full10s_<kind>.npz (all 998 frames of the 10 s clips, computed with the torch float64 restatement,
    oracle/make_golden.py full10s) against what the numpy oracle wrote for the same clips: the 12 frames of logits of
    case_synth10s / case_separator10s, and the waveform the numpy oracle's own reconstruction (SN/apply.py:189-204) gives
    from the stored features + the full logits.  Both files hold float32 roundings of float64 values."""
    for kind, small in (("denoiser", "case_synth10s"), ("separator", "case_separator10s")):
        full, g = load_case("case_full10s_" + kind), load_case(small)
        assert full["logits"].shape == (998, 201) and str(full["features_case"]) == small
        fr = g["frames"].astype(np.int64)
        assert np.abs(full["logits"][fr] - g["logits"]).max() < 1e-6
        den = g["logmag"].astype(np.float64) + full["logits"].astype(np.float64)
        wav = O.recover_samples(den, g["phase"].astype(np.float64))
        assert wav.shape == full["denoised_wav"].shape
        assert np.abs(wav - full["denoised_wav"]).max() < 5e-6          # (float32-rounded features and logits in, float32 golden)
        assert np.isfinite(full["denoised_wav"]).all() and np.abs(full["denoised_wav"]).max() > 0.05
