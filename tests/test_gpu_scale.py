"""GPU tests at BASELINE configuration scale (configs[2]: hundreds of clips, hundreds of
chunks of frame windows, clips straddling chunk boundaries) and of the post-outputs / separator at 10 s.
Oracle runs of that size are out of reach, so the checks are: committed golden frames of a clip placed
at several batch positions, bitwise equality with the single-clip run (batch, chunk and position
invariance), finiteness of every output, and the STFT->iSTFT identity on sampled clips.
"""
import os
import warnings

import numpy as np
import pytest
import torch

import nhans_amd  # noqa: F401
from nhans_amd import apply, engine, hip, spec, synth
from oracle import nhans_oracle as O
from conftest import GOLDEN, load_case

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4
WAV_RMS_TOL = 1e-3


@pytest.fixture(scope="module")
def eng(lib_built, weights_denoiser):
    e = engine.Engine("denoiser", weights_denoiser, precision="f16x3")
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng_sep(lib_built, weights_separator):
    e = engine.Engine("separator", weights_separator, precision="f16x3")
    yield e
    e.close()


def _clip(cid, seconds):
    return (apply.trim_to_frames(apply.normalise(synth.mixture(cid, seconds))),
            apply.normalise(synth.silent()), apply.normalise(synth.noise_context(cid)))


def _golden_10s():
    return _clip(0, 10.0)         # the clip of tests/golden/case_synth10s.npz (oracle/make_golden.py)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_ragged_batch_of_64_clips_across_chunk_boundaries(eng, prec):
    """64 clips of 1..998 frames in one nhans_enhance_clips call with a chunk size (1000) that no
    clip length divides: most clips straddle a chunk boundary.  The golden 10 s clip sits at batch
    positions 0, 31 and 63."""
    eng.set_precision(prec)
    g = load_case("case_synth10s")
    secs = [0.025, 0.035, 0.31, 0.77, 1.0, 1.5, 2.25, 0.5]
    clips = []
    for i in range(64):
        clips.append(_golden_10s() if i in (0, 31, 63) else _clip(100 + i, secs[i % len(secs)]))
    mixes, cas, cbs = [list(x) for x in zip(*clips)]
    nfr = [spec.frames_for_samples(len(m))[1] for m in mixes]
    foff = np.concatenate([[0], np.cumsum(nfr)])
    eng.set_option("frames_per_chunk", 1000)
    try:
        out = eng.enhance(mixes, cas, cbs, want_mixed=True, taps=True)
        assert eng.take_status() == 0
        assert out["logits"].shape == (foff[-1], 201) and foff[-1] > 5000
        assert np.isfinite(out["logits"]).all()
        for w in out["denoised_wav"]:
            assert np.isfinite(w).all()
        for pos in (0, 31, 63):
            lg = out["logits"][foff[pos]:foff[pos + 1]]
            assert np.abs(lg[g["frames"]] - g["logits"]).max() < 5 * LOGIT_TOL, pos   # kernel's own features
            assert np.array_equal(lg, out["logits"][foff[0]:foff[1]])                   # position-invariant, bit for bit
            assert np.array_equal(out["denoised_wav"][pos], out["denoised_wav"][0])
        # sampled clips: the batch run == the clip on its own (other chunking too), bit for bit
        eng.set_option("frames_per_chunk", 1024)                      # (another chunking than the batch run)
        for i in (0, 1, 2, 17, 30, 32, 62):
            single = eng.enhance([mixes[i]], [cas[i]], [cbs[i]], want_mixed=True, taps=True)
            assert np.array_equal(single["logits"], out["logits"][foff[i]:foff[i + 1]]), i
            assert np.array_equal(single["denoised_wav"][0], out["denoised_wav"][i]), i
            assert np.array_equal(single["mixed_wav"][0], out["mixed_wav"][i]), i
    finally:
        eng.set_option("frames_per_chunk", 3776)


def test_batch_of_256_ten_second_clips(eng):
    """BASELINE configs[2] in one call: 256 x 10 s = 255,488 frame windows, 68 chunks of 3,776.  32
    distinct clips repeated 8 times, the golden clip among them: golden frames at three batch
    positions, every repetition bit-identical to the first, every sample finite, round trip intact."""
    eng.set_precision("f16x3")
    g = load_case("case_synth10s")
    distinct = [_golden_10s()] + [_clip(200 + i, 10.0) for i in range(1, 32)]
    clips = [distinct[i % 32] for i in range(256)]
    mixes, cas, cbs = [list(x) for x in zip(*clips)]
    mix_t, mix_off = eng._dev(mixes)
    ca_t, ca_off = eng._dev(cas)
    cb_t, cb_off = eng._dev(cbs)
    res = eng.enhance_device(mix_t, mix_off, ca_t, ca_off, cb_t, cb_off, want_mixed=True, taps=True)
    assert eng.take_status() == 0                                  # no f16 saturation anywhere
    lg = res["logits"].view(256, 998, 201)
    den = res["denoised_wav"].view(256, 159920)
    assert bool(torch.isfinite(lg).all()) and bool(torch.isfinite(den).all())
    gl = torch.from_numpy(g["logits"]).to(lg.device)
    fr = torch.from_numpy(g["frames"].astype(np.int64)).to(lg.device)
    for pos in (0, 96, 224):                                       # golden clip = every 32nd
        assert float((lg[pos][fr] - gl).abs().max()) < 5 * LOGIT_TOL, pos
    first = lg[:32]
    for rep in range(1, 8):                                        # chunk phase differs per repetition: 998*32 % 3776 != 0
        assert torch.equal(lg[32 * rep:32 * rep + 32], first), rep
        assert torch.equal(den[32 * rep:32 * rep + 32], den[:32]), rep
    # single-clip run of three of them == their rows in the batch
    for i in (0, 13, 31):
        single = eng.enhance([mixes[i]], [cas[i]], [cbs[i]], want_mixed=False, taps=True)
        assert np.array_equal(single["logits"], lg[i].cpu().numpy()), i
        assert np.array_equal(single["denoised_wav"][0], den[i].cpu().numpy()), i
    # iSTFT(STFT(x)) == x in the interior, for a sample of clips
    rt = res["mixed_wav"].view(256, 159920)
    for i in (5, 77, 255):
        x = torch.from_numpy(mixes[i]).to(rt.device)
        assert float((rt[i][240:-240] - x[240:-240]).abs().max()) < 2e-4


def test_largest_chunk_the_fast_kernels_address(eng):
    """frames_per_chunk is bounded by the conv kernels' 32-bit element offsets: 4,769 frame windows (the largest
    tensor is then 2,147,194,560 elements) is accepted and runs every layer on the SAME kernels as the default
    chunking (checked through the profile: no launch falls back to the 64-bit-offset LDS-DMA kernel), bit-identical
    to it; one more is refused with NHANS_EINVAL instead of silently running slow (or overflowing int at 305,000)."""
    eng.set_precision("f16x3")
    clips = [_clip(900 + i, 10.0) for i in range(5)]                     # 4,990 frames: chunks of 4,769 + 221
    args = ([c[0] for c in clips], [c[1] for c in clips], [c[2] for c in clips])

    def run():
        eng.set_option("profile", 1)
        eng.profile_reset()
        out = eng.enhance(*args, want_mixed=False, taps=True)
        calls = {k: v["calls"] for k, v in eng.profile().items() if k.startswith("conv_igemm")}
        eng.set_option("profile", 0)
        return out, calls
    ref, ref_calls = run()
    try:
        with pytest.raises(hip.NhansError, match="frames_per_chunk"):
            eng.set_option("frames_per_chunk", 4770)
        with pytest.raises(hip.NhansError, match="frames_per_chunk"):
            eng.set_option("frames_per_chunk", 400000)
        eng.set_option("frames_per_chunk", 4769)
        got, calls = run()
    finally:
        eng.set_option("profile", 0)
        eng.set_option("frames_per_chunk", 3776)
    assert set(calls) == set(ref_calls) and "conv_igemm_dma<64>" not in calls, (calls, ref_calls)
    assert eng.take_status() == 0
    assert np.array_equal(got["logits"], ref["logits"])
    for a, b in zip(got["denoised_wav"], ref["denoised_wav"]):
        assert np.array_equal(a, b)


def test_separator_batch_of_128_ten_second_clips(eng_sep):
    """The per-rank share of BASELINE configs[4] (separator, 1,024 clips over 8 GPUs = 128 x 10 s per GPU,
    SS/apply.py:288-397) in ONE nhans_enhance_clips call: 127,744 frame windows, 34 chunks.  16 distinct speaker
    mixtures repeated 8 times, the golden clip of case_separator10s among them: its golden frames at three batch
    positions, every repetition bit-identical to the first (the chunk phase differs per repetition), everything
    finite, no f16 saturation, and single-clip runs equal to their rows of the batch."""
    eng_sep.set_precision("f16x3")
    g = load_case("case_separator10s")

    def sep_clip(cid):
        return (apply.trim_to_frames(apply.normalise(synth.mixture(cid, 10.0))),
                apply.normalise(synth.speaker_context(cid, low=True)),      # interferer (--neg) -> noise context
                apply.normalise(synth.speaker_context(cid, low=False)))     # target (--pos) -> clean context
    distinct = [sep_clip(5)] + [sep_clip(600 + i) for i in range(1, 16)]     # clip 5 = the golden one
    clips = [distinct[i % 16] for i in range(128)]
    mixes, cas, cbs = [list(x) for x in zip(*clips)]
    mix_t, mix_off = eng_sep._dev(mixes)
    ca_t, ca_off = eng_sep._dev(cas)
    cb_t, cb_off = eng_sep._dev(cbs)
    res = eng_sep.enhance_device(mix_t, mix_off, ca_t, ca_off, cb_t, cb_off, want_mixed=True, taps=True)
    assert eng_sep.take_status() == 0
    lg = res["logits"].view(128, 998, 201)
    den = res["denoised_wav"].view(128, 159920)
    assert bool(torch.isfinite(lg).all()) and bool(torch.isfinite(den).all())
    emb = res["emb"].view(2, 128, 512)
    assert float((emb[0, 0].cpu() - torch.from_numpy(g["emb_a"])).abs().max()) < 1e-4
    assert float((emb[1, 0].cpu() - torch.from_numpy(g["emb_b"])).abs().max()) < 1e-4
    gl = torch.from_numpy(g["logits"]).to(lg.device)
    fr = torch.from_numpy(g["frames"].astype(np.int64)).to(lg.device)
    for pos in (0, 48, 112):                                       # golden clip = every 16th
        assert float((lg[pos][fr] - gl).abs().max()) < 5 * LOGIT_TOL, pos
    for rep in range(1, 8):                                        # 998*16 % 3776 != 0: another chunk phase each time
        assert torch.equal(lg[16 * rep:16 * rep + 16], lg[:16]), rep
        assert torch.equal(den[16 * rep:16 * rep + 16], den[:16]), rep
        assert torch.equal(emb[:, 16 * rep:16 * rep + 16], emb[:, :16]), rep
    for i in (0, 7, 15):
        single = eng_sep.enhance([mixes[i]], [cas[i]], [cbs[i]], want_mixed=False, taps=True)
        assert np.array_equal(single["logits"], lg[i].cpu().numpy()), i
        assert np.array_equal(single["denoised_wav"][0], den[i].cpu().numpy()), i
    rt = res["mixed_wav"].view(128, 159920)
    for i in (3, 127):
        x = torch.from_numpy(mixes[i]).to(rt.device)
        assert float((rt[i][240:-240] - x[240:-240]).abs().max()) < 2e-4


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_separator_ten_second_clip(eng_sep, prec):
    eng_sep.set_precision(prec)
    g = load_case("case_separator10s")
    mix = apply.trim_to_frames(apply.normalise(synth.mixture(5, 10.0)))
    ca = apply.normalise(synth.speaker_context(5, low=True))          # interferer (--neg) -> noise context
    cb = apply.normalise(synth.speaker_context(5, low=False))         # target (--pos) -> clean context
    out = eng_sep.enhance([mix], [ca], [cb], want_mixed=False, taps=True)
    assert out["logits"].shape == (998, 201)
    assert np.abs(out["emb"][0] - g["emb_a"]).max() < 1e-4 and np.abs(out["emb"][1] - g["emb_b"]).max() < 1e-4
    assert np.abs(out["logits"][g["frames"]] - g["logits"]).max() < 5 * LOGIT_TOL
    # identical features in: the strict logit bar
    lm = torch.from_numpy(g["logmag"]).cuda()
    lg, _ = eng_sep.mask_net(lm, [0, 998], torch.from_numpy(g["emb_a"][None]).cuda(), torch.from_numpy(g["emb_b"][None]).cuda())
    assert np.abs(lg.cpu().numpy()[g["frames"]] - g["logits"]).max() < LOGIT_TOL
    # a batch of 4 of them, chunked mid-clip == the single run
    eng_sep.set_option("frames_per_chunk", 700)
    try:
        b = eng_sep.enhance([mix] * 4, [ca] * 4, [cb] * 4, want_mixed=False, taps=True)
    finally:
        eng_sep.set_option("frames_per_chunk", 3776)
    for i in range(4):
        assert np.array_equal(b["logits"][998 * i:998 * (i + 1)], out["logits"])
        assert np.array_equal(b["denoised_wav"][i], out["denoised_wav"][0])


@pytest.mark.parametrize("mode", ["fixed", "ac"])
def test_post_outputs_removed_snr_compensated(eng, mode, tmp_path, capsys):
    """SN/apply.py:456-472 through apply_snc: removed = mixed_processed - denoised, snr_est, and
    compensated with --compensate 0.3 / --ac, against the oracle's restatement."""
    from scipy.io import wavfile
    eng.set_precision("f16x3")
    apply.set_engine("denoiser", eng)
    g = load_case("case_postproc")
    mixp, negp = str(tmp_path / "in.wav"), str(tmp_path / "neg.wav")
    wavfile.write(mixp, 16000, synth.mixture(40, 0.6))
    wavfile.write(negp, 16000, synth.noise_context(40))
    out = str(tmp_path / "clip_denoised.wav")
    apply.FLAGS.ac, apply.FLAGS.compensate = (mode == "ac"), (0.0 if mode == "ac" else 0.3)
    try:
        apply.apply_snc(mixp, str(tmp_path / "Silent.wav"), negp, out)        # absent Silent.wav -> zeros
    finally:
        apply.FLAGS.ac, apply.FLAGS.compensate = False, 0.0
    printed = float(capsys.readouterr().out.split()[0])
    assert abs(printed / float(g["snr_est_" + mode]) - 1) < 1e-3
    for name, key in (("clip_denoised.wav", "denoised_wav"), ("clip_mixed_processed.wav", "mixed_wav"),
                      ("clip_removed.wav", "removed_wav"), ("clip_compensated.wav", "compensated_wav")):
        r, w = wavfile.read(str(tmp_path / name))
        ref = g["%s_%s" % (key, mode)]
        assert r == 16000 and w.dtype == np.float32 and w.shape == ref.shape
        assert np.sqrt(np.mean((w - ref) ** 2)) < WAV_RMS_TOL, name
        assert np.abs(w - ref).max() < 2e-3, name
    _, den = wavfile.read(out)
    _, comp = wavfile.read(str(tmp_path / "clip_compensated.wav"))
    assert not np.array_equal(den, comp)                                       # the factor really is non-zero


def test_tf_written_waveform_round_trips(eng):
    """`*_mixed.wav` of the reference's demo material is inverse_stft(stft(x)) written by TensorFlow
    (SN/main.py:296-306).  In the interior of such a signal STFT -> iSTFT is the identity, so the
    HIP pair must give the TF-written samples back."""
    from scipy.io import wavfile
    r, y = wavfile.read(os.path.join(GOLDEN, "demo_tf_istft_mixed.wav"))
    assert r == 16000 and y.dtype == np.float32 and (len(y) - 400) % 160 == 0
    t = torch.from_numpy(y).cuda()
    lm, ph = eng.stft_features(t, [0, len(y)])
    w, _ = eng.istft(lm, ph, [0, lm.shape[0]])
    w = w.cpu().numpy()
    assert w.shape == y.shape
    assert np.abs(w[240:-240] - y[240:-240]).max() < 2e-4 * max(1.0, np.abs(y).max())
    # and against the oracle on the whole signal, edges included
    ref = O.recover_samples(*O.logmag_phase(O.stft(y)))
    assert np.sqrt(np.mean((w - ref) ** 2)) < 1e-5


def test_directory_mode_is_one_batch_with_distinct_side_files(eng, tmp_path):
    """README.md:59-66: --input/--output/--neg directories paired by file name.  All clips go through
    ONE ragged call; side files are per clip (the reference's save_to[:-12] cut would collide)."""
    from scipy.io import wavfile
    eng.set_precision("f16x3")
    apply.set_engine("denoiser", eng)
    ind, negd, outd = tmp_path / "in", tmp_path / "neg", tmp_path / "out"
    ind.mkdir(); negd.mkdir()
    names = ["a.wav", "bb.wav", "recording_0001.wav", "recording_0002.wav"]
    for i, n in enumerate(names):
        wavfile.write(str(ind / n), 16000, synth.mixture(60 + i, 0.3 + 0.2 * i))
        wavfile.write(str(negd / n), 16000, synth.noise_context(60 + i))
    calls = []
    orig = eng.enhance
    eng.enhance = lambda *a, **k: (calls.append(len(a[0])), orig(*a, **k))[1]
    try:
        apply.main(["--input", str(ind), "--neg", str(negd), "--pos", str(tmp_path / "Silent.wav"),
                    "--output", str(outd), "--weights", "synthetic"])
    finally:
        del eng.enhance
    assert calls == [4]                                                        # one batch
    got = sorted(os.listdir(str(outd)))
    want = sorted(n for s in names for n in (s, s[:-4] + "_mixed_processed.wav", s[:-4] + "_removed.wav",
                                             s[:-4] + "_compensated.wav"))
    assert got == want
    for i, n in enumerate(names):                                              # == the file-mode result
        single = str(tmp_path / ("single%d_denoised.wav" % i))
        apply.apply_snc(str(ind / n), str(tmp_path / "Silent.wav"), str(negd / n), single)
        assert np.array_equal(wavfile.read(single)[1], wavfile.read(str(outd / n))[1])


def test_launch_failure_reaches_the_caller(lib_built):
    """A launch the runtime rejects must come back as NHANS_EHIP, not as NHANS_OK + garbage."""
    lib = hip.load()
    assert lib.nhans_debug_launch_probe(1024, None) == 0
    assert lib.nhans_debug_launch_probe(150 * 1024, None) == 0                # within the 160 KB of a gfx950 CU
    torch.cuda.synchronize()
    rc = lib.nhans_debug_launch_probe(1 << 20, None)                           # 1 MB of LDS does not exist
    assert rc == -2 and b"launch_probe" in lib.nhans_last_error()
    assert lib.nhans_debug_launch_probe(1024, None) == 0                       # the error does not stick
    torch.cuda.synchronize()


def _scaled_block1(weights_denoiser):
    W = dict(weights_denoiser)
    W["resblock1_1_conv1/w"] = (W["resblock1_1_conv1/w"] * np.float32(3.0e5)).astype(np.float32)
    return W


def test_activation_exponents_keep_out_of_range_weights_in_f16x3(lib_built, weights_denoiser):
    """f16x3 carries activations as hi+lo f16: |stored| >= 65504 cannot be represented.  Every stored tensor has a
    power-of-two exponent that nhans_create calibrates (include/nhans_hip.h: "calibrate"): weights scaled so that
    block 1 would overflow by a factor of ~100 run in f16x3 WITHOUT the status flag and without an f32 rerun, and
    agree with the f32 matrix-core result like any other model."""
    W = _scaled_block1(weights_denoiser)
    mix, ca, cb = _clip(7, 0.3)
    e32 = engine.Engine("denoiser", W, precision="f32")
    ref = e32.enhance([mix], [ca], [cb], want_mixed=False, taps=True)
    assert e32.take_status() == 0
    e32.close()
    e16 = engine.Engine("denoiser", W, precision="f16x3")
    exps = e16.activation_exponents()
    assert len(exps) == hip.NUM_ACTIVATIONS and exps[8] >= 10              # stack block 0 conv1 output: ~3e5 x O(10)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                       # the rerun path warns: it must not run
        got = e16.enhance([mix], [ca], [cb], want_mixed=False, taps=True)
    assert e16.take_status() == 0 and e16.precision == "f16x3"
    scale = max(1.0, float(np.abs(ref["logits"]).max()))
    assert np.abs(got["logits"] - ref["logits"]).max() <= 3e-5 * scale
    e16.close()


def test_saturation_is_detected_rerun_in_f32_and_the_exponents_follow(lib_built, weights_denoiser):
    """The backstop: with the exponents forced to zero the same weights must raise the status flag; Engine.enhance
    hands back the f32 matrix-core result of that batch instead of clamped values, raises the exponents from what the
    rerun saw, and the next batch runs in f16x3 again without the flag."""
    W = _scaled_block1(weights_denoiser)
    mix, ca, cb = _clip(7, 0.3)
    e32 = engine.Engine("denoiser", W, precision="f32")
    ref = e32.enhance([mix], [ca], [cb], want_mixed=False, taps=True)
    e32.close()
    e16 = engine.Engine("denoiser", W, precision="f16x3")
    e16.set_activation_exponents([0] * hip.NUM_ACTIVATIONS)
    mix_t, mix_off = e16._dev([mix]); ca_t, ca_off = e16._dev([ca]); cb_t, cb_off = e16._dev([cb])
    e16.enhance_device(mix_t, mix_off, ca_t, ca_off, cb_t, cb_off)
    assert e16.take_status() & hip.STATUS_SATURATED
    assert e16.take_status() == 0                                              # read-and-clear
    with pytest.warns(UserWarning, match="f16 range"):
        got = e16.enhance([mix], [ca], [cb], want_mixed=False, taps=True)
    assert e16.precision == "f16x3"
    assert np.array_equal(got["logits"], ref["logits"])
    assert max(e16.activation_exponents()) >= 10
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        again = e16.enhance([mix], [ca], [cb], want_mixed=False, taps=True)
    assert e16.take_status() == 0
    scale = max(1.0, float(np.abs(ref["logits"]).max()))
    assert np.abs(again["logits"] - ref["logits"]).max() <= 3e-5 * scale
    e16.close()


def test_activation_exponents_stay_inside_the_f16x3_error_envelope(eng, lib_built, weights_denoiser):
    """Scaling a stored tensor by a power of two is exact for every value whose lo half is a normal f16; it only moves
    the threshold below which lo turns subnormal (absolute error 2^-25 of the STORED value, the same order as the
    2^-22 relative error of the normal range -- so the logits move by a few 1e-6, the noise floor of the format).
    With the calibrated exponents, with all of them zero (round 2's storage), shifted by +3 and after a user
    calibration on the clip itself the logits stay within 2e-5 of each other and within 5e-5 of the f32 matrix-core
    path (which is itself the noisier of the two against float64: tools/exponent_accuracy.py); the exponents of
    tensors that share an accumulator stay tied."""
    mix, ca, cb = _clip(11, 0.25)
    e32 = engine.Engine("denoiser", weights_denoiser, precision="f32")
    exact = e32.enhance([mix], [ca], [cb], want_mixed=False, taps=True)["logits"]
    e32.close()
    base = eng.activation_exponents()
    try:
        ref = eng.enhance([mix], [ca], [cb], want_mixed=False, taps=True)["logits"]
        assert eng.take_status() == 0
        assert np.abs(ref - exact).max() <= 5e-5
        assert np.array_equal(ref, eng.enhance([mix], [ca], [cb], want_mixed=False, taps=True)["logits"])
        for exps in ([0] * hip.NUM_ACTIVATIONS, [e + 3 for e in base]):
            eng.set_activation_exponents(exps)
            got = eng.enhance([mix], [ca], [cb], want_mixed=False, taps=True)["logits"]
            assert eng.take_status() == 0
            assert np.abs(got - ref).max() <= 2e-5 and np.abs(got - exact).max() <= 5e-5
        own = eng.calibrate([mix], [ca], [cb])
        amax = eng.activation_amax()
        assert all(0 < a * 2.0 ** -e <= 256.0 for a, e in zip(amax, own))
        got = eng.enhance([mix], [ca], [cb], want_mixed=False, taps=True)["logits"]
        assert np.abs(got - ref).max() <= 2e-5 and np.abs(got - exact).max() <= 5e-5
        odd = list(base)
        odd[1] += 5                                                            # tower block 0 output ...
        eng.set_activation_exponents(odd)
        tied = eng.activation_exponents()
        assert tied[1] == tied[2] == max(base[1] + 5, base[2])                # ... and block 1 conv1 share one
    finally:
        eng.set_activation_exponents(base)


def test_winograd_input_range_raises_the_flag(eng):
    """conv_wino.hip re-splits V = BT d into f16 and that conversion saturates silently: a tensor it reads must stay
    below 65504 / 8.5, so the launch that WRITES such a tensor raises the flag at 7,168 already.  Exponents 6 below
    the calibration store maxima of ~16,000: inside f16, outside what the transform can take -- flagged with the
    Winograd form on, clean (and right) with it off."""
    mix, ca, cb = _clip(11, 0.25)
    base = eng.activation_exponents()
    try:
        ref = eng.enhance([mix], [ca], [cb], want_mixed=False, taps=True)["logits"]
        assert eng.take_status() == 0
        eng.set_activation_exponents([e - 6 for e in base])
        mix_t, mix_off = eng._dev([mix]); ca_t, ca_off = eng._dev([ca]); cb_t, cb_off = eng._dev([cb])
        eng.enhance_device(mix_t, mix_off, ca_t, ca_off, cb_t, cb_off)
        assert eng.take_status() & hip.STATUS_SATURATED
        eng.set_option("winograd", 0)
        got = eng.enhance_device(mix_t, mix_off, ca_t, ca_off, cb_t, cb_off, taps=True)["logits"].cpu().numpy()
        assert eng.take_status() == 0
        assert np.abs(got - ref).max() <= 2e-5
    finally:
        eng.set_option("winograd", 1)
        eng.set_activation_exponents(base)


def test_hip_round_trips_more_tensorflow_written_segments(eng):
    """tests/golden/demo_tf_segments.npz (TensorFlow-written `_mixed.wav` dumps of both models, and the mixture of
    the two triples): nhans_stft_features -> nhans_istft returns TF's samples in the interior; all eight as ONE
    ragged batch."""
    d = dict(np.load(os.path.join(GOLDEN, "demo_tf_segments.npz")))
    sigs = [d["mixed_%d" % i] for i in range(6)] + [d["triple_%d_mixed" % i] for i in range(2)]
    flat, off = eng._dev(sigs)
    lm, ph = eng.stft_features(flat, off)
    nfr = [spec.frames_for_samples(len(s))[1] for s in sigs]
    foff = [0] + list(np.cumsum(nfr))
    out, ooff = eng.istft(lm, ph, foff)
    out = out.cpu().numpy()
    for i, y in enumerate(sigs):
        z = out[ooff[i]:ooff[i + 1]]
        assert z.shape == y.shape
        assert np.abs(z[240:-240] - y[240:-240]).max() < 2e-4, i
