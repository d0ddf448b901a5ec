"""Parity of the HIP path (through the C ABI) against the float64 oracle and the committed golden
vectors, on a real MI355X.  Tolerances (BASELINE.json north_star): mask logits <= 1e-4 max-abs,
reconstructed waveform <= 1e-3 RMS; features are compared in the linear-magnitude domain
(|X| within 1e-5 * max|X|) because log(|X|+1e-5) amplifies float32 FFT rounding at silent bins.
"""
import os

import numpy as np
import pytest
import torch

import nhans_amd  # noqa: F401
from nhans_amd import apply, engine, synth
from oracle import nhans_oracle as O
from conftest import GOLDEN, load_case

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4
WAV_RMS_TOL = 1e-3


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


# every test runs in both arithmetic modes of the conv kernels: exact f32 MFMA and split-f16 x3
@pytest.fixture(params=["f32", "f16x3"])
def eng_d(request, _eng_d):
    _eng_d.set_precision(request.param)
    return _eng_d


@pytest.fixture(params=["f32", "f16x3"])
def eng_s(request, _eng_s):
    _eng_s.set_precision(request.param)
    return _eng_s


def exp2_inputs():
    mix = apply.trim_to_frames(apply.normalise(apply.read_wav(os.path.join(GOLDEN, "exp2_noisy.wav"))))
    return mix, apply.normalise(synth.silent()), apply.normalise(synth.noise_context(0))


def test_native_library_is_loaded(eng_d):
    maps = open("/proc/self/maps").read()
    assert "libnhans_hip.so" in maps


def test_stft_features_exp2(eng_d):
    g = load_case("case_exp2")
    mix, _, _ = exp2_inputs()
    lm, ph = eng_d.stft_features(torch.from_numpy(mix).cuda(), [0, len(mix)])
    lm, ph = lm.cpu().numpy().astype(np.float64), ph.cpu().numpy().astype(np.float64)
    assert lm.shape == (308, 201)
    ref = O.stft(mix)
    z = (np.exp(lm) - 1e-5) * np.exp(1j * ph)
    assert np.abs(z - ref).max() <= 1e-5 * np.abs(ref).max()
    zg = (np.exp(g["logmag"].astype(np.float64)) - 1e-5) * np.exp(1j * g["phase"].astype(np.float64))
    assert np.abs(z - zg).max() <= 2e-5 * np.abs(ref).max()
    # log domain: loose bound only (silent bins), tight where there is energy
    loud = np.abs(ref) > 1e-2
    assert np.abs(lm - np.log(np.abs(ref) + 1e-5))[loud].max() < 2e-4


def test_context_features_truncate_to_200_frames_and_short_context_errors(eng_d):
    ca, cb = apply.normalise(synth.silent()), apply.normalise(synth.noise_context(2))
    cat = torch.from_numpy(np.concatenate([ca, cb])).cuda()
    lm, _ = eng_d.stft_features(cat, [0, len(ca), len(ca) + len(cb)], 200, False)
    lm = lm.cpu().numpy().reshape(2, 200, 201)
    assert np.abs(lm[0] - np.log(1e-5)).max() < 1e-5                      # Silent -> ln(1e-5)
    ref, _ = O.logmag_phase(O.stft(cb))
    assert np.abs(np.exp(lm[1]) - np.exp(ref[:200])).max() < 1e-4
    short = torch.zeros(16000).cuda()
    with pytest.raises(Exception, match="frames"):
        eng_d.stft_features(short, [0, 16000], 200, False)


def test_embedding_tower(eng_d, weights_denoiser):
    g = load_case("case_exp2")
    _, ca, cb = exp2_inputs()
    ctx = np.stack([O.context(O.logmag_phase(O.stft(w))[0]) for w in (ca, cb)])
    emb = eng_d.embed(torch.from_numpy(ctx.astype(np.float32)).cuda()).cpu().numpy()
    assert np.abs(emb[0] - g["emb_a"]).max() < 2e-5 and np.abs(emb[1] - g["emb_b"]).max() < 2e-5
    # batch-size / chunk independence, bit for bit
    eng_d.set_option("contexts_per_chunk", 1)
    emb1 = eng_d.embed(torch.from_numpy(ctx.astype(np.float32)).cuda()).cpu().numpy()
    eng_d.set_option("contexts_per_chunk", 64)
    assert np.array_equal(emb, emb1)


def test_block_outputs_and_logits_exp2(eng_d):
    """Same feature tensor into oracle and kernel (the golden log-magnitudes), per-block and logits."""
    g = load_case("case_exp2")
    lm = torch.from_numpy(g["logmag"]).cuda()
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    tf = [int(f) for f in g["tap_frames"]]
    for b in range(9):
        got = torch.cat([eng_d.block_output(lm, [0, 308], ea, eb, f, 1, b) for f in tf]).cpu().numpy()
        ref = g["block%d" % b]
        sub = got.reshape(-1)[::97]
        assert sub.shape == ref.shape
        assert np.abs(sub - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), "block %d" % b
    lg, den = eng_d.mask_net(lm, [0, 308], ea, eb)
    lg, den = lg.cpu().numpy(), den.cpu().numpy()
    assert np.abs(lg - g["logits"]).max() < LOGIT_TOL
    assert np.abs(den - (g["logmag"] + g["logits"])).max() < LOGIT_TOL


def test_end_to_end_waveform_exp2(eng_d):
    g = load_case("case_exp2")
    mix, ca, cb = exp2_inputs()
    out = eng_d.enhance([mix], [ca], [cb], want_mixed=True, taps=True)
    w = out["denoised_wav"][0]
    assert w.dtype == np.float32 and w.shape == (49520,)
    assert np.sqrt(np.mean((w - g["denoised_wav"]) ** 2)) < WAV_RMS_TOL
    assert np.sqrt(np.mean((out["mixed_wav"][0] - g["mixed_wav"]) ** 2)) < WAV_RMS_TOL
    assert np.abs(out["logits"] - g["logits"]).max() < 5 * LOGIT_TOL     # features differ at silent bins
    assert np.abs(out["emb"][1] - g["emb_b"]).max() < 1e-4


def test_ten_second_clip_selected_frames(eng_d):
    g = load_case("case_synth10s")
    lm = torch.from_numpy(g["logmag"]).cuda()
    assert g["logmag"].shape == (998, 201)
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    lg, _ = eng_d.mask_net(lm, [0, 998], ea, eb)
    fr = g["frames"]
    assert np.abs(lg.cpu().numpy()[fr] - g["logits"]).max() < LOGIT_TOL


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_every_conv_kernel_variant(_eng_d, prec, variant):
    """The three conv kernels (register-staged, LDS-DMA, halo + producer/consumer waves) in both
    arithmetic modes against the same golden logits: 308-frame clip, and 998 frames so that tiles
    span many frame-windows and the ragged last tile of every layer is exercised."""
    _eng_d.set_precision(prec)
    _eng_d.set_option("conv_variant", variant)
    try:
        for case, n in (("case_exp2", 308), ("case_synth10s", 998)):
            g = load_case(case)
            lm = torch.from_numpy(g["logmag"]).cuda()
            ea = torch.from_numpy(g["emb_a"][None]).cuda()
            eb = torch.from_numpy(g["emb_b"][None]).cuda()
            lg = _eng_d.mask_net(lm, [0, n], ea, eb)[0].cpu().numpy()
            ref = g["logits"]
            got = lg[g["frames"]] if "frames" in g else lg
            assert np.abs(got - ref).max() < LOGIT_TOL, (case, prec, variant)
    finally:
        _eng_d.set_option("conv_variant", -1)


def test_mask_net_is_bitwise_reproducible(eng_d):
    """Race screen for the LDS-DMA pipelines (counted vmcnt + raw barriers): a stage read before its
    DMA landed, or overwritten while still being read, shows up as run-to-run differences long before
    it breaks a tolerance.  998 frames = thousands of workgroups per layer, 4 runs, all bits equal."""
    g = load_case("case_synth10s")
    lm = torch.from_numpy(g["logmag"]).cuda()
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    first = eng_d.mask_net(lm, [0, 998], ea, eb)[0].clone()
    for _ in range(3):
        again = eng_d.mask_net(lm, [0, 998], ea, eb)[0]
        assert torch.equal(first, again)


@pytest.mark.parametrize("option, values", [("consumer_interleave", (0, 1)), ("consumer_interleave", (2, 1)), ("epilogue_wide", (0, 1))])
def test_speed_knobs_do_not_change_a_bit(_eng_d, option, values):
    """The A/B options of the C ABI select other instruction orders of the same arithmetic: a 10 s clip (thousands of
    tiles per layer) must give identical logits whichever way they are set."""
    _eng_d.set_precision("f16x3")
    g = load_case("case_synth10s")
    lm = torch.from_numpy(g["logmag"]).cuda()
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    ref = _eng_d.mask_net(lm, [0, 998], ea, eb)[0].clone()
    try:
        _eng_d.set_option(option, values[0])
        got = _eng_d.mask_net(lm, [0, 998], ea, eb)[0]
        assert torch.equal(got, ref), option
    finally:
        _eng_d.set_option(option, values[1])


def test_clip_boundary_inside_the_partial_last_tile(_eng_d):
    """Regression (found by tools/fuzz_batches.py): a batch whose LAST tile is partial and holds the end of one clip
    and the one-frame clip after it.  The epilogue decides "one clip per tile -> load the conditioning bias once"
    from the first and last row of the tile; rows past the end used to look like the first row, so the last clip's
    frame got its neighbour's bias (logits off by up to 7e-2).  Every clip must equal the same clip run alone."""
    _eng_d.set_precision("f16x3")
    secs = (0.1, 0.025, 0.33, 0.025)                        # 8 + 1 + 31 + 1 frames; different conditioning per clip
    mixes = [apply.trim_to_frames(apply.normalise(synth.mixture(70 + i, d))) for i, d in enumerate(secs)]
    ca = [apply.normalise(synth.noise_context(80 + i)) for i in range(len(secs))]
    cb = [apply.normalise(synth.speaker_context(90 + i)) for i in range(len(secs))]
    try:
        alone = [_eng_d.enhance([mixes[i]], [ca[i]], [cb[i]], want_mixed=False, taps=True)["logits"] for i in range(len(secs))]
        for ids in ([0, 1], [2, 3], [0, 1, 2, 3], [3, 2, 1, 0]):
            for fpc in (3776, 9, 4):
                _eng_d.set_option("frames_per_chunk", fpc)
                got = _eng_d.enhance([mixes[i] for i in ids], [ca[i] for i in ids], [cb[i] for i in ids],
                                     want_mixed=False, taps=True)["logits"]
                f0 = 0
                for i in ids:
                    assert np.array_equal(got[f0:f0 + len(alone[i])], alone[i]), (ids, fpc, i)
                    f0 += len(alone[i])
    finally:
        _eng_d.set_option("frames_per_chunk", 3776)


def test_separator_model(eng_s):
    g = load_case("case_separator")
    mix = apply.trim_to_frames(apply.normalise(synth.mixture(3, 2.0)))
    ca = apply.normalise(synth.speaker_context(3, low=False))
    cb = apply.normalise(synth.speaker_context(3, low=True))
    out = eng_s.enhance([mix], [ca], [cb], want_mixed=False, taps=True)
    assert np.abs(out["emb"][0] - g["emb_a"]).max() < 1e-4 and np.abs(out["emb"][1] - g["emb_b"]).max() < 1e-4
    lm = torch.from_numpy(g["logmag"]).cuda()
    lg, _ = eng_s.mask_net(lm, [0, lm.shape[0]], torch.from_numpy(g["emb_a"][None]).cuda(),
                           torch.from_numpy(g["emb_b"][None]).cuda())
    assert np.abs(lg.cpu().numpy()[g["frames"]] - g["logits"]).max() < LOGIT_TOL
    i = list(g["tap_frames"])[0]
    b7 = eng_s.block_output(lm, [0, lm.shape[0]], torch.from_numpy(g["emb_a"][None]).cuda(),
                            torch.from_numpy(g["emb_b"][None]).cuda(), int(i), 1, 7).cpu().numpy()
    assert np.abs(b7.reshape(-1)[::97] - g["block7"]).max() < 1e-3


def test_ragged_batch_edges_and_shard_invariance(eng_d):
    """1-, 2- and 99-frame clips in one batch: window zero padding at both clip ends, no leakage
    between neighbouring clips, and batch == one-by-one, bit for bit."""
    g = load_case("case_ragged")
    mixes, cas, cbs = [], [], []
    for i, n in enumerate(g["lens"]):
        mixes.append(apply.trim_to_frames(apply.normalise(synth.mixture(10 + i, int(n) / 16000.0))))
        cas.append(apply.normalise(synth.silent()))
        cbs.append(apply.normalise(synth.noise_context(10 + i)))
    assert [len(m) for m in mixes] == [400, 560, 16080]
    batch = eng_d.enhance(mixes, cas, cbs, want_mixed=False, taps=True)
    foff = [0, 1, 3, 102]
    assert batch["logits"].shape == (102, 201)
    for i in range(3):
        lg = batch["logits"][foff[i]:foff[i + 1]]
        # logits given the kernel's own features: allow the silent-bin feature difference
        assert np.abs(lg - g["logits_%d" % i]).max() < 5 * LOGIT_TOL, i
        assert np.sqrt(np.mean((batch["denoised_wav"][i] - g["denoised_wav_%d" % i]) ** 2)) < WAV_RMS_TOL
        single = eng_d.enhance([mixes[i]], [cas[i]], [cbs[i]], want_mixed=False, taps=True)
        assert np.array_equal(single["logits"], lg)
        assert np.array_equal(single["denoised_wav"][0], batch["denoised_wav"][i])
    # fake-rank mode: 2 logical shards on one device == the whole batch
    from nhans_amd import dist as nd
    parts = []
    for r in range(2):
        lo, hi = nd.shard_bounds(3, 2, r)
        parts += eng_d.enhance(mixes[lo:hi], cas[lo:hi], cbs[lo:hi], want_mixed=False)["denoised_wav"]
    assert all(np.array_equal(a, b) for a, b in zip(parts, batch["denoised_wav"]))


def test_chunking_is_invisible(eng_d):
    mix = apply.trim_to_frames(apply.normalise(synth.mixture(21, 1.5)))
    ca, cb = apply.normalise(synth.silent()), apply.normalise(synth.noise_context(21))
    a = eng_d.enhance([mix], [ca], [cb], want_mixed=False, taps=True)
    eng_d.set_option("frames_per_chunk", 37)
    b = eng_d.enhance([mix], [ca], [cb], want_mixed=False, taps=True)
    eng_d.set_option("frames_per_chunk", 3776)
    assert np.array_equal(a["logits"], b["logits"]) and np.array_equal(a["denoised_wav"][0], b["denoised_wav"][0])


def test_full_size_properties_10s(eng_d):
    """BASELINE size (10 s, 998 frames): size-independent properties instead of a full oracle run."""
    mix = apply.trim_to_frames(apply.normalise(synth.mixture(0, 10.0)))
    ca, cb = apply.normalise(synth.silent()), apply.normalise(synth.noise_context(0))
    a = eng_d.enhance([mix], [ca], [cb], want_mixed=True, taps=True)
    b = eng_d.enhance([mix], [ca], [cb], want_mixed=True, taps=True)
    assert a["logits"].shape == (998, 201) and len(a["denoised_wav"][0]) == 159920
    assert np.array_equal(a["denoised_wav"][0], b["denoised_wav"][0])            # deterministic
    # STFT -> iSTFT of the unmodified spectrum is the identity in the interior (up to the 1e-5 floor)
    rt = a["mixed_wav"][0]
    assert np.abs(rt[240:-240] - mix[240:-240]).max() < 2e-4
    assert abs(rt[0]) < 1e-6
    # denoised = mixed_central + out, frame by frame
    g = load_case("case_synth10s")
    assert np.abs(a["logits"][g["frames"]] - g["logits"]).max() < 5 * LOGIT_TOL
    assert np.isfinite(a["denoised_wav"][0]).all()


def test_istft_matches_oracle_and_is_linear_in_magnitude(eng_d):
    rng = np.random.default_rng(5)
    T = 57
    lm = rng.normal(-2, 1.5, (T, 201)).astype(np.float32)
    ph = rng.uniform(-np.pi, np.pi, (T, 201)).astype(np.float32)
    w, off = eng_d.istft(torch.from_numpy(lm).cuda(), torch.from_numpy(ph).cuda(), [0, 20, 57])
    w = w.cpu().numpy()
    assert off == [0, 19 * 160 + 400, 19 * 160 + 400 + 36 * 160 + 400]
    r0 = O.recover_samples(lm[:20].astype(np.float64), ph[:20].astype(np.float64))
    r1 = O.recover_samples(lm[20:].astype(np.float64), ph[20:].astype(np.float64))
    assert np.abs(w[:off[1]] - r0).max() < 2e-5 and np.abs(w[off[1]:] - r1).max() < 2e-5
    w2, _ = eng_d.istft(torch.from_numpy(lm + np.float32(np.log(2.0))).cuda(), torch.from_numpy(ph).cuda(), [0, 20, 57])
    assert np.abs(w2.cpu().numpy() - 2 * w).max() < 2e-5                      # exp(lm + ln 2) = 2 exp(lm)


def test_apply_entry_points_write_reference_files(eng_d, eng_s, tmp_path, capsys):
    from scipy.io import wavfile
    apply.set_engine("denoiser", eng_d)
    apply.set_engine("separator", eng_s)
    neg = str(tmp_path / "neg.wav")
    wavfile.write(neg, 16000, synth.noise_context(0)[:16000])                 # 1 s: short-context policy
    out = str(tmp_path / "exp2_denoised.wav")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        apply.apply_denoiser(os.path.join(GOLDEN, "exp2_noisy.wav"), neg, out)
    finally:
        os.chdir(cwd)
    rate, w = wavfile.read(out)
    assert rate == 16000 and w.dtype == np.float32 and len(w) == 49520      # == shipped exp2_denoised.wav geometry
    for side in ("mixed_processed.wav", "removed.wav", "compensated.wav"):
        r, s = wavfile.read(str(tmp_path / ("exp2_" + side)))
        assert r == 16000 and len(s) == 49520 and s.dtype == np.float32
    _, mixed = wavfile.read(str(tmp_path / "exp2_mixed_processed.wav"))
    _, removed = wavfile.read(str(tmp_path / "exp2_removed.wav"))
    _, comp = wavfile.read(str(tmp_path / "exp2_compensated.wav"))
    assert np.array_equal(removed, mixed - w) and np.array_equal(comp, w)   # compensate = 0
    assert "---------------------------" in capsys.readouterr().out         # prints snr_est like the reference
    # separator + CLI
    mixp, posp, negp = [str(tmp_path / n) for n in ("m.wav", "p.wav", "n.wav")]
    wavfile.write(mixp, 16000, synth.mixture(3, 1.0))
    wavfile.write(posp, 16000, synth.speaker_context(3, low=False))
    wavfile.write(negp, 16000, synth.speaker_context(3, low=True))
    sep = str(tmp_path / "sep_denoised.wav")
    apply.main_separator(["--input", mixp, "--pos", posp, "--neg", negp, "--output", sep, "--weights", "synthetic"])
    r, s = wavfile.read(sep)
    assert r == 16000 and s.dtype == np.float32 and len(s) == 15920
    assert os.path.exists(str(tmp_path / "sep_mixed_processed.wav"))


def test_demo_and_eval_mode(_eng_d, _eng_s, weights_denoiser, tmp_path):
    """apply_demo / apply_demo_separator / evaluate_utterance: contexts = first 200 frames of the
    conditioning signals, network on the frames after them (SN/apply.py:212-336, SN/reader.py:398-420,
    SN/main.py:264-353)."""
    from scipy.io import wavfile
    from nhans_amd import mixing
    _eng_d.set_precision("f16x3")
    _eng_s.set_precision("f16x3")
    apply.set_engine("denoiser", _eng_d)
    apply.set_engine("separator", _eng_s)
    paths = {}
    for name, sig in (("speech", synth.mixture(30, 2.6)), ("pos", synth.noise_context(30, 1.0)),
                      ("neg", synth.speaker_context(30, 3.0)), ("spk", synth.speaker_context(31, 2.6, low=False))):
        paths[name] = str(tmp_path / (name + ".wav"))
        wavfile.write(paths[name], 16000, sig)
    out = str(tmp_path / "clip_output_demo.wav")
    apply.apply_demo(paths["speech"], paths["pos"], paths["neg"], out)
    r, den = wavfile.read(out)
    r2, mix = wavfile.read(str(tmp_path / "clip_mixed_demo.wav"))          # save_to[:-15] + 'mixed_demo.wav'
    T = 1 + (len(apply.trim_to_frames(np.zeros(len(synth.mixture(30, 2.6)), np.float32))) - 400) // 160
    assert r == r2 == 16000 and den.dtype == np.float32 and len(den) == len(mix) == (T - 200 - 1) * 160 + 400
    target, pos_sig, neg_sig, mixed = O.demo_signals(*(apply.read_wav(paths[k]) for k in ("speech", "pos", "neg")))
    ref = O.enhance_after_context(mixed, pos_sig, neg_sig, weights_denoiser, "denoiser", frames=[0, 1, 30, T - 201])
    assert np.sqrt(np.mean((mix - ref["mixed_wav"]) ** 2)) < WAV_RMS_TOL
    got = apply._enhance_after_context(_eng_d, mixed, pos_sig, neg_sig)
    assert np.abs(got[3][[0, 1, 30, T - 201]] - ref["logits"][[0, 1, 30, T - 201]]).max() < 5 * LOGIT_TOL
    assert np.array_equal(got[0], den)
    # frame `200` is windowed with zero rows before it, not with frames 183..199
    full = _eng_d.enhance([mixed], [pos_sig], [neg_sig], want_mixed=False, taps=True)
    assert np.abs(full["logits"][200] - got[3][0]).max() > 1e-3
    assert np.abs(full["logits"][200 + 30] - got[3][30]).max() < 1e-6          # interior frames agree
    # separator demo
    sout = str(tmp_path / "sep_output_demo.wav")
    apply.apply_demo_separator(paths["spk"], paths["neg"], sout)
    rs, sden = wavfile.read(sout)
    assert rs == 16000 and np.isfinite(sden).all() and os.path.exists(str(tmp_path / "sep_mixed_demo.wav"))
    # evaluation reader: five wavs with the reference's naming, loss finite
    loss = apply.evaluate_utterance(paths["speech"], paths["pos"], paths["neg"], str(tmp_path / "wav_dump"), "nhans", 5000)
    sp, sn = mixing.eval_snrs(paths["speech"])
    names = sorted(os.listdir(str(tmp_path / "wav_dump")))
    assert names == sorted("nhans_5000_speech_pos_neg_%d_%d_%s.wav" % (sp, sn, k)
                           for k in ("mixed", "denoised", "target", "posNoise", "negNoise"))
    assert np.isfinite(loss) and loss > 0
    _, tw = wavfile.read(str(tmp_path / "wav_dump" / ("nhans_5000_speech_pos_neg_%d_%d_target.wav" % (sp, sn))))
    tgt = mixing.combine_signals(apply.read_wav, paths["speech"], paths["pos"], paths["neg"], snrs=(sp, sn))[0]
    lt, pt = O.logmag_phase(O.stft(tgt))
    assert np.sqrt(np.mean((tw - O.recover_samples(lt[200:], pt[200:])) ** 2)) < WAV_RMS_TOL


@pytest.mark.parametrize("case, n", [("case_exp2", 308), ("case_synth10s", 998)])
def test_winograd_convs_against_golden_logits(_eng_d, case, n):
    """Option winograd (default 1): the stride-1 4x4 convs of the stack run as 1-D Winograd F(5,4) along W
    (conv_wino.hip, 2.5 x fewer MFMAs, 128 tile-pixels per transformed-weight fragment); 0 = the direct halo kernel
    for every conv.  Same golden logits, same bar, on identical features, either way; and the profile shows that the
    Winograd kernel really ran."""
    _eng_d.set_precision("f16x3")
    g = load_case(case)
    lm = torch.from_numpy(g["logmag"]).cuda()
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    _eng_d.set_option("winograd", 0)
    direct = _eng_d.mask_net(lm, [0, n], ea, eb)[0].cpu().numpy()
    try:
        _eng_d.set_option("winograd", 1)
        _eng_d.set_option("profile", 1)
        _eng_d.profile_reset()
        lg = _eng_d.mask_net(lm, [0, n], ea, eb)[0].cpu().numpy()
        again = _eng_d.mask_net(lm, [0, n], ea, eb)[0].cpu().numpy()
        calls = {k: v["calls"] for k, v in _eng_d.profile().items()}
    finally:
        _eng_d.set_option("profile", 0)
        _eng_d.set_option("winograd", 1)
    assert calls.get("conv_wino<128>", 0) >= 10, calls
    got = lg[g["frames"]] if "frames" in g else lg
    print("winograd vs golden %.3e, vs direct %.3e" % (np.abs(got - g["logits"]).max(), np.abs(lg - direct).max()))
    assert np.abs(got - g["logits"]).max() < LOGIT_TOL
    assert np.array_equal(lg, again)                    # bitwise reproducible
    assert _eng_d.take_status() == 0


def test_f32_stored_winograd_tensors(_eng_d):
    """Option winograd_f32_tensors (default 1): the five stack tensors that only Winograd launches read -- conv1's output
    of resblock1_1 / 1_2 / 2_2 and the block outputs of resblock1_1 / 2_1 -- are stored f32 NHWC instead of split NHWC
    (nhans_api.hip: stored_f32); 0 = every tensor split.  Same golden logits at the same bar either way, the block
    outputs of the debug entry point (which has to know each tensor's layout) agree between the two to the split
    format's 22 bits.  (The refusal of a wrong-layout reader: the next test.)"""
    _eng_d.set_precision("f16x3")
    g = load_case("case_exp2")
    lm = torch.from_numpy(g["logmag"]).cuda()
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    out = {}
    try:
        for v in (0, 1):
            _eng_d.set_option("winograd_f32_tensors", v)
            lg = _eng_d.mask_net(lm, [0, 308], ea, eb)[0].cpu().numpy()
            blocks = [_eng_d.block_output(lm, [0, 308], ea, eb, 100, 2, b).cpu().numpy() for b in range(4)]
            assert _eng_d.take_status() == 0
            out[v] = (lg, blocks)
    finally:
        _eng_d.set_option("winograd_f32_tensors", 1)
    for v in (0, 1):
        assert np.abs(out[v][0] - g["logits"]).max() < LOGIT_TOL
    print("f32-stored vs split: logits %.3e" % np.abs(out[0][0] - out[1][0]).max())
    assert np.abs(out[0][0] - out[1][0]).max() < 2e-5
    for b in range(4):
        a0, a1 = out[0][1][b], out[1][1][b]
        assert a0.shape == a1.shape and np.abs(a0 - a1).max() <= 4e-6 * max(1.0, np.abs(a0).max()), b


def test_an_f32_stored_tensor_without_a_winograd_reader_is_refused_not_misread(_eng_d):
    """The layouts follow from what conv_wino_eligible() says about each launch's own arguments (nhans_api.hip:
    run_stack_chunk plans, then launches).  Should plan and kernel ever disagree, the reader of an f32-stored tensor must
    refuse -- a direct kernel would stage f32 words as split halves.  winograd_f32_tensors = 2 is the test value that
    forces the disagreement (f32 storage whatever the readers are); with the Winograd form off every reader is a direct
    kernel: the call comes back NHANS_EHIP naming the refusal, and the context works again afterwards."""
    from nhans_amd import hip
    _eng_d.set_precision("f16x3")
    g = load_case("case_exp2")
    lm = torch.from_numpy(g["logmag"]).cuda()
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    try:
        _eng_d.set_option("winograd", 0)
        _eng_d.set_option("winograd_f32_tensors", 2)
        with pytest.raises(hip.NhansError) as e:
            _eng_d.mask_net(lm, [0, 308], ea, eb)
        assert "f32-stored input without a Winograd form" in str(e.value), str(e.value)
    finally:
        _eng_d.set_option("winograd_f32_tensors", 1)
        _eng_d.set_option("winograd", 1)
    torch.cuda.synchronize()
    _eng_d.take_status()
    lg = _eng_d.mask_net(lm, [0, 308], ea, eb)[0].cpu().numpy()
    assert np.abs(lg - g["logits"]).max() < LOGIT_TOL


def test_a_split_residual_with_an_f32_output_is_refused_by_the_winograd_launcher(_eng_d):
    """conv_wino's epilogue gives a thread channels {4c..4c+3, 32+4c..} for an f32 output and fetches the residual as
    two 16-byte pieces of those channels -- a split-NHWC residual does not hold them contiguously (round-5 advisor: the
    pair was instantiated and silently added the wrong channels).  No plan produces it; winograd_f32_tensors = 3 is the
    test value that does (only resblock1_2's output f32: its conv2 then reads a split residual and writes f32): the
    launcher refuses, the call comes back NHANS_EHIP naming it, nothing later of the pass is launched, and the context
    works again afterwards."""
    from nhans_amd import hip
    _eng_d.set_precision("f16x3")
    g = load_case("case_exp2")
    lm = torch.from_numpy(g["logmag"]).cuda()
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    try:
        _eng_d.set_option("winograd_f32_tensors", 3)
        _eng_d.set_option("profile", 1)
        _eng_d.profile_reset()
        with pytest.raises(hip.NhansError) as e:
            _eng_d.block_output(lm, [0, 308], ea, eb, 100, 2, 1)
        assert "split residual with an f32-stored output" in str(e.value), str(e.value)
        with pytest.raises(hip.NhansError):
            _eng_d.mask_net(lm, [0, 308], ea, eb)
        calls = {k: v["calls"] for k, v in _eng_d.profile().items()}
        # the refusal is block 1's conv2: blocks 2 .. 7 and the head (halo / pointwise / grouped kernels) never ran
        assert not any(k.startswith("conv_igemm_halo_pw") or "grouped" in k for k in calls), calls
    finally:
        _eng_d.set_option("profile", 0)
        _eng_d.set_option("winograd_f32_tensors", 1)
    torch.cuda.synchronize()
    _eng_d.take_status()
    lg = _eng_d.mask_net(lm, [0, 308], ea, eb)[0].cpu().numpy()
    assert np.abs(lg - g["logits"]).max() < LOGIT_TOL


def test_debug_block_output_sees_the_production_layouts(_eng_d):
    """nhans_debug_block_output(block) launches blocks 0 .. block only, but plans the WHOLE stack (round-5 advisor: with a
    plan cut at `block` the requested block's conv2 always wrote split NHWC, never the f32 store production uses for the
    outputs of resblock1_1 / 2_1).  The block outputs agree with the golden taps whichever layout they are stored in."""
    _eng_d.set_precision("f16x3")
    g = load_case("case_exp2")
    lm = torch.from_numpy(g["logmag"]).cuda()
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    tf = [int(f) for f in g["tap_frames"]]
    for b in (0, 2):                                        # f32-stored in production
        got = torch.cat([_eng_d.block_output(lm, [0, 308], ea, eb, f, 1, b) for f in tf]).cpu().numpy()
        ref = g["block%d" % b]
        assert np.abs(got.reshape(-1)[::97] - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), b


@pytest.mark.parametrize("wino", [1, 0])
@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_windows_of_a_clip_that_began_before_the_chunk(_eng_d, prec, wino):
    """The first layer and the 1 -> 64 residual of resblock1_1 read their 35-row windows out of the spectrogram itself
    (WinRows, SN/apply.py:170-186,378), rows counted from the chunk's first frame: for a clip that straddles a chunk
    boundary the first frames of the next chunk look BACK across it (negative row indices that are valid), the last frames of
    a clip see zero rows ahead.  Three clips (40 / 3 / 57 frames) in chunks of 16 frame windows against the same clips in
    one chunk and run alone, bit for bit, in every kernel family that has such a reader: the Winograd epilogue (f16x3),
    the direct kernels' sweep (f16x3 with winograd = 0) and the f32 kernels."""
    _eng_d.set_precision(prec)
    _eng_d.set_option("winograd", wino)
    g = torch.Generator().manual_seed(11)
    nfr = [40, 3, 57]
    lm = (torch.randn(sum(nfr), 201, generator=g) * 2.0 - 4.0).cuda()
    ea = (torch.randn(3, 512, generator=g) * 0.1).cuda()
    eb = (torch.randn(3, 512, generator=g) * 0.1).cuda()
    foff = [0, 40, 43, 100]
    try:
        _eng_d.set_option("frames_per_chunk", 3776)
        whole = _eng_d.mask_net(lm, foff, ea, eb)[0].cpu().numpy()
        _eng_d.set_option("frames_per_chunk", 16)
        chunked = _eng_d.mask_net(lm, foff, ea, eb)[0].cpu().numpy()
        assert np.array_equal(whole, chunked)
        for i in range(3):
            solo = _eng_d.mask_net(lm[foff[i]:foff[i + 1]].contiguous(), [0, nfr[i]], ea[i:i + 1], eb[i:i + 1])[0].cpu().numpy()
            assert np.array_equal(solo, whole[foff[i]:foff[i + 1]]), i
    finally:
        _eng_d.set_option("frames_per_chunk", 3776)
        _eng_d.set_option("winograd", 1)
    assert _eng_d.take_status() == 0


def test_transform_conv_stream_kernel_is_bit_identical_to_the_generic_kernel(_eng_d):
    """conv_1x1_stream.hip runs resblock2_1's stand-alone 1x1 strided `_transform` conv (SN/main.py:176-181) as a stream -- a wave
    per 32 output pixels, weights in registers, results straight from the accumulators -- with the generic kernel's
    arithmetic: same products in the same order.  Option stream_1x1 = 0 puts the launch back on conv_igemm_dma.hip: the
    block outputs from resblock2_1 on and the logits must agree bit for bit, on a frame count that leaves a partial
    32-pixel group (7 frames x 1,818 pixels) and on two chunks; and the profile shows which kernel ran."""
    _eng_d.set_precision("f16x3")
    g = load_case("case_exp2")
    lm = torch.from_numpy(g["logmag"]).cuda()
    ea = torch.from_numpy(g["emb_a"][None]).cuda()
    eb = torch.from_numpy(g["emb_b"][None]).cuda()
    out = {}
    try:
        for v in (0, 1):
            _eng_d.set_option("stream_1x1", v)
            _eng_d.set_option("profile", 1)
            _eng_d.profile_reset()
            blk = _eng_d.block_output(lm, [0, 308], ea, eb, 100, 7, 2).cpu().numpy()
            _eng_d.set_option("frames_per_chunk", 200)
            lg = _eng_d.mask_net(lm, [0, 308], ea, eb)[0].cpu().numpy()
            _eng_d.set_option("frames_per_chunk", 3776)
            names = set(_eng_d.profile())
            assert ("conv_1x1_stream" in names) == bool(v), names
            out[v] = (blk, lg)
    finally:
        _eng_d.set_option("profile", 0)
        _eng_d.set_option("stream_1x1", 1)
        _eng_d.set_option("frames_per_chunk", 3776)
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    assert np.abs(out[1][1] - g["logits"]).max() < LOGIT_TOL
    assert _eng_d.take_status() == 0
