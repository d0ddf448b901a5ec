"""Edges of the C ABI itself (include/nhans_hip.h), called the way a foreign binding would: streams, argument
errors, a clip too short to have a frame in the middle of a batch, two contexts alive at once."""
import ctypes

import numpy as np
import pytest
import torch

import nhans_amd  # noqa: F401
from nhans_amd import apply, engine, hip, spec, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(lib_built, weights_denoiser):
    e = engine.Engine("denoiser", weights_denoiser, precision="f16x3")
    yield e
    e.close()


def _batch(seed, secs):
    mixes = [apply.trim_to_frames(apply.normalise(synth.mixture(seed + i, s))) for i, s in enumerate(secs)]
    ca = [apply.normalise(synth.noise_context(seed + i)) for i in range(len(secs))]
    cb = [apply.normalise(synth.speaker_context(seed + i)) for i in range(len(secs))]
    return mixes, ca, cb


def test_consecutive_calls_on_different_streams_are_ordered(eng):
    """One context, one workspace: a call on stream B issued right behind a call on stream A (no host sync in
    between) must wait for it (the library orders them with an event) -- both results equal their solo runs."""
    b1, b2 = _batch(600, (1.0, 0.3)), _batch(610, (0.5, 0.8, 0.2))
    solo = [eng.enhance(*b, want_mixed=False, taps=True) for b in (b1, b2)]
    dev = [[eng._dev(x) for x in b] for b in (b1, b2)]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(sa):
            r1 = eng.enhance_device(dev[0][0][0], dev[0][0][1], dev[0][1][0], dev[0][1][1], dev[0][2][0], dev[0][2][1], taps=True)
        with torch.cuda.stream(sb):
            r2 = eng.enhance_device(dev[1][0][0], dev[1][0][1], dev[1][1][0], dev[1][1][1], dev[1][2][0], dev[1][2][1], taps=True)
        with torch.cuda.stream(sa):                     # and back, reusing the workspace a third time
            r3 = eng.enhance_device(dev[0][0][0], dev[0][0][1], dev[0][1][0], dev[0][1][1], dev[0][2][0], dev[0][2][1], taps=True)
        torch.cuda.synchronize()
        assert np.array_equal(r1["logits"].cpu().numpy(), solo[0]["logits"])
        assert np.array_equal(r2["logits"].cpu().numpy(), solo[1]["logits"])
        assert np.array_equal(r3["logits"].cpu().numpy(), solo[0]["logits"])
        assert np.array_equal(r2["denoised_wav"].cpu().numpy(), np.concatenate(solo[1]["denoised_wav"]))
    assert eng.take_status() == 0


def test_argument_errors_come_back_as_codes_with_a_message(eng):
    lib = eng.lib
    z = ctypes.c_void_p(None)
    off = hip.i64_array([0, 560])
    assert lib.nhans_enhance_clips(eng.handle, z, off, 1, z, off, z, off, z, z, z, z, z, z, None) == -1     # NHANS_EINVAL
    assert b"null" in lib.nhans_last_error()
    assert lib.nhans_set_option(eng.handle, b"no_such_option", 1) == -1 and b"no_such_option" in lib.nhans_last_error()
    assert lib.nhans_set_option(eng.handle, b"conv_variant", 7) == -1
    assert lib.nhans_set_option(eng.handle, b"frames_per_chunk", 0) == -1
    assert lib.nhans_set_option(None, b"profile", 1) == -1
    # an untrimmed mixture is refused (the reference trims before the STFT, SN/apply.py:157-161), nothing is launched
    wav = torch.zeros(600, device="cuda")
    ctxw = torch.zeros(48000, device="cuda")
    out = torch.zeros(600, device="cuda")
    rc = lib.nhans_enhance_clips(eng.handle, hip.ptr(wav), hip.i64_array([0, 600]), 1, hip.ptr(ctxw), hip.i64_array([0, 48000]),
                                 hip.ptr(ctxw), hip.i64_array([0, 48000]), hip.ptr(out), None, None, None, None, None, None)
    assert rc == -1 and b"whole frame count" in lib.nhans_last_error()
    # a conditioning recording with fewer than 200 frames: NHANS_ESHORT
    short = torch.zeros(16000, device="cuda")
    wav = torch.zeros(560, device="cuda")
    rc = lib.nhans_enhance_clips(eng.handle, hip.ptr(wav), hip.i64_array([0, 560]), 1, hip.ptr(short), hip.i64_array([0, 16000]),
                                 hip.ptr(ctxw), hip.i64_array([0, 48000]), hip.ptr(out), None, None, None, None, None, None)
    assert rc < 0 and b"frames" in lib.nhans_last_error()
    assert eng.take_status() == 0


def test_clip_without_a_frame_inside_a_batch_leaves_its_neighbours_alone(eng):
    """Straight at the ABI (the Python engine refuses such a clip): a 300-sample clip between two real ones has no
    STFT frame; its neighbours must come out exactly as they do without it, its own output stays untouched."""
    mixes, ca, cb = _batch(620, (0.4, 0.3))
    solo = eng.enhance(mixes, ca, cb, want_mixed=False, taps=True)
    stub = np.full(300, 0.25, np.float32)
    mix3 = [mixes[0], stub, mixes[1]]
    ctx_a, ctx_b = [ca[0], ca[0], ca[1]], [cb[0], cb[1], cb[1]]
    mt, mo = eng._dev(mix3)
    at, ao = eng._dev(ctx_a)
    bt, bo = eng._dev(ctx_b)
    out = torch.full((mo[-1],), 7.0, dtype=torch.float32, device="cuda")
    total = sum(spec.frames_for_samples(len(m))[1] for m in (mixes[0], mixes[1]))
    logits = torch.empty((total, spec.BINS), dtype=torch.float32, device="cuda")
    hip.check(eng.lib.nhans_enhance_clips(eng.handle, hip.ptr(mt), hip.i64_array(mo), 3, hip.ptr(at), hip.i64_array(ao),
                                          hip.ptr(bt), hip.i64_array(bo), hip.ptr(out), None, None, None, hip.ptr(logits), None,
                                          eng._stream()))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    assert np.array_equal(out[mo[0]:mo[1]], solo["denoised_wav"][0]) and np.array_equal(out[mo[2]:mo[3]], solo["denoised_wav"][1])
    assert np.all(out[mo[1]:mo[2]] == 7.0)
    assert np.array_equal(logits.cpu().numpy(), solo["logits"])


def test_two_contexts_of_both_models_interleaved(lib_built, weights_denoiser, weights_separator):
    """Contexts are independent (one per model here, alive together, calls interleaved): neither disturbs the other."""
    ed = engine.Engine("denoiser", weights_denoiser, precision="f16x3")
    es = engine.Engine("separator", weights_separator, precision="f16x3")
    try:
        b = _batch(630, (0.6, 0.2))
        rd0 = ed.enhance(*b, want_mixed=False, taps=True)
        rs0 = es.enhance(*b, want_mixed=False, taps=True)
        for _ in range(2):
            rd = ed.enhance(*b, want_mixed=False, taps=True)
            rs = es.enhance(*b, want_mixed=False, taps=True)
            assert np.array_equal(rd["logits"], rd0["logits"]) and np.array_equal(rs["logits"], rs0["logits"])
        assert not np.array_equal(rd0["logits"], rs0["logits"])
    finally:
        ed.close()
        es.close()


def test_istft_output_not_aligned_to_16_bytes(eng):
    """The inverse STFT stores four samples per lane when the clip's output starts on a 16-byte boundary and falls back
    to single stores otherwise (a foreign caller may hand over any float offset): both paths, same bits."""
    mixes, _, _ = _batch(640, (0.7, 0.025, 0.31))
    wav, off = eng._dev(mixes)
    lm, ph = eng.stft_features(wav, off)
    foff = [0]
    for m in mixes:
        foff.append(foff[-1] + spec.frames_for_samples(len(m))[1])
    ref, ooff = eng.istft(lm, ph, foff)
    for shift in (1, 2, 3):
        buf = torch.full((ooff[-1] + 8,), 9.0, dtype=torch.float32, device="cuda")
        view = buf[shift:]                                      # same offsets, base pointer moved by `shift` floats
        hip.check(eng.lib.nhans_istft(eng.handle, hip.ptr(lm), hip.ptr(ph), hip.i64_array(foff), len(mixes),
                                      hip.i64_array(ooff), ctypes.c_void_p(view.data_ptr()), eng._stream()))
        torch.cuda.synchronize()
        assert torch.equal(view[:ooff[-1]], ref), shift
        assert float(buf[:shift].min()) == 9.0 and float(buf[shift + ooff[-1]:].min()) == 9.0     # nothing outside


def test_ten_minute_clip_and_the_per_clip_frame_limit(eng):
    """One long clip (10 minutes = 59,998 frames, 16 chunks of frame windows): finite, unsaturated, and 100 frames from
    its middle bit-equal to the same frames run through the stage-level mask_net from the same features and
    embeddings.  A clip beyond 5,000,000 frames (32-bit offsets inside a clip) is refused before anything is launched."""
    rng = np.random.default_rng(5)
    n = 400 + 160 * 59997
    mix = (0.1 * rng.standard_normal(n)).astype(np.float32)
    ca, cb = apply.normalise(synth.noise_context(650)), apply.normalise(synth.speaker_context(651))
    out = eng.enhance([mix], [ca], [cb], want_mixed=False, taps=True)
    assert out["logits"].shape == (59998, 201) and np.isfinite(out["logits"]).all() and np.isfinite(out["denoised_wav"][0]).all()
    assert eng.take_status() == 0
    lm = torch.from_numpy(out["logmag"][29900:30200].copy()).cuda()
    emb = torch.from_numpy(out["emb"]).cuda()
    lg, _ = eng.mask_net(lm, [0, 300], emb[0:1], emb[1:2])
    assert np.array_equal(lg.cpu().numpy()[100:200], out["logits"][30000:30100])
    off = hip.i64_array([0, 400 + 160 * 5000000])
    rc = eng.lib.nhans_stft_features(eng.handle, hip.ptr(torch.zeros(8, device="cuda")), off, 1, 0,
                                     hip.ptr(torch.zeros(8, device="cuda")), None, None)
    assert rc == -1 and b"5000000" in eng.lib.nhans_last_error()


def test_blob_without_the_tables_two_terms_still_runs(lib_built, weights_denoiser, monkeypatch):
    """A folded blob from before the position table was also emitted as its two terms (`*.tt`, `*.ff`): the Winograd
    kernel is then not eligible (it reads only the two terms) and `direct_conv64` reads the combined table -- same
    logits to the f16x3 noise floor, no error."""
    from nhans_amd import fold
    mix = apply.trim_to_frames(apply.normalise(synth.mixture(5, 0.3)))
    ca, cb = apply.normalise(synth.silent()), apply.normalise(synth.noise_context(5))
    e = engine.Engine("denoiser", weights_denoiser, precision="f16x3")
    ref = e.enhance([mix], [ca], [cb], want_mixed=False, taps=True)["logits"]
    e.close()
    full = fold.fold_arrays

    def without_terms(W, kind, split_f16=True):
        return {k: v for k, v in full(W, kind, split_f16).items() if not (k.endswith(".tt") or k.endswith(".ff"))}

    monkeypatch.setattr(fold, "fold_arrays", without_terms)
    e = engine.Engine("denoiser", weights_denoiser, precision="f16x3")
    e.set_option("profile", 1)
    got = e.enhance([mix], [ca], [cb], want_mixed=False, taps=True)["logits"]
    names = set(e.profile().keys()) if isinstance(e.profile(), dict) else set()
    e.close()
    assert np.abs(got - ref).max() <= 2e-5
    assert not any(n.startswith("conv_wino") for n in names)


def test_blob_of_an_older_packing_version_is_refused(lib_built, weights_denoiser):
    """Advisor finding (round 3): the K order of the packed weights changed while the blob said version 1, so a blob
    folded by an older tree loaded and computed wrong results.  The version now follows the packing conventions
    (fold.BLOB_VERSION); nhans_create refuses any other and says what to do."""
    import struct
    from nhans_amd import fold
    blob = bytearray(fold.fold_weights(weights_denoiser, "denoiser"))
    assert struct.unpack_from("<I", blob, 8)[0] == fold.BLOB_VERSION
    struct.pack_into("<I", blob, 8, 1)
    lib = hip.load()
    buf = (ctypes.c_char * len(blob)).from_buffer(blob)
    handle = ctypes.c_void_p(None)
    rc = lib.nhans_create(hip.KIND_CODE["denoiser"], buf, len(blob), 0, ctypes.byref(handle))
    assert rc == -1 and not handle.value
    msg = lib.nhans_last_error()
    assert b"packing version 1" in msg and b"re-fold" in msg


def _hip_runtime():
    for name in ("libamdhip64.so.7", "libamdhip64.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    pytest.skip("HIP runtime not loadable by soname")


def test_an_error_the_application_left_pending_neither_blocks_nor_is_blamed(eng):
    """Advisor finding (round 3): the entry points refused to run while the runtime's sticky per-thread error held
    something the APPLICATION had left there.  Launches are now checked by their own return code: with an
    out-of-memory error of the caller's pending, a C-ABI call runs, gives the same bits and reports no failure.
    (Everything is allocated beforehand and no torch call sits between: torch itself raises on a pending error.)"""
    rt = _hip_runtime()
    lib = eng.lib
    mix = apply.trim_to_frames(apply.normalise(synth.mixture(3, 0.2)))
    wav = torch.from_numpy(mix).cuda()
    off = hip.i64_array([0, len(mix)])
    n = int(lib.nhans_num_frames(len(mix)))
    lm = [torch.empty((n, spec.BINS), device="cuda") for _ in range(2)]
    ph = [torch.empty((n, spec.BINS), device="cuda") for _ in range(2)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert lib.nhans_stft_features(eng.handle, hip.ptr(wav), off, 1, 0, hip.ptr(lm[0]), hip.ptr(ph[0]), stream) == 0
    p = ctypes.c_void_p(None)
    rc = rt.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 60))
    assert rc != 0                                                    # the application's own failure ...
    pending = rt.hipPeekAtLastError()                                 # ... is now the thread's sticky error (or not, by runtime version)
    rc2 = lib.nhans_stft_features(eng.handle, hip.ptr(wav), off, 1, 0, hip.ptr(lm[1]), hip.ptr(ph[1]), stream)
    rc3 = lib.nhans_debug_launch_probe(1024, None)
    after = rt.hipPeekAtLastError()
    rt.hipGetLastError()                                              # (the application clears its own error)
    assert rc2 == 0 and rc3 == 0, lib.nhans_last_error()
    assert after in (pending, 0)                                      # not turned into something else by the library
    torch.cuda.synchronize()
    assert torch.equal(lm[0], lm[1]) and torch.equal(ph[0], ph[1])
    assert eng.take_status() == 0


def test_a_nan_clip_inside_a_batch_costs_only_itself(eng):
    """Advisor finding (round 3): a batch that raises the saturation flag is rerun in f32 inside a raise-only
    calibration bracket; non-finite maxima used to make the closing call fail and the whole batch was lost.  Two
    things are pinned here: (i) a NaN sample in one clip poisons that clip only -- its neighbours equal the same clips
    run alone, bit for bit, no exception (the ReLUs' v_max drops NaN, so this input does not even raise the flag);
    (ii) a raise-only bracket closed over non-finite maxima reports success and lowers nothing."""
    mixes = [apply.trim_to_frames(apply.normalise(synth.mixture(20 + i, 0.2))) for i in range(3)]
    ca = [apply.normalise(synth.silent()) for _ in range(3)]
    cb = [apply.normalise(synth.noise_context(20 + i)) for i in range(3)]
    alone = [eng.enhance([mixes[i]], [ca[i]], [cb[i]], want_mixed=False)["denoised_wav"][0] for i in (0, 2)]
    exps = eng.activation_exponents()
    bad = mixes[1].copy()
    bad[700] = np.nan
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = eng.enhance([mixes[0], bad, mixes[2]], ca, cb, want_mixed=False)["denoised_wav"]
    assert np.isnan(out[1]).any()
    assert np.array_equal(out[0], alone[0]) and np.array_equal(out[2], alone[1])
    assert eng.activation_exponents() == exps                                          # NaN says nothing about the range
    eng.take_status()
    # (ii) the bracket of an f32 rerun over a batch with +Inf in it (Inf, unlike NaN, survives the ReLUs)
    infmix = mixes[0].copy()
    infmix[900:1400] = np.inf
    for close in (2, 0):
        before = eng.activation_exponents()
        eng.set_option("calibrate", 1)
        eng.set_precision("f32")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            eng.enhance([infmix, mixes[2]], ca[:2], cb[:2], want_mixed=False)
        eng.set_precision("f16x3")
        if close == 2:
            eng.set_option("calibrate", 2)                            # raise-only: non-finite maxima are skipped, not refused
            assert all(a >= b for a, b in zip(eng.activation_exponents(), before))
        else:
            try:
                eng.set_option("calibrate", 0)                        # a calibration proper refuses them ...
            except hip.NhansError as err:
                assert "non-finite" in str(err)
                assert eng.activation_exponents() == before           # ... and changes nothing
    eng.take_status()
    again = eng.enhance([mixes[0]], [ca[0]], [cb[0]], want_mixed=False)["denoised_wav"][0]   # and f16x3 goes on
    assert np.isfinite(again).all()


def test_a_calibration_bracket_that_recorded_nothing_keeps_the_exponents(eng):
    """Advisor finding (round 3): closing a bracket whose pass had failed before its first launch reset all 25
    exponents to 0.  A tensor the pass never wrote keeps its exponent; "calibrate" 3 discards a bracket."""
    exps = eng.activation_exponents()
    assert any(e != 0 for e in exps)                                  # (the built-in calibration of nhans_create)
    eng.set_option("calibrate", 1)
    eng.set_option("calibrate", 0)
    assert eng.activation_exponents() == exps
    eng.set_option("calibrate", 1)
    eng.set_option("calibrate", 3)
    assert eng.activation_exponents() == exps
    with pytest.raises(hip.NhansError):
        eng.set_option("calibrate", 3)                                # no bracket is open


def test_split_k_stress_is_bit_identical_to_the_unsplit_walk(eng):
    """conv_igemm_dma.hip's split-K (one workgroup per (tile, K group), partial tiles through scratch, ONE agent-scope
    release per workgroup, a relaxed ticket, an acquire in the last arriver) against the same launches with option
    split_k = 0 (every workgroup walks its groups itself, no scratch, no ticket) -- the groups and the order of the
    additions depend on the layer only, so the bits must agree, launch after launch:
      * the embedding tower at 10 context images: its 256- and 512-channel convs are 92 and 96 tiles -- the largest
        grids the split rule admits (<= 96) -- 2,000 tower passes back to back = 10,000 split launches;
      * the head's dense layer (K = 13,312 in 32 groups, SN/main.py:237-238) at 700 and 1,500 frames, 150 passes each.
    A lost or stale partial tile (a release that did not cover another wave's stores, a ticket seen before the data)
    shows up as a differing word."""
    g = torch.Generator().manual_seed(5)
    ctx = (torch.randn(10, spec.NOISE_WIN, spec.BINS, generator=g) * 2.0 - 4.0).cuda()
    try:
        eng.set_option("split_k", 0)
        ref_emb = eng.embed(ctx).cpu().numpy()
        eng.set_option("split_k", 1)
        eng.set_option("profile", 1)
        eng.profile_reset()
        first = eng.embed(ctx).cpu().numpy()
        names = eng.profile()
        eng.set_option("profile", 0)
        assert any("grouped" in k for k in names), names        # the grouped / split-K kernel is what ran
        assert np.array_equal(first, ref_emb)
        outs = [eng.embed(ctx) for _ in range(2000)]
        torch.cuda.synchronize()
        ref_t = torch.from_numpy(ref_emb).cuda()
        bad = sum(int(not torch.equal(o, ref_t)) for o in outs)
        assert bad == 0, "%d of 2000 tower passes differ from the unsplit walk" % bad
        del outs
        for nfr in (700, 1500):
            lm = (torch.randn(nfr, spec.BINS, generator=g) * 2.0 - 4.0).cuda()
            ea = torch.randn(1, spec.EMB, generator=g).cuda() * 0.1
            eb = torch.randn(1, spec.EMB, generator=g).cuda() * 0.1
            eng.set_option("split_k", 0)
            ref = eng.mask_net(lm, [0, nfr], ea, eb)[0]
            eng.set_option("split_k", 1)
            bad = 0
            for _ in range(150):
                bad += int(not torch.equal(eng.mask_net(lm, [0, nfr], ea, eb)[0], ref))
            assert bad == 0, "%d of 150 head passes at %d frames differ from the unsplit walk" % (bad, nfr)
    finally:
        eng.set_option("profile", 0)
        eng.set_option("split_k", 1)
    assert eng.take_status() == 0
