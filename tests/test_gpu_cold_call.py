"""configs[1] as a user meets it (SN/apply.py:478-527, setup.py:44-50): `nhans_denoiser` on ONE file in a fresh process.
The command line then runs torch-free (lite.LiteEngine: device memory through the HIP runtime, hiprt.py) on the cached
folded blob (blobcache.py) with the cached activation exponents (nhans_create_ex: no calibration pass).  Checked here:
  * LiteEngine == engine.Engine bit for bit (same library, same entry point), in one process;
  * a fresh process (forked from conftest's fork server: numpy only, no torch) runs the CLI without importing torch, the
    second call hits the cache, skips folding and calibration, and writes the same bytes as the first and as the full
    engine; a stale / truncated cache entry is rebuilt, --no-cache bypasses it.
Wall-clock numbers of the cold call: tools/cold_call.py (profiles/r06)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SECS = 2.0


def _write_inputs(d):
    from scipy.io import wavfile
    import nhans_amd  # noqa: F401
    from nhans_amd import synth
    wavfile.write(os.path.join(d, "in.wav"), 16000, synth.mixture(71, SECS))
    wavfile.write(os.path.join(d, "neg.wav"), 16000, synth.noise_context(71))


def _cli_worker(d, tag, extra, q, prog="denoiser"):
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        os.environ["NHANS_CACHE_DIR"] = os.path.join(d, "cache")
        had_torch = "torch" in sys.modules
        import nhans_amd  # noqa: F401
        from nhans_amd import apply
        import io
        import json
        err, old = io.StringIO(), sys.stderr
        sys.stderr = err
        try:
            if prog == "separator":
                apply.main_separator(["--input", os.path.join(d, "in.wav"), "--neg", os.path.join(d, "neg.wav"), "--pos", os.path.join(d, "pos.wav"),
                                      "--output", os.path.join(d, tag + ".wav"), "--weights", "synthetic", "--timing"] + list(extra))
            else:
                apply.main(["--input", os.path.join(d, "in.wav"), "--neg", os.path.join(d, "neg.wav"), "--pos", os.path.join(d, "Silent.wav"),
                            "--output", os.path.join(d, tag + ".wav"), "--weights", "synthetic", "--timing"] + list(extra))
        finally:
            sys.stderr = old
        timing = None
        for line in err.getvalue().splitlines():
            if line.startswith("nhans timing: "):
                timing = json.loads(line[len("nhans timing: "):])
        q.put((tag, had_torch, "torch" in sys.modules, timing, None))
    except Exception as e:
        import traceback
        q.put((tag, None, None, None, traceback.format_exc() + repr(e)))


def _run_cli(d, tag, extra=(), prog="denoiser"):
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    p = ctx.Process(target=_cli_worker, args=(d, tag, extra, q, prog))
    p.start()
    try:
        res = q.get(timeout=600)
    finally:
        p.join(timeout=120)
        if p.is_alive():
            p.kill()
    assert res[4] is None, res[4]
    return res


def test_lite_engine_equals_the_full_engine_bit_for_bit(lib_built, weights_denoiser):
    import nhans_amd  # noqa: F401
    from nhans_amd import apply, engine, lite, synth
    mixes = [apply.trim_to_frames(apply.normalise(synth.mixture(80 + i, s))) for i, s in enumerate((1.0, 0.3, 2.5))]
    ca = [apply.normalise(synth.silent()) for _ in mixes]
    cb = [apply.normalise(synth.noise_context(80 + i)) for i in range(len(mixes))]
    full = engine.Engine("denoiser", weights_denoiser, precision="f16x3")
    ref = full.enhance(mixes, ca, cb, want_mixed=True)
    exps = full.activation_exponents()
    full.close()
    le = lite.LiteEngine("denoiser", weights_denoiser)
    try:
        assert le.activation_exponents() == exps                     # (same built-in calibration)
        got = le.enhance(mixes, ca, cb, want_mixed=True)
        blob = le.blob
    finally:
        le.close()
    # and created from the blob + exponents, no calibration pass (nhans_create_ex)
    le2 = lite.LiteEngine("denoiser", blob=blob, exponents=exps)
    try:
        got2 = le2.enhance(mixes, ca, cb, want_mixed=True)
        assert le2.activation_exponents() == exps
    finally:
        le2.close()
    for i in range(len(mixes)):
        for k in ("denoised_wav", "mixed_wav"):
            assert np.array_equal(got[k][i], ref[k][i]), (k, i)
            assert np.array_equal(got2[k][i], ref[k][i]), (k, i)


def test_fresh_process_cli_is_torch_free_and_uses_the_cache(lib_built, tmp_path):
    from scipy.io import wavfile
    d = str(tmp_path)
    _write_inputs(d)
    first = _run_cli(d, "first")                                     # cache miss: folds, calibrates, stores
    second = _run_cli(d, "second")                                   # cache hit
    nocache = _run_cli(d, "nocache", ["--no-cache"])
    for tag, had, has, timing, _ in (first, second, nocache):
        assert had is False and has is False, (tag, "torch was imported by the single-file command line")
        assert timing and timing["engine"] == "LiteEngine" and timing["torch_imported"] is False, timing
    assert first[3]["cache"] == "miss" and "fold_s" in first[3]
    assert second[3]["cache"] == "hit" and "fold_s" not in second[3]
    assert nocache[3]["cache"] == "bypassed" and "fold_s" in nocache[3]
    print("cold CLI on a %.0f s clip: miss %.2f s (fold %.2f, create %.2f), hit %.2f s (cache read %.2f, create %.2f, enhance %.2f)"
          % (SECS, first[3]["run_cli_s"], first[3]["fold_s"], first[3]["create_s"], second[3]["run_cli_s"],
             second[3]["cache_read_s"], second[3]["create_s"], second[3]["enhance_s"]))
    files = sorted(f for f in os.listdir(d) if f.startswith("first"))
    assert len(files) == 4, files                                    # denoised + mixed_processed / removed / compensated
    for f in files:
        a = open(os.path.join(d, f), "rb").read()
        for tag in ("second", "nocache"):
            assert a == open(os.path.join(d, f.replace("first", tag)), "rb").read(), (f, tag)
    # the same bytes as the full torch engine in THIS process
    import nhans_amd  # noqa: F401
    from nhans_amd import apply, engine, synth, weights
    eng = engine.Engine("denoiser", weights.synthetic_weights("denoiser", 7), precision="f16x3")
    mix = apply.trim_to_frames(apply.normalise(apply.read_wav(os.path.join(d, "in.wav"))))
    neg = apply.normalise(apply.extend_context(apply.read_wav(os.path.join(d, "neg.wav"))))
    pos = apply.normalise(np.zeros(len(synth.silent()), dtype=np.int16))
    ref = eng.enhance([mix], [pos], [neg], want_mixed=False)["denoised_wav"][0]
    eng.close()
    got = wavfile.read(os.path.join(d, "first.wav"))[1]
    assert np.array_equal(got, ref)
    # a truncated cache entry is ignored and rebuilt, not trusted
    cache = os.path.join(d, "cache")
    blobs = [f for f in os.listdir(cache) if f.endswith(".blob")]
    assert len(blobs) == 1
    with open(os.path.join(cache, blobs[0]), "r+b") as f:
        f.truncate(1 << 20)
    third = _run_cli(d, "third")
    assert third[3]["cache"] == "miss"
    assert open(os.path.join(d, "third.wav"), "rb").read() == open(os.path.join(d, "first.wav"), "rb").read()
    assert os.path.getsize(os.path.join(cache, blobs[0])) > (100 << 20)


def test_fresh_process_separator_cli_equals_the_full_engine(lib_built, tmp_path):
    """`nhans_separator` (setup.py:48, SS/apply.py:288-397) through the torch-free engine and its own cache entry: the same
    bytes as engine.Engine with the separator's (interferer, target) conditioning order."""
    from scipy.io import wavfile
    import nhans_amd  # noqa: F401
    from nhans_amd import apply, engine, synth, weights
    d = str(tmp_path)
    _write_inputs(d)
    wavfile.write(os.path.join(d, "pos.wav"), 16000, synth.speaker_context(72, low=False))
    first = _run_cli(d, "sep1", prog="separator")
    second = _run_cli(d, "sep2", prog="separator")
    assert first[2] is False and first[3]["engine"] == "LiteEngine" and first[3]["cache"] == "miss"
    assert second[3]["cache"] == "hit"
    assert open(os.path.join(d, "sep1.wav"), "rb").read() == open(os.path.join(d, "sep2.wav"), "rb").read()
    eng = engine.Engine("separator", weights.synthetic_weights("separator", 7), precision="f16x3")
    mix = apply.trim_to_frames(apply.normalise(apply.read_wav(os.path.join(d, "in.wav"))))
    neg = apply.normalise(apply.extend_context(apply.read_wav(os.path.join(d, "neg.wav"))))     # interferer: context a
    pos = apply.normalise(apply.extend_context(apply.read_wav(os.path.join(d, "pos.wav"))))     # target: context b
    ref = eng.enhance([mix], [neg], [pos], want_mixed=False)["denoised_wav"][0]
    eng.close()
    assert np.array_equal(wavfile.read(os.path.join(d, "sep1.wav"))[1], ref)
