"""CPU tests of the host side: checkpoint inventory and bundle reader, folding/packing, the
apply_* front end, the C-ABI library's exported surface (no compute without a GPU), the FFT
building blocks compiled for the host."""
import ctypes
import json
import os
import re
import struct
import subprocess
import sys

import numpy as np
import pytest
from scipy.io import wavfile

import nhans_amd  # noqa: F401
from nhans_amd import apply, fold, hip, spec, synth, tfbundle, weights
from conftest import GOLDEN, ROOT


# ------------------------------------------------------------------------------ inventory / bundle
@pytest.mark.parametrize("kind", ["denoiser", "separator"])
def test_inventory_matches_shipped_index(kind):
    ent = tfbundle.read_index(os.path.join(GOLDEN, kind + ".index"))
    want = json.load(open(os.path.join(GOLDEN, "checkpoint_index.json")))[kind]
    assert [[k, list(v["shape"]), v["dtype"], v["offset"], v["size"]] for k, v in ent.items()] == want
    shapes = spec.variable_shapes(kind)
    extra = set(ent) - set(shapes)
    assert extra == (set() if kind == "denoiser" else {"Variable"})       # SS global_step
    for n, s in shapes.items():
        assert tuple(ent[n]["shape"]) == tuple(s) and ent[n]["dtype"] == tfbundle.DT_FLOAT
    assert sum(int(np.prod(s)) for s in shapes.values()) == 28999881
    assert max(e["offset"] + e["size"] for e in ent.values()) == (115999524 if kind == "denoiser" else 115999528)


def test_bundle_reader_roundtrip_and_lfs_pointer(tmp_path, weights_separator):
    """Write a data shard laid out by the shipped separator index, read it back."""
    prefix = str(tmp_path / "81457_2-545000")
    idx = os.path.join(GOLDEN, "separator.index")
    os.symlink(idx, prefix + ".index")
    ent = tfbundle.read_index(idx)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(b"version https://git-lfs.github.com/spec/v1\noid sha256:0\nsize 1\n")
    with pytest.raises(FileNotFoundError):
        weights.load_checkpoint(prefix, "separator")
    buf = bytearray(max(e["offset"] + e["size"] for e in ent.values()))
    for n, e in ent.items():
        if n == "Variable":
            buf[e["offset"]:e["offset"] + 4] = struct.pack("<i", 545000)
        else:
            buf[e["offset"]:e["offset"] + e["size"]] = weights_separator[n].tobytes()
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(buf)
    # the shipped index records the crc32c of the TRAINED tensors: synthetic bytes must be refused ...
    with pytest.raises(ValueError, match="crc32c"):
        weights.load_checkpoint(prefix, "separator")
    # ... and load when the check is switched off
    got = weights.load_checkpoint(prefix, "separator", verify_crc=False)
    assert list(got) == list(spec.variable_shapes("separator"))
    for n in got:
        assert np.array_equal(got[n], weights_separator[n])
    assert int(tfbundle.load_checkpoint(prefix, verify_crc=False)["Variable"]) == 545000


def test_crc32c_pinned_to_tensorflow_written_values(tmp_path, lib_built):
    """The reference's own `.index` files carry TF-written masked CRC-32Cs: one per table block
    (verified by read_index) and one per tensor.  The only tensor whose bytes are known without the
    LFS blob is the separator's int32 scalar `Variable`, which is 0."""
    assert tfbundle.crc32c(b"123456789") == 0xE3069283                      # the CRC-32C check value
    ent = tfbundle.read_index(os.path.join(GOLDEN, "separator.index"))      # block trailers verified inside
    assert ent["Variable"]["crc32c"] == tfbundle.masked_crc32c(struct.pack("<i", 0))
    assert ent["Variable"]["crc32c"] != tfbundle.masked_crc32c(struct.pack("<i", 545000))
    # a flipped byte inside a data block of the index is caught by its trailer
    raw = bytearray(open(os.path.join(GOLDEN, "denoiser.index"), "rb").read())
    raw[100] ^= 0x40
    bad = str(tmp_path / "bad.index")
    open(bad, "wb").write(raw)
    with pytest.raises(ValueError, match="crc32c"):
        tfbundle.read_index(bad)
    # the library's slicing-by-8 host helper == the plain table loop
    blob = np.random.default_rng(3).integers(0, 256, 100003, dtype=np.uint8).tobytes()
    lib = hip.load()
    buf = np.frombuffer(blob, dtype=np.uint8)
    fast = lib.nhans_crc32c(0, buf.ctypes.data, len(buf))
    slow = 0
    for i in range(0, len(blob), 1000):          # below the 4096-byte switch: pure Python, chained
        slow = tfbundle.crc32c(blob[i:i + 1000], slow)
    assert fast == slow == tfbundle.crc32c(blob)


def test_synthetic_weights_deterministic_and_shared():
    a = weights.synthetic_weights("denoiser", 7)
    b = weights.synthetic_weights("denoiser", 7)
    s = weights.synthetic_weights("separator", 7)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert np.array_equal(a["resblock3_1_conv2/w"], s["resblock3_1_conv2/w"])
    assert "resblock1_1_conv1_noise_pos_emb/w" in a and "resblock1_1_conv1_clean_emb/w" in s
    assert not np.array_equal(a["last_dense/w"], weights.synthetic_weights("denoiser", 8)["last_dense/w"])


# ------------------------------------------------------------------------------ folding
def test_pack_igemm_layout():
    rng = np.random.default_rng(0)
    k, n, npad = 64, 40, 64
    w = rng.standard_normal((k, n))
    p = fold.pack_igemm(w, npad).reshape(k // 32, npad // 32, 4, 64, 4)
    for chunk, nt, q, lane, e in [(0, 0, 0, 0, 0), (1, 1, 3, 63, 3), (1, 0, 2, 37, 1), (0, 1, 1, 5, 2)]:
        kk = 32 * chunk + 8 * q + 4 * (lane >> 5) + e
        nn = 32 * nt + (lane & 31)
        want = np.float32(w[kk, nn]) if nn < n else np.float32(0)
        assert p[chunk, nt, q, lane, e] == want


def test_fold_blob_structure(weights_denoiser):
    arrs = fold.fold_arrays(weights_denoiser, "denoiser")
    assert arrs["cond.w"].size == 1024 * 3840 and arrs["cond.base"].size == 3840
    assert arrs["m0.c1.w"].size == 16 * 64 and arrs["t0.c1.w"].size == 32 * 64
    assert arrs["m2.c2.wpk_t"].size == 64 * 128 and "m1.c2.wpk_t" not in arrs
    assert arrs["head.dense.wpk"].size == 13312 * 256
    assert arrs["m0.c1.tf"].size == 35 * 201 * 64 and arrs["m7.c2.tf"].size == 5 * 26 * 512
    # ... and its two terms on their own: tf[h, w] == tt[h] + ff[w] up to the float32 rounding of the sum
    tt, ff = arrs["m1.c2.tt"].reshape(35, 64), arrs["m1.c2.ff"].reshape(201, 64)
    assert np.abs(arrs["m1.c2.tf"].reshape(35, 201, 64) - (tt[:, None, :] + ff[None, :, :])).max() <= 2e-6
    # BN scale really is folded: conv1 weights of block 1 equal w * gamma/sqrt(var+eps)
    W = weights_denoiser
    sc = (W["resblock1_2_conv1/gamma"].astype(np.float64) /
          np.sqrt(W["resblock1_2_conv1/pop_variance"].astype(np.float64) + 1e-3)).reshape(-1)
    w4 = W["resblock1_2_conv1/w"].astype(np.float64)
    w = fold.kmat(w4) * sc
    # K order: 32-channel chunk, filter row, filter column, channel
    assert w.shape == (4 * 4 * 64, 64) and w[((1 * 4 + 1) * 4 + 2) * 32 + 5, 7] == w4[1, 2, 32 + 5, 7] * sc[7]
    np.testing.assert_array_equal(arrs["m1.c1.wpk"], fold.pack_igemm(w))
    blob = fold.write_blob({"b": np.arange(5, dtype=np.float32), "a": np.ones(3, dtype=np.float32)})
    magic, ver, n, total = struct.unpack_from("<8sIIQ", blob, 0)
    assert (magic, ver, n, total) == (b"NHANSFW1", fold.BLOB_VERSION, 2, len(blob)) and fold.BLOB_VERSION == 2
    name, off, cnt = struct.unpack_from("<48sQQ", blob, 24)
    assert name.rstrip(b"\0") == b"a" and off % 256 == 0 and cnt == 3
    assert np.frombuffer(blob, np.float32, 3, off).tolist() == [1, 1, 1]


# ------------------------------------------------------------------------------ apply front end
def test_read_normalise_trim_on_reference_example():
    x = apply.read_wav(os.path.join(GOLDEN, "exp2_noisy.wav"))
    assert x.dtype == np.int16 and len(x) == 49600
    y = apply.trim_to_frames(apply.normalise(x))
    assert y.dtype == np.float32 and len(y) == 49520 and abs(np.abs(y).max() - 1.0) < 1e-6
    from oracle import nhans_oracle as O
    np.testing.assert_array_equal(y, O.trim_to_frames(O.normalise(O.read_wav(os.path.join(GOLDEN, "exp2_noisy.wav")))))


def test_read_wav_rejects_wrong_format(tmp_path):
    p = str(tmp_path / "a.wav")
    wavfile.write(p, 8000, np.zeros(100, dtype=np.int16))
    with pytest.raises(AssertionError):
        apply.read_wav(p)
    wavfile.write(p, 16000, np.zeros(100, dtype=np.float32))
    with pytest.raises(AssertionError):
        apply.read_wav(p)
    wavfile.write(p, 16000, np.stack([np.full(50, 100, np.int16), np.full(50, 300, np.int16)], 1))
    assert apply.read_wav(p).tolist() == [200.0] * 50                  # stereo -> mean
    assert apply.handle_signals(str(tmp_path / "missing.wav"), p, p) is None   # 'error in threads'


def test_short_context_policy(tmp_path):
    x = np.arange(16000, dtype=np.int16)
    y = apply.extend_context(x)
    assert len(y) == spec.MIN_CTX_SAMPLES == 32240
    assert np.array_equal(y[:16000], x) and np.array_equal(y[16000:32000], x) and np.array_equal(y[32000:], x[:240])
    z = np.arange(40000, dtype=np.int16)
    assert apply.extend_context(z) is z
    assert spec.frames_for_samples(32240)[1] == 200


def test_cli_flag_surface():
    a = apply._parse(["--input", "i.wav", "--neg", "n.wav", "--output", "o_denoised.wav", "--ac", "--compensate", "0.5"],
                     "nhans_denoiser")
    assert (a.input, a.neg, a.output, a.ac, a.compensate) == ("i.wav", "n.wav", "o_denoised.wav", True, 0.5)
    assert a.pos.endswith("Silent.wav")
    assert list(apply._pairs(a)) == [("i.wav", a.pos, "n.wav", "o_denoised.wav")]
    apply.FLAGS.ac, apply.FLAGS.compensate = False, 0.0


def test_side_file_names():
    """SN/apply.py:457-470 cuts 12 characters off the output path; kept where that is meaningful."""
    assert apply.side_prefix("./audio_examples/denoised.wav") == "./audio_examples/"           # the reference's default
    assert apply.side_prefix("out/exp2_denoised.wav") == "out/exp2_"
    assert apply.side_prefix("out/a.wav") == "out/a_"                                            # not 'out/a.wav'[:-12] == ''
    assert apply.side_prefix("out/recording_0001.wav") != apply.side_prefix("out/recording_0002.wav")


def test_directory_mode_batches_and_names_side_files(tmp_path, monkeypatch):
    """Directory mode without a GPU: a stand-in engine records that all clips arrive in ONE call and
    returns recognisable waveforms; the files written must be per clip."""
    from scipy.io import wavfile as wf

    class FakeEngine:
        calls = []

        def enhance(self, mixes, ca, cb, want_mixed=True, taps=False):
            FakeEngine.calls.append(len(mixes))
            return {"denoised_wav": [m * np.float32(0.5) for m in mixes], "mixed_wav": [m.copy() for m in mixes]}

    ind, negd, outd = tmp_path / "in", tmp_path / "neg", tmp_path / "out"
    ind.mkdir()
    negd.mkdir()
    names = ["a.wav", "bb.wav", "recording_0001.wav", "recording_0002.wav"]
    for i, n in enumerate(names):
        wf.write(str(ind / n), 16000, synth.mixture(60 + i, 0.1 + 0.05 * i))
        wf.write(str(negd / n), 16000, synth.noise_context(60 + i, 1.0))
    wf.write(str(ind / "tiny.wav"), 16000, np.ones(300, dtype=np.int16))       # < one window: reported and skipped
    wf.write(str(negd / "tiny.wav"), 16000, synth.noise_context(1, 1.0))
    (ind / "notes.txt").write_text("not audio")
    monkeypatch.setitem(apply._engines, "denoiser", FakeEngine())
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    apply.main(["--input", str(ind), "--neg", str(negd), "--pos", str(tmp_path / "Silent.wav"), "--output", str(outd),
                "--weights", "synthetic", "--compensate", "0.25"])
    apply.FLAGS.compensate = 0.0
    assert FakeEngine.calls == [4]
    want = sorted(n for s in names for n in (s, s[:-4] + "_mixed_processed.wav", s[:-4] + "_removed.wav",
                                             s[:-4] + "_compensated.wav"))
    assert sorted(os.listdir(str(outd))) == want
    for i, n in enumerate(names):
        x = apply.trim_to_frames(apply.normalise(synth.mixture(60 + i, 0.1 + 0.05 * i)))
        assert np.array_equal(wf.read(str(outd / n))[1], x * np.float32(0.5))
        # compensated = denoised + (mixed - denoised) * 0.25
        assert np.allclose(wf.read(str(outd / (n[:-4] + "_compensated.wav")))[1], x * 0.5 + (x - x * 0.5) * 0.25, atol=1e-7)


def test_missing_checkpoint_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(apply.FLAGS, "weights", "checkpoint")
    monkeypatch.setattr(apply.FLAGS, "model_dir", str(tmp_path))
    with pytest.raises((FileNotFoundError, OSError)):
        apply._load_weights("denoiser")


# ------------------------------------------------------------------------------ C ABI surface
def test_library_exports_every_declared_symbol(lib_built):
    header = open(os.path.join(ROOT, "include", "nhans_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(nhans_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(hip.EXPORTS)
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    h = hip.load()
    assert h.nhans_abi_version() == hip.ABI_VERSION == 5
    assert h.nhans_num_frames(399) == 0 and h.nhans_num_frames(400) == 1 and h.nhans_num_frames(159920) == 998
    bad = ctypes.create_string_buffer(b"x" * 64, 64)
    out = ctypes.c_void_p()
    assert h.nhans_create(0, bad, 64, 0, ctypes.byref(out)) == -1            # NHANS_EINVAL: bad magic
    assert b"magic" in h.nhans_last_error()
    # nhans_create_ex (ABI 5): the exponents are checked before anything touches a device
    ok = (ctypes.c_int * hip.NUM_ACTIVATIONS)(*([0] * hip.NUM_ACTIVATIONS))
    assert h.nhans_create_ex(0, bad, 64, 0, ok, hip.NUM_ACTIVATIONS - 1, ctypes.byref(out)) == -1
    assert b"NHANS_NUM_ACTIVATIONS" in h.nhans_last_error()
    far = (ctypes.c_int * hip.NUM_ACTIVATIONS)(*([0] * (hip.NUM_ACTIVATIONS - 1) + [61]))
    assert h.nhans_create_ex(0, bad, 64, 0, far, hip.NUM_ACTIVATIONS, ctypes.byref(out)) == -1
    assert b"outside [-60, 60]" in h.nhans_last_error()
    assert h.nhans_create_ex(0, bad, 64, 0, ok, hip.NUM_ACTIVATIONS, ctypes.byref(out)) == -1 and b"magic" in h.nhans_last_error()
    assert h.nhans_create_ex(0, bad, 64, 0, None, 0, ctypes.byref(out)) == -1 and b"magic" in h.nhans_last_error()


def test_launch_failure_returns_negative_code(lib_built):
    """The launch-error channel end to end: a kernel launch that cannot happen (here: no device at
    all on the CPU box; on the GPU box tests/test_gpu_scale.py asks for 1 MB of LDS) must come back
    as a negative code with the kernel's name, never as NHANS_OK."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_scale.py::test_launch_failure_reaches_the_caller")
    h = hip.load()
    rc = h.nhans_debug_launch_probe(1 << 20, None)
    assert rc == -2 and b"launch_probe" in h.nhans_last_error()
    flags = ctypes.c_int(0)
    assert h.nhans_take_status(None, ctypes.byref(flags), None) == -1          # null context: EINVAL, not a crash


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nhans_amd import engine
    with pytest.raises(hip.NhansError):
        engine.Engine("denoiser", {})


def test_product_code_never_imports_the_oracle():
    """Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may touch oracle/: neither the package
    nor the developer tools do."""
    for top in ("n-hans_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert "import oracle" not in src and "from oracle" not in src, f
    src = open(os.path.join(ROOT, "bench.py")).read()
    for line in src.splitlines():                       # bench.py: only inside cpu_baseline() / rms_check()
        if "from oracle" in line or "import oracle" in line:
            assert line.startswith("    "), "bench.py imports the oracle at module level: " + line


# ------------------------------------------------------------------------------ FFT blocks on the host
def test_fft400_host_build_matches_numpy(tmp_path):
    so = str(tmp_path / "libfft400_host.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "n-hans_amd", "csrc", "fft400_host.cpp")])
    lib = ctypes.CDLL(so)
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(400) + 1j * rng.standard_normal(400)).astype(np.complex64)
    out = np.zeros(400, np.complex64)
    for inv in (0, 1):
        lib.nhans_fft400_host(x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), inv)
        ref = np.fft.ifft(x.astype(np.complex128)) * 400 if inv else np.fft.fft(x.astype(np.complex128))
        assert np.abs(out - ref).max() < 5e-6 * np.abs(ref).max()
    # the real-input analysis transform of the STFT kernel (rows 0..10 + conjugate mirror) against numpy's rfft
    xr = rng.standard_normal(400).astype(np.float32)
    outr = np.full(201, np.nan + 0j, np.complex64)
    lib.nhans_rfft400_host(xr.ctypes.data_as(ctypes.c_void_p), outr.ctypes.data_as(ctypes.c_void_p))
    refr = np.fft.rfft(xr.astype(np.float64))
    assert np.isfinite(outr).all() and np.abs(outr - refr).max() < 5e-6 * np.abs(refr).max()


def test_synthetic_audio_is_deterministic():
    a, b = synth.mixture(3, 0.5), synth.mixture(3, 0.5)
    assert a.dtype == np.int16 and np.array_equal(a, b) and np.abs(a).max() == 32767
    assert len(synth.noise_context(0)) == 48000 >= spec.MIN_CTX_SAMPLES
    assert not np.array_equal(synth.mixture(4, 0.5), a)


# ------------------------------------------------------------------------------ converter / load_*
def test_format_converter(tmp_path):
    t = np.arange(8000) / 8000.0
    x = np.sin(2 * np.pi * 440 * t)
    p = str(tmp_path / "a.wav")
    wavfile.write(p, 16000, np.round(x * 20000).astype(np.int16))
    assert np.array_equal(apply.read_wav_any(p), apply.read_wav(p))           # already standard: untouched
    wavfile.write(p, 8000, x.astype(np.float32))                              # 8 kHz float -> 16 kHz int16
    y = apply.read_wav_any(p)
    assert y.dtype == np.int16 and len(y) == 16000
    ref = np.sin(2 * np.pi * 440 * np.arange(16000) / 16000.0) * 32768
    assert np.abs(y[2000:14000] - ref[2000:14000]).max() < 400
    wavfile.write(p, 48000, np.stack([np.round(np.tile(x, 6) * 2 ** 30).astype(np.int32)] * 2, 1))   # 48 kHz int32 stereo
    z = apply.read_wav_any(p)
    assert z.ndim == 1 and len(z) == 16000 and 15000 < np.abs(z).max() < 17000
    with pytest.raises(AssertionError):
        apply.read_wav(p)
    a = apply._parse(["--input", p, "--neg", p, "--output", "o_denoised.wav"], "nhans_denoiser")
    assert a.convert is True and apply._reader() is apply.read_wav_any
    apply._parse(["--input", p, "--neg", p, "--output", "o_denoised.wav", "--no-convert"], "nhans_denoiser")
    assert apply._reader() is apply.read_wav
    apply.FLAGS.convert = False


def test_load_model_verifies_user_checkpoint(tmp_path, weights_denoiser, capsys):
    from nhans_amd import load_model
    name, size, sha = load_model.BUNDLES["denoiser"]
    assert (name, size) == ("81448_0-1000000", 115999524) and len(sha) == 64
    with pytest.raises(FileNotFoundError):
        load_model.verify("denoiser", str(tmp_path))
    os.symlink(os.path.join(GOLDEN, "denoiser.index"), str(tmp_path / (name + ".index")))
    data = str(tmp_path / (name + ".data-00000-of-00001"))
    open(data, "w").write("version https://git-lfs.github.com/spec/v1\noid sha256:%s\nsize %d\n" % (sha, size))
    with pytest.raises(FileNotFoundError, match="LFS"):
        load_model.verify("denoiser", str(tmp_path))
    ent = tfbundle.read_index(os.path.join(GOLDEN, "denoiser.index"))
    buf = bytearray(size)
    for n, e in ent.items():
        buf[e["offset"]:e["offset"] + e["size"]] = weights_denoiser[n].tobytes()
    open(data, "wb").write(buf)
    with pytest.raises(ValueError, match="sha256"):          # right size, not the trained weights
        load_model.verify("denoiser", str(tmp_path))
    load_model.main(["--model_dir", str(tmp_path), "--no-hash"])
    assert "571 tensors, 28999881 parameters" in capsys.readouterr().out


# ------------------------------------------------------------------------------ demo / eval mixing
def test_domixing_matches_oracle_and_reference_quirks():
    from nhans_amd import mixing
    from oracle import nhans_oracle as O
    sig = {"c": synth.mixture(7, 3.0), "p": synth.noise_context(7, 1.0), "n": synth.speaker_context(7, 4.0)}
    t1, p1, n1, m1 = O.demo_signals(sig["c"], sig["p"], sig["n"])
    t2, p2, n2, m2, sp, sn = mixing.combine_signals(lambda k: sig[k], "c", "p", "n")
    for a, b in ((t1, t2), (p1, p2), (n1, n2), (m1, m2)):
        assert a.dtype == np.float32 and np.array_equal(a, b)
    assert (int(sp), int(sn)) == (0, 0) and len(m2) == 47920 == len(p2) == len(n2)      # trimmed, noise tiled/cut
    assert abs(np.abs(m2).max() - 1.0) < 1e-5
    assert np.abs(t2).max() > 1.2            # sic: target is divided by the NORMALISED mixture's peak
    # 0 dB: scaled noise power == speech power (before the final normalisation they share one factor)
    clean = O.trim_to_frames(O.normalise(sig["c"]))
    assert abs(np.mean(p2.astype(np.float64) ** 2) / np.mean(clean.astype(np.float64) ** 2) - 1) < 1e-3
    # evaluation reader: md5-derived SNRs are deterministic and in the table
    a, b = mixing.eval_snrs("/data/test/6930-76324-0008.wav")
    assert (a, b) == mixing.eval_snrs(b"/data/test/6930-76324-0008.wav") and a in mixing.SNRS_DENOISER and b in mixing.SNRS_DENOISER
    t3, _, n3, m3, sp3, sn3 = mixing.combine_signals(lambda k: sig[k], "c", "p", "n", snrs=(8, -3))
    ratio = np.mean(n3.astype(np.float64) ** 2) / np.mean(np.asarray(n2, np.float64) ** 2)
    assert (int(sp3), int(sn3)) == (8, -3) and not np.array_equal(m3, m2) and ratio > 1.0


def test_separator_mixing_and_trim_quirk():
    from nhans_amd import mixing
    sig = {"c": synth.speaker_context(1, 2.0, low=False), "n": synth.speaker_context(1, 1.0, low=True)}
    clean, noise_k, mixed, snr = mixing.combine_signals_separator(lambda k: sig[k], "c", "n")
    assert len(clean) == len(mixed) == 31920 and len(noise_k) == 16000 and int(snr) == 0
    assert abs(np.abs(mixed).max() - 1.0) < 1e-5
    # a clean recording that already holds a whole number of frames is emptied by the reference's
    # unconditional `[:-0]` slice (SS/apply.py:98); reproduced, and it fails loudly downstream
    sig["c"] = sig["c"][:400 + 160 * 50]
    with pytest.raises(Exception):
        mixing.combine_signals_separator(lambda k: sig[k], "c", "n")


def test_bench_device_sampler_degrades_to_none_and_reads_hwmon(tmp_path, monkeypatch):
    """bench.py's clock/power sampler: no GPU / no hwmon files -> device_state null (never an exception); with
    hwmon-shaped files it averages freq1_input (Hz) and power1_input (uW)."""
    import sys
    import time
    sys.path.insert(0, ROOT)
    import bench
    s = bench.DeviceSampler(0)
    s.start()
    assert s.stop() is None
    for name, val in (("freq1_input", "1950000000"), ("power1_input", "1336000000"), ("power1_cap", "1400000000")):
        (tmp_path / name).write_text(val + "\n")
    import threading
    s = bench.DeviceSampler(0, period=0.01)
    s.files = {k: str(tmp_path / k) for k in ("freq1_input", "power1_input", "power1_cap")}
    s.thread = threading.Thread(target=s._run, daemon=True)
    s.start()
    time.sleep(0.1)
    st = s.stop()
    assert st["samples"] >= 2 and abs(st["sclk_mhz_mean"] - 1950.0) < 1e-6
    assert abs(st["socket_power_w_mean"] - 1336.0) < 1e-6 and st["power_cap_w"] == 1400.0


def test_bench_gpus_flag_is_honoured_or_refused(monkeypatch, capfd):
    import sys
    """bench.py --gpus N: under a launcher (WORLD_SIZE set) a different N is an error (exit 2) instead of a silently
    different run; without a launcher N > 1 starts N rank processes and the exit code is non-zero when a rank fails
    (here: no HIP device, so every rank does)."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    assert bench.main(["--gpus", "8"]) == 2
    assert "WORLD_SIZE=1" in capfd.readouterr().err
    import torch
    if torch.cuda.device_count() > 0:
        return                                   # (the GPU suite runs the two-rank launch for real)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.main(["--gpus", "2", "--clips-per-gpu", "1", "--seconds", "1", "--steps", "1", "--warmup", "0"]) == 1
    assert "rank(s) failed" in capfd.readouterr().err


def test_bench_self_launcher_ends_the_siblings_of_a_dead_rank(monkeypatch, capfd):
    """bench.py --gpus N without a launcher polls ALL its children: when one rank dies (here rank 2, at start-up) while
    another would sit in communicator initialisation for ever (rank 0 sleeps), the rest is terminated and the launcher
    returns 1 within seconds; a job that produces nothing within --rank-timeout ends the same way."""
    import sys
    import time
    sys.path.insert(0, ROOT)
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    t0 = time.time()
    assert bench.main(["--gpus", "3", "--debug-fail-rank", "2", "--debug-hang-rank", "0"]) == 1
    err = capfd.readouterr().err
    assert "rank(s) failed" in err and "(2, 3)" in err and time.time() - t0 < 30
    t0 = time.time()
    assert bench.main(["--gpus", "2", "--debug-hang-rank", "0", "--debug-fail-rank", "-1", "--rank-timeout", "2"]) == 1
    err = capfd.readouterr().err
    assert "rank(s) failed" in err and time.time() - t0 < 60


# ------------------------------------------------------------------------------ Winograd folding (conv_wino.hip)
def test_winograd_matrices_are_exact_and_match_the_kernel_constants():
    """fold.wino_matrices: Cook-Toom F(5,4) / F(6,3) on the points 0, +-1, +-2, +-1/2, infinity in exact rationals;
    y = AT ((G g) * (BT d)) equals the correlation, BT is the table hard-coded in the kernel's input transform, AT is
    the powers-of-the-points table its output transform spells out."""
    from nhans_amd import fold
    rng = np.random.default_rng(0)
    for m, r in ((5, 4), (6, 3)):
        AT, G, BT = fold.wino_matrices(m, r)
        assert np.array_equal(BT, fold.WINO_BT) and AT.shape == (m, 8) and G.shape == (8, r)
        pts = [0, 1, -1, 2, -2, 0.5, -0.5]
        for i in range(m):
            assert [AT[i, p] for p in range(7)] == [float(q) ** i for q in pts] and AT[i, 7] == (1.0 if i == m - 1 else 0.0)
        g, d = rng.standard_normal(r), rng.standard_normal(8)
        ref = np.array([d[i:i + r] @ g for i in range(m)])
        assert np.abs(AT @ ((G @ g) * (BT @ d)) - ref).max() < 1e-13
    assert fold.wino_outputs(4) == 5 and fold.wino_outputs(3) == 6


def test_winograd_weight_pack_reproduces_the_direct_convolution():
    """fold.pack_wino -> the fragment order conv_wino.hip streams ([N/64][p][C/8][KH/2][nt][hi|lo][lane][8]: a k-step
    of the MFMA is 8 channels x two filter rows): unpack it with the kernel's own index arithmetic, run the 1-D Winograd algorithm in float64 (V = BT d along W, M_p = sum over
    filter rows and channels, Y = AT M) and compare with the direct SAME convolution."""
    from nhans_amd import fold
    rng = np.random.default_rng(1)
    KH, C, N, H, W = 4, 32, 64, 7, 23
    w4 = rng.standard_normal((KH, KH, C, N)) / np.sqrt(KH * KH * C)
    x = rng.standard_normal((H, W, C))
    pk, ws = fold.pack_wino(w4)
    halfs = pk.view(np.float16).astype(np.float64).reshape(N // 64, 8, C // 8, KH // 2, 2, 2, 64, 8)   # nb p c8 s nt h lane e
    U = np.zeros((8, KH, C, N))
    for lane in range(64):
        for e in range(8):
            # lane l, element e of fragment (nb, p, c8, s, nt): filter row 2 s + (l >> 5), channel 8 c8 + e,
            # column 64 nb + 32 nt + (l & 31)
            U[:, (lane >> 5)::2, e::8, (lane & 31)::32] = (halfs[:, :, :, :, :, 0, lane, e] + halfs[:, :, :, :, :, 1, lane, e]) \
                .transpose(1, 3, 2, 0, 4).reshape(8, KH // 2, C // 8, (N // 64) * 2)
    U *= ws[None, None, None, :]                                   # undo the per-channel power-of-two scale
    m = fold.wino_outputs(KH)
    AT, G, BT = fold.wino_matrices(m, KH)
    assert np.abs(U - np.einsum("pk,hkcn->phcn", G, w4)).max() < 1e-5 * np.abs(U).max()
    pt, pl = spec.same_pad(H, KH, 1)[0], spec.same_pad(W, KH, 1)[0]
    ntile = -(-W // m)
    xp = np.zeros((H + KH - 1, ntile * m + 8, C))
    xp[pt:pt + H, pl:pl + W] = x
    out = np.zeros((H, ntile * m, N))
    for j in range(ntile):
        V = np.einsum("px,hxc->phc", BT, xp[:, j * m:j * m + 8])   # [8, H+KH-1, C]
        M = sum(np.einsum("phc,pcn->phn", V[:, kh:kh + H], U[:, kh]) for kh in range(KH))
        out[:, j * m:(j + 1) * m] = np.einsum("ip,phn->hin", AT, M)
    ref = np.zeros((H, W, N))
    for kh in range(KH):
        for kw in range(KH):
            ref += np.einsum("hwc,cn->hwn", xp[kh:kh + H, kw:kw + W], w4[kh, kw])
    assert np.abs(out[:, :W] - ref).max() < 2e-6 * np.abs(ref).max() + 1e-6


def test_winograd_kernel_isa_keeps_its_hand_counted_waits():
    """conv_wino.hip requests its weights inside inline asm and waits for them with its own s_waitcnt; the compiler
    believes such a register is valid when the asm statement ends.  tools/check_wino_isa.py compiles the kernel and
    verifies that nothing touches a requested register before the wait, that no spill and no compiler-generated
    vector-memory wait or LDS access sits in the K loop (needs hipcc only)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_wino_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


def test_winograd_isa_checker_catches_what_it_is_there_for(tmp_path):
    """tools/check_wino_isa.py is part of the build (Makefile: it checks the assembly of the object being linked).  A
    checker that cannot fail is no check: five mutations of real kernel assembly -- a compiler-style copy INTO a weight
    register between its request and the multiply block, a multiply block that opens with another wait count, one
    residual request fewer than the epilogue's counted wait stands for, a VALU write of a scalar address register right in front
    of an asm request that reads it -- must each fail it, the unmutated text pass."""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "check_wino_isa.py")
    src = os.path.join(ROOT, "n-hans_amd", "csrc", ".isa", "conv_wino-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(src):                       # (not built through the Makefile in this checkout: compile it)
        src = str(tmp_path / "w.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only",
                        "-fno-slp-vectorize", "-S", os.path.join(ROOT, "n-hans_amd", "csrc", "conv_wino.hip"), "-o", src], check=True)
    text = open(src).read()

    def check(t):
        f = tmp_path / "m.s"
        f.write_text(t)
        return subprocess.run([sys.executable, tool, "--asm", str(f)], capture_output=True, text=True)
    r = check(text)
    assert r.returncode == 0, r.stdout
    lines = text.split("\n")
    # (a) the first weight-request asm block inside a K loop: a write to its first destination register right behind it
    k = next(i for i, l in enumerate(lines) if "global_load_dwordx4" in l and "s[" in l and i > 300)
    reg = re.search(r"global_load_dwordx4 v\[(\d+):", lines[k]).group(1)
    end = next(i for i in range(k, k + 20) if "#ASMEND" in lines[i])
    mut = lines[:end + 1] + ["\tv_mov_b32_e32 v%s, 0" % reg] + lines[end + 1:]
    r = check("\n".join(mut))
    assert r.returncode != 0 and "touches a weight register" in r.stdout, r.stdout[-400:]
    # (b) a multiply block opening with vmcnt(5)
    m = next(i for i, l in enumerate(lines) if "s_waitcnt vmcnt(4)" in l and "v_mfma" in "".join(lines[i:i + 12]))
    mut = list(lines)
    mut[m] = mut[m].replace("vmcnt(4)", "vmcnt(5)")
    r = check("\n".join(mut))
    assert r.returncode != 0 and "does not open with s_waitcnt vmcnt(4)" in r.stdout, r.stdout[-400:]
    # (c) the epilogue: drop one residual request in front of a counted wait `s_waitcnt vmcnt(4)` that follows the constants
    w = [i for i, l in enumerate(lines) if re.search(r"s_waitcnt vmcnt\(4\)\s*$", l) and "ASMSTART" in lines[i - 1] and "v_mfma" not in "".join(lines[i:i + 12])]
    assert w, "no counted epilogue wait found"
    j = next(i for i in range(w[0], w[0] - 200, -1) if re.match(r"\s*(global|buffer)_load_dwordx4", lines[i]) and "ASM" not in lines[i - 1])
    mut = lines[:j] + lines[j + 1:]
    r = check("\n".join(mut))
    assert r.returncode != 0 and "vector-memory instructions between the constants" in r.stdout, r.stdout[-400:]
    # (d) a VALU write of the scalar base register right in front of a weight request inside asm (the wait states the compiler
    # inserts for its own instructions are not inserted for the contents of inline asm)
    sreg = re.search(r"global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[(\d+):\d+\]", lines[k]).group(1)
    start = next(i for i in range(k, k - 20, -1) if "#ASMSTART" in lines[i])
    mut = lines[:start] + ["\tv_readfirstlane_b32 s%s, v0" % sreg] + lines[start:]
    r = check("\n".join(mut))
    assert r.returncode != 0 and "wait states earlier" in r.stdout, r.stdout[-400:]
    # (e) a write into a register of the epilogue's FIRST constants request, right behind that request (far in front of the
    # counted wait, with the second request block in between): what the compiler does with a requested register the source
    # never reads -- the latent fault round 5 found in variants of the epilogue
    c0 = next(i for i, l in enumerate(lines) if re.search(r"global_load_dwordx4 v\[\d+:\d+\], v\[\d+:\d+\], off\s*$", l.split(";")[0]) and "ASMSTART" in lines[i - 1])
    creg = re.search(r"global_load_dwordx4 v\[(\d+):", lines[c0]).group(1)
    cend = next(i for i in range(c0, c0 + 8) if "#ASMEND" in lines[i])
    mut = lines[:cend + 1] + ["\tv_mov_b32_e32 v%s, 0" % creg] + lines[cend + 1:]
    r = check("\n".join(mut))
    assert r.returncode != 0 and "is touched before the wait" in r.stdout, r.stdout[-400:]


def _wino_schedule_from_source():
    """What the queue model below needs, READ FROM conv_wino.hip instead of restated: the order of the request / wait /
    multiply / transform events of a period (the token sequence of the X_PERIOD macro), the immediates of the hand-written
    `s_waitcnt vmcnt(N)`, and how many requests each macro issues (counted in the macros' own text).  A reordered or
    recounted source changes the model's input -- or fails the structural asserts here -- instead of silently leaving a
    stale restatement green (round-4 advisor finding)."""
    src = open(os.path.join(ROOT, "n-hans_amd", "csrc", "conv_wino.hip")).read()

    def macro(name):
        """text of `#define <name> ...` with its continuation lines"""
        lines = src.split("\n")
        i = next(k for k, l in enumerate(lines) if l.startswith("#define " + name))
        j = i
        while lines[j].rstrip().endswith("\\"):
            j += 1
        return "\n".join(lines[i:j + 1])

    dma2, dma, b2, b2c, b1c, mult, xform, period = (macro(n) for n in ("X_DMA2(CC, RB, K0)", "X_DMA(CC, RB)", "X_LOAD_B2(S, J, UPTR)",
                                                                          "X_LOAD_B2C(S, J, UPTR, SKIP)", "X_LOAD_B1C(CC, SKIP)",
                                                                          "X_MULTIPLY(VB, CN, LAST)", "X_TRANSFORM(RB, VB, BC, DC, DB, DO_T, REQ)",
                                                                          "X_PERIOD(C_, VB)"))
    n = {}
    assert dma2.count("raw_ptr_buffer_load_lds") == 1 and "k < (K0) + 2" in dma2
    n["dma2"] = 2
    n["dma"] = dma.count("X_DMA2(") * n["dma2"]
    n["b2"] = b2.count("global_load_dwordx4")
    assert b2c.count("global_load_dwordx4") == n["b2"]
    n["b1c"] = b1c.count("X_LOAD_B2C(") * n["b2"]
    n["mult_loads"] = mult.count("global_load_dwordx4")              # the next chunk's k-step-0 weights, inside the block
    n["mult_wait"] = int(re.search(r"s_waitcnt vmcnt\((\d+)\)", mult).group(1))
    # requests of a transform by its REQ argument, from the macro's own conditionals (ep = 0, 1)
    for want in ("if ((REQ) == 2) X_LOAD_B2(0, 0, u_)", "if (ep == 0) { if ((REQ) == 2) X_LOAD_B2(0, 1, u_) } else if (REQ) { X_DMA2(DC, DB, 0) }",
                 "if (REQ) { if (ep == 0) { X_LOAD_B2(1, 0, u_) } else { X_DMA2(DC, DB, 2) } }", "if (REQ && ep == 0) { X_LOAD_B2(1, 1, u_) }"):
        assert want in xform, want
    # (in issue order) REQ 2: k-step 0 (2 x b2), k-step 1 (2 x b2), then the tile (2 x dma2); REQ 1: k-step 1, the tile
    n["xform"] = {2: [("w0", 2 * n["b2"]), ("w1", 2 * n["b2"]), ("tile", 2 * n["dma2"])], 1: [("w1", 2 * n["b2"]), ("tile", 2 * n["dma2"])], 0: []}
    toks = re.findall(r"if \(early\) X_MULTIPLY|if \(!early && more_\) X_DMA\(c_ \+ 3|if \(more_\)|X_TRANSFORM\([^)]*\(early \? 1 : 0\)\)|if \(!early\)|"
                      r"if \(!more_\) asm volatile\(\"s_waitcnt vmcnt\((\d+)\)\"|X_MULTIPLY\(VB|X_LOAD_B1C\(c_ \+ 1|s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)|s_barrier", period)
    kinds = re.findall(r"if \(early\) X_MULTIPLY|if \(!early && more_\) X_DMA|if \(more_\)|X_TRANSFORM|if \(!early\)|if \(!more_\) asm|X_MULTIPLY\(VB|X_LOAD_B1C|vmcnt\(\d+\) lgkmcnt|s_barrier", period)
    assert kinds == ["if (early) X_MULTIPLY", "if (!early && more_) X_DMA", "if (more_)", "X_TRANSFORM", "if (!early)", "if (!more_) asm", "X_MULTIPLY(VB",
                     "X_LOAD_B1C", "vmcnt(12) lgkmcnt" if False else kinds[8], "s_barrier"], kinds
    n["late_last_wait"] = int([t[0] for t in toks if t[0]][0])
    n["period_wait"] = int([t[1] for t in toks if t[1]][0])
    # prologue: X_DMA(0, 0); vmcnt(0); barrier; X_TRANSFORM(0, 0, 0, 1, 1, true, 2); X_DMA(2, 2); vmcnt(N)
    pro = src[src.index("// ---- prologue: ONLY tile 0"):src.index("// ---- K loop.")]
    assert re.search(r"X_DMA\(0, 0\)", pro) and "X_TRANSFORM(0, 0, 0, 1, 1, true, 2)" in pro and "X_DMA(2, 2)" in pro
    n["prologue_wait"] = int(re.findall(r"s_waitcnt vmcnt\((\d+)\) lgkmcnt\(0\)", pro)[0])
    return n


def test_winograd_kernel_wait_counts_by_model():
    """The counts of conv_wino.hip's hand-written `s_waitcnt vmcnt(N)` as a queue model (requests of a wave return in
    order; `vmcnt(N)` = at most the N youngest may still be in flight), checked for both kinds of wave and every chunk
    count: when a multiply block starts, the chunk's 8 weight requests have landed; when a period ends, the tile the NEXT
    period transforms has.  The ISA check looks at registers, not at counts: the first version of "the second-multiplying
    waves request their tile at the start of the period" passed it and let the last chunk's k-step-1 weights stay in
    flight (round 4; the bitwise-reproducibility test caught it on the GPU).  The event order, the wait immediates and the
    requests per macro come from the SOURCE (_wino_schedule_from_source)."""
    S = _wino_schedule_from_source()

    def run(nc, early):
        q, done = [], set()                      # in-flight requests in issue order; landed tags

        def issue(tag, n):
            q.extend([tag] * n)

        def wait(n):
            while len(q) > n:
                done.add(q.pop(0))

        def landed(tag):                             # ALL requests of the tag (a tag has several)
            return tag in done and tag not in q
        # prologue: tile 0, wait; transform chunk 0 with all of chunk 0's weights and tile 1 behind it; tile 2; wait for tile 1
        issue(("tile", 0), S["dma"]); wait(0)
        for kind, cnt in S["xform"][2]:
            issue((kind, 0) if kind != "tile" else ("tile", 1), cnt)
        issue(("tile", 2), S["dma"])
        wait(S["prologue_wait"])
        assert landed(("tile", 1)) and landed(("w0", 0)) and landed(("w1", 0))
        for c in range(nc):
            more = c + 1 < nc

            def multiply():
                if not early and not more:
                    wait(S["late_last_wait"])
                wait(S["mult_wait"])                 # opens the multiply block
                assert landed(("w0", c)) and landed(("w1", c)), (nc, early, c, list(q))
                if more:
                    issue(("w0", c + 1), S["mult_loads"])   # k-step 0 of the next chunk, inside the block

            def transform():
                if more:
                    assert landed(("tile", c + 1)), (nc, early, c)
                    for kind, cnt in S["xform"][1 if early else 0]:
                        issue((kind, c + 1) if kind != "tile" else ("tile", c + 3), cnt)
            if early:
                multiply(); transform()
            else:
                if more:
                    issue(("tile", c + 3), S["dma"])  # at the start of the period
                transform(); multiply()
                if more:
                    issue(("w1", c + 1), S["b1c"])    # k-step 1
            wait(S["period_wait"])                    # end of the period
            if c + 2 < nc:
                assert landed(("tile", c + 2)), (nc, early, c)
        wait(0)

    for nc in (2, 4, 8, 16, 32):
        for early in (True, False):
            run(nc, early)


def test_blob_cache_round_trip_and_staleness(tmp_path, monkeypatch):
    """blobcache.py: an entry comes back byte for byte with its exponents; a truncated blob, a blob of another packing
    version and an entry without its json are ignored (None -> the caller folds again); the key moves with the model kind,
    the seed, the ABI version and fold.BLOB_VERSION."""
    from nhans_amd import blobcache, fold
    monkeypatch.setenv("NHANS_CACHE_DIR", str(tmp_path))
    arrays = {"zero": np.zeros(64, np.float32), "tw400": np.arange(800, dtype=np.float32)}
    blob = fold.write_blob(arrays)
    assert fold.blob_header_ok(blob) and not fold.blob_header_ok(blob[:-1]) and not fold.blob_header_ok(b"x" * 100)
    assert blobcache.load("k1") is None
    assert blobcache.store("k1", blob, list(range(hip.NUM_ACTIVATIONS)))
    got = blobcache.load("k1")
    assert got is not None and got[0].tobytes() == blob and got[1] == list(range(hip.NUM_ACTIVATIONS))
    with open(os.path.join(str(tmp_path), "k1.blob"), "r+b") as f:
        f.truncate(len(blob) - 256)
    assert blobcache.load("k1") is None
    other = bytearray(blob)
    other[8:12] = (fold.BLOB_VERSION + 1).to_bytes(4, "little")
    assert blobcache.store("k2", bytes(other)) and blobcache.load("k2") is None
    assert blobcache.store("k3", blob) and blobcache.load("k3")[1] is None          # (no exponents yet)
    os.remove(os.path.join(str(tmp_path), "k3.json"))
    assert blobcache.load("k3") is None
    keys = {blobcache.key_for_synthetic("denoiser", 7), blobcache.key_for_synthetic("separator", 7), blobcache.key_for_synthetic("denoiser", 8)}
    monkeypatch.setattr(fold, "BLOB_VERSION", fold.BLOB_VERSION + 1)
    keys.add(blobcache.key_for_synthetic("denoiser", 7))
    monkeypatch.setattr(hip, "ABI_VERSION", hip.ABI_VERSION + 1)
    keys.add(blobcache.key_for_synthetic("denoiser", 7))
    assert len(keys) == 5
    # a TensorFlow bundle is keyed by its .index bytes (shapes, offsets, per-tensor crc32c) and its data size
    import shutil
    pre = os.path.join(str(tmp_path), "m")
    shutil.copyfile(os.path.join(GOLDEN, "denoiser.index"), pre + ".index")
    open(pre + ".data-00000-of-00001", "wb").write(b"0" * 134)
    k_a = blobcache.key_for_checkpoint(pre, "denoiser")
    open(pre + ".data-00000-of-00001", "wb").write(b"0" * 135)
    assert blobcache.key_for_checkpoint(pre, "denoiser") != k_a


def test_the_command_line_modules_do_not_import_torch():
    """lite.py / hiprt.py / blobcache.py / apply.py must stay importable without PyTorch: the one-file command line never
    pays for `import torch` (checked end to end on the GPU box: tests/test_gpu_cold_call.py)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import nhans_amd; from nhans_amd import apply, blobcache, hiprt, lite; "
            "assert 'torch' not in sys.modules, 'torch imported'; print('ok')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr
