"""LiteEngine: the whole-path entry point of the C ABI (nhans_enhance_clips) WITHOUT PyTorch -- device memory comes from
the HIP runtime through ctypes (hiprt.py), launches go to the null stream.  It is what the single-process command line
(`nhans_denoiser` / `nhans_separator` on a file or a directory, SN/apply.py:478-527, setup.py:44-50) runs on: importing
torch alone costs more wall clock than the rest of a one-file call.  Same library, same entry point, same bits as
engine.Engine.enhance (tests/test_gpu_cold_call.py); everything else -- stage-level entry points, streams, the sharded
multi-GPU path -- stays on engine.Engine.  No CPU fallback here either: no HIP device, no library -> an error."""
import ctypes
import warnings

import numpy as np

from . import blobcache, fold, hip, hiprt, spec


class LiteEngine:
    def __init__(self, kind, weights=None, device=0, blob=None, exponents=None, precision="f16x3"):
        """weights: checkpoint dict (folded here) -- or `blob`: an already folded blob (bytes / uint8 array, e.g. from
        blobcache) with, optionally, the activation exponents a previous context calibrated for it (the calibration
        pass of nhans_create is then skipped)."""
        self.lib = hip.load()
        if hiprt.device_count() <= device:
            raise hip.NhansError("no HIP device %d visible: the N-HANS hot path has no CPU fallback" % device)
        hiprt.check(hiprt.rt().hipSetDevice(device), "hipSetDevice")
        self.kind, self.device_index = kind, device
        if blob is None:
            blob = fold.fold_weights(weights, kind)
        arr = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
        self.blob = arr                                     # (kept for blobcache.store by the caller)
        handle = ctypes.c_void_p()
        exps = None if exponents is None else (ctypes.c_int * hip.NUM_ACTIVATIONS)(*[int(e) for e in exponents])
        hip.check(self.lib.nhans_create_ex(hip.KIND_CODE[kind], arr.ctypes.data_as(ctypes.c_void_p), arr.size, device,
                                           exps, hip.NUM_ACTIVATIONS if exps is not None else 0, ctypes.byref(handle)))
        self.handle = handle
        self.precision = precision
        self.set_option("precision", {"f32": 0, "f16x3": 1}[precision])

    def set_option(self, key, value):
        hip.check(self.lib.nhans_set_option(self.handle, key.encode(), int(value)))

    def activation_exponents(self):
        e = (ctypes.c_int * hip.NUM_ACTIVATIONS)()
        hip.check(self.lib.nhans_get_activation_exponents(self.handle, e, hip.NUM_ACTIVATIONS))
        return list(e)

    def take_status(self):
        flags = ctypes.c_int(0)
        hip.check(self.lib.nhans_take_status(self.handle, ctypes.byref(flags), None))
        return flags.value

    def close(self):
        if getattr(self, "handle", None):
            self.lib.nhans_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _flat(arrays):
        off = [0]
        for a in arrays:
            off.append(off[-1] + len(a))
        flat = np.concatenate([np.asarray(a, dtype=np.float32) for a in arrays]) if arrays else np.zeros(0, np.float32)
        return np.ascontiguousarray(flat), off

    def _run(self, mix, ca, cb, want_mixed):
        (mf, moff), (af, aoff), (bf, boff) = self._flat(mix), self._flat(ca), self._flat(cb)
        n = len(mix)
        for i in range(n):
            if int(self.lib.nhans_num_frames(moff[i + 1] - moff[i])) == 0:
                raise ValueError("mixture clip %d has fewer than %d samples: no STFT frame" % (i, spec.WIN))
        d_mix, d_ca, d_cb = hiprt.DevBuf.from_array(mf), hiprt.DevBuf.from_array(af), hiprt.DevBuf.from_array(bf)
        d_den = hiprt.DevBuf(mf.nbytes, zero=True)
        d_rt = hiprt.DevBuf(mf.nbytes, zero=True) if want_mixed else None
        hip.check(self.lib.nhans_enhance_clips(
            self.handle, d_mix.ptr, hip.i64_array(moff), n, d_ca.ptr, hip.i64_array(aoff), d_cb.ptr, hip.i64_array(boff),
            d_den.ptr, d_rt.ptr if d_rt else None, None, None, None, None, None))
        status = self.take_status()                       # (waits for the null stream)
        den = d_den.to_array(np.empty_like(mf))
        rt = d_rt.to_array(np.empty_like(mf)) if want_mixed else None
        for b in (d_mix, d_ca, d_cb, d_den, d_rt):
            if b is not None:
                b.free()
        return den, rt, moff, status

    def enhance(self, mixes, ctx_a, ctx_b, want_mixed=True):
        """Lists of normalised float32 waveforms (mixtures trimmed) -> {"denoised_wav": [...], "mixed_wav": [...]}; the same
        saturation fallback as engine.Engine.enhance (the batch redone on the exact-f32 matrix path, exponents raised)."""
        den, rt, off, status = self._run(mixes, ctx_a, ctx_b, want_mixed)
        if status & hip.STATUS_SATURATED and self.precision == "f16x3":
            warnings.warn("N-HANS f16x3 path: an activation left the f16 range; batch recomputed in f32 MFMA mode "
                          "and the activation exponents raised")
            self.set_option("calibrate", 1)
            try:
                self.set_option("precision", 0)
                den, rt, off, _ = self._run(mixes, ctx_a, ctx_b, want_mixed)
            except BaseException:
                try:
                    self.set_option("calibrate", 3)
                finally:
                    self.set_option("precision", 1)
                raise
            try:
                self.set_option("calibrate", 2)
            except hip.NhansError as err:
                warnings.warn("N-HANS: activation exponents not updated after the f32 rerun: %s" % err)
            finally:
                self.set_option("precision", 1)
        out = {"denoised_wav": [den[off[i]:off[i + 1]] for i in range(len(mixes))], "mixed_wav": []}
        if want_mixed:
            out["mixed_wav"] = [rt[off[i]:off[i + 1]] for i in range(len(mixes))]
        return out


def cached_engine(kind, key, make_weights, device=0, use_cache=True, timing=None):
    """LiteEngine for `kind` from the blob cache entry `key` (blobcache.key_for_*), or -- no entry, a stale one,
    use_cache False -- from make_weights() folded now, the result (blob + the exponents nhans_create calibrated)
    stored for the next process.  `timing`: optional dict that receives the seconds of each step."""
    import time
    t = time.perf_counter()
    hit = blobcache.load(key) if use_cache and key else None
    if timing is not None:
        timing["cache_read_s"] = time.perf_counter() - t
        timing["cache"] = "hit" if hit else ("miss" if use_cache else "bypassed")
    if hit:
        t = time.perf_counter()
        eng = LiteEngine(kind, device=device, blob=hit[0], exponents=hit[1])
        if timing is not None:
            timing["create_s"] = time.perf_counter() - t
        if hit[1] is None:                               # (an entry written without exponents: complete it)
            blobcache.store(key, hit[0], eng.activation_exponents())
        return eng
    t = time.perf_counter()
    W = make_weights()
    t1 = time.perf_counter()
    blob = fold.fold_weights(W, kind)
    t2 = time.perf_counter()
    eng = LiteEngine(kind, device=device, blob=blob)
    if timing is not None:
        timing.update(load_weights_s=t1 - t, fold_s=t2 - t1, create_s=time.perf_counter() - t2)
    if use_cache and key:
        blobcache.store(key, blob, eng.activation_exponents())
    return eng
