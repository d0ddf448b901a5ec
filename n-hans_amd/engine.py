"""Device-side engine: owns one libnhans_hip context and moves ragged clip batches through it.
PyTorch is used only for device memory and streams; all arithmetic runs in the HIP library."""
import ctypes
import os
import warnings

import numpy as np
import torch

from . import fold, hip, spec, weights as weights_mod


def _offsets(lengths):
    off = [0]
    for n in lengths:
        off.append(off[-1] + int(n))
    return off


class Engine:
    """kind: 'denoiser' | 'separator'.  weights: checkpoint dict (name -> float32 array) or None for
    the seeded synthetic weights.  Conditioning order everywhere is (a, b) = resnet_block argument
    order: denoiser (pos, neg); separator (noise = --neg, clean = --pos)."""

    PRECISIONS = {"f32": 0, "f16x3": 1}

    def __init__(self, kind=spec.DENOISER, weights=None, device=0, seed=7, frames_per_chunk=None,
                 precision=None):
        if not torch.cuda.is_available():
            raise hip.NhansError("no HIP device visible: the N-HANS hot path has no CPU fallback")
        self.lib = hip.load()
        self.kind = kind
        self.device = torch.device("cuda", device)
        if weights is None:
            weights = weights_mod.synthetic_weights(kind, seed)
        blob = fold.fold_weights(weights, kind)
        handle = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(blob, len(blob))
        hip.check(self.lib.nhans_create(hip.KIND_CODE[kind], buf, len(blob), device, ctypes.byref(handle)))
        self.handle = handle
        if frames_per_chunk:
            self.set_option("frames_per_chunk", frames_per_chunk)
        self.set_precision(precision)
        if "NHANS_CONV_VARIANT" in os.environ:
            self.set_option("conv_variant", int(os.environ["NHANS_CONV_VARIANT"]))

    def set_precision(self, precision):
        """'f32': exact f32 matrix-core path.  'f16x3': split-f16 (hi+lo, three products) on the f16
        matrix cores -- FP32-class accuracy, needs |activations| < 65504."""
        if precision is None:
            precision = os.environ.get("NHANS_PRECISION", "f16x3")
        self.set_option("precision", self.PRECISIONS[precision])
        self.precision = precision

    def close(self):
        if getattr(self, "handle", None):
            self.lib.nhans_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        hip.check(self.lib.nhans_set_option(self.handle, key.encode(), int(value)))

    # ---- activation exponents of the f16x3 mode (include/nhans_hip.h: "calibrate") ----------
    def activation_exponents(self):
        e = (ctypes.c_int * hip.NUM_ACTIVATIONS)()
        hip.check(self.lib.nhans_get_activation_exponents(self.handle, e, hip.NUM_ACTIVATIONS))
        return list(e)

    def set_activation_exponents(self, exps):
        e = (ctypes.c_int * hip.NUM_ACTIVATIONS)(*[int(v) for v in exps])
        hip.check(self.lib.nhans_set_activation_exponents(self.handle, e, hip.NUM_ACTIVATIONS))

    def activation_amax(self):
        """Largest |x| of every exponent-carrying tensor in the last finished calibration."""
        a = (ctypes.c_float * hip.NUM_ACTIVATIONS)()
        hip.check(self.lib.nhans_get_activation_amax(self.handle, a, hip.NUM_ACTIVATIONS))
        return list(a)

    def calibrate(self, mixes, ctx_a, ctx_b, raise_only=False):
        """Sets the activation exponents from the caller's own clips (nhans_create has calibrated on a built-in
        signal): one f32-mode pass with every tensor's maximum recorded."""
        before = self.precision
        self.set_option("calibrate", 1)
        try:
            self.set_precision("f32")
            mix_t, mix_off = self._dev(mixes)
            ca_t, ca_off = self._dev(ctx_a)
            cb_t, cb_off = self._dev(ctx_b)
            self.enhance_device(mix_t, mix_off, ca_t, ca_off, cb_t, cb_off)
        except BaseException:
            # the pass failed: close the bracket FIRST (3 = keep the exponents, learn nothing) -- an open bracket keeps
            # recording maxima on every later call --, then put the precision back; neither may mask the original error
            try:
                self.set_option("calibrate", 3)
            finally:
                self.set_precision(before)
            raise
        try:
            self.set_option("calibrate", 2 if raise_only else 0)
        finally:
            self.set_precision(before)
        return self.activation_exponents()

    def take_status(self):
        """Waits for the current stream; returns and clears the sticky device status bits
        (hip.STATUS_SATURATED: a split-f16 activation left the f16 range and was clamped)."""
        flags = ctypes.c_int(0)
        hip.check(self.lib.nhans_take_status(self.handle, ctypes.byref(flags), self._stream()))
        return flags.value

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, arrays):
        """list of 1-D float32 numpy arrays -> (concatenated device tensor, offsets)."""
        off = _offsets([len(a) for a in arrays])
        flat = np.concatenate([np.asarray(a, dtype=np.float32) for a in arrays]) if arrays else np.zeros(0, np.float32)
        return torch.from_numpy(flat).to(self.device), off

    # ---- stage-level entry points (used by the parity tests) --------------------------------
    def stft_features(self, wav_t, sample_off, max_frames=0, want_phase=True):
        n = len(sample_off) - 1
        tot = 0
        for i in range(n):
            t = int(self.lib.nhans_num_frames(sample_off[i + 1] - sample_off[i]))
            tot += min(t, max_frames) if max_frames > 0 else t
        lm = torch.empty((tot, spec.BINS), dtype=torch.float32, device=self.device)
        ph = torch.empty_like(lm) if want_phase else None
        hip.check(self.lib.nhans_stft_features(self.handle, hip.ptr(wav_t), hip.i64_array(sample_off), n,
                                               max_frames, hip.ptr(lm), hip.ptr(ph), self._stream()))
        return lm, ph

    def embed(self, ctx_lm):
        n = ctx_lm.shape[0]
        out = torch.empty((n, spec.EMB), dtype=torch.float32, device=self.device)
        hip.check(self.lib.nhans_embed(self.handle, hip.ptr(ctx_lm.contiguous()), n, hip.ptr(out), self._stream()))
        return out

    def mask_net(self, logmag, frame_off, emb_a, emb_b, want_logits=True):
        den = torch.empty_like(logmag)
        lg = torch.empty_like(logmag) if want_logits else None
        hip.check(self.lib.nhans_mask_net(self.handle, hip.ptr(logmag), hip.i64_array(frame_off), len(frame_off) - 1,
                                          hip.ptr(emb_a.contiguous()), hip.ptr(emb_b.contiguous()), hip.ptr(lg),
                                          hip.ptr(den), self._stream()))
        return lg, den

    def block_output(self, logmag, frame_off, emb_a, emb_b, frame0, nframes, block):
        g = (spec.main_geometry() + [dict(hout=1, wout=26, cout=512)])[block]
        out = torch.empty((nframes, g["hout"], g["wout"], g["cout"]), dtype=torch.float32, device=self.device)
        hip.check(self.lib.nhans_debug_block_output(
            self.handle, hip.ptr(logmag), hip.i64_array(frame_off), len(frame_off) - 1, hip.ptr(emb_a.contiguous()),
            hip.ptr(emb_b.contiguous()), frame0, nframes, block, hip.ptr(out), self._stream()))
        return out

    def istft(self, logmag, phase, frame_off):
        lens = [(frame_off[i + 1] - frame_off[i] - 1) * spec.HOP + spec.WIN if frame_off[i + 1] > frame_off[i] else 0
                for i in range(len(frame_off) - 1)]
        ooff = _offsets(lens)
        out = torch.zeros(ooff[-1], dtype=torch.float32, device=self.device)
        hip.check(self.lib.nhans_istft(self.handle, hip.ptr(logmag), hip.ptr(phase), hip.i64_array(frame_off),
                                       len(frame_off) - 1, hip.i64_array(ooff), hip.ptr(out), self._stream()))
        return out, ooff

    # ---- whole path -------------------------------------------------------------------------
    def enhance_device(self, mix_t, mix_off, ca_t, ca_off, cb_t, cb_off, want_mixed=False, taps=False):
        """All inputs already in HBM.  Returns dict of device tensors."""
        n = len(mix_off) - 1
        nfr = [int(self.lib.nhans_num_frames(mix_off[i + 1] - mix_off[i])) for i in range(n)]
        if any(t == 0 for t in nfr):
            # (the reference dies on such a clip too: its STFT has no frames to stack)
            raise ValueError("mixture clip %d has fewer than %d samples: no STFT frame" % (nfr.index(0), spec.WIN))
        total = sum(nfr)
        res = {"denoised_wav": torch.zeros(mix_off[-1], dtype=torch.float32, device=self.device)}
        res["mixed_wav"] = torch.zeros_like(res["denoised_wav"]) if want_mixed else None
        if taps:
            for k in ("logmag", "phase", "logits"):
                res[k] = torch.empty((total, spec.BINS), dtype=torch.float32, device=self.device)
            res["emb"] = torch.empty((2 * n, spec.EMB), dtype=torch.float32, device=self.device)
        hip.check(self.lib.nhans_enhance_clips(
            self.handle, hip.ptr(mix_t), hip.i64_array(mix_off), n, hip.ptr(ca_t), hip.i64_array(ca_off),
            hip.ptr(cb_t), hip.i64_array(cb_off), hip.ptr(res["denoised_wav"]), hip.ptr(res["mixed_wav"]),
            hip.ptr(res.get("logmag")), hip.ptr(res.get("phase")), hip.ptr(res.get("logits")),
            hip.ptr(res.get("emb")), self._stream()))
        return res

    def enhance(self, mixes, ctx_a, ctx_b, want_mixed=True, taps=False):
        """Lists of normalised float32 waveforms (mixtures trimmed) -> per-clip numpy results."""
        mix_t, mix_off = self._dev(mixes)
        ca_t, ca_off = self._dev(ctx_a)
        cb_t, cb_off = self._dev(ctx_b)
        res = self.enhance_device(mix_t, mix_off, ca_t, ca_off, cb_t, cb_off, want_mixed, taps)
        if self.take_status() & hip.STATUS_SATURATED and self.precision == "f16x3":
            # the split-f16 layout holds |activation * 2^-e| < 65504 and this batch is further from the calibration
            # than the 2^8 of headroom: redo it on the exact f32 matrix-core path (same library, no CPU involved)
            # with the tensors' maxima recorded, and raise the exponents so that the batches after it fit
            warnings.warn("N-HANS f16x3 path: an activation left the f16 range; batch recomputed in f32 MFMA mode "
                          "and the activation exponents raised")
            self.set_option("calibrate", 1)
            try:
                self.set_precision("f32")
                res = self.enhance_device(mix_t, mix_off, ca_t, ca_off, cb_t, cb_off, want_mixed, taps)
                self.take_status()
            except BaseException:
                try:
                    self.set_option("calibrate", 3)      # close the bracket first, then restore the precision
                finally:
                    self.set_precision("f16x3")
                raise
            try:
                # (raise-only; maxima that are not finite -- the flag is also raised by a NaN / Inf INPUT -- are skipped)
                self.set_option("calibrate", 2)
            except hip.NhansError as err:          # the f32 result stands whatever the exponent update says
                warnings.warn("N-HANS: activation exponents not updated after the f32 rerun: %s" % err)
            finally:
                self.set_precision("f16x3")
        torch.cuda.synchronize(self.device)
        out = {"denoised_wav": [], "mixed_wav": []}
        den = res["denoised_wav"].cpu().numpy()
        mixed = res["mixed_wav"].cpu().numpy() if want_mixed else None
        for i in range(len(mixes)):
            out["denoised_wav"].append(den[mix_off[i]:mix_off[i + 1]])
            if want_mixed:
                out["mixed_wav"].append(mixed[mix_off[i]:mix_off[i + 1]])
        if taps:
            for k in ("logmag", "phase", "logits", "emb"):
                out[k] = res[k].cpu().numpy()
        return out

    def profile(self):
        return hip.profile_dict(self.handle)

    def profile_reset(self):
        hip.check(self.lib.nhans_profile_reset(self.handle))
