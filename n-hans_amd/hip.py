"""ctypes binding of libnhans_hip.so (C ABI in include/nhans_hip.h).

There is no CPU fallback: if the shared library is missing or fails to load, every use raises.
"""
import ctypes
import json
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# ($NHANS_LIB: another build of the same library for a same-box A/B of two kernels -- a developer convenience of this
# Python binding; the library itself reads no environment)
LIB_PATH = os.environ.get("NHANS_LIB") or os.path.join(_HERE, "csrc", "libnhans_hip.so")

DENOISER, SEPARATOR = 0, 1
KIND_CODE = {"denoiser": DENOISER, "separator": SEPARATOR}

EXPORTS = [
    "nhans_abi_version", "nhans_last_error", "nhans_num_frames", "nhans_create", "nhans_create_ex", "nhans_destroy",
    "nhans_set_option", "nhans_workspace_bytes", "nhans_stft_features", "nhans_embed",
    "nhans_mask_net", "nhans_istft", "nhans_enhance_clips", "nhans_debug_block_output",
    "nhans_profile_json", "nhans_profile_reset", "nhans_take_status", "nhans_debug_launch_probe", "nhans_crc32c",
    "nhans_debug_mfma_ceiling", "nhans_set_activation_exponents", "nhans_get_activation_exponents",
    "nhans_get_activation_amax",
]
STATUS_SATURATED = 1
NUM_ACTIVATIONS = 25
ABI_VERSION = 5

_lib = None


class NhansError(RuntimeError):
    pass


def load():
    """Load the HIP library once; raises NhansError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NhansError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "or `make -C n-hans_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    # With torch in the process its bundled libamdhip64.so.7 must be THE HIP runtime (two runtimes in one process do not
    # share devices or streams): torch is imported first unless the caller has said this process stays torch-free
    # (NHANS_NO_TORCH=1, set by the single-process command line: lite.py) -- importing it costs more than a one-file call.
    if os.environ.get("NHANS_NO_TORCH") != "1" or "torch" in sys.modules:
        import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)
    lib.nhans_abi_version.restype = ctypes.c_int
    lib.nhans_last_error.restype = ctypes.c_char_p
    lib.nhans_num_frames.restype = ctypes.c_int64
    lib.nhans_num_frames.argtypes = [ctypes.c_int64]
    lib.nhans_create.argtypes = [ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(vp)]
    lib.nhans_create_ex.argtypes = [ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                    ctypes.POINTER(vp)]
    lib.nhans_destroy.argtypes = [vp]
    lib.nhans_destroy.restype = None
    lib.nhans_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64]
    lib.nhans_workspace_bytes.argtypes = [vp, ctypes.c_int64, ctypes.c_int]
    lib.nhans_workspace_bytes.restype = ctypes.c_size_t
    lib.nhans_stft_features.argtypes = [vp, vp, i64p, ctypes.c_int, ctypes.c_int, vp, vp, vp]
    lib.nhans_embed.argtypes = [vp, vp, ctypes.c_int, vp, vp]
    lib.nhans_mask_net.argtypes = [vp, vp, i64p, ctypes.c_int, vp, vp, vp, vp, vp]
    lib.nhans_istft.argtypes = [vp, vp, vp, i64p, ctypes.c_int, i64p, vp, vp]
    lib.nhans_enhance_clips.argtypes = [vp, vp, i64p, ctypes.c_int, vp, i64p, vp, i64p, vp, vp, vp, vp, vp, vp, vp]
    lib.nhans_debug_block_output.argtypes = [vp, vp, i64p, ctypes.c_int, vp, vp, ctypes.c_int64, ctypes.c_int,
                                             ctypes.c_int, vp, vp]
    lib.nhans_profile_json.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
    lib.nhans_profile_reset.argtypes = [vp]
    lib.nhans_take_status.argtypes = [vp, ctypes.POINTER(ctypes.c_int), vp]
    lib.nhans_debug_launch_probe.argtypes = [ctypes.c_size_t, vp]
    lib.nhans_debug_mfma_ceiling.argtypes = [ctypes.c_double, vp, ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    lib.nhans_debug_mfma_ceiling.restype = ctypes.c_int
    lib.nhans_set_activation_exponents.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.nhans_get_activation_exponents.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.nhans_get_activation_amax.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.c_int]
    lib.nhans_crc32c.argtypes = [ctypes.c_uint32, vp, ctypes.c_size_t]
    lib.nhans_crc32c.restype = ctypes.c_uint32
    for name in ("nhans_create", "nhans_create_ex", "nhans_set_option", "nhans_stft_features", "nhans_embed", "nhans_mask_net",
                 "nhans_istft", "nhans_enhance_clips", "nhans_debug_block_output", "nhans_profile_json",
                 "nhans_profile_reset", "nhans_take_status", "nhans_debug_launch_probe",
                 "nhans_set_activation_exponents", "nhans_get_activation_exponents", "nhans_get_activation_amax"):
        getattr(lib, name).restype = ctypes.c_int
    if lib.nhans_abi_version() != ABI_VERSION:
        raise NhansError("libnhans_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc < 0:
        raise NhansError("libnhans_hip: %s (code %d)" % (load().nhans_last_error().decode(), rc))
    return rc


def i64_array(values):
    arr = (ctypes.c_int64 * len(values))(*[int(v) for v in values])
    return arr


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def mfma_ceiling(seconds, stream=None):
    """-> dict(sustained_tflops, first_launch_tflops, launches): the f16 MFMA rate the current device
    holds at its power cap (nhans_debug_mfma_ceiling)."""
    sus, first, n = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int(0)
    check(load().nhans_debug_mfma_ceiling(float(seconds), stream, ctypes.byref(sus), ctypes.byref(first), ctypes.byref(n)))
    return {"sustained_tflops": sus.value, "first_launch_tflops": first.value, "launches": n.value}


def profile_dict(handle):
    lib = load()
    n = lib.nhans_profile_json(handle, None, 0)
    buf = ctypes.create_string_buffer(n + 1)
    lib.nhans_profile_json(handle, buf, n + 1)
    return json.loads(buf.value.decode())
