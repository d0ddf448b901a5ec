"""The handful of HIP runtime calls the torch-free CLI path needs (device memory and copies), bound with ctypes to the
SAME libamdhip64 that libnhans_hip.so is linked against: dlopen by soname returns the copy already in the process
(torch's, if torch was imported first; /opt/rocm's otherwise).  PyTorch stays what it is elsewhere in this package --
plumbing for device memory, streams and torch.distributed -- but importing it costs ~1.5 s, more than everything else a
single-file `nhans_denoiser` call does together (cold call)."""
import ctypes

from . import hip

_rt = None
H2D, D2H = 1, 2


class HipError(hip.NhansError):
    pass


def rt():
    global _rt
    if _rt is None:
        hip.load()                                   # (pulls libamdhip64.so.7 in through its DT_NEEDED / RUNPATH)
        lib = ctypes.CDLL("libamdhip64.so.7")
        lib.hipGetErrorString.restype = ctypes.c_char_p
        lib.hipGetErrorString.argtypes = [ctypes.c_int]
        lib.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        lib.hipFree.argtypes = [ctypes.c_void_p]
        lib.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        lib.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        lib.hipSetDevice.argtypes = [ctypes.c_int]
        lib.hipGetDeviceCount.argtypes = [ctypes.POINTER(ctypes.c_int)]
        _rt = lib
    return _rt


def check(rc, what):
    if rc != 0:
        raise HipError("%s: %s" % (what, rt().hipGetErrorString(rc).decode()))


def device_count():
    n = ctypes.c_int(0)
    rc = rt().hipGetDeviceCount(ctypes.byref(n))
    return n.value if rc == 0 else 0


class DevBuf:
    """nbytes of device memory; freed with the object."""

    def __init__(self, nbytes, zero=False):
        self.ptr = ctypes.c_void_p()
        self.nbytes = int(nbytes)
        check(rt().hipMalloc(ctypes.byref(self.ptr), max(self.nbytes, 4)), "hipMalloc(%d)" % self.nbytes)
        if zero and self.nbytes:
            check(rt().hipMemset(self.ptr, 0, self.nbytes), "hipMemset")

    @classmethod
    def from_array(cls, a):
        """contiguous numpy array -> device copy"""
        b = cls(a.nbytes)
        if a.nbytes:
            check(rt().hipMemcpy(b.ptr, a.ctypes.data_as(ctypes.c_void_p), a.nbytes, H2D), "hipMemcpy H2D")
        return b

    def to_array(self, out):
        """device -> the contiguous numpy array `out` (blocks until every earlier launch on the null stream is done)"""
        if out.nbytes:
            check(rt().hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), self.ptr, out.nbytes, D2H), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            rt().hipFree(self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
