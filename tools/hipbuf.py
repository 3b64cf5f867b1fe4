"""Device buffers without torch (measurement scripts that run under rocprofv3, where torch's own kernels crash the
profiler on this image): hipMalloc / hipMemcpy / hipFree through ctypes."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipDeviceSynchronize.argtypes = []
    return _hip


class DevBuf:
    def __init__(self, arr=None, nbytes=None):
        self.nbytes = int(arr.nbytes if arr is not None else nbytes)
        self.p = C.c_void_p()
        rc = hip().hipMalloc(C.byref(self.p), max(self.nbytes, 16))
        if rc != 0:
            raise RuntimeError("hipMalloc failed: %d" % rc)
        if arr is not None:
            a = np.ascontiguousarray(arr)
            rc = hip().hipMemcpy(self.p, a.ctypes.data_as(C.c_void_p), self.nbytes, 1)  # hipMemcpyHostToDevice
            if rc != 0:
                raise RuntimeError("hipMemcpy H2D failed: %d" % rc)

    @property
    def ptr(self):
        return self.p.value

    def to_host(self, dtype=np.uint32):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype)
        rc = hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.p, self.nbytes, 2)  # hipMemcpyDeviceToHost
        if rc != 0:
            raise RuntimeError("hipMemcpy D2H failed: %d" % rc)
        return out

    def free(self):
        if self.p:
            hip().hipFree(self.p)
            self.p = C.c_void_p()
