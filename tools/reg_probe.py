#!/usr/bin/env python3
"""What moving a caller's pageable arrays costs on this box (why rio_gp_place_pending registers big host batches for the call):
hipHostRegister / hipHostUnregister per call, hipMemcpy H2D / D2H from registered against pageable memory, plain memcpy."""
import ctypes as C, time, numpy as np
hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p
hip.hipHostRegister.argtypes = [vp, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [vp]
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
hip.hipStreamSynchronize.argtypes = [vp]
hip.hipInit(0); hip.hipSetDevice(0)
d = vp(); hip.hipMalloc(C.byref(d), 8 << 20)
for mb in (0.25, 1, 2):
    n = int(mb * (1 << 20))
    a = np.ones(n, np.uint8); b = np.empty(n, np.uint8)
    rec = {}
    for reg in (False, True):
        ts, tr = [], []
        for _ in range(30):
            t0 = time.perf_counter()
            if reg:
                assert hip.hipHostRegister(a.ctypes.data, n, 0) == 0 and hip.hipHostRegister(b.ctypes.data, n, 0) == 0
            t1 = time.perf_counter()
            assert hip.hipMemcpyAsync(d, a.ctypes.data, n, 1, None) == 0
            assert hip.hipMemcpyAsync(b.ctypes.data, d, n, 2, None) == 0
            hip.hipStreamSynchronize(None)
            t2 = time.perf_counter()
            if reg:
                hip.hipHostUnregister(a.ctypes.data); hip.hipHostUnregister(b.ctypes.data)
            t3 = time.perf_counter()
            ts.append(t2 - t1); tr.append((t1 - t0) + (t3 - t2))
        rec["registered" if reg else "pageable"] = "H2D+D2H %.1f us, register+unregister %.1f us" % (np.median(ts) * 1e6, np.median(tr) * 1e6)
    assert (b == 1).all()
    t0 = time.perf_counter()
    for _ in range(30): C.memmove(b.ctypes.data, a.ctypes.data, n)
    rec["memcpy"] = "%.1f us" % ((time.perf_counter() - t0) / 30 * 1e6)
    print(mb, "MB each way:", rec)
