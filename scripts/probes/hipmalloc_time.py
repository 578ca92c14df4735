"""GPU box: wall time of hipMalloc / hipFree by size (what the first prefilter block pays for its working buffers)."""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), 1 << 20); hip.hipFree(p)
for gb in (0.25, 1, 2, 4, 8, 16):
    n = int(gb * (1 << 30))
    t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), n); t1 = time.perf_counter()
    hip.hipMemset(p, 0, ctypes.c_size_t(n)); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    hip.hipFree(p); t3 = time.perf_counter()
    print("%.2f GB: hipMalloc %.1f ms (rc %d), first memset %.1f ms, hipFree %.1f ms" % (gb, (t1 - t0) * 1e3, rc, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
