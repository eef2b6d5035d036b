'''GPU box: host-to-device copy rate of 96 MB from pageable and from page-locked memory (what mpt_build_tree's upload could gain)'''
import ctypes as C, time, numpy as np
hip = C.CDLL('/opt/rocm/lib/libamdhip64.so')
n = 96 * 1000 * 1000
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), C.c_size_t(n)) == 0
a = np.ones(n, np.uint8)
h = C.c_void_p()
assert hip.hipHostMalloc(C.byref(h), C.c_size_t(n), 0) == 0
C.memmove(h, a.ctypes.data, n)
for name, src in (('pageable', C.c_void_p(a.ctypes.data)), ('pinned', h)):
    ts = []
    for _ in range(6):
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        assert hip.hipMemcpy(d, src, C.c_size_t(n), 1) == 0
        hip.hipDeviceSynchronize()
        ts.append(time.perf_counter() - t0)
    print(name, 'ms', [round(t * 1e3, 3) for t in ts], 'GB/s', round(n / min(ts) / 1e9, 1))
