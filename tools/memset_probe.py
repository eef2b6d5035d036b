#!/usr/bin/env python3
'''Diagnostic: does hipMemset on device memory return before the zeros are there?  Times the call and the synchronisation after it.'''
import ctypes as C
import time
hip = C.CDLL('libamdhip64.so')
p = C.c_void_p()
for gib in (0.25, 2.0):
    nbytes = int(gib * (1 << 30))
    assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
    hip.hipDeviceSynchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        assert hip.hipMemset(p, 0, C.c_size_t(nbytes)) == 0
        t1 = time.perf_counter()
        hip.hipDeviceSynchronize()
        t2 = time.perf_counter()
        print('hipMemset %.2f GiB: call returned after %.3f ms, device idle after another %.3f ms' % (gib, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    hip.hipFree(p)
