#!/usr/bin/env python3
'''Diagnostic: when do the workgroups of a solo 1/8-slab launch start, per XCD (workgroup index mod 8), after different idle gaps
and with something else running just before?  usage: start_probe.py [parts]'''
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
from ptina_amd import scenes, common  # noqa: E402
from ptina_amd.common import ctx  # noqa: E402
from helpers import setup_engine  # noqa: E402

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, spp = 512, 32
r = parts // 2
eng = setup_engine(scenes.scene_s978(), n, n, mode='fast', slab=(r * n // parts, (r + 1) * n // parts))
c = ctx()
c.set_option('batch', spp)
c.set_option('timeline', 1)


def starts():
    nw = C.c_int(0)
    buf = (C.c_ulonglong * (8 * 4096))()
    c.call('mpt_get_timeline', buf, 4096, C.byref(nw))
    t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)[:nw.value].astype(np.int64)
    per_wg = t.shape[0] // 256
    s = t[:, 0].reshape(256, per_wg)[:, 0]
    e = t[:, 3].reshape(256, per_wg).max(axis=1)
    s0 = s.min()
    return [round(float((s[x::8].mean() - s0) / 100.0), 1) for x in range(8)], round(float((e.max() - s0) / 100.0), 1)


for label, gap, warm in (('gap 0', 0.0, False), ('gap 0', 0.0, False), ('gap 1 ms', 0.001, False), ('gap 20 ms', 0.02, False),
                         ('gap 0, probe kernel first', 0.0, True), ('gap 20 ms, probe kernel first', 0.02, True)):
    for rep in range(2):
        eng.render(spp)
        c.call('mpt_synchronize')
    time.sleep(gap)
    if warm:
        ms = C.c_double(0)
        c.call('mpt_probe_kernel', 256, 1024, C.byref(ms))
    eng.render(spp)
    c.call('mpt_synchronize')
    print(label, 'mean start per XCD (us):', *starts(), flush=True)
common.reset_all()
