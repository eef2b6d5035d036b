#!/usr/bin/env python3
'''GPU-box soak of mpt_build_tree (round 6): the LBVH fit's and the prims kernel's hand-overs between workgroups are ordered by hand
(agent-scope atomic accesses + s_waitcnt, no L2 write-back: lbvh_build.hip fit_boxes_kernel, sah_build.hip sb_prims_kernel), so
they are checked on DATA and under load: every model is built REPS times, beside a stream of device-to-device copies
(mpt_stress_copies), and the reference-shaped LBVH (child, bmin, bmax, depth) and the 4-wide records the kernels walk must be the
first build's bit for bit every time.
usage: tools/build_soak.py [reps]'''
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ctypes as C          # noqa: E402
import numpy as np          # noqa: E402
from ptina_amd import scenes, _lib                     # noqa: E402
from ptina_amd.common import ctx, reset_all            # noqa: E402
from ptina_amd.things import init_things, ModelPool, BVHTree   # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
t_start = time.time()
for n, seed in ((1000000, 12346), (99382, 0), (20000, 5), (1025, 6), (700, 7)):
    reset_all()
    if seed == 0:
        v, m, _, _ = scenes.get_scene('c4')
        n = m.shape[0]
    else:
        v, m, _, _ = scenes.scene_random_tris(n, seed=seed, edge=0.05 if n < 500000 else 0.02)
    init_things(max_faces=n + 1)
    ModelPool().load(v, m)
    c = ctx()
    c.set_option('sah_build', 1)
    first = None
    for r in range(reps):
        if r % 2 == 1:
            try:
                c.call('mpt_stress_copies', 256, 8)          # 8 copies of 256 MiB beside the build
            except Exception:
                pass
        if r % 5 == 4:
            ModelPool().load(v, m)                          # (the upload path as well)
        BVHTree().build()
        t = BVHTree().to_numpy()
        nw = C.c_int(0)
        c.call('mpt_get_wide', None, None, 0, C.byref(nw))
        w = np.zeros((nw.value, 8, 4), np.float32)
        q = np.zeros((nw.value, 4, 4), np.float32)
        c.call('mpt_get_wide', _lib.fptr(w), _lib.fptr(q), nw.value, C.byref(nw))
        cur = (t['child'].copy(), t['bmin'].view(np.uint32).copy(), t['bmax'].view(np.uint32).copy(), int(t['depth']), w.view(np.uint32).copy(),
               q.view(np.uint32).copy(), c.get_option('fast_depth'))
        if first is None:
            first = cur
        else:
            same = all(np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b for a, b in zip(first, cur))
            if not same:
                bad += 1
                print(f'n {n}: build {r} differs from build 0', flush=True)
    print(f'n {n}: {reps} builds, depth {first[3]} / {first[6]}, {first[4].shape[0]} wide nodes, fallback {c.get_option("sah_fallback")}, '
          f'{time.time() - t_start:.0f} s', flush=True)
reset_all()
print('bad', bad)
sys.exit(1 if bad else 0)
