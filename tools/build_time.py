#!/usr/bin/env python3
'''GPU: wall time of BVHTree().build() (mpt_build_tree: upload, device LBVH, SAH re-partition, 4-wide collapse, triangle
records) for the big configurations, by option.  usage: tools/build_time.py [n_triangles ...]'''
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from ptina_amd import scenes                      # noqa: E402
from ptina_amd.common import ctx, reset_all       # noqa: E402
from ptina_amd.things import BVHTree              # noqa: E402
from helpers import setup_engine                  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [100000, 1000000]:
    scene = scenes.get_scene('c5', n=n)
    reset_all()
    setup_engine(scene, 16, 16, mode='fast', max_faces=n + 1)
    c = ctx()
    for tree, wide_build, sah_dev in ((0, 1, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1)):
        try:
            c.set_option('sah_build', sah_dev)
        except RuntimeError:
            if sah_dev:
                continue
        c.set_option('tree', tree)
        c.set_option('wide_build', wide_build)
        ts = []
        for _ in range(3):
            c.call('mpt_synchronize')
            t0 = time.perf_counter()
            BVHTree().build()
            c.call('mpt_synchronize')
            ts.append(time.perf_counter() - t0)
        print(f'n {n}: tree {"SAH" if tree else "LBVH"} ({"device" if sah_dev else "host"} SAH pass), collapse on the {"device" if wide_build else "host"}: '
              f'build {min(ts) * 1e3:.1f} ms (runs {[round(t * 1e3, 1) for t in ts]}), depth {c.get_option("fast_depth")}, wide nodes {c.get_option("wide_nodes")}', flush=True)
reset_all()
