#!/usr/bin/env python3
'''One GPU's share of an N-way tiled film (no gather): `steps` pipelined 32-spp steps of slab `rank`
of `parts`.  Run under `rocprofv3 --kernel-trace` to see the launch timeline.  Diagnostics only.'''
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from ptina_amd import scenes  # noqa: E402
from ptina_amd.common import ctx  # noqa: E402
from helpers import setup_engine  # noqa: E402

parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else parts // 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n, spp = 512, 32
stripe = int(os.environ.get('STRIPE', '0'))
eng = setup_engine(scenes.get_scene('s978'), n, n, mode='fast',
                   slab=None if stripe else (rank * n // parts, (rank + 1) * n // parts))
c = ctx()
if stripe:
    c.call('mpt_set_stripes', stripe, rank, parts)
c.set_option('batch', spp)
for kv in filter(None, os.environ.get('MIPTINA_OPTS', '').split(',')):
    key, val = kv.split('=')
    c.set_option(key, int(val))
eng.render(spp)
c.call('mpt_synchronize')
c.kernel_time()
t0 = time.perf_counter()
for _ in range(steps):
    eng.render(spp)
    c.call('mpt_flush')
    c.call('mpt_resolve', 0)
host = (time.perf_counter() - t0) / steps * 1e3
c.call('mpt_synchronize')
dt = (time.perf_counter() - t0) / steps * 1e3
kms, nl = c.kernel_time()
print(os.environ.get('MIPTINA_OPTS', ''), f'stripe {stripe} slab {rank}/{parts}: step {dt:.3f} ms (host issue {host:.3f}), kernel {kms / nl:.3f} ms', flush=True)
