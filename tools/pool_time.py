#!/usr/bin/env python3
'''GPU: launch time (HIP events) and lane use of the pooled LDS kernel on the 512x512x32 headline launch, for the library
MIPTINA_LIB names.  usage: tools/pool_time.py [shader-wave counts ...]   (0 = the unpooled kernel)'''
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from ptina_amd import scenes                      # noqa: E402
from ptina_amd.common import ctx, reset_all       # noqa: E402
from helpers import setup_engine                  # noqa: E402

tag = os.path.basename(os.environ.get('MIPTINA_LIB', 'libmiptina.so'))
scene = scenes.get_scene(os.environ.get('POOL_SCENE', 's978'))
n = int(os.environ.get('POOL_N', '512'))
for s in [int(a) for a in sys.argv[1:]] or [0, 3]:
    reset_all()
    eng = setup_engine(scene, n, n, mode='fast')
    c = ctx()
    c.set_option('pool', 1 if s else 0)
    if s:
        c.set_option('pool_shaders', s)
    c.set_option('batch', 32)
    c.set_option('count', 1)
    c.call('mpt_reset_counters')
    eng.render(32)
    cc = c.counters()
    c.set_option('count', 0)
    for _ in range(3):
        eng.render(32)
        c.call('mpt_synchronize')
    c.kernel_time()
    for _ in range(10):
        eng.render(32)
        c.call('mpt_synchronize')
    ms, k = c.kernel_time()
    smp = cc['samples']
    line = (f'{tag} shaders {s}: {ms / k:.4f} ms | per 64 samples NODE {64 * cc["it_node"] / smp:.1f} at {cc["n_node"] / max(cc["it_node"], 1):.1f} lanes, '
            f'LEAF {64 * cc["it_leaf"] / smp:.1f} at {cc["n_tri"] / max(cc["it_leaf"], 1):.1f}, SHADE {64 * cc["it_shade"] / smp:.2f} at {cc["n_shade"] / max(cc["it_shade"], 1):.1f}')
    if s:
        line += (f' | batch {cc["pl_batch_lanes"] / max(cc["pl_batches"], 1):.1f}, local {cc["pl_local"] / max(cc["n_shade"], 1):.1%}, trips/64 {64 * cc["pl_trips"] / smp:.1f}, '
                 f'rays/trip {cc["pl_taken"] / max(cc["pl_trips"], 1):.1f}, tidle/64 {64 * cc["pl_tidle"] / smp:.1f}, sidle/64 {64 * cc["pl_sidle"] / smp:.1f}')
    print(line, flush=True)
reset_all()
