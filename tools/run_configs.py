#!/usr/bin/env python3
'''Measure every BASELINE.json configuration that fits one GPU (BASELINE.md section 3's table):
Msamples/s, counted algorithmic bytes -> GB/s (served mostly from LDS / L2: not an HBM figure), traversal counters.
Writes gpurun_out/configs.json.'''
import json
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
from bench import algorithmic_bytes  # noqa: E402

CONFIGS = [
    ('C1 s34 512x512x32', 's34', {}, 512, 32, None),
    ('C2 s978 512x512x32', 's978', {}, 512, 32, None),
    ('C3-film s978 2048x2048x64 (one GPU)', 's978', {}, 2048, 64, None),
    ('C4 99k-tri blob + env light 1024x1024x64', 'c4', {}, 1024, 64, ([1.0, 1.0, 1.0, 1.0], 0)),
    ('C5 1M random tris 1024x1024x16', 'c5', {'n': 1000000}, 1024, 16, None),
]
only = sys.argv[1:]          # e.g. `run_configs.py C4 C5`
if only:
    CONFIGS = [c for c in CONFIGS if any(c[0].startswith(o) for o in only)]
out = {}
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
for title, name, kw, n, spp, world in CONFIGS:
    common.reset_all()
    t0 = time.time()
    scene = scenes.get_scene(name, **kw)
    if os.environ.get('MIPTINA_SAH_MAX'):
        from ptina_amd.things import init_things
        init_things(max_filmsize=max(n * n, 1 << 21))
        ctx().set_option('sah_max', int(os.environ['MIPTINA_SAH_MAX']))
    eng = setup_engine(scene, n, n, mode='fast', world=world, max_filmsize=max(n * n, 1 << 21))
    c = ctx()
    setup_s = time.time() - t0
    c.set_option('batch', 32)
    for kv in filter(None, os.environ.get('MIPTINA_OPTS', '').split(',')):
        c.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    if os.environ.get('MIPTINA_OPTS'):
        from ptina_amd.things import BVHTree
        BVHTree().build()                     # (options that change the tree take effect at the next build)
    if os.environ.get('MIPTINA_WIDE'):
        c.set_option('wide', int(os.environ['MIPTINA_WIDE']))
    eng.render(1)
    c.call('mpt_synchronize')
    c.set_option('count', 1)
    c.call('mpt_reset_counters')
    eng.render(min(spp, 32))
    cnt = c.counters()
    c.set_option('count', 0)
    c.call('mpt_synchronize')
    c.kernel_time()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        eng.render(spp)
        c.call('mpt_resolve', 0)
    c.call('mpt_synchronize')
    dt = (time.perf_counter() - t0) / reps
    kms, nl = c.kernel_time()
    bps = algorithmic_bytes(cnt) / cnt['samples']
    ms = n * n * spp / dt / 1e6
    out[title] = {'ntri': int(scene[1].shape[0]), 'setup_s': round(setup_s, 3), 'msamples_s': round(ms, 1),
                  'ms_per_step': round(dt * 1e3, 3), 'kernel': ('gather', 'lds', 'gather4', 'lds_pool', 'gather8', 'lds4')[c.get_option('last_kernel')],
                  'bytes_per_sample': round(bps, 1), 'achieved_GBs': round(bps * ms * 1e6 / 1e9, 1),
                  'rays_per_sample': round(cnt['rays'] / cnt['samples'], 2),
                  'nodes_per_ray': round(cnt['n_node'] / cnt['rays'], 2), 'tris_per_ray': round(cnt['n_tri'] / cnt['rays'], 2),
                  'mrays_s': round(cnt['rays'] / cnt['samples'] * ms, 1),
                  'tree_depth': [c.get_option('tree_depth'), c.get_option('fast_depth'), c.get_option('wide_depth')]}
    print(title, json.dumps(out[title]), flush=True)
    # (a run restricted to some configurations -- the profiler passes name C4 C5 -- does not overwrite the table of all of them)
    default_out = 'configs.json' if len(sys.argv) <= 1 else 'configs_subset.json'
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', os.environ.get('CONFIGS_OUT', default_out)), 'w'), indent=1)
common.reset_all()
