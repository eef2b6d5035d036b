#!/usr/bin/env python3
'''GPU: the pooled LDS kernel (option "pool") against the unpooled one -- same film bit for bit on growing sizes, then the
512x512x32 launch time of both (HIP events), for a list of shader-wave counts.  usage: tools/pool_check.py [shaders ...]'''
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from ptina_amd import scenes                      # noqa: E402
from ptina_amd.common import ctx, reset_all       # noqa: E402
from ptina_amd.things import FilmTable            # noqa: E402
from helpers import setup_engine                  # noqa: E402


def film(scene, nx, ny, spp, pool, shaders=3, count=0):
    reset_all()
    eng = setup_engine(scene, nx, ny, mode='fast')
    c = ctx()
    c.set_option('pool', pool)
    c.set_option('pool_shaders', shaders)
    c.set_option('count', count)
    if count:
        c.call('mpt_reset_counters')
    eng.render(spp)
    raw = FilmTable().get_raw().copy()
    return raw, c.get_option('last_kernel'), (c.counters() if count else None)


def main():
    shaders = [int(a) for a in sys.argv[1:]] or [3]
    ok = True
    for name, nx, ny, spp in (('s34', 24, 16, 2), ('s34', 64, 64, 4), ('s978', 52, 43, 3), ('s978', 128, 128, 8), ('s978', 512, 512, 32)):
        scene = scenes.get_scene(name)
        ref, k0, _ = film(scene, nx, ny, spp, 0)
        for s in shaders:
            t0 = time.time()
            got, k1, _ = film(scene, nx, ny, spp, 1, s)
            same = np.array_equal(ref, got)
            ok &= same and k1 == 3
            print(f'{name} {nx}x{ny}x{spp} shaders {s}: kernels {k0}/{k1} film {"EQUAL" if same else "DIFFERENT"} '
                  f'counts ok {bool(np.all(got[:, 3] == spp))} ({time.time() - t0:.2f} s)', flush=True)
            if not same:
                d = (ref != got).any(axis=1)
                print('   pixels differing', int(d.sum()), 'of', len(d), 'max abs', float(np.abs(ref - got).max()), flush=True)
        if not ok:
            break
    # counters agree (the same rays, boxes, triangles, bounces: only the order in time differs)
    if ok:
        scene = scenes.get_scene('s978')
        _, _, c0 = film(scene, 128, 128, 8, 0, count=1)
        _, _, c1 = film(scene, 128, 128, 8, 1, 3, count=1)
        keys = ('samples', 'rays', 'n_box', 'n_tri', 'n_shade', 'n_draws', 'bounces', 'n_node')
        print('counters', {k: (c0[k], c1[k]) for k in keys}, flush=True)
        print('steps   ', {k: (c0[k], c1[k]) for k in ('it_node', 'it_leaf', 'it_shade', 'it_new')}, flush=True)
    # how the two kinds of waves spent a full-size launch
    scene = scenes.get_scene('s978')
    for pool, s in [(0, 0)] + [(1, s) for s in shaders]:
        _, _, cc = film(scene, 512, 512, 32, pool, max(s, 1), count=1)
        smp = cc['samples']
        print(f'pool {pool} shaders {s}: per 64 samples: NODE steps {64 * cc["it_node"] / smp:.1f} at {cc["n_node"] / max(cc["it_node"], 1):.1f} lanes, '
              f'LEAF steps {64 * cc["it_leaf"] / smp:.1f} at {cc["n_tri"] / max(cc["it_leaf"], 1):.1f}, SHADE stages {64 * cc["it_shade"] / smp:.2f} at '
              f'{cc["n_shade"] / max(cc["it_shade"], 1):.1f}', flush=True)
        if pool and os.environ.get('MIPTINA_POOL_STAMPS'):
            tl, sl = max(cc['pl_trips'], 1), max(cc['pl_tidle'], 1)
            print(f'    tracer waves: traversal {cc["pl_batches"] / tl:.1%}, trips to the pools {cc["pl_batch_lanes"] / tl:.1%}, idle {cc["pl_prim"] / tl:.1%} of their lifetime; '
                  f'shader waves: SHADE batches {cc["pl_local"] / sl:.1%}, primary rays {cc["pl_taken"] / sl:.1%}, idle {cc["pl_sidle"] / sl:.1%}', flush=True)
        elif pool:
            print('   ', {k: v for k, v in cc.items() if k.startswith('pl_')},
                  f'shader batch {cc["pl_batch_lanes"] / max(cc["pl_batches"], 1):.1f} lanes; local bounces {cc["pl_local"] / max(cc["n_shade"], 1):.1%}; '
                  f'rays per trip {cc["pl_taken"] / max(cc["pl_trips"], 1):.2f}', flush=True)
    # timing
    scene = scenes.get_scene('s978')
    for pool, s in [(0, 0)] + [(1, s) for s in shaders]:
        reset_all()
        eng = setup_engine(scene, 512, 512, mode='fast')
        c = ctx()
        c.set_option('pool', pool)
        if pool:
            c.set_option('pool_shaders', s)
        c.set_option('batch', 32)
        for _ in range(3):
            eng.render(32)
            c.call('mpt_synchronize')
        c.kernel_time()
        for _ in range(10):
            eng.render(32)
            c.call('mpt_synchronize')
        ms, n = c.kernel_time()
        print(f'pool {pool} shaders {s}: render kernel {ms / n:.4f} ms per 512x512x32 launch ({n} launches), kernel {c.get_option("last_kernel")}', flush=True)
    reset_all()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
