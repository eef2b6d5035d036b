#!/usr/bin/env python3
'''GPU-box diagnostics: parity statistics (strict / fast vs oracle) and a timing sweep.
Writes gpurun_out/diag.json.  Not part of the product or the tests.'''
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import oracle  # noqa: E402
from ptina_amd import scenes, common, _lib  # noqa: E402
from ptina_amd.common import ctx  # noqa: E402
from ptina_amd.things import FilmTable  # noqa: E402
from helpers import setup_engine, setup_oracle, image_stats  # noqa: E402

out = {}
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)


def save():
    '''gpurun_out/diag.json (everything this run measured) and ONE FILE PER DIAGNOSTIC, gpurun_out/diag_<key>.json, which
    tools/collect_profiles.py copies to profiles/<round>_<key>.json: a later run of another diagnostic then cannot overwrite what
    DESIGN.md cites (r04_diag.json held only the last run's stamps)'''
    with open(os.path.join(ROOT, 'gpurun_out', 'diag.json'), 'w') as f:
        json.dump(out, f, indent=1)
    for key, val in out.items():
        with open(os.path.join(ROOT, 'gpurun_out', 'diag_%s.json' % key), 'w') as f:
            json.dump({key: val, 'library': os.environ.get('MIPTINA_LIB', 'ptina_amd/libmiptina.so')}, f, indent=1)


def parity(name, nx, ny, spp):
    scene = scenes.get_scene(name)
    ref = setup_oracle(oracle, scene, nx, ny)
    t0 = time.time()
    ref.render(spp)
    tor = time.time() - t0
    want = ref.get_image()
    res = {'oracle_s': tor, 'oracle_counters': ref.counters()}
    for mode in ('strict', 'fast'):
        common.reset_all()
        eng = setup_engine(scene, nx, ny, mode=mode)
        ctx().set_option('count', 1)
        eng.render(spp)
        img = FilmTable().get_image()
        d, refn, rel = image_stats(img, want)
        res[mode] = {'rel_rmse': rel, 'exact_frac': float((d == 0).mean()),
                     'frac_gt_1e-5': float((d > 1e-5 * (1 + refn)).mean()),
                     'frac_gt_1e-4': float((d > 1e-4 * (1 + refn)).mean()),
                     'frac_gt_1e-3': float((d > 1e-3 * (1 + refn)).mean()),
                     'frac_gt_1e-2': float((d > 1e-2 * (1 + refn)).mean()),
                     'max': float(d.max()), 'mean_img': float(img[..., :3].mean()),
                     'mean_ref': float(want[..., :3].mean()), 'counters': ctx().counters()}
    common.reset_all()
    return res


def timing(name, mode, chunk, steps=5, spp=32, n=512, lds=1, sched=(1, 1)):
    common.reset_all()
    eng = setup_engine(scenes.get_scene(name), n, n, mode=mode)
    c = ctx()
    c.set_option('batch', spp)
    c.set_option('chunk', chunk)
    c.set_option('lds', lds)
    eng.render(spp)
    c.call('mpt_synchronize')
    c.kernel_time()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.render(spp)
        c.call('mpt_resolve', 0)
    c.call('mpt_synchronize')
    dt = time.perf_counter() - t0
    kms, nl = c.kernel_time()
    lk = c.get_option('last_kernel')
    common.reset_all()
    return {'wall_ms_per_step': dt / steps * 1e3, 'kernel_ms': kms / nl, 'msamples_s': n * n * spp * steps / dt / 1e6,
            'lds_kernel': lk}


def ab(variants=((0, 2, 1), (0, 3, 2), (0, 1, 1)),
       rounds=5, spp=32, n=512, name='s978'):
    '''interleaved A/B of kernel variants in ONE process (median and min kernel ms)'''
    common.reset_all()
    eng = setup_engine(scenes.get_scene(name), n, n, mode='fast')
    c = ctx()
    c.set_option('batch', spp)
    res = {v: [] for v in variants}
    for r in range(rounds + 1):
        for v in variants:
            eng.render(spp)
            c.call('mpt_synchronize')
            ms, nl = c.kernel_time()
            if r > 0:
                res[v].append(ms / nl)
    common.reset_all()
    return {str(v): {'median': float(np.median(t)), 'min': float(np.min(t))} for v, t in res.items()}


def tree_ab(name='s978', rounds=5, spp=32, n=512):
    from ptina_amd.things import BVHTree
    common.reset_all()
    eng = setup_engine(scenes.get_scene(name), n, n, mode='fast')
    c = ctx()
    c.set_option('batch', spp)
    res = {}
    for r in range(rounds + 1):
        for kind in (0, 1):
            c.set_option('tree', kind)
            BVHTree().build()
            c.set_option('count', 1 if r == 0 else 0)
            c.call('mpt_reset_counters')
            eng.render(spp)
            c.call('mpt_synchronize')
            ms, nl = c.kernel_time()
            if r == 0:
                cnt = c.counters()
                res[kind] = {'depth': c.get_option('fast_depth'), 'lds_kernel': c.get_option('last_kernel'),
                             'node_per_ray': cnt['n_node'] / cnt['rays'], 'tri_per_ray': cnt['n_tri'] / cnt['rays'], 'ms': []}
            else:
                res[kind]['ms'].append(ms / nl)
    common.reset_all()
    for k in res:
        res[k]['median_ms'] = float(np.median(res[k]['ms']))
    return res


def big(name, res_px, spp, **kw):
    '''large scenes through the gather kernel: build time, render time, counters, fast-vs-strict'''
    from ptina_amd.things import FilmTable as FT
    t0 = time.time()
    scene = scenes.get_scene(name, **kw)
    tgen = time.time() - t0
    world = ([1.0, 1.0, 1.0, 1.0], 0) if scene[3] else None
    res = {'ntri': int(scene[1].shape[0]), 'gen_s': tgen}
    imgs = {}
    for mode in ('fast', 'strict'):
        common.reset_all()
        t0 = time.time()
        n = res_px
        eng = setup_engine(scene, n, n, mode=mode, world=world)
        res[mode + '_setup_s'] = time.time() - t0
        c = ctx()
        res[mode + '_depth'] = [c.get_option('tree_depth'), c.get_option('fast_depth')]
        c.set_option('batch', min(spp, 32))
        eng.render(1)
        c.call('mpt_synchronize')
        c.kernel_time()
        reps = 1 if mode == 'strict' else 3
        use = spp if mode == 'fast' else min(spp, 4)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.render(use)
        c.call('mpt_synchronize')
        dt = time.perf_counter() - t0
        kms, nl = c.kernel_time()
        res[mode] = {'msamples_s': n * n * use * reps / dt / 1e6, 'kernel_ms_per_launch': kms / nl, 'spp': use,
                     'lds_kernel': c.get_option('last_kernel')}
        if mode == 'fast':
            c.set_option('count', 1)
            c.call('mpt_reset_counters')
            eng.render(1)
            cnt = c.counters()
            res['per_ray'] = {k: cnt[k] / max(cnt['rays'], 1) for k in ('n_node', 'n_box', 'n_tri')}
            res['per_sample'] = {k: cnt[k] / max(cnt['samples'], 1) for k in cnt}
            c.set_option('count', 0)
        from ptina_amd.sampling.sobol import SobolSampler
        SobolSampler().reset()
        FT().clear()
        eng.render(4)
        imgs[mode] = FT().get_image()
    d, refn, rel = image_stats(imgs['fast'], imgs['strict'])
    res['fast_vs_strict'] = {'rel_rmse': rel, 'frac_gt_1e-3': float((d > 1e-3 * (1 + refn)).mean()), 'max': float(d.max())}
    common.reset_all()
    return res


def util(name='s978', spp=32, n=512):
    '''lanes per issued stage (of 64) for a few scheduler thresholds'''
    res = {}
    for sched in ((1, 1), (2, 1), (4, 1)):
        common.reset_all()
        eng = setup_engine(scenes.get_scene(name), n, n, mode='fast')
        c = ctx()
        c.set_option('batch', spp)
        c.set_option('count', 1)
        eng.render(spp)
        k = c.counters()
        res[f'{sched[0]}:{sched[1]}'] = {
            'node_lanes': k['n_node'] / max(k['it_node'], 1), 'leaf_lanes': k['n_tri'] / max(k['it_leaf'], 1),
            'shade_lanes': k['n_shade'] / max(k['it_shade'], 1), 'new_lanes': k['samples'] / max(k['it_new'], 1),
            'it_node_per_wave_sample': k['it_node'] * 64 / k['samples'], 'it_leaf_pws': k['it_leaf'] * 64 / k['samples'],
            'it_shade_pws': k['it_shade'] * 64 / k['samples'], 'it_new_pws': k['it_new'] * 64 / k['samples']}
    common.reset_all()
    return res


def slabs(name='s978', spp=32, n=512, steps=20, tiles=((3, 3),), lds=(1, 0)):
    '''per-rank cost of a 1/N column slab (what one GPU of N does per step, without the gather)'''
    res = {}
    for parts, tile, use_lds in [(pp, tt, ll) for pp in (1, 2, 4, 8) for tt in tiles for ll in lds]:
        per = []
        for r in sorted(set((0, parts // 2,))):
            common.reset_all()
            x0, x1 = r * n // parts, (r + 1) * n // parts
            eng = setup_engine(scenes.get_scene(name), n, n, mode='fast', slab=(x0, x1))
            c = ctx()
            c.set_option('batch', spp)
            c.set_option('lds', use_lds)
            c.set_option('tile_w_shift', tile[0])
            c.set_option('tile_h_shift', tile[1])
            eng.render(spp)
            c.call('mpt_resolve', 0)
            c.call('mpt_synchronize')
            c.kernel_time()
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.render(spp)
                c.call('mpt_flush')
                c.call('mpt_resolve', 0)
            c.call('mpt_synchronize')
            dt = (time.perf_counter() - t0) / steps * 1e3
            kms, nl = c.kernel_time()
            per.append({'rank': r, 'step_ms': dt, 'kernel_ms': kms / nl})
        res[f'{parts}:lds{use_lds}'] = per
    common.reset_all()
    return res


def shares_sync(name='s978', spp=32, n=512, steps=20):
    '''what ONE rank of N does per benchmark step when the step ends with a read-back (bench.py's `value`:
    one launch at a time): its stripes' share rendered, resolved and read back, without the gather'''
    res = {}
    for parts, forced in ((1, 0), (2, 0), (2, 2), (4, 0), (4, 2), (8, 0), (8, 4)):
        common.reset_all()
        eng = setup_engine(scenes.get_scene(name), n, n, mode='fast')
        c = ctx()
        c.set_option('batch', spp)
        if forced:
            c.set_option('grid_div', forced)      # what the samples-per-lane rule alone would pick (before: always)
        for kv in filter(None, os.environ.get('MIPTINA_OPTS', '').split(',')):
            c.set_option(kv.split('=')[0], int(kv.split('=')[1]))
        if parts > 1:
            c.call('mpt_set_stripes', 16, 0, parts)
        for _ in range(3):
            eng.render(spp)
            FilmTable().get_image()
        c.kernel_time()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.render(spp)
            FilmTable().get_image()
        dt = (time.perf_counter() - t0) / steps * 1e3
        kms, nl = c.kernel_time()
        # the same step as a rank that is NOT the root ends it in bench.py (render, flush, wait: no image), and the read-back alone
        # (resolve + the 4 MiB image over PCIe) on a film nobody is rendering to: what `step - kernel` of a share is made of
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.render(spp)
            c.call('mpt_flush')
            c.call('mpt_synchronize')
        dt_nonroot = (time.perf_counter() - t0) / steps * 1e3
        kms2, nl2 = c.kernel_time()
        FilmTable().get_image()
        t0 = time.perf_counter()
        for _ in range(steps):
            FilmTable().get_image()
        dt_read = (time.perf_counter() - t0) / steps * 1e3
        key = f'{parts}' + (f' grid_div={forced}' if forced else '')
        res[key] = {'step_ms': round(dt, 4), 'kernel_ms': round(kms / max(nl, 1), 4), 'last_div': c.get_option('last_div'),
                    'cur_div': c.get_option('cur_div'), 'speedup_vs_1': None,
                    'step_without_image_ms': round(dt_nonroot, 4), 'kernel_without_image_ms': round(kms2 / max(nl2, 1), 4),
                    'get_image_alone_ms': round(dt_read, 4)}
        res[key]['speedup_vs_1'] = round(res.get('1', res[key])['step_ms'] / dt, 3)
        print('shares_sync', key, json.dumps(res[key]), flush=True)
    out['shares_sync'] = res
    save()
    common.reset_all()


def blk(name='s978', spp=32, n=512, steps=20, opt='lds_block', values=(1024, 768, 512, 256)):
    '''an option (threads per persistent workgroup, reserved CUs) vs launch size: pipelined step and solo kernel time'''
    res = {}
    for parts in (1, 2, 4, 8):
        for block in values:
            common.reset_all()
            r = parts // 2
            eng = setup_engine(scenes.get_scene(name), n, n, mode='fast', slab=(r * n // parts, (r + 1) * n // parts))
            c = ctx()
            c.set_option('batch', spp)
            c.set_option(opt, block)
            eng.render(spp)
            c.call('mpt_synchronize')
            c.kernel_time()
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.render(spp)
                c.call('mpt_flush')
                c.call('mpt_resolve', 0)
            c.call('mpt_synchronize')
            dt = (time.perf_counter() - t0) / steps * 1e3
            c.kernel_time()
            for _ in range(5):
                eng.render(spp)
                c.call('mpt_synchronize')
            kms, nl = c.kernel_time()
            res[f'{parts}:{block}'] = {'step_ms': round(dt, 4), 'solo_kernel_ms': round(kms / nl, 4)}
            print(opt, parts, block, res[f'{parts}:{block}'], flush=True)
    common.reset_all()
    return res


def pipe(name='s978', spp=32, n=512, steps=30):
    '''launches of 1/G of the CUs, D batches in flight: pipelined step time per slab size'''
    res = {}
    for parts in (1, 2, 4, 8):
        for depth, div in ((0, 0), (2, 1), (4, 2), (6, 4)):
            common.reset_all()
            r = parts // 2
            eng = setup_engine(scenes.get_scene(name), n, n, mode='fast', slab=(r * n // parts, (r + 1) * n // parts))
            c = ctx()
            c.set_option('batch', spp)
            c.set_option('pipe_depth', depth)
            c.set_option('grid_div', div)
            for _ in range(6):
                eng.render(spp)
            c.call('mpt_synchronize')
            c.kernel_time()
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.render(spp)
                c.call('mpt_flush')
                c.call('mpt_resolve', 0)
            c.call('mpt_synchronize')
            dt = (time.perf_counter() - t0) / steps * 1e3
            kms, nl = c.kernel_time()
            res[f'{parts}:D{depth}G{div}'] = {'step_ms': round(dt, 4), 'kernel_ms': round(kms / nl, 4)}
            print('pipe parts', parts, 'depth', depth, 'grid_div', div, res[f'{parts}:D{depth}G{div}'], flush=True)
    common.reset_all()
    return res


def timeline(name='s978', spp=32, n=512):
    '''per-wave timestamps of one solo launch: when the queues ran dry and when waves / workgroups left'''
    import ctypes as C
    res = {}
    for parts in (1, 8):
        common.reset_all()
        r = parts // 2
        eng = setup_engine(scenes.get_scene(name), n, n, mode='fast', slab=(r * n // parts, (r + 1) * n // parts))
        c = ctx()
        c.set_option('batch', spp)
        c.set_option('timeline', 1)
        for kv in filter(None, os.environ.get('MIPTINA_OPTS', '').split(',')):
            c.set_option(kv.split('=')[0], int(kv.split('=')[1]))
        for _ in range(3):
            eng.render(spp)
            c.call('mpt_synchronize')
        nw = C.c_int(0)
        buf = (C.c_ulonglong * (8 * 4096))()
        c.call('mpt_get_timeline', buf, 4096, C.byref(nw))
        t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8)[:nw.value].astype(np.int64)
        np.save(os.path.join(ROOT, 'gpurun_out', f'timeline_raw_{parts}.npy'), t)
        t0 = t[:, 0].min()
        us = (t[:, :4] - t0) / 100.0
        if t[:, 6].max() == 0 and t[:, 4].max() > 0:     # product build with the tail finalisation: when each wave left it, tiles it did
            fin_end = (t[:, 4] - t0) / 100.0
            extra = {'fin_end': [round(float(x), 1) for x in np.percentile(fin_end, [0, 10, 50, 90, 100])],
                     'fin_us_per_wave': [round(float(x), 1) for x in np.percentile(fin_end - us[:, 3], [0, 10, 50, 90, 100])],
                     'tiles_per_wave': [int(x) for x in np.percentile(t[:, 5], [0, 10, 50, 90, 100])],
                     'waves_that_finalised': int((t[:, 5] > 0).sum()), 'tiles': int(t[:, 5].sum()),
                     'last_trace_exit': round(float(us[:, 3].max()), 1), 'last_fin_exit': round(float(fin_end.max()), 1)}
            print('timeline_fin', parts, json.dumps(extra), flush=True)
        elif t[:, 5].max() > 0:            # diagnostic build -DMPT_X_TIMELINE2: last pull, items, lanes in flight at empty, passes after
            lastpull = (t[:, 4] - t0) / 100.0
            extra = {'last_pull': [round(float(x), 1) for x in np.percentile(lastpull, [0, 10, 50, 90, 100])],
                     'last_item_us': [round(float(x), 1) for x in np.percentile(us[:, 2] - lastpull, [0, 10, 50, 90, 100])],
                     'items_per_wave': [int(x) for x in np.percentile(t[:, 5], [0, 10, 50, 90, 100])],
                     'lanes_in_flight_at_empty': [int(x) for x in np.percentile(t[:, 6], [0, 10, 50, 90, 100])],
                     'passes_after_empty': [int(x) for x in np.percentile(t[:, 7], [0, 10, 50, 90, 100])]}
            print('timeline2', parts, json.dumps(extra), flush=True)
        wg_exit = us[:, 3].reshape(-1, 16).max(axis=1)
        q = lambda a: [round(float(x), 1) for x in np.percentile(a, [0, 10, 50, 90, 100])]
        res[str(parts)] = {'start': q(us[:, 0]), 'ready': q(us[:, 1]), 'queue_empty': q(us[:, 2]),
                           'wave_exit': q(us[:, 3]), 'wg_exit': q(wg_exit),
                           'wave_drain': q(us[:, 3] - us[:, 2]),
                           'lane_time_after_empty_frac': float((us[:, 3] - us[:, 2]).sum() / (us[:, 3] - us[:, 0]).sum())}
        print('timeline', parts, json.dumps(res[str(parts)]), flush=True)
    common.reset_all()
    return res


def fixedcost(name='s978', n=512, steps=12):
    '''solo launches of B frames (each synchronised before the next): kernel ms against B.  The slope is the steady-state cost
    of a frame of THIS film, the intercept what a launch costs beyond it (fill + drain)'''
    res = {}
    common.reset_all()
    eng = setup_engine(scenes.get_scene(name), n, n, mode='fast')
    c = ctx()
    for kv in filter(None, os.environ.get('MIPTINA_OPTS', '').split(',')):
        c.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    for B in (4, 8, 16, 32, 48, 64):
        c.set_option('batch', B)
        for _ in range(2):
            eng.render(B)
            c.call('mpt_synchronize')
        c.kernel_time()
        for _ in range(steps):
            eng.render(B)
            c.call('mpt_synchronize')
        kms, nl = c.kernel_time()
        res[B] = round(kms / max(nl, 1), 4)
    bs = np.array(sorted(res), float)
    ts = np.array([res[int(b)] for b in bs])
    slope, icpt = np.polyfit(bs[2:], ts[2:], 1)
    res['fit_ms_per_frame'] = round(float(slope), 5)
    res['fit_intercept_ms'] = round(float(icpt), 4)
    print('fixedcost', json.dumps(res), flush=True)
    common.reset_all()
    return res


def c3(n=2048, spp=64):
    '''config 3's film on one GPU: S978 at 2048x2048 (needs max_filmsize = 2^22)'''
    common.reset_all()
    eng = setup_engine(scenes.get_scene('s978'), n, n, mode='fast', max_filmsize=n * n)
    c = ctx()
    c.set_option('batch', 32)
    eng.render(1)
    c.call('mpt_synchronize')
    c.kernel_time()
    t0 = time.perf_counter()
    eng.render(spp)
    c.call('mpt_resolve', 0)
    c.call('mpt_synchronize')
    dt = time.perf_counter() - t0
    kms, nl = c.kernel_time()
    from ptina_amd.things import FilmTable as FT
    raw = FT().get_raw()
    ok = bool(np.all(raw[:, 3] == spp + 1) and np.isfinite(raw).all())
    common.reset_all()
    return {'msamples_s': n * n * spp / dt / 1e6, 'kernel_ms': kms / nl, 'launches': nl, 'film_ok': ok}


def c5sah(n=1024, spp=16):
    '''config 5 with the SAH re-partition forced on (host pass over 1M leaves): build time and rate'''
    from ptina_amd.things import BVHTree
    res = {}
    scene = scenes.get_scene('c5', n=1000000)
    for sah_max in (1 << 18, 1 << 21):
        common.reset_all()
        from ptina_amd.things import init_things
        init_things()
        ctx().set_option('sah_max', sah_max)
        t0 = time.time()
        eng = setup_engine(scene, n, n, mode='fast')
        c = ctx()
        res_key = 'sah' if sah_max > 1000000 else 'lbvh'
        setup = time.time() - t0
        c.set_option('batch', 16)
        eng.render(1)
        c.call('mpt_synchronize')
        c.set_option('count', 1)
        c.call('mpt_reset_counters')
        eng.render(4)
        cnt = c.counters()
        c.set_option('count', 0)
        c.call('mpt_synchronize')
        t0 = time.perf_counter()
        for _ in range(3):
            eng.render(spp)
        c.call('mpt_synchronize')
        dt = (time.perf_counter() - t0) / 3
        res[res_key] = {'setup_s': setup, 'msamples_s': n * n * spp / dt / 1e6, 'nodes_per_ray': cnt['n_node'] / cnt['rays'],
                        'tris_per_ray': cnt['n_tri'] / cnt['rays'], 'depth': c.get_option('fast_depth')}
        print('c5', res_key, json.dumps(res[res_key]), flush=True)
    common.reset_all()
    return res


def probe(name='s978', spp=32, n=512, world=8, rounds=40):
    '''how long does a small foreign kernel (stand-in for RCCL's send/recv kernel) wait for a CU while this
    rank's share of the film is rendered by G overlapped persistent launches?  One 256-lane workgroup with
    16 KiB of LDS on a stream of its own, launched right after a burst of render launches was enqueued;
    wall time until it completes, against the same probe on an idle GPU.'''
    import ctypes as C
    res = {}
    for share, label, reserve in (((16, 0, world), f'1/{world} share (G launches overlapped)', 0),
                                  ((16, 0, world), f'1/{world} share, 2 CUs reserved', 2),
                                  (None, 'whole film', 0), (None, 'whole film, 2 CUs reserved', 2)):
        common.reset_all()
        eng = setup_engine(scenes.get_scene(name), n, n, mode='fast')
        c = ctx()
        c.set_option('batch', spp)
        c.set_option('reserve_cus', reserve)
        if share:
            c.call('mpt_set_stripes', *share)
        us = C.c_double(0)
        idle = []
        c.call('mpt_synchronize')
        for _ in range(10):
            c.call('mpt_probe_kernel', 256, 16384, C.byref(us))
            idle.append(us.value)
        busy = []
        for _ in range(rounds):
            for _ in range(6):
                eng.render(spp)             # six launches enqueued: the ring is full, G resident
            c.call('mpt_probe_kernel', 256, 16384, C.byref(us))
            busy.append(us.value)
        c.call('mpt_synchronize')
        t0 = time.perf_counter()
        for _ in range(20):
            eng.render(spp)
        c.call('mpt_synchronize')
        step_us = (time.perf_counter() - t0) / 20 * 1e6
        res[label] = {'idle_us_median': float(np.median(idle)), 'busy_us_median': float(np.median(busy)),
                      'busy_us_p90': float(np.percentile(busy, 90)), 'busy_us_max': float(np.max(busy)),
                      'step_us': step_us, 'grid_div': c.get_option('cur_div'), 'pipe_depth': c.get_option('cur_depth')}
        print('probe', label, res[label], flush=True)
    out['probe'] = res
    save()
    common.reset_all()


def sync_sweep(name='s978', spp=32, n=512, steps=15):
    '''the benchmark's step (32 x render + get_image, synchronous) under work-item granularities:
    frames per item (chunk) and tile shape; ms per step and render-kernel ms'''
    res = {}
    variants = [('default', {})] + [(f'chunk={ch}', {'chunk': ch}) for ch in (1, 2, 4)] + \
               [(f'chunk=1 tile={1 << w}x{1 << h}', {'chunk': 1, 'tile_w_shift': w, 'tile_h_shift': h}) for w, h in ((3, 2), (2, 2))] + \
               [(f'tile={1 << w}x{1 << h}', {'tile_w_shift': w, 'tile_h_shift': h}) for w, h in ((3, 2), (2, 2))]
    extra = [kv.split('=') for kv in filter(None, os.environ.get('MIPTINA_OPTS', '').split(','))]
    for label, opts in variants:
        common.reset_all()
        eng = setup_engine(scenes.get_scene(name), n, n, mode='fast')
        c = ctx()
        c.set_option('batch', spp)
        for k, v in list(opts.items()) + [(k, int(v)) for k, v in extra]:
            c.set_option(k, v)
        for _ in range(3):
            eng.render(spp)
            FilmTable().get_image()
        c.kernel_time()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.render(spp)
            FilmTable().get_image()
        dt = (time.perf_counter() - t0) / steps
        kms, nl = c.kernel_time()
        res[label] = {'ms_per_step': round(dt * 1e3, 4), 'kernel_ms': round(kms / max(nl, 1), 4),
                      'msamples_s': round(n * n * spp / dt / 1e6, 1)}
        print('sync_sweep', label, res[label], flush=True)
    out['sync_sweep'] = res
    save()
    common.reset_all()


def readback(name='s978', spp=32, n=512, rounds=50):
    '''what the end of a step costs on an otherwise idle GPU: get_image() (resolve kernel + D2H + the numpy
    array) against resolve + synchronize alone'''
    common.reset_all()
    eng = setup_engine(scenes.get_scene(name), n, n, mode='fast')
    c = ctx()
    c.set_option('batch', spp)
    eng.render(spp)
    FilmTable().get_image()
    res = {}
    t0 = time.perf_counter()
    for _ in range(rounds):
        FilmTable().get_image()
    res['get_image_us'] = (time.perf_counter() - t0) / rounds * 1e6
    t0 = time.perf_counter()
    for _ in range(rounds):
        c.call('mpt_resolve', 0)
        c.call('mpt_synchronize')
    res['resolve_sync_us'] = (time.perf_counter() - t0) / rounds * 1e6
    t0 = time.perf_counter()
    for _ in range(rounds):
        c.call('mpt_synchronize')
    res['sync_only_us'] = (time.perf_counter() - t0) / rounds * 1e6
    res['image_bytes'] = n * n * 16
    print('readback', json.dumps(res), flush=True)
    out['readback'] = res
    save()
    common.reset_all()


def stamps(name='s978', spp=32, n=512):
    '''needs a library built with -DMPT_X_STAMPS=1 (MIPTINA_LIB): shader-clock shares of the stages of the
    counting LDS kernel, per wave: NODE steps, LEAF steps, shadow-ray restarts, SHADE, NEW (+ pull), the rest
    (loop headers, ballots)'''
    common.reset_all()
    name = os.environ.get('STAMP_SCENE', name)          # s978 | c4 | c5 (the gather kernel's stages)
    kw, world = {}, None
    if name == 'c5':
        kw, n, spp = {'n': 1000000}, 1024, 16
    if name == 'c4':
        n, spp, world = 1024, 32, ([1.0, 1.0, 1.0, 1.0], 0)
    eng = setup_engine(scenes.get_scene(name, **kw), n, n, mode='fast', world=world, max_filmsize=max(n * n, 1 << 21))
    c = ctx()
    c.set_option('batch', spp)
    for kv in filter(None, os.environ.get('MIPTINA_OPTS', '').split(',')):    # A/B switches, e.g. lds_wide=1
        c.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    eng.render(spp)
    c.call('mpt_synchronize')
    c.set_option('count', 1)
    c.call('mpt_reset_counters')
    eng.render(spp)
    k = c.counters()
    c.set_option('count', 0)
    tot = k['n_node']
    res = {'node': k['n_box'] / tot, 'leaf': k['n_tri'] / tot, 'shadow_done': k['n_draws'] / tot, 'shade': k['n_shade'] / tot,
           'new': k['bounces'] / tot}
    res['rest'] = 1.0 - sum(res.values())
    res['stages_per_64_samples'] = {s: k['it_' + s] / k['samples'] * 64 for s in ('node', 'leaf', 'shade', 'new')}
    res['cycles_per_stage'] = {s: v * 256 / max(k['it_' + s], 1) for s, v in (('node', k['n_box']), ('leaf', k['n_tri']), ('shade', k['n_shade']), ('new', k['bounces']))}
    if os.environ.get('STAMPS3'):     # -DMPT_X_STAMPS=3: of NEW's cycles, the pull (once per work item) and the preparation of 64 primary rays
        res['new_segments_cycles_per_64_samples'] = {'pull': k['pl_local'] * 16 / k['samples'] * 64, 'prepare': k['pl_batches'] * 16 / k['samples'] * 64,
                                                     'new_total': k['bounces'] * 256 / k['samples'] * 64}
    elif k.get('pl_local', 0):         # -DMPT_X_STAMPS=2: the segments of SHADE, cycles per SHADE stage
        seg = (('entry_and_gather_issue', 'pl_trips'), ('lights_hit', 'pl_local'), ('geometry_material_after_gathers', 'pl_batches'), ('light_sample', 'pl_batch_lanes'),
               ('bsdf_eval_mis', 'pl_prim'), ('bsdf_sample', 'pl_tidle'), ('ray_start', 'pl_sidle'))
        res['shade_segments_cycles'] = {a: k[b] * 16 / max(k['it_shade'], 1) for a, b in seg}
    res['scene'] = name
    print('stamps', json.dumps(res), flush=True)
    out['stamps_' + name] = res
    save()
    common.reset_all()


def lane_hist():
    '''how many of a wave's 64 lanes take part in an issued NODE / LEAF / SHADE stage, and whose lanes they are
    (bounce depth, closest-hit or shadow ray) -- counting kernels with option lane_hist, for the headline scene (LDS-resident 4-wide
    kernel) and the two gather scenes.  Writes gpurun_out/lane_histogram.json'''
    import ctypes as C
    res = {}
    for key, name, kw, n, spp, world in (('s978', 's978', {}, 512, 32, None), ('c4', 'c4', {}, 1024, 16, ([1.0, 1.0, 1.0, 1.0], 0)),
                                         ('c5', 'c5', {'n': 1000000}, 1024, 8, None)):
        common.reset_all()
        eng = setup_engine(scenes.get_scene(name, **kw), n, n, mode='fast', world=world, max_filmsize=max(n * n, 1 << 21))
        c = ctx()
        c.set_option('batch', min(spp, 32))
        eng.render(1)
        c.call('mpt_synchronize')
        c.set_option('count', 1)
        c.set_option('lane_hist', 1)
        c.call('mpt_reset_counters')
        eng.render(spp)
        h = (C.c_uint64 * 255)()
        c.call('mpt_get_lane_hist', h, 255)
        c.set_option('count', 0)
        c.set_option('lane_hist', 0)
        h = np.array(list(h), dtype=np.float64)
        out_k = {'kernel': ('gather', 'lds', 'gather4', 'lds_pool', 'gather8', 'lds4')[c.get_option('last_kernel')], 'film': [n, n], 'spp': spp}
        for si, stage in enumerate(('NODE', 'LEAF', 'SHADE')):
            hist = h[si * 65:(si + 1) * 65]
            stages = hist.sum()
            lanes = (hist * np.arange(65)).sum()
            comp = h[195 + si * 12:195 + (si + 1) * 12].reshape(6, 2)
            out_k[stage] = {
                'stages_issued': int(stages), 'mean_lanes': round(lanes / max(stages, 1), 2),
                'stages_by_lanes_1_8__57_64': [round(float(hist[1 + 8 * b:9 + 8 * b].sum() / max(stages, 1)), 4) for b in range(8)],
                'lane_steps_by_depth_closest': [round(float(comp[d, 0] / max(lanes, 1)), 4) for d in range(6)],
                'lane_steps_by_depth_shadow': [round(float(comp[d, 1] / max(lanes, 1)), 4) for d in range(6)]}
        idh = h[231:255]
        if idh.sum() > 0:
            cum = np.cumsum(idh) / idh.sum()
            out_k['node_steps_with_number_below'] = {str(1 << k): round(float(cum[k]), 4) for k in range(6, 21)}
            out_k['wide_nodes'] = c.get_option('wide_nodes')
        res[key] = out_k
        print('lane_hist', key, json.dumps(out_k), flush=True)
    out['lane_hist'] = res
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'lane_histogram.json'), 'w'), indent=1)
    save()
    common.reset_all()


if __name__ == '__main__':
    what = sys.argv[1:] or ['parity', 'timing']
    if 'lane_hist' in what:
        lane_hist()
    if 'stamps' in what:
        stamps()
    if 'readback' in what:
        readback()
    if 'shares_sync' in what:
        shares_sync()
    if 'sync_sweep' in what:
        sync_sweep()
    if 'probe' in what:
        probe()
    if 'c5sah' in what:
        out['c5sah'] = c5sah()
        save()
    if 'c3' in what:
        out['c3'] = c3()
        print('c3 2048x2048', json.dumps(out['c3']), flush=True)
        save()
    if 'slabs' in what:
        out['slabs'] = slabs()
        for k, v in out['slabs'].items():
            print('slabs', k, json.dumps(v), flush=True)
        save()
    if 'blk' in what:
        out['blk'] = blk()
        save()
    if 'fixedcost' in what:
        out['fixedcost'] = fixedcost()
        save()
    if 'timeline' in what:
        out['timeline'] = timeline()
        save()
    if 'pipe' in what:
        out['pipe'] = pipe()
        save()
    if 'util' in what:
        out['util'] = util()
        for k, v in out['util'].items():
            print('util sched', k, json.dumps(v), flush=True)
        save()
    if 'big' in what:
        for nm, n, spp, kw in (('c4', 512, 16, {}), ('c5', 512, 8, {'n': 1000000})):
            out['big_' + nm] = big(nm, n, spp, **kw)
            print('big', nm, json.dumps(out['big_' + nm]), flush=True)
            save()
    if 'tree' in what:
        for nm in ('s978', 's34'):
            out['tree_' + nm] = tree_ab(nm)
            print('tree A/B', nm, json.dumps(out['tree_' + nm]), flush=True)
            save()
    if 'ab' in what:
        out['ab'] = ab()
        print('A/B variants (kernel ms):', json.dumps(out['ab']), flush=True)
        save()
    if 'parity' in what:
        for name, nx, ny, spp in (('s34', 64, 64, 8), ('s978', 96, 96, 8), ('s978', 128, 128, 32)):
            out[f'parity_{name}_{nx}x{ny}x{spp}'] = parity(name, nx, ny, spp)
            print(name, nx, ny, spp, json.dumps(out[f'parity_{name}_{nx}x{ny}x{spp}'])[:600], flush=True)
            save()
    if 'timing' in what:
        for sched in ((8, 1), (4, 1), (3, 1), (2, 1), (3, 2), (1, 1)):
            for ch in (1, 2, 4):
                r = timing('s978', 'fast', ch, sched=sched)
                out[f'sched_{sched[0]}_{sched[1]}_chunk{ch}'] = r
                print('s978 sched', sched, 'chunk', ch, r, flush=True)
                save()
        for name in ('s978', 's34'):
            for lds in (1, 0):
                for ch in (0,):
                    r = timing(name, 'fast', ch, lds=lds)
                    out[f'timing_{name}_fast_lds{lds}_chunk{ch}'] = r
                    print(name, 'fast lds', lds, 'chunk', ch, r, flush=True)
                    save()
            r = timing(name, 'strict', 32)
            out[f'timing_{name}_strict'] = r
            print(name, 'strict', r, flush=True)
            save()
    save()
