#!/usr/bin/env python3
'''GPU: BVHTree().build() (mpt_build_tree) of one big configuration, for `rocprofv3 --kernel-trace --stats`.
One untimed build (allocations), then REPS timed ones; prints the wall time of each and the library's own phase times
(option "build_phase_us_<k>": upload | LBVH | SAH pass | triangle records | 4-wide collapse).
usage: rocprofv3 --kernel-trace --stats -d gpurun_out/build_c5 -- python3 tools/build_profile.py c5 [reps]'''
import json
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

PHASES = ('upload', 'lbvh', 'sah', 'tri_records', 'wide', 'total')


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'c5'
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    kw = {'n': 1000000} if name == 'c5' else {}
    scene = scenes.get_scene(name, **kw)
    n = int(scene[1].shape[0])
    reset_all()
    setup_engine(scene, 16, 16, mode='fast', max_faces=n + 1)      # (builds once: allocations, first-call costs)
    c = ctx()
    for kv in filter(None, os.environ.get('MIPTINA_OPTS', '').split(',')):
        c.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    try:
        c.set_option('build_phases', int(os.environ.get('BUILD_PHASES', '1')))
    except RuntimeError:
        pass                                      # (a library from before round 6: wall times only)
    out = {'scene': name, 'ntri': n, 'runs': []}
    from ptina_amd.things import ModelPool
    for _ in range(reps):
        if os.environ.get('BUILD_RESIDENT', '0') != '1':
            ModelPool().load(scene[0], scene[1])          # (the model in host memory: the build uploads it, as rounds 1-5 timed it)
        c.call('mpt_synchronize')
        t0 = time.perf_counter()
        BVHTree().build()
        c.call('mpt_synchronize')
        dt = time.perf_counter() - t0
        run = {'wall_ms': round(dt * 1e3, 3)}
        try:
            for k, ph in enumerate(PHASES):
                run[ph + '_ms'] = c.get_option(f'build_phase_us_{k}') / 1e3
        except RuntimeError:
            pass
        try:
            for k in ('sah_levels', 'sah_kelems', 'sah_chunks', 'sah_segments', 'sah_part_kwords', 'sah_tasks_small', 'sah_tasks_big',
                      'sah_t_sort_k', 'sah_t_loop_k', 'sah_t_max_k', 'sah_task_levels', 'sah_task_levels_max'):
                run[k] = c.get_option(k)
        except RuntimeError:
            pass
        run['fast_depth'] = c.get_option('fast_depth')
        run['wide_nodes'] = c.get_option('wide_nodes')
        run['wide_depth'] = c.get_option('wide_depth')
        out['runs'].append(run)
        print(json.dumps(run), flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', f'build_profile_{name}{os.environ.get("BUILD_TAG", "")}.json'), 'w'), indent=1)
    reset_all()


if __name__ == '__main__':
    main()
