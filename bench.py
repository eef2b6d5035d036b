#!/usr/bin/env python3
'''
bench.py -- Msamples/s of the path-trace hot path on the 978-triangle cornell scene,
512x512, 32 spp (BASELINE.json configs[1]), on N MI355X of one node.

One "step" = the timed region of the reference's exams/benchmark.py:29-35: 32 x PathEngine.render()
(each = Sobol update + one sample per pixel) followed by the film resolve of get_image().  The
scene, BVH, Sobol tables and film are resident in HBM before the timed region starts; `value`
stops at the resolved image in HBM, the D2H-inclusive rate is reported beside it.

N > 1 (launched by torch.distributed.run, one rank per GPU): the film columns are dealt out in
stripes of 16, rank r renders stripes r, r+N, ... of the replicated scene and the stripes are
gathered to rank 0 with grouped ncclSend/ncclRecv (RCCL over xGMI) before the resolve -- strong
scaling of the same 512x512x32 job.  No PyTorch anywhere in the process: rendezvous of the RCCL unique id is a file,
barriers and the max-over-ranks are RCCL all-reduces.
'''

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402

NX = NY = 512
SPP = 32
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(c):
    '''film float4 read+write, 4 B per Sobol draw, 32 B per box test
    (bmin+bmax+children), 40 B per triangle test (leaf id + 3 positions), and per shaded hit 64 B
    of normals/uvs/mtlid + the 64-B packed material record'''
    return (32 * c['samples'] + 4 * c['n_draws'] + 32 * c['n_box'] + 40 * c['n_tri']
            + 128 * c['n_shade'])


def measured_traffic():
    '''HBM bytes per render launch from the committed rocprofv3 PMC passes of this same command
    (profiles/: FETCH_SIZE and WRITE_SIZE are in KiB; scattered 16-B accesses, so the guide's x2
    read correction for wide coalesced streams is not applied), or None'''
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')))
        for k, v in d.items():
            if 'render_kernel_lds<false' in k:
                return int((v['FETCH_SIZE']['median'] + v['WRITE_SIZE']['median']) * 1024)
    except Exception:
        pass
    return None


def measured_valu(avg_kernel_s, n_simd=1024, clock_hz=2.4e9):
    '''what actually bounds the LDS kernel, from the same committed PMC passes: the fraction of SIMD
    cycles with a VALU instruction in flight (SQ_ACTIVE_INST_VALU counts 4-cycle issue slots) and the
    fraction of lanes those instructions had switched on'''
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')))
        for k, v in d.items():
            if 'render_kernel_lds<false' in k:
                active = v['SQ_ACTIVE_INST_VALU']['median']
                return {'valu_busy_frac': round(active * 4.0 / (n_simd * clock_hz * avg_kernel_s), 3),
                        'lane_utilisation': round(v['SQ_THREAD_CYCLES_VALU']['median'] / (64.0 * active), 3),
                        'valu_insts_per_launch': int(v['SQ_INSTS_VALU']['median'])}
    except Exception:
        pass
    return None


def host_threads():
    '''threads for the CPU baseline: the cores this process may run on, capped at the GPU box's
    per-GPU CPU share (16)'''
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get('MIPTINA_CPU_THREADS', '16'))))


def cpu_baseline(scene, camera, budget_s=15.0):
    '''the CPU restatement (oracle, kind "port") on this host's cores, on a column window of the
    same 512x512 workload sized to ~budget_s of wall time'''
    import oracle
    from helpers import setup_oracle
    threads = host_threads()
    o = setup_oracle(oracle, scene, NX, NY, camera=camera, threads=threads)
    o.set_window(248, 248 + threads)
    o.render(1)                                   # thread start-up, page faults
    t0 = time.time()
    o.render(1)
    per_col_frame = (time.time() - t0) / threads
    o.sobol_reset(64)
    frames = SPP
    full = per_col_frame * NX * frames            # estimated wall time of the whole workload
    if full <= 2 * budget_s:
        cols, x0 = NX, 0                          # the whole 512x512x32 job
    else:
        cols = int(max(threads, NX * budget_s / full))
        cols -= cols % threads
        x0 = (NX - cols) // 2
    o.set_window(x0, x0 + cols)
    o.clear()
    o.render(1)                                   # warm-up frame of exams/benchmark.py:25-27
    o.clear()
    t0 = time.time()
    o.render(frames)
    dt = time.time() - t0
    samples = cols * NY * frames
    # single-thread rate on 8 centre columns x 4 spp (SURVEY 8d asks for both figures)
    o1 = setup_oracle(oracle, scene, NX, NY, camera=camera, threads=1)
    o1.set_window(252, 260)
    o1.render(1)
    t1 = time.time()
    o1.render(4)
    one = 8 * NY * 4 / (time.time() - t1) / 1e6
    return {'value': round(samples / dt / 1e6, 4), 'unit': 'Msamples/s', 'cores': threads, 'kind': 'port',
            'value_1thread': round(one, 4),
            'sample': f'columns [{x0},{x0 + cols}) of the 512x512 film x {frames} spp = {samples} samples '
                      f'in {dt:.1f} s (C restatement of PTina\'s algorithm, OpenMP, {threads} threads)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--scene', default='s978')
    ap.add_argument('--mode', default='fast')
    ap.add_argument('--chunk', type=int, default=-1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-comm', action='store_true', help='create the RCCL communicator even for one rank')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    from ptina_amd import scenes, _lib
    from ptina_amd.common import ctx
    from ptina_amd.things import FilmTable
    from ptina_amd.dist import RcclFilm
    from helpers import setup_engine

    scene = scenes.get_scene(args.scene)
    eng = setup_engine(scene, NX, NY, mode=args.mode)
    c = ctx()
    if args.chunk >= 0:
        c.set_option('chunk', args.chunk)
    c.set_option('batch', SPP)
    comm = RcclFilm(rank, world) if (world > 1 or args.force_comm) else None
    if comm:
        comm.set_stripes(NX)                  # every world-th stripe of 16 columns: even load

    def step():
        eng.render(SPP)                       # 32 x (Sobol update + 1 spp), one fused launch
        c.call('mpt_flush')
        if comm:
            comm.gather(0, 0)
        if rank == 0:
            c.call('mpt_resolve', 0)

    def barrier():
        c.call('mpt_synchronize')
        if comm:
            comm.barrier()

    # exams/benchmark.py:25-27: warm-up frame, read back, clear
    eng.render()
    FilmTable().get_image()
    FilmTable().clear()

    # warm-up steps double as the counting pass for the roofline's algorithmic bytes
    c.set_option('count', 1)
    c.call('mpt_reset_counters')
    W = max(args.warmup, 1)
    for _ in range(W):
        step()
    barrier()
    cnt = c.counters()
    c.set_option('count', 0)
    c.kernel_time()
    bytes_per_launch = algorithmic_bytes(cnt) / W
    for _ in range(1):                        # one untimed step of the production (non-counting) kernel
        step()
    barrier()
    c.kernel_time()

    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    kms, nlaunch = c.kernel_time()
    if comm:
        dt = comm.allreduce_max(dt)

    # D2H-inclusive variant of the same step (the reference's get_image returns a host array)
    d2h = None
    if rank == 0 and world == 1:
        reps = max(args.steps // 4, 3)
        eng.render(SPP)
        FilmTable().get_image()               # untimed: first read-back allocates the staging buffer
        t1 = time.perf_counter()
        for _ in range(reps):
            eng.render(SPP)
            FilmTable().get_image()
        d2h = NX * NY * SPP * reps / (time.perf_counter() - t1) / 1e6

    if rank == 0:
        total = NX * NY * SPP * args.steps
        value = total / dt / 1e6
        avg_kernel_s = kms / 1e3 / max(nlaunch, 1)
        # small launches (N > 1 slabs) take 1/G of the CUs each and G of them are resident at once
        concurrent = max(c.get_option('cur_div'), 1)
        achieved = bytes_per_launch * concurrent / avg_kernel_s / 1e9
        out = {
            'metric': 'Msamples/sec (pixels x spp / s), 512x512x32spp cornell-monkey (978 tri)',
            'value': round(value, 3), 'unit': 'Msamples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': W, 'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.scene}: 978-tri synthetic cornell + bumpy sphere, {NX}x{NY}, {SPP} spp, '
                                   'unidirectional MIS path tracer, depth<=5', 'film': [NX, NY], 'spp': SPP,
                       'mode': args.mode, 'parallelism': f'film columns in 16-wide stripes over {world} GPUs, RCCL gather' if world > 1 else 'single GPU'},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 3), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 6), 'traffic': measured_traffic(),
                         'kernel': ('render_kernel_lds' if c.get_option('last_kernel') else 'render_kernel_fast')
                         if args.mode == 'fast' else 'render_kernel_strict',
                         'avg_kernel_ms': round(avg_kernel_s * 1e3, 4), 'launches': nlaunch, 'concurrent_launches': concurrent,
                         'algorithmic_bytes_per_launch': int(bytes_per_launch),
                         'bytes_per_sample': round(bytes_per_launch / (cnt['samples'] / W), 2),
                         'note': 'algorithmic bytes (SURVEY 8d) over kernel time; the 125 KB of nodes+triangles are '
                                 'served from LDS, so this exceeds what HBM could deliver and HBM is not the binding '
                                 'limit: the kernel is VALU-issue bound (profiles/, DESIGN.md)'},
            'valu': measured_valu(avg_kernel_s) if (world == 1 and args.mode == 'fast' and args.scene == 's978') else None,
            'counters_per_sample': {k: round(v / max(cnt['samples'], 1), 3) for k, v in cnt.items() if k != 'samples'},
            'mrays_per_s': round(cnt['rays'] / W / avg_kernel_s / 1e6, 1),
        }
        if d2h is not None:
            out['value_incl_d2h'] = round(d2h, 3)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(scene, scenes.BENCH_CAMERA)
        print(json.dumps(out))
    if comm:
        comm.barrier()
        comm.close()


if __name__ == '__main__':
    main()
