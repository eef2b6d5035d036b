#!/usr/bin/env python3
'''
bench.py -- Msamples/s of the path-trace hot path on the 978-triangle cornell scene,
512x512, 32 spp (BASELINE.json configs[1]), on N MI355X of one node.

One "step" = the timed region of the reference's exams/benchmark.py:29-36: PathEngine.render(32) -- the
reference's 32 x render(), each = Sobol update + one sample per pixel, enqueued by ONE call and fused into one
launch (result-identical to 32 calls: tests/test_parity_gpu.py::test_batching_does_not_change_the_film) --
followed by FilmTable.get_image(): the resolve AND the read-back of the 4 MiB image into a host array.  The scene, BVH, Sobol tables and film are
resident in HBM before the timed region starts.  `value` is that D2H-inclusive rate (the metric
SURVEY.md 8d defines); `value_resolve_only` (stops at the resolved image in HBM, consecutive steps
overlapped) is reported beside it.

N > 1, one rank per GPU: launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or -- when WORLD_SIZE is not set -- by this script itself, which then only spawns N
fresh copies of itself (ptina_amd.dist.launch_ranks) before anything has touched the GPU and relays
rank 0's line.  The film columns are dealt out in stripes of 16, rank r renders stripes r, r+N, ...
of the replicated scene and the stripes are gathered to rank 0 -- each rank's share packed into ONE
message, one ncclSend per rank / N - 1 ncclRecv on the root (RCCL over xGMI), scattered into the film in
front of the resolve: strong scaling of the same 512x512x32 job (`value`, `config`: unchanged by N).
After the headline steps an N > 1 run also times BASELINE configs[2]'s film -- s978 2048x2048, 32 spp per
step, striped and gathered the same way -- and adds `c3` to the line: the configuration on which
near-linear scaling is reachable (16 x the work per launch; the headline film is 3.6 ms of work for ONE GPU).
No PyTorch in the process; the RCCL unique id travels through a file, barriers and the max-over-ranks are
RCCL all-reduces.

Roofline (N = 1): the dominant kernel is bound by VALU issue at partial lane occupancy, not by HBM
and not by MFMA -- its scene lives in LDS.  The `roofline` object therefore prices VALU lane-operations
per second against the chip's f32 vector lane rate (256 CUs x 4 SIMD-32 x clock), with the counters
collected IN THIS RUN: before the parent touches the GPU it runs itself twice under `rocprofv3 --pmc`
(separate passes, a few seconds each) and reads SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU /
SQ_THREAD_CYCLES_VALU and FETCH_SIZE / WRITE_SIZE of the render kernel from the CSVs.  `hbm` holds the
measured HBM traffic and its GB/s.  Without rocprofv3 the committed summary of the same command
(profiles/) is used and `counters_from` says so.
'''

import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

NX = NY = 512
SPP = 32
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SIMD = 256 * 4             # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32
SIMD_LANES = 32              # a wave64 VALU instruction issues over 2 cycles on a SIMD-32
PROFILE_FALLBACK = os.path.join(ROOT, 'profiles', 'r06_pmc_summary.json')
if not os.path.exists(PROFILE_FALLBACK):
    PROFILE_FALLBACK = os.path.join(ROOT, 'profiles', 'r05_pmc_summary.json')
# The gather kernels' roofline (scenes that do not fit LDS): 64-byte records per second against what tools/microbench/gather_microbench
# reaches with nothing else to do -- a dependent chain of random 64-byte records, four 16-byte loads each, 5 workgroups of 256 lanes per
# CU, 58 % of the lanes taking part in a step (profiles/r06_gather_microbench.log, MI355X, re-measured this round): 194.0 G records/s inside L2, and per size of
# the table beyond it.  A traversal step reads one 64-byte node (four gathers), a triangle test one 48-byte record (three: 0.75 record).
GATHER_IN_L2_GRECS = 193.98                           # ... at the kernels' own lane activity (58 %), 5 workgroups per CU
GATHER_BEYOND_L2_GRECS = {'c4': 79.31, 'c5': 59.18}   # 16.8 MB table (C4's records: ~15 MB) / 67 MB table (C5's: ~80 MB), profiles/r06_gather_microbench.log
# `frac` is quoted against the FULL-LANE ceiling of the same access form (every lane of every wave taking part:
# what the kernel would reach if its steps were never short of lanes): 211.4 G records/s inside L2, 79.1 / 58.7 beyond it
# (profiles/r06_gather_microbench.log, rows "own ... W 5 active 1.00")
GATHER_FULL_IN_L2_GRECS = 211.41
GATHER_FULL_BEYOND_L2_GRECS = {'c4': 79.06, 'c5': 58.72}
GATHER_L2_HIT = {'c4': 0.67, 'c5': 0.90}              # fallback when the run collects no counters: TCC_HIT / (TCC_HIT + TCC_MISS), profiles/r05_pmc_big_summary.json
CONFIG_PMC_PASSES = [['FETCH_SIZE', 'GRBM_GUI_ACTIVE'], ['WRITE_SIZE', 'TCC_HIT_sum', 'TCC_MISS_sum'],
                     ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_THREAD_CYCLES_VALU']]
C3_N = 2048                  # BASELINE.json configs[2]: 2048 x 2048 film ...
C3_SPP = 256                 # ... at 256 spp: one c3 step = render(256) (eight launches of 32 frames, pipelined) + gather + get_image()
# the launch model of DESIGN.md section 6, measured on ONE MI355X (tools/gpu_diag.py shares_sync): a launch of 1/N of
# a film costs a / N + b -- b = the end-of-launch drain, independent of N
MODEL = {'headline': {'a_ms': 2.18, 'b_ms': 0.17}, 'c3': {'a_ms': 34.0, 'b_ms': 0.17},
         'from': 'one-GPU share measurements, DESIGN.md section 6 (render_kernel_lds4: 2.35 / 1.30 / 0.74 / 0.45 ms per launch for N = 1 / 2 / 4 / 8; profiles/r06_shares_sync.json)'}

PMC_PASSES = [
    ['SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_THREAD_CYCLES_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES',
     'SQ_BUSY_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY'],
    ['FETCH_SIZE', 'GRBM_GUI_ACTIVE'],
    ['WRITE_SIZE', 'TCC_HIT_sum', 'TCC_MISS_sum'],
    # the instruction mix, and how busy the LDS pipe is (round 5)
    ['SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_CVT', 'SQ_INSTS_VALU_TRANS_F32',
     'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT'],
]


def is_counting_kernel(name):
    '''the counting instantiation of a render kernel (COUNT = true), by its template arguments:
    render_kernel_lds<COUNT>, render_kernel_pool<COUNT>, render_kernel_wide<COUNT, QUANT>, render_kernel_fast<STACK, COUNT>.
    (Round 3 tested for ", true>" and with it threw away the production gather kernel render_kernel_wide<false, true>.)'''
    import re
    m = re.search(r'render_kernel_(\w+?)<([^>]*)>', name)
    if not m:
        return False
    kind, args = m.group(1), [a.strip() for a in m.group(2).split(',')]
    if kind.startswith('wide'):
        return args[0] == 'true'
    if kind.startswith('fast') or kind.startswith('strict'):
        return args[-1] == 'true'
    return args[0] == 'true'                      # lds, pool


def render_kernel_name(mode, last_kernel):
    if mode != 'fast':
        return 'render_kernel_strict'
    return ('render_kernel_fast', 'render_kernel_lds', 'render_kernel_wide', 'render_kernel_pool', 'render_kernel_oct', 'render_kernel_lds4')[last_kernel]


def collect_pmc(argv_tail, budget_s=150, passes=None, role='pmc-child'):
    '''run this script (role pmc-child: the same steps, nothing printed) under `rocprofv3 --pmc`, one
    pass per counter group, and return {counter: median per 32-spp render launch}.  Called BEFORE the
    parent has made any HIP call; rocprofv3 gets the interpreter itself after `--`.'''
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    out = {}
    work = tempfile.mkdtemp(prefix='miptina_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    # the profiler's preloaded tool initialises HIP before libmiptina can ask for more hardware queues: ask here, so
    # that the counters come from the queue configuration the timed run uses (option "hw_queues")
    env.setdefault('GPU_MAX_HW_QUEUES', '12')
    try:
        passes = PMC_PASSES if passes is None else passes
        for i, counters in enumerate(passes):
            d = os.path.join(work, f'pass{i}')
            cmd = [exe, '--pmc', *counters, '--output-format', 'csv', '-d', d, '--',
                   sys.executable, os.path.abspath(__file__), '--role', role, *argv_tail]
            try:
                p = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                     start_new_session=True)
                try:
                    _, err = p.communicate(timeout=budget_s)
                except subprocess.TimeoutExpired:
                    os.killpg(p.pid, 15)
                    try:
                        p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        os.killpg(p.pid, 9)
                    return None, f'rocprofv3 pass {i} exceeded {budget_s}s'
                if p.returncode != 0:
                    return None, f'rocprofv3 pass {i} exited {p.returncode}: {err.decode("utf-8", "replace")[-300:]}'
            except OSError as e:
                return None, f'rocprofv3 could not start: {e}'
            rows = []
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                rows += list(csv.DictReader(open(f)))
            per = {}
            for r in rows:
                k = r['Kernel_Name']
                if 'render_kernel' not in k or is_counting_kernel(k):              # not the counting build
                    continue
                per.setdefault((k, r['Counter_Name']), {}).setdefault(r['Dispatch_Id'], 0.0)
                per[(k, r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
            for (k, cn), by_dispatch in per.items():
                vals = sorted(by_dispatch.values())
                # the 1-spp warm-up frame of the benchmark sequence is a render launch too: keep the big ones
                big = [v for v in vals if v >= 0.5 * vals[-1]] if vals[-1] > 0 else vals
                out[cn] = {'median': big[len(big) // 2], 'n': len(big), 'kernel': k}
        return out, 'rocprofv3 --pmc in this run (%d passes, %s)' % (len(passes), os.path.basename(exe))
    finally:
        shutil.rmtree(work, ignore_errors=True)


def committed_pmc(kernel):
    try:
        d = json.load(open(PROFILE_FALLBACK))
        for k, v in d.items():
            if kernel in k and not is_counting_kernel(k):
                return {cn: dict(x, kernel=k) for cn, x in v.items()}, 'profiles/' + os.path.basename(PROFILE_FALLBACK)
    except Exception:
        pass
    return None, 'no counters available'


# What a wave64 VALU instruction costs a SIMD, measured on MI355X (tools/microbench/valu_microbench, exec_microbench; profiles/r05_*_microbench.log):
# the issue port takes one instruction per 2.2-2.3 cycles whatever its kind -- a stream that alternates v_fma_f32 with v_max_f32 or
# v_cvt_f32_ubyte runs at the pure FMA stream's rate -- while compares, min / max, conversions, v_perm, three-operand and 64-bit integer
# forms keep a second unit busy for 4.1-4.2 cycles each (a stream of those alone runs at half rate), transcendentals for 8.1.  A mix
# is therefore bound by max(all instructions x 2.25, half-rate ones x 4.2), not by the sum (round 5 first priced it by the sum:
# DESIGN.md 3.1).  The hardware counts add, mul, fma, 32-bit integer, conversion and transcendental instructions separately; the
# half-rate share cannot be read off those classes, so the line reports the port alone.
ISSUE_CYCLES = 2.25
MIX_COUNTERS = ('SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_CVT',
                'SQ_INSTS_VALU_TRANS_F32')


def instruction_mix(g, insts):
    if g('SQ_INSTS_VALU_FMA_F32') is None:
        return None
    mix = {k: g(k) or 0.0 for k in MIX_COUNTERS}
    mix['other'] = max(insts - sum(mix.values()), 0.0)
    return {k.replace('SQ_INSTS_VALU_', '').lower(): round(v / insts, 4) for k, v in mix.items()}


def roofline_blocks(pmc, source, kernel, avg_kernel_s, clock_hz, concurrent):
    '''VALU roofline + HBM traffic of the dominant kernel from its PMC counters (per launch) and its
    average launch duration measured with HIP events in this run.

      wave-instructions      = SQ_INSTS_VALU
      lane occupancy         = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
      achieved lane-ops/s    = SQ_INSTS_VALU * 64 * lane occupancy / kernel time
      peak lane-ops/s        = 1024 SIMD-32 x 32 lanes x clock (= 157.3 TFLOP/s / 2 at 2.4 GHz)
      issue fraction         = SQ_INSTS_VALU * 2.25 cycles / (1024 SIMDs * clock * kernel time)
      LDS pipe fraction      = SQ_LDS_IDX_ACTIVE / (256 CUs * clock * kernel time)
    (2.25 cycles per wave64 VALU instruction of any kind at the issue port: ISSUE_CYCLES above)'''
    peak = N_SIMD * SIMD_LANES * clock_hz
    if not pmc or 'SQ_INSTS_VALU' not in pmc:
        return ({'bound': 'valu', 'achieved': None, 'peak': round(peak / 1e12, 3), 'unit': 'Tlane-op/s', 'frac': None,
                 'traffic': None, 'kernel': kernel, 'avg_kernel_ms': round(avg_kernel_s * 1e3, 4),
                 'counters_from': source}, None)
    g = lambda c: pmc[c]['median'] if c in pmc else None
    insts = g('SQ_INSTS_VALU')
    occ = g('SQ_THREAD_CYCLES_VALU') / (64.0 * g('SQ_ACTIVE_INST_VALU'))
    lane_ops = insts * 64.0 * occ
    achieved = lane_ops * concurrent / avg_kernel_s
    fetch, write = g('FETCH_SIZE'), g('WRITE_SIZE')
    traffic = int((fetch + write) * 1024) if fetch is not None and write is not None else None
    roof = {
        'bound': 'valu', 'achieved': round(achieved / 1e12, 4), 'peak': round(peak / 1e12, 3), 'unit': 'Tlane-op/s',
        'frac': round(achieved / peak, 4), 'traffic': traffic,
        'kernel': kernel, 'avg_kernel_ms': round(avg_kernel_s * 1e3, 4), 'concurrent_launches': concurrent,
        'clock_ghz': round(clock_hz / 1e9, 3),
        'valu_insts_per_launch': int(insts), 'lane_occupancy': round(occ, 4),
        'valu_issue_frac': round(insts * ISSUE_CYCLES * concurrent / (N_SIMD * clock_hz * avg_kernel_s), 4),
        'lds_pipe_frac': round(g('SQ_LDS_IDX_ACTIVE') * concurrent / (N_SIMD / 4 * clock_hz * avg_kernel_s), 4) if g('SQ_LDS_IDX_ACTIVE') else None,
        'salu_per_valu': round(g('SQ_INSTS_SALU') / insts, 3) if g('SQ_INSTS_SALU') else None,
        'wave_cycles_valu_frac': round(g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'), 3) if g('SQ_WAVE_CYCLES') else None,
        'wave_cycles_wait_inst_frac': round(g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'), 3) if g('SQ_WAVE_CYCLES') and g('SQ_WAIT_INST_ANY') else None,
        'counters_from': source,
        'instruction_mix': instruction_mix(g, insts),
        'note': 'f32 vector lane-operations per second against 256 CU x 4 SIMD-32 x clock; the node and triangle records '
                'are LDS-resident, so HBM is not the binding limit (see "hbm") and there is no contraction '
                'for MFMA',
    }
    hbm = None
    if traffic is not None:
        gbs = traffic * concurrent / avg_kernel_s / 1e9
        hbm = {'traffic_bytes': traffic, 'fetch_bytes': int(fetch * 1024), 'write_bytes': int(write * 1024),
               'GB/s': round(gbs, 2), 'peak': HBM_PEAK_GBS, 'frac': round(gbs / HBM_PEAK_GBS, 5),
               'l2_hit_rate': round(g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')), 4) if g('TCC_HIT_sum') else None,
               'note': 'FETCH_SIZE + WRITE_SIZE (KiB) per render launch; scattered 16-B accesses, so the guide\'s x2 '
                       'correction for wide coalesced reads does not apply'}
    return roof, hbm


def algorithmic_flops(c):
    '''~30 flop per box test (6 divides included), ~60 per triangle test, 150-300 per shaded
    hit (225 taken), of the traversal actually run (counting build).  Unlike the executed-instruction figures of
    `roofline` this one does not fall when an optimisation removes instructions: it is work / time.'''
    return 30.0 * c['n_box'] + 60.0 * c['n_tri'] + 225.0 * c['n_shade']


def algorithmic_bytes(c):
    '''film float4 read+write, 4 B per Sobol draw, 32 B per box test
    (bmin+bmax+children), 40 B per triangle test (leaf id + 3 positions), and per shaded hit 64 B
    of normals/uvs/mtlid + the 64-B packed material record'''
    return (32 * c['samples'] + 4 * c['n_draws'] + 32 * c['n_box'] + 40 * c['n_tri']
            + 128 * c['n_shade'])


def host_threads():
    '''threads for the CPU baseline: the CPUs this process can really use (affinity mask capped by the cgroup quota: a GPU box
    shows the host's 256 cores and gives one GPU's share of 16), capped at MIPTINA_CPU_THREADS (default 16)'''
    import oracle
    return max(1, min(oracle.usable_cpus(), int(os.environ.get('MIPTINA_CPU_THREADS', '16'))))


def _cpu_rate(oracle, scene, camera, threads, budget_s):
    '''Msamples/s of the CPU restatement with `threads` OpenMP threads on a column window of the 512x512x32 workload sized to
    ~budget_s of wall time (the whole job if it fits twice that); returns (rate, samples, seconds, x0, columns)'''
    from helpers import setup_oracle
    o = setup_oracle(oracle, scene, NX, NY, camera=camera, threads=threads)
    probe = min(threads, NX)
    o.set_window((NX - probe) // 2, (NX - probe) // 2 + probe)
    o.render(1)                                   # thread start-up, page faults
    t0 = time.time()
    o.render(1)
    per_col_frame = (time.time() - t0) / probe
    o.sobol_reset(64)
    frames = SPP
    full = per_col_frame * NX * frames            # estimated wall time of the whole workload
    if full <= 2 * budget_s:
        cols, x0 = NX, 0                          # the whole 512x512x32 job
    else:
        cols = int(max(probe, NX * budget_s / full))
        cols -= cols % probe
        x0 = (NX - cols) // 2
    o.set_window(x0, x0 + cols)
    o.clear()
    o.render(1)                                   # warm-up frame of exams/benchmark.py:25-27
    o.clear()
    t0 = time.time()
    o.render(frames)
    dt = time.time() - t0
    samples = cols * NY * frames
    return samples / dt / 1e6, samples, dt, x0, cols


def cpu_baseline(scene, camera, budget_s=12.0):
    '''the CPU restatement (oracle, kind "port") on this host's cores: with the GPU box's per-GPU CPU share (16 threads; `value`,
    `cores`), with every core this process may run on (`value_all_cores`, `cores_all`: BASELINE.md section 2 asks for both the
    1-thread and the all-thread figure; on a GPU box the cgroup quota IS 16 CPUs of the host's 256, and more threads than CPUs only
    slow the run down: measured 0.52 Msamples/s with 256 threads against 2.5 with 16) and single-threaded -- each on a column window of the same 512x512x32 workload'''
    import oracle
    from helpers import setup_oracle
    threads = host_threads()
    rate, samples, dt, x0, cols = _cpu_rate(oracle, scene, camera, threads, budget_s)
    allc = max(1, min(oracle.usable_cpus(), int(os.environ.get('MIPTINA_CPU_THREADS_ALL', '1024'))))
    all_rate, all_samples, all_dt = rate, samples, dt
    if allc != threads:
        all_rate, all_samples, all_dt, _, _ = _cpu_rate(oracle, scene, camera, allc, budget_s / 2)
    # single-thread rate on 8 centre columns x 4 spp (SURVEY 8d asks for both figures)
    o1 = setup_oracle(oracle, scene, NX, NY, camera=camera, threads=1)
    o1.set_window(252, 260)
    o1.render(1)
    t1 = time.time()
    o1.render(4)
    one = 8 * NY * 4 / (time.time() - t1) / 1e6
    return {'value': round(rate, 4), 'unit': 'Msamples/s', 'cores': threads, 'host_cores': os.cpu_count(),
            'kind': 'port',
            'value_all_cores': round(all_rate, 4), 'cores_all': allc,
            'value_1thread': round(one, 4),
            'sample': f'columns [{x0},{x0 + cols}) of the 512x512 film x {SPP} spp = {samples} samples '
                      f'in {dt:.1f} s (C restatement of PTina\'s algorithm, OpenMP, {threads} threads); all cores: {all_samples} samples '
                      f'in {all_dt:.1f} s with {allc} threads = every CPU this process can use (affinity mask capped by the cgroup quota; '
                      f'the host has {os.cpu_count()} cores)'}


# BASELINE.json's other configurations on ONE GPU (driver-timed, in the N = 1 line's `configs`).  A step is what
# the headline's is -- PathEngine.render(spp) + FilmTable.get_image(), resolve and read-back included -- after one untimed
# frame + clear (exams/benchmark.py:25-27); scene generation, upload and mpt_build_tree are outside it and reported beside it.
OTHER_CONFIGS = [
    # key, title, scene, scene kwargs, film side, spp, world light, timed steps
    ('c1', 'BASELINE configs[0]: s34 34-tri cornell two-boxes, 512x512, 32 spp', 's34', {}, 512, 32, None, 20),
    ('c3_film_1gpu', 'BASELINE configs[2] on ONE GPU: s978 2048x2048, 256 spp (eight pipelined launches of 32 frames)', 's978', {}, 2048, 256, None, 1),
    ('c4', 'BASELINE configs[3]: 99 382-tri displaced blob in cornell + equirect env light, MIS, 1024x1024, 64 spp', 'c4', {}, 1024, 64,
     ([1.0, 1.0, 1.0, 1.0], 0), 1),
    ('c5', 'BASELINE configs[4]: 1M random triangles, on-GPU LBVH + SAH + 4-wide build, 1024x1024, 16 spp', 'c5', {'n': 1000000}, 1024, 16, None, 2),
]


CONFIG_PMC = {}      # key -> (counters per launch of the production render kernel, where they come from): filled by main() before the GPU is touched


def run_config_child(key, mode):
    '''role pmc-config-child (under rocprofv3 --pmc): the render launches of one of OTHER_CONFIGS and nothing else'''
    from ptina_amd import scenes, common
    from ptina_amd.common import ctx
    from ptina_amd.things import FilmTable
    from helpers import setup_engine
    cfg = [c for c in OTHER_CONFIGS if c[0] == key]
    if not cfg:
        raise SystemExit(f'no such configuration: {key}')
    _, _, name, kw, n, spp, world, _ = cfg[0]
    eng = setup_engine(scenes.get_scene(name, **kw), n, n, mode=mode, world=world, max_filmsize=max(n * n, 1 << 21))
    ctx().set_option('batch', SPP)
    film = FilmTable()
    eng.render()
    film.get_image()
    film.clear()
    for _ in range(3):
        eng.render(min(spp, SPP))             # one launch of 32 (c5: 16) frames at a time: the counters are per launch
        film.get_image()
    common.reset_all()


def gather_roofline(key, name, cnt, n, spp, dt, frames):
    '''the roofline object of a gather-kernel leg (c4 / c5): 64-byte records per second against tools/microbench/gather_microbench's
    dependent random 64-byte records, blended harmonically by the L2 hit rate of the leg's own launches'''
    recs_per_sample = (cnt['n_node'] + 0.75 * cnt['n_tri']) / max(cnt['samples'], 1)
    achieved = recs_per_sample * n * n * spp / dt / 1e9       # G records/s over the whole step (its launches overlap; the read-back is in it)
    got, src = CONFIG_PMC.get(key, (None, 'not collected in this run'))
    g = (lambda c: got[c]['median'] if got and c in got else None)
    h, h_src = GATHER_L2_HIT[name], 'profiles/r05_pmc_big_summary.json (fallback: ' + src + ')'
    if g('TCC_HIT_sum') is not None and g('TCC_MISS_sum') is not None and g('TCC_HIT_sum') + g('TCC_MISS_sum') > 0:
        h, h_src = g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')), src
    peak_full = 1.0 / (h / GATHER_FULL_IN_L2_GRECS + (1.0 - h) / GATHER_FULL_BEYOND_L2_GRECS[name])
    peak_own = 1.0 / (h / GATHER_IN_L2_GRECS + (1.0 - h) / GATHER_BEYOND_L2_GRECS[name])
    traffic = int((g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024) if g('FETCH_SIZE') is not None and g('WRITE_SIZE') is not None else None
    roof = {
        'bound': 'gather', 'achieved': round(achieved, 2), 'peak': round(peak_full, 2), 'unit': 'G 64-B records/s', 'frac': round(achieved / peak_full, 4),
        'traffic': traffic, 'traffic_unit': f'HBM bytes (FETCH_SIZE + WRITE_SIZE) per launch of {frames} frames' if traffic is not None else None,
        'peak_at_own_lane_activity': round(peak_own, 2), 'frac_at_own_lane_activity': round(achieved / peak_own, 4),
        'records_per_sample': round(recs_per_sample, 2), 'node_steps_per_ray': round(cnt['n_node'] / max(cnt['rays'], 1), 2),
        'triangle_tests_per_ray': round(cnt['n_tri'] / max(cnt['rays'], 1), 2), 'l2_hit_rate': round(h, 4), 'l2_hit_rate_from': h_src,
        'wave_cycles_wait_frac': round(g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'), 3) if g('SQ_WAIT_ANY') and g('SQ_WAVE_CYCLES') else None,
        'lane_occupancy': round(g('SQ_THREAD_CYCLES_VALU') / (64.0 * g('SQ_ACTIVE_INST_VALU')), 4) if g('SQ_THREAD_CYCLES_VALU') and g('SQ_ACTIVE_INST_VALU') else None,
        'note': 'records = 4-wide node steps + 0.75 x triangle tests of one step (counting build) / the step\'s wall time; peak = '
                'tools/microbench/gather_microbench (random dependent 64-byte records, 5 workgroups per CU) with EVERY lane taking part: '
                f'{GATHER_FULL_IN_L2_GRECS} G records/s inside L2 and {GATHER_FULL_BEYOND_L2_GRECS[name]} beyond, blended harmonically by the L2 hit rate of '
                'these launches; peak_at_own_lane_activity = the same at the 58 percent of the lanes the kernel\'s steps have '
                f'({GATHER_IN_L2_GRECS} / {GATHER_BEYOND_L2_GRECS[name]}). HBM bytes are not the limit: `traffic` / launch time is 7-14 percent of the peak'}
    return roof


def run_other_configs(mode, stub=False):
    '''{key: {msamples_s, ms_per_step, kernel, avg_kernel_ms, ...}} for OTHER_CONFIGS; drops the headline's context first'''
    out = {}
    if stub:
        for key, title, name, kw, n, spp, world, steps in OTHER_CONFIGS:
            out[key] = {'workload': title, 'msamples_s': 1.0, 'ms_per_step': 1.0, 'kernel': 'stub', 'avg_kernel_ms': 1.0,
                        'steps': steps, 'build_tree_ms': 1.0, 'msamples_s_incl_build': 1.0}
        return out
    from ptina_amd import scenes, common
    from ptina_amd.common import ctx
    from ptina_amd.things import FilmTable, BVHTree
    from helpers import setup_engine
    for key, title, name, kw, n, spp, world, steps in OTHER_CONFIGS:
        try:
            common.reset_all()
            t0 = time.perf_counter()
            scene = scenes.get_scene(name, **kw)
            gen_s = time.perf_counter() - t0
            eng = setup_engine(scene, n, n, mode=mode, world=world, max_filmsize=max(n * n, 1 << 21))
            c = ctx()
            film = FilmTable()
            c.set_option('batch', SPP)
            # mpt_build_tree timed alone, as round 5 timed it: the model in host memory (ModelPool.load again), so the PCIe upload is in it
            from ptina_amd.things import ModelPool
            ModelPool().load(scene[0], scene[1])
            c.call('mpt_synchronize')
            t0 = time.perf_counter()
            BVHTree().build()                     # upload + LBVH (+ SAH + 4-wide collapse)
            c.call('mpt_synchronize')
            build_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            BVHTree().build()                     # again with the model resident on the device (a rebuild: another option, another tree kind)
            c.call('mpt_synchronize')
            rebuild_s = time.perf_counter() - t0
            # the build by phase (once more from host memory, with a synchronisation after every phase: its total is not the figure above)
            c.set_option('build_phases', 1)
            ModelPool().load(scene[0], scene[1])
            BVHTree().build()
            phases = {ph: round(c.get_option(f'build_phase_us_{k}') / 1e3, 3)
                      for k, ph in enumerate(('upload_ms', 'lbvh_ms', 'sah_ms', 'triangle_records_ms', 'wide_collapse_ms'))}
            c.set_option('build_phases', 0)
            eng.render()                          # exams/benchmark.py:25-27
            film.get_image()
            film.clear()
            eng.render(min(spp, SPP))             # untimed: the sample slabs and the host array are allocated here
            film.get_image()
            film.clear()
            c.call('mpt_synchronize')
            c.kernel_time()
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.render(spp)
                img = film.get_image()
            dt = (time.perf_counter() - t0) / steps
            kms, nl = c.kernel_time()
            assert img.shape == (n, n, 4) and float(img[..., 3].min()) == 1.0
            kernel = render_kernel_name(mode, c.get_option('last_kernel'))
            # launches of one step are pipelined (up to `pipe_depth` in flight, each on its own stream): avg_kernel_ms is a launch's own
            # duration while it shares the chip, so avg_kernel_ms x launches_per_step / concurrent_launches <= ms_per_step is the check
            # ... `launch_slots` = how many may be in flight (cur_depth: 2 for whole-chip launches), `concurrent_launches` = how many ARE on
            # average: the launches' summed durations over the step's wall time (1 when a step is one launch)
            slots = max(1, min(nl // steps, c.get_option('cur_depth')))
            conc = max(1.0, round(kms / max(steps, 1) / (dt * 1e3), 3)) if nl // steps > 1 else 1
            out[key] = {'workload': title, 'ntri': int(scene[1].shape[0]), 'msamples_s': round(n * n * spp / dt / 1e6, 1),
                        'ms_per_step': round(dt * 1e3, 3), 'kernel': kernel, 'avg_kernel_ms': round(kms / max(nl, 1), 4),
                        'launches_per_step': nl // steps, 'concurrent_launches': conc, 'launch_slots': slots, 'steps': steps,
                        'build_tree_ms': round(build_s * 1e3, 2),
                        'rebuild_tree_ms_model_resident': round(rebuild_s * 1e3, 2),
                        'msamples_s_incl_build': round(n * n * spp / (dt + build_s) / 1e6, 1), 'scene_generation_s': round(gen_s, 2)}
            ntri = int(scene[1].shape[0])
            if kernel in ('render_kernel_lds4', 'render_kernel_lds'):
                # the same VALU roofline object as the headline's, from this configuration's own counters (collected in this run)
                got, src = CONFIG_PMC.get(key, (None, 'not collected in this run'))
                try:
                    roof, hbm = roofline_blocks(got, src, kernel, kms / 1e3 / max(nl, 1), c.get_option('clock_khz') * 1e3 or 2.4e9, conc)
                    out[key]['roofline'] = roof
                    if hbm:
                        out[key]['hbm'] = hbm
                except Exception as e:
                    out[key]['roofline'] = {'bound': 'valu', 'achieved': None, 'counters_from': f'counters unusable ({type(e).__name__}: {e})'}
            if ntri > 8192:
                # mpt_build_tree against what it has to move: the model in once (96-byte vertices + material id), the records the
                # kernels walk out once (reference nodes 32 B, binary nodes 64 B, triangle records 64 + 64 + 48 B, 4-wide nodes 128 + 64 B)
                nw = c.get_option('wide_nodes')
                bbytes = ntri * (100 + 64 + 64 + 48) + (ntri - 1) * 96 + nw * 192
                out[key]['build'] = {
                    'bytes': bbytes, 'ms': round(build_s * 1e3, 2), 'GB/s': round(bbytes / build_s / 1e9, 1),
                    'frac': round(bbytes / build_s / 1e9 / HBM_PEAK_GBS, 4), 'peak': HBM_PEAK_GBS, 'phases': phases,
                    'sah_levels': c.get_option('sah_levels'), 'fast_depth': c.get_option('fast_depth'), 'wide_nodes': nw,
                    'note': 'bytes = the model read once + every record written once; ms = the whole mpt_build_tree call with the model in host '
                            'memory (the PCIe upload is phases.upload_ms); per-kernel table: profiles/r06_build_table_*.json'}
            if kernel == 'render_kernel_wide' and name in GATHER_L2_HIT:
                # what bounds these kernels is the vector memory path's rate of divergent 16-byte gathers, not HBM bytes (their L2 hit
                # rate is 67 / 90 %) and not MFMA: records of one launch (counting build) / its duration, against the micro-benchmark's
                # rate blended by the launches' measured L2 hit rate
                frames = min(spp, SPP)
                film.clear()
                c.set_option('count', 1)
                c.call('mpt_reset_counters')
                eng.render(frames)
                c.call('mpt_synchronize')
                cnt = c.counters()
                c.set_option('count', 0)
                out[key]['roofline'] = gather_roofline(key, name, cnt, n, spp, dt, frames)
        except Exception as e:                    # a configuration that fails must not cost the run its headline
            out[key] = {'workload': title, 'error': f'{type(e).__name__}: {e}'}
    common.reset_all()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--scene', default='s978')
    ap.add_argument('--mode', default='fast')
    ap.add_argument('--chunk', type=int, default=-1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='N = 1: do not time the other BASELINE configurations (`configs`)')
    ap.add_argument('--no-pmc', action='store_true', help='do not run the rocprofv3 --pmc passes (counters from profiles/)')
    ap.add_argument('--force-comm', action='store_true', help='create the RCCL communicator even for one rank')
    ap.add_argument('--c3-steps', type=int, default=2, help='N > 1: timed steps of the 2048x2048 leg, each --c3-spp samples per pixel (0 = skip it)')
    ap.add_argument('--c3-spp', type=int, default=C3_SPP, help='samples per pixel of one c3 step (BASELINE configs[2]: 256)')
    ap.add_argument('--host-gather', action='store_true',
                    help='dress rehearsal on ONE GPU: every rank renders its stripes on device 0 at the same time and the shares travel '
                         'through files instead of RCCL (which refuses two ranks on one device); not a measurement of anything multi-GPU')
    ap.add_argument('--save-film', default='', help='rank 0 saves the raw headline film (numpy, [nx * ny, 4] f32) here after the timed steps')
    ap.add_argument('--stub-hang', default='', help='tests only (--stub): "RANK:PHASE" -- that rank stops responding in that phase')
    ap.add_argument('--stub', action='store_true',
                    help='tests only: a stand-in renderer that touches no GPU (checks the launcher and the line format)')
    ap.add_argument('--role', default='main', choices=['main', 'pmc-child', 'pmc-config-child'],
                    help='pmc-child: the same render steps with nothing else around them (run under rocprofv3); pmc-config-child: the '
                         'render launches of one of the other configurations (--config)')
    ap.add_argument('--config', default='', help='pmc-config-child: which of OTHER_CONFIGS (c4, c5)')
    return ap.parse_args(argv)


class _Stub:
    '''TESTS ONLY (--stub): stands in for the engine / context / film / communicator with objects that touch no GPU
    and no RCCL, so that the rank launcher and the shape of the result line can be checked on a CPU box
    (tests/test_dist_cpu.py).  A "render" sleeps a / N + b of the launch model; nothing is computed.'''

    def __init__(self, rank, world, phases=None, hang=''):
        self.rank, self.world, self.nx, self.lib = rank, world, NX, self
        self.opts = {'nranks': world, 'last_div': 1, 'last_kernel': 1, 'clock_khz': 2400000, 'hw_queues': 0}
        self.phases, self.hang, self._gathers = phases, hang, 0
        for ph in ('unique_id', 'CommInitRank', 'first barrier', 'communicator ready'):
            self._phase(ph)

    def _phase(self, name):
        if self.phases is not None and self.world > 1:
            self.phases.enter(name)
            if self.hang == f'{self.rank}:{name}':
                time.sleep(3600)                  # a rank that never comes back from a collective
            if self.hang == f'{self.rank}:{name}:raise':
                raise RuntimeError('ncclCommInitRank failed: unhandled system error (stand-in)')

    # context
    def mpt_device_count(self): return self.world
    def call(self, name, *a): pass
    def set_option(self, k, v): pass
    def get_option(self, k): return self.opts.get(k, 0)
    def kernel_time(self): return 1.0, 1

    def counters(self):
        return dict(samples=NX * NY * SPP, rays=1, n_box=1, n_tri=1, n_shade=1, n_draws=1, bounces=1, n_node=1,
                    it_node=1, it_leaf=1, it_shade=1, it_new=1)

    # engine
    def render(self, n=1):
        m = MODEL['c3' if self.nx == C3_N else 'headline']
        time.sleep((m['a_ms'] / self.world + m['b_ms']) * 1e-5 * n / SPP)      # 1 % of the modelled time

    # film
    def get_image(self, id=0):
        import numpy as np
        return np.ones((self.nx, self.nx, 4), np.float32)

    def clear(self, id=0): pass
    def set_size(self, nx, ny): self.nx = nx

    # communicator
    def set_stripes(self, nx): pass
    def gather(self, id=0, root=0):
        if self._gathers == 0:
            self._phase('first gather')
            self._phase('first gather done')
        self._gathers += 1

    def barrier(self): pass
    def allreduce_max(self, v): return v
    def close(self): pass


def run_rank(args, rank, world, pmc, pmc_source):
    '''one rank of the benchmark.  Every phase of the multi-rank protocol is named in the PhaseLog: a rank that stays in one
    for MIPTINA_PHASE_TIMEOUT seconds, or an RCCL call that fails, ends the run with a non-zero exit code and one line saying
    which phase every rank had reached'''
    from ptina_amd.dist import PhaseLog
    phases = PhaseLog(rank, world)
    try:
        _run_rank(args, rank, world, pmc, pmc_source, phases)
    except SystemExit:
        raise
    except BaseException as e:
        phases.report(f"failed in phase '{phases.name}': {type(e).__name__}: {e}")
        phases.done = True
        raise
    phases.finish()


def _run_rank(args, rank, world, pmc, pmc_source, phases):
    import numpy as np  # noqa: F401
    phases.enter('setup (scene, tree, film)')
    if args.stub:
        eng = c = film = _Stub(rank, world, phases, args.stub_hang)
        comm = c if (world > 1 or args.force_comm) else None
        scene = None
    else:
        from ptina_amd import scenes, _lib
        from ptina_amd.common import ctx
        from ptina_amd.things import FilmTable
        from ptina_amd.dist import RcclFilm
        from helpers import setup_engine

        scene = scenes.get_scene(args.scene)
        # N > 1 also renders BASELINE configs[2]'s 2048 x 2048 film: capacity at init_things, as in the reference
        eng = setup_engine(scene, NX, NY, mode=args.mode, max_filmsize=(C3_N * C3_N if world > 1 else 2**21))
        c = ctx()
        if c.lib.mpt_device_count() < world and not _lib.devices_isolated() and not args.host_gather:
            raise SystemExit(f'--gpus {world} but only {c.lib.mpt_device_count()} GPU(s) visible')
        if args.host_gather:
            from ptina_amd.dist import HostFilm
            comm = HostFilm(rank, world, phases)
        else:
            comm = RcclFilm(rank, world, phases) if (world > 1 or args.force_comm) else None
        film = FilmTable()
    if args.chunk >= 0:
        c.set_option('chunk', args.chunk)
    for kv in filter(None, os.environ.get('MIPTINA_OPTS', '').split(',')):    # A/B switches (tools/gpu_round.sh), e.g. finalise=0
        c.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    c.set_option('batch', SPP)
    if comm:
        comm.set_stripes(NX)                  # every world-th stripe of 16 columns: even load
    n_gpus = (comm.world if args.host_gather else c.get_option('nranks')) if comm else 1
    if n_gpus != world:
        raise SystemExit(f'communicator has {n_gpus} ranks, expected {world}')

    def step(d2h):
        '''exams/benchmark.py:29-36: 32 x render() then get_image()'''
        eng.render(SPP)                       # render(32) = 32 x (Sobol update + 1 spp), one fused launch
        if comm:
            c.call('mpt_flush')
            comm.gather(0, 0)
        if rank == 0:
            if d2h:
                return film.get_image()       # resolve + D2H into a host array, blocking
            c.call('mpt_resolve', 0)
        elif d2h:
            c.call('mpt_synchronize')
        return None

    def barrier(name=None):
        if name:
            phases.enter('barrier: ' + name)
        c.call('mpt_synchronize')
        if comm:
            comm.barrier()

    # exams/benchmark.py:25-27: warm-up frame, read back, clear
    phases.enter('warm-up frame')
    eng.render()
    film.get_image()
    film.clear()

    if args.role == 'pmc-child':
        for _ in range(max(args.steps, 1)):
            step(True)
        barrier()
        return

    # warm-up steps double as the counting pass (algorithmic counters of the traversal actually run)
    c.set_option('count', 1)
    c.call('mpt_reset_counters')
    W = max(args.warmup, 1)
    phases.enter('warm-up steps (counting kernel)')
    for _ in range(W):
        step(True)
    barrier('after warm-up')
    cnt = c.counters()
    c.set_option('count', 0)
    c.kernel_time()
    step(True)                                # one untimed step of the production (non-counting) kernel
    barrier()
    c.kernel_time()

    # ---- timed region: K steps, each ending with the image in host memory (max over ranks)
    barrier('before the timed steps')
    phases.enter('timed steps')
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = step(True)
    barrier('after the timed steps')
    dt = time.perf_counter() - t0
    kms, nlaunch = c.kernel_time()
    if comm:
        dt = comm.allreduce_max(dt)

    # ---- the same K steps stopping at the resolved image in HBM (consecutive launches overlap)
    barrier()
    phases.enter('resolve-only steps')
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    barrier('after the resolve-only steps')
    dt_resolve = time.perf_counter() - t1
    c.kernel_time()
    if comm:
        dt_resolve = comm.allreduce_max(dt_resolve)

    if args.save_film and rank == 0:
        # (tests: the film after every step so far -- the assembled one of a rehearsal, the device's otherwise)
        import numpy
        if comm and world > 1 and not args.host_gather:
            c.call('mpt_flush')
        numpy.save(args.save_film, comm.film if (args.host_gather and comm and comm.film is not None) else film.get_raw())

    # ---- N > 1: BASELINE configs[2] as stated: 2048 x 2048 film, 256 spp per step, striped and gathered the same way
    c3 = None
    if comm and world > 1 and args.c3_steps > 0:
        phases.enter('c3: set-up')
        c3_spp = max(args.c3_spp, 1)

        def step3(spp):
            eng.render(spp)                   # launches of 32 frames each, pipelined; the film sums in HBM
            c.call('mpt_flush')
            comm.gather(0, 0)
            if rank == 0:
                return film.get_image()
            c.call('mpt_synchronize')
            return None

        film.set_size(C3_N, C3_N)             # (back to one slab: the stripes are dealt again for this width)
        comm.set_stripes(C3_N)
        film.clear()
        step3(SPP)                            # untimed: sample slabs, gather buffers and the host array are allocated here
        barrier('c3 warm-up')
        c.kernel_time()
        phases.enter('c3: timed steps')
        t3 = time.perf_counter()
        for _ in range(args.c3_steps):
            film.clear()
            img3 = step3(c3_spp)
        barrier('after the c3 steps')
        dt3 = comm.allreduce_max(time.perf_counter() - t3)
        kms3, nl3 = c.kernel_time()
        if rank == 0:
            if args.host_gather:              # (a rehearsal: the device film of rank 0 holds its own stripes only; the assembled film is on the host)
                assert comm.film.shape == (C3_N * C3_N, 4) and float(comm.film[:, 3].min()) == float(comm.film[:, 3].max()) == c3_spp
            else:
                assert img3 is not None and img3.shape == (C3_N, C3_N, 4) and float(img3[..., 3].min()) == 1.0
            m = MODEL['c3']
            launches = (c3_spp + SPP - 1) // SPP
            c3 = {'msamples_s': round(C3_N * C3_N * c3_spp * args.c3_steps / dt3 / 1e6, 3),
                  'ms_per_step': round(dt3 / args.c3_steps * 1e3, 4), 'n_gpus': n_gpus, 'steps': args.c3_steps, 'spp': c3_spp,
                  'workload': f'BASELINE configs[2]: {args.scene} {C3_N}x{C3_N}, {c3_spp} spp per step (= {launches} pipelined launches of '
                              f'{SPP} frames), stripes of 16 columns per rank, film cleared, rendered, gathered in one message per rank, '
                              'resolved and the 64 MiB image read back, every step',
                  'render_kernel_ms_rank0': round(kms3 / max(nl3, 1), 4), 'render_launches_rank0': nl3,
                  'model_ms_per_step': round(launches * m['a_ms'] / n_gpus + m['b_ms'], 3),
                  'model': f"launches x a / N + b with a = {m['a_ms']} ms per 32-frame launch of the whole film, b = {m['b_ms']} ms "
                           f"(the drain of the last launch; the others overlap theirs) ({MODEL['from']}); "
                           'the clear, the gather and the 64 MiB read-back are not in the model'}

    if rank == 0:
        assert img is not None and img.shape == (NX, NY, 4) and (args.host_gather or float(img[..., 3].min()) == 1.0)
        total = NX * NY * SPP * args.steps
        avg_kernel_s = kms / 1e3 / max(nlaunch, 1)
        # small launches (N > 1 shares) take 1/G of the CUs each and G of them are resident at once
        concurrent = max(c.get_option('last_div'), 1)
        kernel = render_kernel_name(args.mode, c.get_option('last_kernel'))
        clock_hz = c.get_option('clock_khz') * 1e3 or 2.4e9
        if pmc is None:
            pmc, pmc_source = committed_pmc(kernel)
            if pmc is not None and world > 1:
                # N > 1: counters are not collected in the run (one profiler per rank, and the driver's launcher owns the ranks):
                # the committed per-launch counters of the whole-film launch, times this rank's share of the film -- instruction
                # counts and bytes scale with the samples a launch traces, the lane occupancy ratio is kept.  The kernel TIME is
                # this run's (rank 0, HIP events).  Labelled as such in counters_from
                share = samples_share = 1.0 / n_gpus
                pmc = {cn: dict(x, median=x['median'] * samples_share) for cn, x in pmc.items()}
                pmc_source = (f'{pmc_source}: per-launch counters of the N = 1 launch x {share:.4f} (rank 0\'s share of the film); '
                              'kernel time measured in this run on rank 0; NOT collected in this run')
        try:
            roof, hbm = roofline_blocks(pmc, pmc_source, kernel, avg_kernel_s, clock_hz, concurrent)
        except Exception as e:                # an incomplete counter set must not cost the run its result line
            roof, hbm = roofline_blocks(None, f'counters unusable ({type(e).__name__}: {e})', kernel, avg_kernel_s,
                                        clock_hz, concurrent)
        samples_per_launch = cnt['samples'] / W
        out = {
            'metric': ('STUB, NOT A MEASUREMENT: ' if args.stub else '') + 'Msamples/sec (pixels x spp / s), 512x512x32spp cornell-monkey (978 tri)',
            'value': round(total / dt / 1e6, 3), 'unit': 'Msamples/s', 'n_gpus': n_gpus, 'steps': args.steps,
            'warmup': W, 'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'STUB (no GPU, tests only)' if args.stub else 'synthetic',
            'config': {'workload': f'{args.scene}: 978-tri synthetic cornell + bumpy sphere, {NX}x{NY}, {SPP} spp, '
                                   'unidirectional MIS path tracer, depth<=5; step = 32 x render() + get_image() '
                                   '(resolve + D2H), exams/benchmark.py:29-36', 'film': [NX, NY], 'spp': SPP,
                       'mode': args.mode,
                       'parallelism': (f'REHEARSAL: {n_gpus} processes on ONE GPU, film columns in 16-wide stripes, shares over files (--host-gather)'
                                       if args.host_gather else
                                       f'film columns in 16-wide stripes over {n_gpus} GPUs, RCCL gather' if n_gpus > 1 else 'single GPU')},
            'value_resolve_only': round(total / dt_resolve / 1e6, 3),
            'ms_per_step_resolve_only': round(dt_resolve / args.steps * 1e3, 4),
            'roofline': roof,
            'hbm': hbm,
            'algorithmic': {'bytes_per_sample': round(algorithmic_bytes(cnt) / max(cnt['samples'], 1), 2),
                            'bytes_per_launch': int(algorithmic_bytes(cnt) / W),
                            'GB/s': round(algorithmic_bytes(cnt) / W * concurrent / avg_kernel_s / 1e9, 1),
                            'note': 'SURVEY 8(d) bytes of the traversal actually run (counting build), served from LDS: '
                                    'NOT an HBM figure and not a roofline fraction'},
            'algorithmic_flops': {'flop_per_sample': round(algorithmic_flops(cnt) / max(cnt['samples'], 1), 1),
                                  'TFLOP/s': round(algorithmic_flops(cnt) / W * concurrent / avg_kernel_s / 1e12, 3),
                                  'peak': round(2 * N_SIMD * SIMD_LANES * clock_hz / 1e12, 1),
                                  'frac': round(algorithmic_flops(cnt) / W * concurrent / avg_kernel_s / (2 * N_SIMD * SIMD_LANES * clock_hz), 4),
                                  'note': 'SURVEY 8(d) per-unit flops (30 / box test, 60 / triangle test, 225 / shaded hit) x the '
                                          'counted units of one launch / its duration, against the f32 vector FMA peak: '
                                          'rises with speed whatever the instruction count, unlike roofline.frac'},
            'counters_per_sample': {k: round(v / max(cnt['samples'], 1), 3) for k, v in cnt.items() if k != 'samples'},
            'mrays_per_s': round(cnt['rays'] / W * concurrent / avg_kernel_s / 1e6, 1),
            'samples_per_launch': int(samples_per_launch),
            # hardware queues the HIP runtime multiplexes this process's streams onto: the library asks for 12 before
            # the runtime starts; under rocprofv3 --pmc the GPU is initialised first and the request has no effect
            'hw_queues': c.get_option('hw_queues') or None,
        }
        if n_gpus > 1:
            m = MODEL['headline']
            out['model_ms_per_step'] = round(m['a_ms'] / n_gpus + m['b_ms'], 3)
            out['model'] = f"a / N + b with a = {m['a_ms']} ms, b = {m['b_ms']} ms per launch ({MODEL['from']}), before the gather"
        if c3 is not None:
            out['c3'] = c3
        if world == 1 and comm is None and not args.no_configs and args.scene == 's978':
            out['configs'] = run_other_configs(args.mode, stub=args.stub)      # (drops this rank's context: nothing below needs it)
        if world == 1 and not args.no_cpu_baseline and not args.stub:
            from ptina_amd import scenes
            out['cpu_baseline'] = cpu_baseline(scene, scenes.BENCH_CAMERA)
        print(json.dumps(out), flush=True)
    if comm:
        comm.barrier()
        comm.close()


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # nobody launched ranks for us: do it here, before this process makes any HIP call
        from ptina_amd.dist import launch_ranks
        env = None
        if args.host_gather:                      # every rank on device 0 (ptina_amd._lib.rank_device reads MIPTINA_DEVICE)
            env = dict(os.environ, MIPTINA_DEVICE=os.environ.get('MIPTINA_DEVICE', '0'))
        rc, out = launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                               timeout=float(os.environ.get('MIPTINA_LAUNCH_TIMEOUT', '900')), env=env)
        sys.stdout.write(out)
        sys.stdout.flush()
        sys.exit(rc)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        # dmabuf IPC is what RCCL needs on this driver; a launcher that did not export it (torchrun passes its own
        # environment on) would fail in hipIpcGetMemHandle.  Set before this process makes its first HIP call.
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    pmc, pmc_source = None, 'not collected'
    if args.role == 'main' and world == 1 and args.mode == 'fast' and not args.no_pmc and not args.stub \
            and os.environ.get('MIPTINA_BENCH_PMC', '1') != '0':
        # before anything in this process touches the GPU
        tail = ['--steps', '3', '--warmup', '1', '--scene', args.scene, '--mode', args.mode, '--no-cpu-baseline']
        pmc, pmc_source = collect_pmc(tail)
        if pmc is not None:
            try:
                os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
                with open(os.path.join(ROOT, 'gpurun_out', 'bench_pmc.json'), 'w') as f:
                    json.dump(pmc, f, indent=1, sort_keys=True)
            except OSError:
                pass
        else:
            print('bench.py: ' + pmc_source, file=sys.stderr)
        # the gather kernels of BASELINE configs 4 and 5: HBM bytes, L2 hit rate and wait fraction of their launches, collected in this run
        # too (the TA_* / TCP_* counters hung rocprofv3 on this pool and are not asked for)
        if not args.no_configs and args.scene == 's978':
            # (bounded: a pass normally takes 2-4 s; one that hangs costs its 45 s, and once 150 s are gone the remaining legs fall back
            # to the committed L2 hit rates -- the default run must stay within minutes whatever the profiler does)
            t_pmc = time.time()
            for key in ('c4', 'c5', 'c1', 'c3_film_1gpu'):
                if time.time() - t_pmc > 150:
                    CONFIG_PMC[key] = (None, 'not collected: the counter passes of this run had used up their time')
                    continue
                got, src = collect_pmc(['--config', key, '--mode', args.mode], budget_s=45, role='pmc-config-child',
                                       passes=CONFIG_PMC_PASSES if key in ('c4', 'c5') else PMC_PASSES[:3])
                CONFIG_PMC[key] = (got, src)
                if got is None:
                    print(f'bench.py: {key}: {src}', file=sys.stderr)
            try:
                with open(os.path.join(ROOT, 'gpurun_out', 'bench_pmc_configs.json'), 'w') as f:
                    json.dump({k: {'counters': v[0], 'from': v[1]} for k, v in CONFIG_PMC.items()}, f, indent=1, sort_keys=True)
            except OSError:
                pass
    if args.role == 'pmc-config-child':
        run_config_child(args.config, args.mode)
        return
    run_rank(args, rank, world, pmc, pmc_source)


if __name__ == '__main__':
    main()
