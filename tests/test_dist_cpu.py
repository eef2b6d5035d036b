'''
world_size-2 gloo test of the slab-tiling host logic: each rank renders its column slab (with the
CPU oracle -- there is no GPU here), the slabs are gathered to rank 0, and the result must be
bit-identical to the single-rank render.
'''

import os
import time
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_bounds_partition():
    from ptina_amd.dist import slab_bounds
    for nx in (1, 7, 512, 2048, 100):
        for world in (1, 2, 3, 8):
            b = [slab_bounds(nx, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == nx
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(x1 - x0 for x0, x1 in b) - min(x1 - x0 for x0, x1 in b) <= 1


def test_stripe_columns_partition():
    from ptina_amd.dist import stripe_columns
    for nx in (1, 7, 50, 512, 2048):
        for world in (1, 2, 3, 8):
            cols = [stripe_columns(nx, world, r) for r in range(world)]
            assert sorted(np.concatenate(cols).tolist()) == list(range(nx))
            if nx >= 16 * world * 4:
                assert max(map(len, cols)) - min(map(len, cols)) <= 16


def test_comm_plan_is_the_split_the_renderer_uses():
    '''mpt_comm_plan (pure C function of libmiptina: no context, no GPU) against the Python statement of the split
    (dist.slab_bounds / dist.stripe_columns, which the stripe / slab render tests hold the kernels to), for
    R in {1, 2, 3, 8}, ragged widths and both split kinds; the ranges partition the film; packed order is ascending x'''
    from ptina_amd.dist import comm_plan, slab_bounds, stripe_columns
    for ny in (1, 5, 37):
        for nx in (1, 7, 16, 50, 70, 512, 2048, 100):
            for world in (1, 2, 3, 8):
                for stripe in (0, 16, 32):
                    owned = np.zeros(nx * ny, np.int32)
                    for r in range(world):
                        plan = comm_plan(nx, ny, stripe, r, world)
                        cols = np.arange(*slab_bounds(nx, world, r)) if stripe == 0 else stripe_columns(nx, world, r, stripe)
                        want = np.zeros(nx * ny, bool)
                        want.reshape(nx, ny)[cols] = True
                        got = np.zeros(nx * ny, bool)
                        last = -1
                        for o, n in plan:
                            assert n > 0 and o > last and o % ny == 0 and n % ny == 0
                            got[o:o + n] = True
                            last = o + n - 1
                        assert np.array_equal(got, want), (nx, ny, world, stripe, r)
                        assert len(plan) <= (1 if stripe == 0 else (nx + stripe * world - 1) // (stripe * world))
                        owned += got
                    assert np.all(owned == 1)
    # BASELINE config 3 on 8 GPUs: 16 stripes of 16 columns per rank = 16 pieces, one message
    plan = comm_plan(2048, 2048, 16, 3, 8)
    assert len(plan) == 16 and all(n == 16 * 2048 for _, n in plan) and plan[0][0] == 3 * 16 * 2048
    with pytest.raises(ValueError):
        comm_plan(512, 512, 16, 8, 8)
    with pytest.raises(ValueError):
        comm_plan(512, 512, -1, 0, 8)


def test_packed_gather_reassembles_any_film_in_process():
    '''the whole gather replayed in numpy for R in {2, 3, 8} (pack by the sender's plan, one message, scatter by the
    same plan on the root): whatever the root held in the peers' columns is overwritten, its own columns are kept'''
    from ptina_amd.dist import comm_plan
    from dist_helpers import pack_share, scatter_share
    rng = np.random.default_rng(3)
    for nx, ny, stripe in ((70, 9, 16), (512, 4, 16), (50, 7, 0), (33, 3, 32)):
        truth = rng.normal(size=(nx * ny, 4)).astype(np.float32)
        for world in (2, 3, 8):
            for root in (0, world - 1):
                plans = [comm_plan(nx, ny, stripe, r, world) for r in range(world)]
                films = []
                for r in range(world):            # every rank holds only its share (the rest: garbage)
                    f = np.full((nx * ny, 4), np.float32(-7.0 - r))
                    for o, n in plans[r]:
                        f[o:o + n] = truth[o:o + n]
                    films.append(f)
                out = films[root].copy()
                nmsg = 0
                for r in range(world):
                    if r == root or not plans[r]:
                        continue
                    msg = pack_share(films[r], plans[r])
                    assert msg.shape[0] == sum(n for _, n in plans[r])
                    scatter_share(out, plans[r], msg)
                    nmsg += 1
                assert nmsg <= world - 1
                assert np.array_equal(out, truth), (nx, ny, stripe, world, root)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import oracle
    from ptina_amd import scenes
    from ptina_amd.dist import slab_bounds
    from dist_helpers import gather_film_torch
    from helpers import setup_oracle
    dist.init_process_group('gloo', rank=rank, world_size=world)
    nx, ny, spp = 30, 20, 2
    o = setup_oracle(oracle, scenes.scene_s34(), nx, ny, threads=1)
    o.set_window(*slab_bounds(nx, world, rank))
    o.render(spp)
    film = gather_film_torch(o.get_film_raw(), nx, ny, rank, world)
    # the striped split of the same film (ragged: 70 columns = 4 stripes of 16 + one of 6)
    nx2 = 70
    o2 = setup_oracle(oracle, scenes.scene_s34(), nx2, ny, threads=1)
    o2.set_stripes(16, rank, world)
    o2.render(spp)
    film2 = gather_film_torch(o2.get_film_raw(), nx2, ny, rank, world, stripe=16)
    if rank == 0:
        np.save(out, film)
        np.save(out + '.stripes.npy', film2)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_slab_gather_is_bit_identical(tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    from ptina_amd import scenes
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import setup_oracle
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'film.npy')
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    tiled = np.load(out)
    o = setup_oracle(oracle_mod, scenes.scene_s34(), 30, 20, threads=1)
    o.render(2)
    full = o.get_film_raw()
    assert np.array_equal(tiled, full)
    assert np.all(tiled[:, 3] == 2.0)
    striped = np.load(out + '.stripes.npy')
    o = setup_oracle(oracle_mod, scenes.scene_s34(), 70, 20, threads=1)
    o.render(2)
    assert np.array_equal(striped, o.get_film_raw())
    assert np.all(striped[:, 3] == 2.0)


def test_unique_id_file_rendezvous(tmp_path, monkeypatch):
    from ptina_amd import dist as D
    monkeypatch.setenv('MIPTINA_RDZV_DIR', str(tmp_path))
    uid = bytes(range(128))
    assert D.rendezvous_path() == str(tmp_path / 'rccl_uid')
    assert D.exchange_unique_id(lambda: uid, 0, 2) == uid
    assert D.exchange_unique_id(lambda: b'', 1, 2, timeout=2.0) == uid
    other = tmp_path / 'other_job'
    other.mkdir()
    monkeypatch.setenv('MIPTINA_RDZV_DIR', str(other))
    with pytest.raises(RuntimeError, match='no RCCL unique id'):
        D.exchange_unique_id(lambda: b'', 1, 2, timeout=0.2)


def test_rendezvous_needs_a_common_key(monkeypatch):
    '''ranks that are neither children of launch_ranks() nor torchrun workers have nothing to derive a
    common path from: that must be an error, not a guess'''
    from ptina_amd import dist as D
    for k in ('MIPTINA_RDZV_DIR', 'TORCHELASTIC_RUN_ID', 'TORCHELASTIC_RESTART_COUNT'):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(RuntimeError, match='MIPTINA_RDZV_DIR'):
        D.exchange_unique_id(lambda: bytes(128), 1, 2, timeout=0.1)
    monkeypatch.setenv('TORCHELASTIC_RUN_ID', 'none')        # torchrun: siblings under one agent
    monkeypatch.setenv('MASTER_PORT', '29511')
    # ... in a directory of this user's own (mode 0700, not a symlink, not somebody else's: round-4 ADVICE), keyed by port, agent and run
    import stat
    import tempfile
    path = D.rendezvous_path()
    base = os.path.join(tempfile.gettempdir(), 'miptina_%d' % os.getuid())
    assert path == os.path.join(base, 'job_29511_%d_none_a0' % os.getppid(), 'rccl_uid')
    # an elastic restart of the same agent gets a directory of its own: a unique id the dead attempt left is not found (round-5 ADVICE)
    monkeypatch.setenv('TORCHELASTIC_RESTART_COUNT', '2')
    assert D.rendezvous_path() == os.path.join(base, 'job_29511_%d_none_a2' % os.getppid(), 'rccl_uid')
    import shutil as _sh
    _sh.rmtree(os.path.dirname(D.rendezvous_path()))
    monkeypatch.delenv('TORCHELASTIC_RESTART_COUNT')
    st = os.lstat(base)
    assert stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and not (st.st_mode & 0o077)
    assert D.phase_dir() == os.path.dirname(path)
    # a phase file an earlier job left under the same key is not this rank's phase
    with open(os.path.join(D.phase_dir(), 'phase_1'), 'w') as f:
        f.write('first gather 1.0')
    log = D.PhaseLog(1, 2, timeout=1000)
    assert D.read_phases(D.phase_dir(), 2)[1].startswith('start')
    log.finish()
    import shutil
    shutil.rmtree(os.path.dirname(path))


STUB_RANK = r'''
import json, os, sys, time
sys.path.insert(0, os.environ['MIPTINA_TEST_ROOT'])
from ptina_amd.dist import exchange_unique_id
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
assert int(os.environ['LOCAL_RANK']) == rank and os.environ['MASTER_ADDR'] == '127.0.0.1'
mode = os.environ.get('STUB_MODE', 'ok')
if mode == 'fail' and rank == world - 1:
    sys.exit(3)                                   # e.g. no device for this LOCAL_RANK
uid = exchange_unique_id(lambda: bytes([7]) * 128, rank, world, timeout=20.0)   # the stub "renderer": rendezvous only
assert uid == bytes([7]) * 128
if mode == 'fail':
    time.sleep(60)                                # a rank stuck in a collective its peer never joins
if rank == 0:
    print(json.dumps({'metric': 'stub', 'n_gpus': world, 'rdzv': os.environ['MIPTINA_RDZV_DIR']}))
'''


def test_launcher_starts_ranks_and_relays_rank0(tmp_path):
    '''ptina_amd.dist.launch_ranks -- what `bench.py --gpus N` runs when nobody launched ranks for it:
    N fresh processes with RANK / LOCAL_RANK / WORLD_SIZE and a private rendezvous directory, rank 0's
    line relayed, the directory removed afterwards'''
    import json
    from ptina_amd.dist import launch_ranks
    script = tmp_path / 'stub_rank.py'
    script.write_text(STUB_RANK)
    env = dict(os.environ, MIPTINA_TEST_ROOT=ROOT)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MIPTINA_RDZV_DIR'):
        env.pop(k, None)
    rc, out = launch_ranks(4, [sys.executable, str(script)], timeout=60, env=env)
    assert rc == 0
    line = json.loads(out.strip().splitlines()[-1])
    assert line['n_gpus'] == 4
    assert not os.path.exists(line['rdzv'])


def test_launcher_fails_loudly_when_a_rank_fails(tmp_path):
    import time
    from ptina_amd.dist import launch_ranks
    script = tmp_path / 'stub_rank.py'
    script.write_text(STUB_RANK)
    env = dict(os.environ, MIPTINA_TEST_ROOT=ROOT, STUB_MODE='fail')
    t0 = time.time()
    rc, out = launch_ranks(2, [sys.executable, str(script)], timeout=60, env=env)
    assert rc == 3 and out == ''
    assert time.time() - t0 < 30                  # the stuck rank was stopped, not waited for


def test_bench_self_launch_is_decided_before_any_gpu_call(monkeypatch, tmp_path):
    '''`python bench.py --gpus 2` without WORLD_SIZE goes through launch_ranks with its own argv (checked
    here with the launcher stubbed out: no GPU, no children)'''
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    from ptina_amd import dist as D
    seen = {}

    def fake(world, argv, timeout=None, env=None):
        seen.update(world=world, argv=argv)
        return 0, '{"n_gpus": %d}\n' % world
    monkeypatch.setattr(D, 'launch_ranks', fake)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '5'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert seen['world'] == 2 and seen['argv'][1].endswith('bench.py') and seen['argv'][2:] == ['--gpus', '2', '--steps', '5']
    # under a launcher that set WORLD_SIZE the flag must agree with it
    monkeypatch.setenv('WORLD_SIZE', '4')
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'WORLD_SIZE=4' in str(e.value.code)


def test_bench_n_gpu_line_carries_the_config3_leg(tmp_path):
    '''`bench.py --gpus 2` through its own launcher with the stand-in renderer (--stub: no GPU, no RCCL): rank 0's
    line keeps the headline metric / config and adds `c3` (BASELINE configs[2]'s 2048 x 2048 film) with the
    launch-model prediction beside the measured step'''
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MIPTINA_RDZV_DIR'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--stub', '--c3-steps', '2'], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['scaling'] == 'strong'
    assert '512x512x32spp' in line['metric'] and line['config']['film'] == [512, 512] and line['config']['spp'] == 32
    assert line['metric'].startswith('STUB') and 'STUB' in line['data']          # never mistaken for a measurement
    c3 = line['c3']
    assert c3['n_gpus'] == 2 and c3['steps'] == 2 and '2048x2048' in c3['workload']
    assert c3['spp'] == 256 and '256 spp' in c3['workload'] and 'BASELINE configs[2]' in c3['workload']   # the stated config
    assert c3['msamples_s'] > 0 and c3['ms_per_step'] > 0
    import bench
    m1, m3 = bench.MODEL['headline'], bench.MODEL['c3']
    assert abs(c3['model_ms_per_step'] - (8 * m3['a_ms'] / 2 + m3['b_ms'])) < 1e-3 and 'a / N + b' in c3['model']
    assert abs(line['model_ms_per_step'] - (m1['a_ms'] / 2 + m1['b_ms'])) < 1e-3
    # the N > 1 line is not "unmeasured": a roofline block from rank 0's kernel time and the committed counters x its share
    roof = line['roofline']
    assert roof['frac'] is not None and roof['avg_kernel_ms'] > 0 and 'x 0.5000' in roof['counters_from']
    assert 'NOT collected in this run' in roof['counters_from'] and line['hbm']['traffic_bytes'] > 0
    # one GPU: no c3 leg, no model keys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '1', '--stub'],
                       env=env, capture_output=True, text=True, timeout=120)
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and line['n_gpus'] == 1 and 'c3' not in line and 'model_ms_per_step' not in line
    # ... and the other BASELINE configurations timed in the same driver-run line
    assert set(line['configs']) == {'c1', 'c3_film_1gpu', 'c4', 'c5'}
    for key, v in line['configs'].items():
        assert {'workload', 'msamples_s', 'ms_per_step', 'kernel', 'avg_kernel_ms'} <= set(v), key
    assert 'build_tree_ms' in line['configs']['c5'] and '256 spp' in line['configs']['c3_film_1gpu']['workload']


def test_bench_rank_that_never_returns_is_reported_with_every_ranks_phase():
    '''a multi-rank run never hangs silently.  A stand-in rank (--stub) stops responding in a phase of the
    protocol ("first gather": a collective one rank never joins blocks the others inside RCCL, where no exception reaches them);
    its own watchdog thread ends it after MIPTINA_PHASE_TIMEOUT seconds with ONE line naming the phase every rank had reached,
    the launcher stops the other ranks, and the exit code is not 0'''
    import subprocess
    env = dict(os.environ, MIPTINA_PHASE_TIMEOUT='2')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MIPTINA_RDZV_DIR'):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--stub',
                        '--stub-hang', '1:first gather', '--c3-steps', '0'], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and time.time() - t0 < 60
    assert "stuck in phase 'first gather'" in r.stderr, r.stderr
    assert 'rank 0:' in r.stderr and 'rank 1: first gather' in r.stderr       # which phase EVERY rank reached
    assert 'launch_ranks: rank' in r.stderr and 'phases:' in r.stderr         # the launcher's own line
    # and a rank that fails (an exception: what a failed RCCL call raises) names its phase as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--stub',
                        '--stub-hang', '0:CommInitRank:raise', '--c3-steps', '0'], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "failed in phase 'CommInitRank'" in r.stderr, r.stderr


def _hostfilm_rank(rank, world, d, nx, ny, q):
    '''one rank of the HostFilm protocol test: a fake context whose film holds this rank's stripes only'''
    import ctypes as C
    os.environ['MIPTINA_RDZV_DIR'] = d
    from ptina_amd import dist, common

    full = (np.arange(nx * ny * 4, dtype=np.float32).reshape(nx * ny, 4) * 0.5 + 1.0)

    class FakeCtx:
        def call(self, name, *a):
            if name == 'mpt_get_size':
                C.cast(a[0], C.POINTER(C.c_int))[0] = nx
                C.cast(a[1], C.POINTER(C.c_int))[0] = ny
            elif name == 'mpt_get_film_raw':
                out = np.ctypeslib.as_array(C.cast(a[1], C.POINTER(C.c_float)), shape=(nx * ny, 4))
                out[:] = 0.0
                for o, n in dist.comm_plan(nx, ny, self.w, rank, world):
                    out[o:o + n] = full[o:o + n]
            elif name == 'mpt_set_stripes':
                self.w = a[0]
    fake = FakeCtx()
    common.ctx = lambda: fake
    dist_ctx = dist.__dict__
    import ptina_amd.common
    ptina_amd.common.ctx = lambda: fake
    hf = dist.HostFilm(rank, world)
    hf.set_stripes(nx)
    got = []
    for k in range(3):
        hf.gather(0, 0)
        got.append(hf.allreduce_max(float(rank * 10 + k)))
        hf.barrier()
    q.put((rank, got, None if hf.film is None else bool(np.array_equal(hf.film, full))))


@pytest.mark.parametrize('world', [2, 3])
def test_host_film_rehearsal_transport(tmp_path, world):
    '''dist.HostFilm (the file transport of `bench.py --host-gather`: R processes on one GPU, because RCCL refuses two ranks on one
    device): barriers, max-reductions and three gathers in a row over the job directory, world_size 2 and 3, every rank its own
    process; rank 0 assembles the whole film from the ranks' stripes (the split is mpt_comm_plan's), the others keep none'''
    import multiprocessing as mp
    ctxm = mp.get_context('spawn')
    q = ctxm.Queue()
    nx, ny = 70, 9
    ps = [ctxm.Process(target=_hostfilm_rank, args=(r, world, str(tmp_path), nx, ny, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    for r, got, film_ok in res:
        assert got == [float((world - 1) * 10 + k) for k in range(3)]
        assert film_ok is (True if r == 0 else None)
