'''
multi-GPU film tiling: one process per GPU, each renders its share of the film columns of a
replicated scene -- one contiguous slab, or (default of bench.py) every world-th stripe of 16
columns, which evens out the load -- and one gather of those column ranges to rank 0
(SURVEY.md 8e).  The reference has no multi-device code at all, so this module
has no counterpart there.

Film index is x*ny + y (reference filmtable.py:38): the columns [x0, x1) of a slab are one
contiguous float4 range, and because pixel hashes use global (i, j) and every rank advances the
same Sobol index, the tiled image is bit-identical to the single-GPU image.

Transport: mpt_comm_* in libmiptina.so -- one ncclSend per rank / R - 1 ncclRecv on the root, over
xGMI: a share of several stripes is packed side by side on the device, travels as one message and is
scattered into the root's film by one kernel.  Which float4 ranges of the film a rank owns, and in which
order they are packed, is the pure function mpt_comm_plan (comm_plan() below; no GPU needed), which the
world_size-2 CPU tests drive over gloo with the oracle as the per-rank renderer (tests/dist_helpers.py).
'''

import os
import time

import numpy as np


def slab_bounds(nx, world, rank):
    '''columns [x0, x1) of `rank`; the same split libmiptina uses in mpt_comm_gather_film'''
    return rank * nx // world, (rank + 1) * nx // world


STRIPE = 16     # columns per stripe: the strict build's tile width, two work-item tiles of the fast build


def stripe_columns(nx, world, rank, width=STRIPE):
    '''columns of `rank` when the film is dealt out in stripes (mpt_set_stripes(width, rank, world))'''
    x = np.arange(nx)
    return x[(x // width) % world == rank]


def comm_plan(nx, ny, stripe_w, rank, world):
    '''libmiptina's own split (mpt_comm_plan, a pure function: no context, no GPU): the (offset, count) float4
    ranges of the film -- index x*ny + y -- that `rank` of `world` owns, in the order they are packed into its one
    message of the gather.  stripe_w = 0: one slab; > 0: stripes of that many columns dealt round-robin'''
    import ctypes as C
    from . import _lib
    lib = _lib.load_library()
    n = lib.mpt_comm_plan(int(nx), int(ny), int(stripe_w), int(rank), int(world), None, None, 0)
    if n < 0:
        raise ValueError(f'no such split: nx={nx} ny={ny} stripe_w={stripe_w} rank={rank} world={world}')
    off = (C.c_int64 * max(n, 1))()
    cnt = (C.c_int64 * max(n, 1))()
    lib.mpt_comm_plan(int(nx), int(ny), int(stripe_w), int(rank), int(world), off, cnt, n)
    return [(int(off[i]), int(cnt[i])) for i in range(n)]


def env_rank():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), \
        int(os.environ.get('LOCAL_RANK', '0'))


def _torchrun_job_dir():
    '''under torch.distributed.run the workers are children of ONE agent process: its pid, MASTER_PORT and the run id name the
    job.  The directory lives under a per-user directory of mode 0700 that must be a real directory owned by this user (nobody
    else can pre-create, symlink or read what the ranks leave there); None when not under torchrun'''
    if not ('TORCHELASTIC_RUN_ID' in os.environ or 'TORCHELASTIC_RESTART_COUNT' in os.environ):
        return None
    import stat
    import tempfile
    base = os.path.join(tempfile.gettempdir(), 'miptina_%d' % os.getuid())
    try:
        os.mkdir(base, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(base)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError('%s is not a private directory of this user: remove it or set MIPTINA_RDZV_DIR' % base)
    run = ''.join(ch if ch.isalnum() else '_' for ch in os.environ.get('TORCHELASTIC_RUN_ID', 'none'))[:32]
    # (the attempt number too: an elastic restart of the same agent must not find the unique id a dead attempt left behind --
    # it is 128 bytes like a good one, and RCCL's initialisation would hang on it: round-5 ADVICE)
    attempt = ''.join(ch for ch in os.environ.get('TORCHELASTIC_RESTART_COUNT', '0') if ch.isdigit()) or '0'
    d = os.path.join(base, 'job_%s_%d_%s_a%s' % (os.environ.get('MASTER_PORT', '0'), os.getppid(), run, attempt))
    os.makedirs(d, mode=0o700, exist_ok=True)
    return d


def rendezvous_path():
    '''where rank 0 leaves the RCCL unique id for the other ranks of this node.

    * MIPTINA_RDZV_DIR set (launch_ranks() below makes a private directory per job and hands it to
      every rank through the environment): <dir>/rccl_uid -- nothing to collide with, nothing stale;
    * under torch.distributed.run the workers are children of ONE agent process, so its pid plus
      MASTER_PORT (and the run id) names the job: <tmp>/miptina_<uid>/job_<port>_<agent pid>_<run id>/rccl_uid, the parent
      a directory of mode 0700 owned by this user (_torchrun_job_dir);
    * ranks started any other way have no common key to derive: they must be given MIPTINA_RDZV_DIR.'''
    d = os.environ.get('MIPTINA_RDZV_DIR')
    if d:
        return os.path.join(d, 'rccl_uid')
    d = _torchrun_job_dir()
    if d:
        return os.path.join(d, 'rccl_uid')
    raise RuntimeError('multi-rank run without a rendezvous: start the ranks with bench.py --gpus N, '
                       'ptina_amd.dist.launch_ranks() or torch.distributed.run, or give every rank the same '
                       'MIPTINA_RDZV_DIR')


def exchange_unique_id(make_uid, rank, world, timeout=120.0):
    '''hand rank 0's 128-byte ncclUniqueId to the other ranks of this node through a file
    (the only non-RCCL step of a multi-GPU run)'''
    if world == 1:
        return make_uid()
    path = rendezvous_path()
    if rank == 0:
        uid = make_uid()
        tmp = path + '.tmp'
        with open(tmp, 'wb') as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while True:
        try:
            with open(path, 'rb') as f:
                uid = f.read()
            if len(uid) == 128:
                return uid
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise RuntimeError(f'rank {rank}: no RCCL unique id at {path} after {timeout}s')
        time.sleep(0.01)


def phase_dir():
    '''where the ranks of this job leave the name of the phase they are in (PhaseLog): the launcher's private directory,
    or under torch.distributed.run the job's directory under this user's private one (_torchrun_job_dir)'''
    d = os.environ.get('MIPTINA_RDZV_DIR')
    if d:
        return d
    return _torchrun_job_dir()


def read_phases(d, world):
    '''{rank: "phase (seconds in it)"} from the files the ranks keep in `d`; "?" for a rank that wrote none'''
    out = {}
    now = time.time()
    for r in range(world):
        try:
            with open(os.path.join(d, 'phase_%d' % r)) as f:
                name, t = f.read().rsplit(' ', 1)
            out[r] = '%s (%.0f s)' % (name, now - float(t))
        except (OSError, ValueError):
            out[r] = '?'
    return out


class PhaseLog:
    '''A multi-rank run must never hang silently: every rank names the phase it enters (unique_id, CommInitRank, first
    gather, barrier, ...) in a small file, and a watchdog thread of its own ends the rank -- printing which phase EVERY rank
    had reached -- when it stays in one phase longer than `timeout` seconds (MIPTINA_PHASE_TIMEOUT, default 300).  A collective
    that one rank never joins blocks the others inside RCCL, where no Python exception can reach them: hence a thread and
    os._exit, not an exception.  The launcher (launch_ranks, or torchrun's agent) then stops the remaining ranks.'''

    def __init__(self, rank, world, timeout=None):
        import threading
        self.rank, self.world = rank, world
        self.dir = phase_dir() if world > 1 else None
        self.timeout = float(os.environ.get('MIPTINA_PHASE_TIMEOUT', '300')) if timeout is None else float(timeout)
        self.name, self.t0, self.done = 'start', time.time(), False
        if self.dir:
            os.makedirs(self.dir, exist_ok=True)
            try:                                    # (a file an earlier job with the same key left behind is not this rank's phase)
                os.remove(os.path.join(self.dir, 'phase_%d' % self.rank))
            except OSError:
                pass
            self.enter('start')
            th = threading.Thread(target=self._watch, daemon=True)
            th.start()

    def enter(self, name):
        self.name, self.t0 = name, time.time()
        if self.dir:
            try:
                tmp = os.path.join(self.dir, 'phase_%d.tmp' % self.rank)
                with open(tmp, 'w') as f:
                    f.write('%s %.3f' % (name, self.t0))
                os.replace(tmp, os.path.join(self.dir, 'phase_%d' % self.rank))
            except OSError:
                pass

    def report(self, why):
        phases = read_phases(self.dir, self.world) if self.dir else {self.rank: self.name}
        sys_err('rank %d: %s; phase reached by every rank: %s' % (
            self.rank, why, ', '.join('rank %d: %s' % (r, phases[r]) for r in sorted(phases))))

    def finish(self):
        self.done = True
        self.enter('done')

    def _watch(self):
        while not self.done:
            time.sleep(0.25)
            if not self.done and time.time() - self.t0 > self.timeout:
                self.report("stuck in phase '%s' for more than %.0f s, giving up" % (self.name, self.timeout))
                os._exit(3)


def launch_ranks(world, argv, timeout=None, env=None):
    '''start `world` fresh processes of `argv` (one rank per GPU of this node), wait for them and
    return (exit code, rank 0's stdout).  The caller must not have touched the GPU: the children are
    new programs, and a process that has initialised HIP must never exec another (nor fork one that does).

    Each child gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT like a torchrun worker,
    plus MIPTINA_RDZV_DIR = a directory made for this job (the RCCL unique id travels through it).
    The first rank that fails takes the others down with it: the exit code is that rank's, never 0
    unless every rank returned 0.'''
    import shutil
    import socket
    import subprocess
    import tempfile
    if world < 1:
        raise ValueError('world must be >= 1')
    rdzv = tempfile.mkdtemp(prefix='miptina_rdzv_')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                MIPTINA_RDZV_DIR=rdzv)
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL needs on this driver
    procs = []
    try:
        for r in range(world):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            # rank 0's stdout goes to a file (a pipe nobody drains while we poll could fill up)
            so = open(os.path.join(rdzv, 'rank0.out'), 'wb') if r == 0 else subprocess.DEVNULL
            procs.append(subprocess.Popen(list(argv), env=e, stdout=so, start_new_session=True))
            if r == 0:
                so.close()
        t0 = time.time()
        rc = None
        live = set(range(world))
        while live and rc is None:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    rc = code if code > 0 else 128 - code
                    sys_err(f'launch_ranks: rank {r} exited with {code}; stopping the other ranks; phases: {read_phases(rdzv, world)}')
                    break
            if rc is None and timeout is not None and time.time() - t0 > timeout:
                rc = 124
                sys_err(f'launch_ranks: no result after {timeout}s; stopping every rank; phases: {read_phases(rdzv, world)}')
            if live and rc is None:
                time.sleep(0.02)
        out0 = b''
        if rc is None:
            rc = 0
            with open(os.path.join(rdzv, 'rank0.out'), 'rb') as f:
                out0 = f.read()
        return rc, out0.decode('utf-8', 'replace')
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 15)                     # the rank's own process group, nobody else's
                except OSError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    pass
        shutil.rmtree(rdzv, ignore_errors=True)


def sys_err(msg):
    import sys
    print(msg, file=sys.stderr, flush=True)


class RcclFilm:
    '''slab tiling over RCCL for the current ptina_amd context'''

    def __init__(self, rank=None, world=None, phases=None):
        import ctypes as C
        from . import _lib
        from .common import ctx
        r, w, _ = env_rank()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.ctx = ctx()
        self.phases = phases
        self._gathers = 0

        def make_uid():
            buf = C.create_string_buffer(128)
            _lib.check(self.ctx.lib.mpt_comm_unique_id(buf))
            return buf.raw
        self._phase('unique_id')
        uid = exchange_unique_id(make_uid, self.rank, self.world)
        self._phase('CommInitRank')
        self.ctx.call('mpt_comm_init', uid, self.world, self.rank)
        self._phase('first barrier')
        self.barrier()
        self._phase('communicator ready')
        if self.rank == 0 and self.world > 1:
            try:
                os.remove(rendezvous_path())
            except OSError:
                pass

    def set_slab(self, nx):
        x0, x1 = slab_bounds(nx, self.world, self.rank)
        self.ctx.call('mpt_set_slab', x0, x1)
        return x0, x1

    def set_stripes(self, nx, width=STRIPE):
        self.ctx.call('mpt_set_stripes', int(width), self.rank, self.world)
        return stripe_columns(nx, self.world, self.rank, width)

    def _phase(self, name):
        if self.phases is not None:
            self.phases.enter(name)

    def gather(self, id=0, root=0):
        if self.world > 1:
            if self._gathers == 0:
                self._phase('first gather')
            self.ctx.call('mpt_comm_gather_film', int(id), int(root))
            if self._gathers == 0:
                self.ctx.call('mpt_synchronize')          # (the first one only: so that "first gather" means the messages arrived)
                self._phase('first gather done')
            self._gathers += 1

    def barrier(self):
        self.ctx.call('mpt_comm_barrier')

    def allreduce_max(self, value):
        import ctypes as C
        v = C.c_double(float(value))
        self.ctx.call('mpt_comm_allreduce_max', C.byref(v))
        return v.value

    def close(self):
        self.ctx.call('mpt_comm_destroy')


class HostFilm:
    '''The transport of a dress rehearsal on ONE GPU (bench.py --host-gather; tests/test_parity_gpu.py): RCCL refuses two ranks
    on one device, so R processes -- each with its own HIP context on device 0, its stripes of the film
    (mpt_set_stripes(16, r, R)), rendering at the same time -- hand their shares to rank 0 through files in the job's private
    directory instead of ncclSend / ncclRecv.  Same interface as RcclFilm, the same split (comm_plan = mpt_comm_plan), the same
    phases in the PhaseLog; everything but the RCCL calls themselves is the code a multi-GPU run executes.  Rank 0 keeps the
    film it assembled last in `.film` ([nx * ny, 4] f32, index x * ny + y).'''

    def __init__(self, rank=None, world=None, phases=None):
        from .common import ctx
        r, w, _ = env_rank()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.ctx = ctx()
        self.phases = phases
        self.dir = phase_dir()
        if self.world > 1 and not self.dir:
            raise RuntimeError('HostFilm needs the job directory of launch_ranks (MIPTINA_RDZV_DIR)')
        self.seq = 0
        self.stripe_w = 0
        self.film = None
        self._gathers = 0
        self._phase('first barrier')
        self.barrier()
        self._phase('communicator ready')

    def _phase(self, name):
        if self.phases is not None:
            self.phases.enter(name)

    def _put(self, tag, payload):
        path = os.path.join(self.dir, '%s_%d_%d' % (tag, self.seq, self.rank))
        with open(path + '.tmp', 'wb') as f:
            f.write(payload)
        os.replace(path + '.tmp', path)

    def _get(self, tag, r, remove=False):
        path = os.path.join(self.dir, '%s_%d_%d' % (tag, self.seq, r))
        while True:                                   # (a rank that never comes is the PhaseLog watchdog's business)
            try:
                with open(path, 'rb') as f:
                    data = f.read()
                if remove:
                    os.remove(path)
                return data
            except FileNotFoundError:
                time.sleep(0.0005)

    def _allgather(self, tag, payload):
        if self.world == 1:
            return [payload]
        self._put(tag, payload)
        out = [self._get(tag, r) for r in range(self.world)]
        self.seq += 1
        # everybody has passed exchange seq - 2 by now (it took part in seq - 1): its file can go
        old = os.path.join(self.dir, '%s_%d_%d' % (tag, self.seq - 3, self.rank))
        try:
            os.remove(old)
        except OSError:
            pass
        return out

    def set_stripes(self, nx, width=STRIPE):
        self.stripe_w = int(width)
        self.ctx.call('mpt_set_stripes', int(width), self.rank, self.world)
        return stripe_columns(nx, self.world, self.rank, width)

    def gather(self, id=0, root=0):
        import ctypes as C
        from ._lib import fptr
        if self._gathers == 0:
            self._phase('first gather')
        nx, ny = C.c_int(0), C.c_int(0)
        self.ctx.call('mpt_get_size', C.byref(nx), C.byref(ny))
        nx, ny = nx.value, ny.value
        raw = np.empty((nx * ny, 4), np.float32)
        self.ctx.call('mpt_get_film_raw', int(id), fptr(raw))
        mine = comm_plan(nx, ny, self.stripe_w, self.rank, self.world)
        if self.world > 1 and self.rank != root:
            self._put('share', b''.join(raw[o:o + n].tobytes() for o, n in mine))
        elif self.world > 1:
            for r in range(self.world):
                if r == root:
                    continue
                data = np.frombuffer(self._get('share', r, remove=True), np.float32).reshape(-1, 4)
                at = 0
                for o, n in comm_plan(nx, ny, self.stripe_w, r, self.world):
                    raw[o:o + n] = data[at:at + n]
                    at += n
                assert at == data.shape[0]
        if self.rank == root:
            self.film = raw
        self.seq += 1
        if self._gathers == 0:
            self._phase('first gather done')
        self._gathers += 1

    def barrier(self):
        self._allgather('barrier', b'')

    def allreduce_max(self, value):
        import struct
        return max(struct.unpack('d', b)[0] for b in self._allgather('max', struct.pack('d', float(value))))

    def close(self):
        pass
