'''
multi-GPU film tiling: one process per GPU, each renders its share of the film columns of a
replicated scene -- one contiguous slab, or (default of bench.py) every world-th stripe of 16
columns, which evens out the load -- and one gather of those column ranges to rank 0
(SURVEY.md 8e).  The reference has no multi-device code at all, so this module
has no counterpart there.

Film index is x*ny + y (reference filmtable.py:38): the columns [x0, x1) of a slab are one
contiguous float4 range, and because pixel hashes use global (i, j) and every rank advances the
same Sobol index, the tiled image is bit-identical to the single-GPU image.

Two transports:
  * 'rccl'  : mpt_comm_* in libmiptina.so -- grouped ncclSend/ncclRecv straight between film
              buffers over xGMI (the product path);
  * 'torch' : any initialised torch.distributed group on host arrays (gloo on CPU; used by the
              world_size-2 CPU tests, where the per-rank renderer is the oracle).
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


def env_rank():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), \
        int(os.environ.get('LOCAL_RANK', '0'))


def exchange_unique_id(make_uid, rank, world, timeout=120.0):
    '''hand rank 0's 128-byte ncclUniqueId to the other ranks of this node through a file
    (the only non-RCCL step; torchrun's workers share a parent pid and MASTER_PORT)'''
    if world == 1:
        return make_uid()
    tag = '%s_%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.getppid(),
                        os.environ.get('TORCHELASTIC_RUN_ID', 'none'))
    path = os.path.join(os.environ.get('MIPTINA_RDZV_DIR', '/tmp'), f'miptina_uid_{tag}')
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


class RcclFilm:
    '''slab tiling over RCCL for the current ptina_amd context'''

    def __init__(self, rank=None, world=None):
        import ctypes as C
        from . import _lib
        from .common import ctx
        r, w, _ = env_rank()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.ctx = ctx()

        def make_uid():
            buf = C.create_string_buffer(128)
            _lib.check(self.ctx.lib.mpt_comm_unique_id(buf))
            return buf.raw
        uid = exchange_unique_id(make_uid, self.rank, self.world)
        self.ctx.call('mpt_comm_init', uid, self.world, self.rank)
        self.barrier()
        if self.rank == 0 and self.world > 1:
            tag = '%s_%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.getppid(),
                                os.environ.get('TORCHELASTIC_RUN_ID', 'none'))
            try:
                os.remove(os.path.join(os.environ.get('MIPTINA_RDZV_DIR', '/tmp'), f'miptina_uid_{tag}'))
            except OSError:
                pass

    def set_slab(self, nx):
        x0, x1 = slab_bounds(nx, self.world, self.rank)
        self.ctx.call('mpt_set_slab', x0, x1)
        return x0, x1

    def set_stripes(self, nx, width=STRIPE):
        self.ctx.call('mpt_set_stripes', int(width), self.rank, self.world)
        return stripe_columns(nx, self.world, self.rank, width)

    def gather(self, id=0, root=0):
        if self.world > 1:
            self.ctx.call('mpt_comm_gather_film', int(id), int(root))

    def barrier(self):
        self.ctx.call('mpt_comm_barrier')

    def allreduce_max(self, value):
        import ctypes as C
        v = C.c_double(float(value))
        self.ctx.call('mpt_comm_allreduce_max', C.byref(v))
        return v.value

    def close(self):
        self.ctx.call('mpt_comm_destroy')


def gather_film_torch(film_raw, nx, ny, rank, world, root=0, stripe=None):
    '''host-array gather through torch.distributed: film_raw is this rank's [nx*ny, 4] raw
    film of which only its share (slab, or stripes of `stripe` columns) is meaningful; returns the
    assembled film on root'''
    import torch
    import torch.distributed as dist
    if stripe:
        cols = [stripe_columns(nx, world, r, stripe) for r in range(world)]
    else:
        cols = [np.arange(*slab_bounds(nx, world, r)) for r in range(world)]
    most = max(len(c) for c in cols)               # gloo's gather wants equal shapes: pad
    mine = torch.zeros((most, ny, 4), dtype=torch.float32)
    mine[:len(cols[rank])] = torch.from_numpy(np.ascontiguousarray(film_raw.reshape(nx, ny, 4)[cols[rank]]))
    if rank == root:
        bufs = [torch.empty((most, ny, 4), dtype=torch.float32) for _ in cols]
        dist.gather(mine, bufs, dst=root)
        out = np.zeros((nx, ny, 4), np.float32)
        for c, b in zip(cols, bufs):
            out[c] = b.numpy()[:len(c)]
        return out.reshape(nx * ny, 4)
    dist.gather(mine, None, dst=root)
    return None
