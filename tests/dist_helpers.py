'''
TEST INFRASTRUCTURE: the film gather of ptina_amd/csrc/comm.cpp replayed on host arrays over any initialised
torch.distributed group (gloo on the CPU), driven by the SAME split function the product uses (mpt_comm_plan through
ptina_amd.dist.comm_plan): every rank packs the ranges of its share side by side into one message, the root
receives one message per peer and scatters it back by the peer's plan.  The product path never imports this
(no PyTorch there); the world_size-2 CPU tests do, with the oracle as the per-rank renderer.
'''

import numpy as np


def pack_share(film_raw, plan):
    '''film_raw [nx*ny, 4] -> the share's ranges side by side, as copy_pieces packs them before the send'''
    return np.concatenate([film_raw[o:o + n] for o, n in plan]) if plan else np.zeros((0, 4), np.float32)


def scatter_share(film_raw, plan, msg):
    '''the root's half: one received message back into the film by the sender's plan'''
    at = 0
    for o, n in plan:
        film_raw[o:o + n] = msg[at:at + n]
        at += n
    assert at == len(msg)


def gather_film_torch(film_raw, nx, ny, rank, world, root=0, stripe=0):
    '''one message per peer, as mpt_comm_gather_film: returns the assembled film on root, None elsewhere'''
    import torch
    import torch.distributed as dist
    from ptina_amd.dist import comm_plan
    plans = [comm_plan(nx, ny, stripe, r, world) for r in range(world)]
    if rank != root:
        msg = pack_share(film_raw, plans[rank])
        if len(msg):
            dist.send(torch.from_numpy(np.ascontiguousarray(msg)), dst=root)
        return None
    out = film_raw.copy()
    for r in range(world):
        n = sum(c for _, c in plans[r])
        if r == root or n == 0:
            continue
        buf = torch.empty((n, 4), dtype=torch.float32)
        dist.recv(buf, src=r)
        scatter_share(out, plans[r], buf.numpy())
    return out
