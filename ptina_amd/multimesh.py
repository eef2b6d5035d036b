'''
compose several meshes (each with its own world matrix and material id) into the single
[3n, 8] vertex array + [n] material ids that ModelPool.load takes (reference multimesh.py:9-87:
rows are  posx posy posz nrmx nrmy nrmz texu texv, f64)
'''

import numpy as np


def _transform(p, n, t, world, mtl):
    assert world is not None and p is not None and n is not None
    p = np.asarray(p, np.float64).reshape(-1, 3)
    n = np.asarray(n, np.float64).reshape(-1, 3)
    # (the reference's t = None branch builds one row per face and then fails its own reshape,
    #  multimesh.py:47-54; here missing texcoords are zeros per corner)
    t = np.zeros((p.shape[0], 2)) if t is None else np.asarray(t, np.float64).reshape(-1, 2)
    assert p.shape[0] == n.shape[0] == t.shape[0] and p.shape[0] % 3 == 0
    w = np.asarray(world, np.float64)
    ph = np.concatenate([p, np.ones((p.shape[0], 1))], axis=1) @ w.T
    p = ph[:, :3] / ph[:, 3:4]
    n = (np.concatenate([n, np.zeros((n.shape[0], 1))], axis=1) @ w.T)[:, :3]   # direction: w component 0
    n = n / np.linalg.norm(n, axis=1, keepdims=True)
    return np.concatenate([p, n, t], axis=1), np.full(p.shape[0] // 3, -1 if mtl is None else mtl)


def compose_multiple_meshes(primitives):
    '''primitives: iterable of (p [k,3,3], n [k,3,3], t [k,3,2] | None, world 4x4, material id | None)'''
    parts = [_transform(*prim) for prim in primitives]
    assert parts, 'no primitives'
    vertices = np.concatenate([a for a, _ in parts], axis=0)
    mtlids = np.concatenate([m for _, m in parts], axis=0)
    assert len(vertices) == 3 * len(mtlids)
    return vertices, mtlids
