#!/usr/bin/env python3
'''Golden vectors from the reference's own Taichi-free modules (the only parts of archibate/ptina that
import in the build container: ptina.tools.readobj, ptina.multimesh).  They pin the data formats on
the caller side of the hot path: the OBJ dictionary and the [3n, 8] vertex / [n] material-id arrays
that ModelPool.load consumes.

Run in the build container only (needs /root/reference; nothing here is needed at test time):
    cd /tmp && python3 /root/repo/tests/golden/make_reference_golden.py
writes tests/golden/reference_hosttools.npz -- inputs and the reference's outputs, data only.'''
import io
import os
import sys

import numpy as np

REF = os.environ.get('PTINA_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path = [p for p in sys.path if os.path.abspath(p or '.') != os.path.dirname(os.path.dirname(HERE))]

from ptina.tools import readobj as R   # noqa: E402
from ptina import multimesh as M       # noqa: E402
import ptina                           # noqa: E402
assert os.path.abspath(ptina.__file__).startswith(os.path.abspath(REF)), ptina.__file__

OBJ = b"""# golden: v/vt/vn and v//vn corners (the reference needs three fields per corner), a quad, a pentagon, two materials
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 2 0.25
v -1 0.5 2
vt 0 0
vt 1 0
vt 1 1
vt 0.25 0.75
vn 0 0 1
vn 0 1 0
vn 0.6 0 0.8
usemtl red
f 1/1/1 2/2/1 3/3/1 4/4/1
f 1//2 2//2 6//3
usemtl blue
f 1/1/1 2/2/2 3/3/3 4/4/1 5/1/2
f 3/3/3 4/4/3 5/1/3
"""

out = {'obj_text': np.frombuffer(OBJ, np.uint8)}
rng = np.random.default_rng(20261003)

# ---- readobj with the option combinations the reference's callers use
for tag, kw in (('default', {}), ('zxy', {'orient': 'zxy'}), ('flipped', {'orient': '-xZy'}), ('scaled', {'scale': 2.5}),
                ('auto', {'scale': 'auto'}), ('nomtl', {'usemtl': False})):
    obj = R.readobj(io.BytesIO(OBJ), **kw)
    for k in ('v', 'vt', 'vn', 'f'):
        out[f'readobj_{tag}_{k}'] = np.asarray(obj[k])
    if 'usemtl' in obj:
        um = obj['usemtl']
        out[f'readobj_{tag}_usemtl_start'] = np.array([u[0] for u in um], np.int64)
        out[f'readobj_{tag}_usemtl_name'] = np.array([bytes(u[1]) for u in um])
v, tri = R.readobj(io.BytesIO(OBJ), simple=True)
out['readobj_simple_v'] = np.asarray(v)
out['readobj_simple_f'] = np.asarray(tri)
obj = R.readobj(io.BytesIO(OBJ))
out['objverts'] = np.asarray(R.objverts(obj))
out['objnorms'] = np.asarray(R.objnorms(obj))
out['objcoors'] = np.asarray(R.objcoors(obj))
out['objmtlids'] = np.asarray(R.objmtlids(obj))
parts = R.objunpackmtls(obj)
out['objunpackmtls_names'] = np.array([bytes(k) for k in parts.keys()])
for k, part in parts.items():
    out['objunpackmtls_f_' + k.decode()] = np.asarray(part['f'])
obj2 = R.readobj(io.BytesIO(OBJ))
R.objmknorm(obj2)
out['objmknorm_vn'] = np.asarray(obj2['vn'])
out['objmknorm_f'] = np.asarray(obj2['f'])
# usemtl names repeating: ids follow first appearance, 1-based (0 = faces before any usemtl)
OBJ2 = b"""v 0 0 0
v 1 0 0
v 0 1 0
f 1/1/1 2/1/1 3/1/1
usemtl a
f 1/1/1 2/1/1 3/1/1
usemtl b
f 1/1/1 2/1/1 3/1/1
f 1/1/1 2/1/1 3/1/1
usemtl a
f 1/1/1 2/1/1 3/1/1
"""
out['obj2_text'] = np.frombuffer(OBJ2, np.uint8)
out['obj2_mtlids'] = np.asarray(R.objmtlids(R.readobj(io.BytesIO(OBJ2))))

# ---- compose_multiple_meshes: two primitives, one with a non-uniform scale + rotation + translation
def rand_prim(ntri, seed):
    g = np.random.default_rng(seed)
    p = g.uniform(-1, 1, (ntri, 3, 3))
    n = g.normal(size=(ntri, 3, 3))
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    t = g.uniform(0, 1, (ntri, 3, 2))
    return p, n, t

th = 0.7
rot = np.array([[np.cos(th), -np.sin(th), 0, 0], [np.sin(th), np.cos(th), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
world1 = np.eye(4)
world2 = rot @ np.diag([2.0, 0.5, 1.5, 1.0])
world2[:3, 3] = [0.3, -1.2, 4.0]
prims = []
for i, (ntri, world, mtl) in enumerate(((3, world1, 0), (5, world2, 2), (2, world2 @ world2, None))):
    p, n, t = rand_prim(ntri, 100 + i)
    out[f'compose_in{i}_p'], out[f'compose_in{i}_n'] = p, n
    if t is not None:
        out[f'compose_in{i}_t'] = t
    out[f'compose_in{i}_world'] = world
    if mtl is not None:
        out[f'compose_in{i}_mtl'] = np.int64(mtl)
    prims.append((p, n, t, world, mtl))
res = M.compose_multiple_meshes(prims)
if isinstance(res, tuple):
    for k, a in enumerate(res):
        out[f'compose_out{k}'] = np.asarray(a)
else:
    out['compose_out0'] = np.asarray(res)

np.savez_compressed(os.path.join(HERE, 'reference_hosttools.npz'), **out)
print('wrote', os.path.join(HERE, 'reference_hosttools.npz'), 'with', len(out), 'arrays')
for k, a in out.items():
    print(' ', k, getattr(a, 'shape', None), getattr(a, 'dtype', None))
