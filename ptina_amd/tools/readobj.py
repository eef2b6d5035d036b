'''
Wavefront OBJ reader with the reference's result layout (tools/readobj.py:21-104): a dict of
'v' [nv,3], 'vt' [nt,2], 'vn' [nn,3] f32 and 'f' [nf,3,3] i32 whose last axis is
(vertex, texcoord, normal) index, 0-based; quads become (0,1,2),(2,3,0), longer polygons a fan.
ModelPool.load(path | dict) consumes it (reference model.py:63-73).
'''

import numpy as np


def _triangulate(corners):
    k = len(corners)
    if k == 3:
        return [corners]
    if k == 4:
        a, b, c, d = corners
        return [[a, b, c], [c, d, a]]
    if k > 4:
        return [[corners[0], corners[i], corners[i + 1]] for i in range(1, k - 1)]
    raise ValueError(f'face with {k} corners')


def readobj(path, orient='xyz', scale=None, simple=False, usemtl=True, quadok=False):
    if callable(getattr(path, 'read', None)):
        text = path.read()
    else:
        with open(path, 'rb') as fh:
            text = fh.read()
    if isinstance(text, bytes):
        text = text.decode('utf-8', 'replace')

    pools = {'v': [], 'vt': [], 'vn': []}
    faces, groups, mtllib = [], [], None
    for raw in text.splitlines():
        parts = raw.split('#', 1)[0].split()
        if len(parts) < 2:
            continue
        tag, args = parts[0], parts[1:]
        if tag in pools:
            try:
                pools[tag].append([float(a) for a in args])
            except ValueError:
                pass
        elif tag == 'mtllib':
            mtllib = args[0].encode()
        elif tag == 'usemtl':
            groups.append([len(faces), args[0].encode()])
        elif tag == 'f':
            corners = []
            for a in args:
                idx = [int(t) - 1 if t else 0 for t in a.split('/')]
                corners.append((idx + [0, 0, 0])[:3])
            faces.extend([corners] if quadok else _triangulate(corners))

    def arr(rows, width):
        if not rows:
            return np.zeros((1, width), np.float32)
        return np.array([(r + [0.0] * width)[:width] for r in rows], np.float32)

    obj = {'v': arr(pools['v'], 3), 'vt': arr(pools['vt'], 2), 'vn': arr(pools['vn'], 3),
           'f': np.array(faces, np.int32) if faces else np.zeros((1, 3, 3), np.int32)}
    if usemtl:
        obj['usemtl'] = groups
        obj['mtllib'] = mtllib
    if orient is not None:
        objorient(obj, orient)
    if scale is not None:
        if scale == 'auto':
            objautoscale(obj)
        else:
            obj['v'] *= scale
    if simple:
        return obj['v'], obj['f'][:, :, 0]
    return obj


def objorient(obj, orient):
    '''axis permutation / flips, e.g. 'xyz' (identity), 'xzy', '-xyz' (reference readobj.py:177-206)'''
    flip = orient.startswith('-')
    axes = orient.lstrip('-+').lower()
    order = ['xyz'.index(ch) for ch in axes]
    for key in ('v', 'vn'):
        if key in obj:
            obj[key] = np.ascontiguousarray(obj[key][:, order])
    if flip:
        obj['f'] = np.ascontiguousarray(obj['f'][:, ::-1, :])
        if 'vn' in obj:
            obj['vn'] = -obj['vn']


def objautoscale(obj):
    v = obj['v']
    lo, hi = v.min(axis=0), v.max(axis=0)
    obj['v'] = (v - (lo + hi) / 2) / max(float((hi - lo).max()) / 2, 1e-30)


def writeobj(path, obj):
    close = not callable(getattr(path, 'write', None))
    fh = open(path, 'w') if close else path
    for key in ('v', 'vt', 'vn'):
        for row in obj.get(key, []):
            fh.write(key + ' ' + ' '.join(repr(float(x)) for x in row) + '\n')
    for face in obj['f']:
        fh.write('f ' + ' '.join('/'.join(str(int(i) + 1) for i in corner) for corner in face) + '\n')
    if close:
        fh.close()
