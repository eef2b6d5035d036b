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
    '''reference readobj.py:177-208: `orient` names, for output axes x, y, z in turn, the input axis
    each one is taken from; an upper-case letter also negates that output axis (positions and
    normals); a leading '-' reverses the corner order of every face'''
    reverse = orient.startswith('-')
    letters = orient[1:] if reverse else orient
    src = ['xyz'.index(ch.lower()) for ch in letters]
    for key in ('v', 'vn'):
        a = obj[key][:, src].copy()
        for axis, ch in enumerate(letters):
            if ch.isupper():
                a[:, axis] = -a[:, axis]
        obj[key] = np.ascontiguousarray(a)
    if reverse:
        obj['f'] = np.ascontiguousarray(obj['f'][:, ::-1, :])


def objautoscale(obj):
    '''reference readobj.py:172-174: centre on the mean vertex, then scale the largest |coordinate| to 1'''
    v = obj['v'] - np.average(obj['v'], axis=0)
    obj['v'] = (v / np.max(np.abs(v))).astype(obj['v'].dtype)


def objverts(obj):
    '''[nf, 3, 3] corner positions (reference readobj.py:160-161)'''
    return obj['v'][obj['f'][:, :, 0]]


def objcoors(obj):
    '''[nf, 3, 2] corner texture coordinates (readobj.py:168-169)'''
    return obj['vt'][obj['f'][:, :, 1]]


def objnorms(obj):
    '''[nf, 3, 3] corner normals (readobj.py:164-165)'''
    return obj['vn'][obj['f'][:, :, 2]]


def _usemtl_ranges(obj):
    starts = [g[0] for g in obj['usemtl']]
    stops = starts[1:] + [len(obj['f'])]
    return [(b, e, g[1]) for b, e, g in zip(starts, stops, obj['usemtl'])]


def objmtlids(obj):
    '''material id per face from the usemtl groups: ids are 1-based in order of first appearance of
    a name, faces before the first usemtl keep 0 (reference readobj.py:141-157)'''
    ids = np.zeros(len(obj['f']), np.int32)
    seen = []
    for beg, end, name in _usemtl_ranges(obj):
        if name not in seen:
            seen.append(name)
        ids[beg:end] = seen.index(name) + 1
    return ids


def objunpackmtls(obj):
    '''one sub-object per material name, sharing the vertex pools (reference readobj.py:117-138)'''
    faces = {}
    for beg, end, name in _usemtl_ranges(obj):
        chunk = obj['f'][beg:end]
        faces[name] = np.concatenate([faces[name], chunk], axis=0) if name in faces else chunk
    return {name: {'f': f, 'v': obj['v'], 'vn': obj['vn'], 'vt': obj['vt']} for name, f in faces.items()}


def objmknorm(obj):
    '''replace the normals by one flat normal per face, cross(p2 - p0, p1 - p0) normalised
    (reference readobj.py:211-222)'''
    ip, it = obj['f'][:, :, 0], obj['f'][:, :, 1]
    p = obj['v'][ip]
    nrm = np.cross(p[:, 2] - p[:, 0], p[:, 1] - p[:, 0])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    own = np.repeat(np.arange(len(ip))[:, None], 3, axis=1)
    obj['vn'] = nrm
    obj['f'] = np.stack([ip, it, own], axis=2)


def readply(path):
    '''(vertices f32 [nv,3], faces i32 [nf,k]) through the optional `plyfile` package (reference readobj.py:225-234)'''
    try:
        from plyfile import PlyData
    except ImportError as e:
        raise ImportError('readply needs the `plyfile` package') from e
    ply = PlyData.read(path)
    verts = np.array([list(v)[:3] for v in ply.elements[0]], np.float32)
    faces = np.array([list(f[0]) for f in ply.elements[1]], np.int32)
    return verts, faces


def writeobj(path, obj):
    close = not callable(getattr(path, 'write', None))
    fh = open(path, 'w') if close else path
    for key in ('v', 'vt', 'vn'):
        for row in obj.get(key, []):
            fh.write(key + ' ' + ' '.join(repr(float(x)) for x in row) + '\n')
    faces = obj['f']
    for face in faces:
        if faces.ndim >= 3:
            fh.write('f ' + ' '.join('/'.join(str(int(i) + 1) for i in corner) for corner in face) + '\n')
        else:                     # 'simple' faces: one index per corner, used for all three fields
            fh.write('f ' + ' '.join('/'.join([str(int(i) + 1)] * 3) for i in face) + '\n')
    if close:
        fh.close()
