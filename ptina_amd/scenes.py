'''
procedural benchmark scenes

The reference's benchmark assets (assets/cornell.gltf, assets/monkey_cornell.gltf,
exams/benchmark.py:12) are git-ignored and absent, so the scenes the
metric is quoted on are re-synthesised here, deterministically, with the same
triangle counts the reference README quotes (34 and 978, README.md:40,46).

Every scene is returned in the exact form the reference loaders hand to the pools
(tools/readgltf.py:240): (vertices [3n,8] f32, mtlids [n] i32, materials, images),
vertices being pos3 nrm3 uv2 per corner (multimesh.py:25-27).
'''

import numpy as np

# the literal world->clip matrix of exams/benchmark.py:18-23
BENCH_CAMERA = np.array([
    [1.73205081e+00, 0.00000000e+00, 0.00000000e+00, 1.01348227e-02],
    [0.00000000e+00, 1.73205081e+00, -1.73205081e-05, -3.36860025e+00],
    [0.00000000e+00, -1.00020002e-05, -1.00020002e+00, 5.27350023e+00],
    [0.00000000e+00, -1.00000000e-05, -1.00000000e+00, 5.37243564e+00],
])

# MaterialPool parameter order, mtllib.py:61-74, and its defaults, mtllib.py:82-93
PARAM_NAMES = ('basecolor', 'metallic', 'roughness', 'specular', 'specularTint',
               'subsurface', 'sheen', 'sheenTint', 'clearcoat', 'clearcoatGloss',
               'transmission', 'ior')
PARAM_DEFAULTS = dict(basecolor=(0.8, 0.8, 0.8), metallic=0.0, roughness=0.4,
                      specular=0.5, specularTint=0.4, subsurface=0.0, sheen=0.0,
                      sheenTint=0.4, clearcoat=0.0, clearcoatGloss=0.5,
                      transmission=0.0, ior=1.45)


def material(**kw):
    '''a full 12-parameter material as a list of (fac, tex) pairs, tex = -1'''
    p = dict(PARAM_DEFAULTS)
    p.update(kw)
    return [(list(p[k]) if k == 'basecolor' else float(p[k]), -1) for k in PARAM_NAMES]


def gltf_compat_material(basecolor, metallic, roughness):
    '''what MaterialPool.load leaves behind for a readgltf 3-tuple:
    the first three parameters set, the other nine at field-zero'''
    return [(list(basecolor), -1), (float(metallic), -1), (float(roughness), -1)] + \
           [(0.0, -1)] * 9


def _pack(tris_p, tris_n, tris_t=None):
    tris_p = np.asarray(tris_p, np.float64).reshape(-1, 3, 3)
    tris_n = np.asarray(tris_n, np.float64).reshape(-1, 3, 3)
    if tris_t is None:
        tris_t = np.zeros((tris_p.shape[0], 3, 2))
    v = np.concatenate([tris_p, tris_n, tris_t], axis=2)
    return v.reshape(-1, 8).astype(np.float32)


def quad(p0, p1, p2, p3, normal):
    '''two triangles (p0,p1,p2), (p0,p2,p3) with a flat normal and unit-square uvs'''
    p = np.array([[p0, p1, p2], [p0, p2, p3]], np.float64)
    n = np.broadcast_to(np.asarray(normal, np.float64), (2, 3, 3))
    t = np.array([[[0, 0], [1, 0], [1, 1]], [[0, 0], [1, 1], [0, 1]]], np.float64)
    return p, n, t


def cornell_walls():
    '''[-2,2] x [0,4] x [-2,2], open at +z: 5 quads = 10 triangles.
    mtlids: 0 white (floor, ceiling, back), 1 red (left), 2 green (right)'''
    a, b = -2.0, 2.0
    y0, y1 = 0.0, 4.0
    parts = [
        (quad((a, y0, a), (b, y0, a), (b, y0, b), (a, y0, b), (0, 1, 0)), 0),   # floor
        (quad((a, y1, a), (a, y1, b), (b, y1, b), (b, y1, a), (0, -1, 0)), 0),  # ceiling
        (quad((a, y0, a), (a, y1, a), (b, y1, a), (b, y0, a), (0, 0, 1)), 0),   # back
        (quad((a, y0, a), (a, y0, b), (a, y1, b), (a, y1, a), (1, 0, 0)), 1),   # left, red
        (quad((b, y0, a), (b, y1, a), (b, y1, b), (b, y0, b), (-1, 0, 0)), 2),  # right, green
    ]
    P = np.concatenate([q[0][0] for q in parts])
    N = np.concatenate([q[0][1] for q in parts])
    T = np.concatenate([q[0][2] for q in parts])
    M = np.concatenate([[q[1]] * 2 for q in parts]).astype(np.int32)
    return P, N, T, M


def box(center, half, yaw_deg, mtl):
    '''an oriented box: 6 quads = 12 triangles with outward flat normals'''
    c = np.asarray(center, np.float64)
    h = np.asarray(half, np.float64)
    th = np.radians(yaw_deg)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    faces = []
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            n = np.zeros(3)
            n[axis] = sgn
            u = np.zeros(3)
            v = np.zeros(3)
            u[(axis + 1) % 3] = 1.0
            v[(axis + 2) % 3] = 1.0
            if sgn < 0:
                u, v = v, u
            corners = [n - u - v, n + u - v, n + u + v, n - u + v]
            corners = [c + R @ (k * h) for k in corners]
            faces.append(quad(*corners, R @ n))
    P = np.concatenate([f[0] for f in faces])
    N = np.concatenate([f[1] for f in faces])
    T = np.concatenate([f[2] for f in faces])
    M = np.full(12, mtl, np.int32)
    return P, N, T, M


def bumpy_sphere(center=(0.0, 1.5, 0.0), radius=1.0, segments=22, rings=23,
                 bump=0.2, mtl=3):
    '''closed UV sphere with a radial displacement r (1 + bump sin 5 theta sin 4 phi)
    and smooth (area-weighted) vertex normals: 2 * segments * (rings - 1) triangles
    (22 x 23 -> 968, so that the 10 wall triangles make 978)'''
    c = np.asarray(center, np.float64)

    def pt(ring, seg):
        phi = np.pi * ring / rings                 # 0 .. pi, pole to pole
        theta = 2 * np.pi * (seg % segments) / segments
        r = radius * (1 + bump * np.sin(5 * theta) * np.sin(4 * phi))
        return c + r * np.array([np.sin(phi) * np.cos(theta), np.cos(phi),
                                 np.sin(phi) * np.sin(theta)])

    def key(ring, seg):
        if ring == 0:
            return (0, 0)
        if ring == rings:
            return (rings, 0)
        return (ring, seg % segments)

    tris = []
    for s in range(segments):
        tris.append([(0, 0), (1, s + 1), (1, s)])                       # top fan
        for r in range(1, rings - 1):
            tris.append([(r, s), (r, s + 1), (r + 1, s + 1)])
            tris.append([(r, s), (r + 1, s + 1), (r + 1, s)])
        tris.append([(rings, 0), (rings - 1, s), (rings - 1, s + 1)])   # bottom fan
    P = np.array([[pt(*v) for v in t] for t in tris])

    acc = {}
    fn = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    for t, n in zip(tris, fn):
        for v in t:
            acc[key(*v)] = acc.get(key(*v), 0) + n
    N = np.array([[acc[key(*v)] / np.linalg.norm(acc[key(*v)]) for v in t] for t in tris])
    # normals must point outward (away from the centre)
    outward = np.einsum('ijk,ijk->ij', N, P - c)
    if np.mean(outward) < 0:
        N = -N
    T = np.array([[[(v[1] % (segments + 1)) / segments, v[0] / rings] for v in t] for t in tris])
    M = np.full(len(tris), mtl, np.int32)
    return P, N, T, M


def _compose(parts):
    P = np.concatenate([p[0] for p in parts])
    N = np.concatenate([p[1] for p in parts])
    T = np.concatenate([p[2] for p in parts])
    M = np.concatenate([p[3] for p in parts]).astype(np.int32)
    return _pack(P, N, T), M


WALL_MATERIALS = [
    material(basecolor=(0.8, 0.8, 0.8), roughness=0.5),     # 0 white
    material(basecolor=(0.8, 0.05, 0.05), roughness=0.5),   # 1 red
    material(basecolor=(0.05, 0.8, 0.05), roughness=0.5),   # 2 green
]


def scene_s34():
    '''config C1: 34-triangle cornell with two boxes (README.md:40)'''
    parts = [cornell_walls(),
             box((-0.7, 1.2, -0.6), (0.6, 1.2, 0.6), 18.0, 3),
             box((0.75, 0.6, 0.55), (0.6, 0.6, 0.6), -17.0, 4)]
    vertices, mtlids = _compose(parts)
    assert mtlids.shape[0] == 34
    materials = WALL_MATERIALS + [
        material(basecolor=(0.75, 0.75, 0.75), roughness=0.5),
        material(basecolor=(0.7, 0.6, 0.3), roughness=0.35, metallic=0.2),
    ]
    return vertices, mtlids, materials, []


def scene_s978():
    '''configs C2/C3: 10 wall triangles + a 968-triangle smooth closed mesh standing in
    for the reference's Suzanne (README.md:46)'''
    parts = [cornell_walls(), bumpy_sphere()]
    vertices, mtlids = _compose(parts)
    assert mtlids.shape[0] == 978
    materials = WALL_MATERIALS + [
        material(basecolor=(0.8, 0.6, 0.2), roughness=0.3, metallic=0.1, specular=0.5),
    ]
    return vertices, mtlids, materials, []


def heightfield_blob(n_side, mtl=3, center=(0.0, 1.6, 0.0), radius=1.1):
    '''a displaced cube-sphere with 12 * n_side^2 triangles (n_side = 91 -> 99372)'''
    c = np.asarray(center, np.float64)
    faces = []
    lin = np.linspace(-1, 1, n_side + 1)
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            u, v = np.meshgrid(lin, lin, indexing='ij')
            w = np.full_like(u, sgn)
            cube = [None] * 3
            cube[axis] = w
            cube[(axis + 1) % 3] = u if sgn > 0 else v
            cube[(axis + 2) % 3] = v if sgn > 0 else u
            q = np.stack(cube, axis=-1)
            d = q / np.linalg.norm(q, axis=-1, keepdims=True)
            r = radius * (1 + 0.08 * np.sin(9 * d[..., 0]) * np.sin(7 * d[..., 1] + 1.0)
                          * np.cos(8 * d[..., 2]) + 0.04 * np.sin(23 * d[..., 1]))
            faces.append((c + d * r[..., None], d))
    tris_p, tris_n = [], []
    for p, d in faces:
        a = p[:-1, :-1], p[1:, :-1], p[1:, 1:], p[:-1, 1:]
        tris_p.append(np.stack([a[0], a[1], a[2]], axis=-2).reshape(-1, 3, 3))
        tris_p.append(np.stack([a[0], a[2], a[3]], axis=-2).reshape(-1, 3, 3))
    P = np.concatenate(tris_p)
    # smooth normals from the analytic neighbourhood: normalised gradient via face normals
    fn = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    fn /= np.linalg.norm(fn, axis=1, keepdims=True) + 1e-30
    flip = np.einsum('ij,ij->i', fn, P.mean(axis=1) - c) < 0
    P[flip] = P[flip][:, ::-1]
    fn[flip] = -fn[flip]
    # vertex-normal smoothing by hashing quantised positions
    keys = np.round(P.reshape(-1, 3) * 1e5).astype(np.int64)
    _, inv = np.unique(keys, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    acc = np.zeros((inv.max() + 1, 3))
    np.add.at(acc, inv, np.repeat(fn, 3, axis=0))
    acc /= np.linalg.norm(acc, axis=1, keepdims=True) + 1e-30
    N = acc[inv].reshape(-1, 3, 3)
    T = np.zeros((P.shape[0], 3, 2))
    d = (P - c) / np.linalg.norm(P - c, axis=-1, keepdims=True)
    T[..., 0] = np.arctan2(d[..., 2], d[..., 0]) / (2 * np.pi) + 0.5
    T[..., 1] = np.arccos(np.clip(d[..., 1], -1, 1)) / np.pi
    M = np.full(P.shape[0], mtl, np.int32)
    return P, N, T, M


def env_image(nx=512, ny=256):
    '''procedural equirect environment (sun lobe + sky gradient), [nx,ny,4] f32'''
    s = (np.arange(nx) + 0.5) / nx
    t = (np.arange(ny) + 0.5) / ny
    S, Tt = np.meshgrid(s, t, indexing='ij')
    az = (S - 0.5) * 2 * np.pi
    el = (Tt - 0.5) * np.pi
    d = np.stack([np.cos(el) * np.cos(az), np.sin(el), np.cos(el) * np.sin(az)], axis=-1)
    sun = np.array([0.3, 0.8, 0.52])
    sun /= np.linalg.norm(sun)
    mu = np.clip(d @ sun, 0, 1)
    sky = 0.25 + 0.55 * np.clip(d[..., 1], 0, 1)[..., None] * np.array([0.6, 0.8, 1.0])
    img = sky + (12.0 * mu ** 64)[..., None] * np.array([1.0, 0.9, 0.7])
    out = np.ones((nx, ny, 4), np.float32)
    out[..., :3] = img
    return out


def scene_c4(n_side=91):
    '''config C4: cornell walls + ~100k-triangle displaced blob, env light as image 0'''
    parts = [cornell_walls(), heightfield_blob(n_side)]
    vertices, mtlids = _compose(parts)
    materials = WALL_MATERIALS + [
        material(basecolor=(0.7, 0.7, 0.75), roughness=0.25, metallic=0.6),
    ]
    return vertices, mtlids, materials, [env_image()]


def scene_random_tris(n=1_000_000, seed=12345, edge=0.02):
    '''config C5: n random small triangles, centroids ~ U([-2,2] x [0,4] x [-2,2])'''
    rng = np.random.default_rng(seed)
    cen = rng.uniform([-2, 0, -2], [2, 4, 2], size=(n, 3))
    off = rng.normal(size=(n, 3, 3)) * edge
    P = cen[:, None, :] + off
    fn = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    fn /= np.linalg.norm(fn, axis=1, keepdims=True) + 1e-30
    N = np.repeat(fn[:, None, :], 3, axis=1)
    M = rng.integers(0, 3, size=n).astype(np.int32)
    vertices = _pack(P, N)
    return vertices, M, list(WALL_MATERIALS), []


SCENES = {'s34': scene_s34, 's978': scene_s978, 'c4': scene_c4, 'c5': scene_random_tris}


def get_scene(name, **kw):
    return SCENES[name](**kw)
