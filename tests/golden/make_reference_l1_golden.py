#!/usr/bin/env python3
'''
Runs the reference's OWN per-sample math -- the bodies of its `@ti.func`s, imported from
/root/reference with the pure-Python `taichi` stand-in of tests/golden/taichi_standin on sys.path --
on seeded inputs, in single and in double precision, and writes inputs and outputs to
tests/golden/reference_l1.npz.  tests/test_reference_l1_cpu.py holds the C oracle to these vectors.

Build container only: neither the reference nor the stand-in is needed at test time.

What these vectors are: the reference's source logic (branch structure, operand order, constants,
which sample feeds which lobe) evaluated by numpy.  What they are not: Taichi's arithmetic -- so the
oracle's parity with real PTina output stays formally UNPINNED.

Functions covered (reference file:line):
  materials/microfacet.py:9-78    schlickFresnel, dielectricFresnel, GTR1, GTR2, smithGGX, sample_GTR1/2
  common.py:213-260               tanspace, spherical, dir2tex, reflect, refract
  geometries.py:24-177            Box / Face / Sphere / Area .intersect, Face.normal / texcoord
  materials/disney.py:13-233      Disney.__init__, .brdf, .bounce (every Choice branch, materials/__init__.py:37-48)
  sampling/__init__.py:9-23       wanghash, wanghash2
  engine/path.py:11-15            power_heuristic

usage: python3 tests/golden/make_reference_l1_golden.py   (from the repo root)
'''

import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('PTINA_REFERENCE', '/root/reference')
sys.path.insert(0, os.path.join(HERE, 'taichi_standin'))
sys.path.insert(0, REF)
warnings.filterwarnings('ignore', category=RuntimeWarning)        # u32 wrap-around, sqrt of negatives: wanted

import taichi as ti                                    # noqa: E402  (the stand-in)
import ptina.common as C                               # noqa: E402
import ptina.materials as MAT                          # noqa: E402
import ptina.materials.microfacet as MF                # noqa: E402
import ptina.materials.disney as DIS                   # noqa: E402
import ptina.geometries as GEO                         # noqa: E402
import ptina.sampling as SAMP                          # noqa: E402
from ptina.engine.path import power_heuristic          # noqa: E402

assert 'taichi_standin' in ti.__file__

MATERIALS = {       # the 12 Disney parameters in mtllib.py:44-56 order (basecolor first)
    'default': dict(basecolor=(0.8, 0.8, 0.8), metallic=0.0, roughness=0.4, specular=0.5, specularTint=0.4,
                    subsurface=0.0, sheen=0.0, sheenTint=0.4, clearcoat=0.0, clearcoatGloss=0.5, transmission=0.0, ior=1.45),
    'glass': dict(basecolor=(0.9, 0.95, 1.0), roughness=0.08, transmission=0.9, ior=1.5, specular=0.5),
    'rough_glass': dict(basecolor=(0.8, 0.9, 0.8), roughness=0.45, transmission=0.6, ior=1.33, metallic=0.1),
    'clearcoat': dict(basecolor=(0.7, 0.1, 0.1), roughness=0.5, clearcoat=1.0, clearcoatGloss=0.9),
    'coat_on_metal': dict(basecolor=(0.9, 0.7, 0.3), roughness=0.3, metallic=0.9, clearcoat=0.5, clearcoatGloss=0.2),
    'cloth': dict(basecolor=(0.3, 0.2, 0.7), roughness=0.9, sheen=1.0, sheenTint=0.8, subsurface=0.7, specular=0.1),
    'tinted_spec': dict(basecolor=(0.1, 0.6, 0.2), roughness=0.2, specular=1.0, specularTint=1.0),
    'mirror': dict(basecolor=(0.95, 0.95, 0.95), roughness=0.0, metallic=1.0),
    'black': dict(basecolor=(0.0, 0.0, 0.0), roughness=0.5),
    'gltf_compat': dict(basecolor=(0.8, 0.05, 0.05), metallic=0.0, roughness=0.5, specular=0.0, specularTint=0.0,
                        subsurface=0.0, sheen=0.0, sheenTint=0.0, clearcoat=0.0, clearcoatGloss=0.0, transmission=0.0, ior=0.0),
}
ORDER = ('metallic', 'roughness', 'specular', 'specularTint', 'subsurface', 'sheen', 'sheenTint', 'clearcoat',
         'clearcoatGloss', 'transmission', 'ior')


def params14(name):
    d = dict(MATERIALS['default'])
    d.update(MATERIALS[name])
    return [*d['basecolor'], *[d[k] for k in ORDER]]


def vec(a, T):
    return C.V(*[T(x) for x in a])


def arr(v):
    if isinstance(v, ti.Matrix):
        return [float(e) for e in v.entries]
    return float(v)


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def generate(T, tag, out):
    ti.set_default_fp(T)
    rng = np.random.default_rng(20261004)
    put = lambda k, a, dt=np.float64: out.__setitem__(f'{tag}/{k}', np.asarray(a, dt))   # noqa: E731
    # values are stored as f64 (exact for both precisions); the test casts inputs back to the run's type

    # ---------------------------------------------------------------- microfacet.py
    x = T(1) * rng.uniform(-0.2, 1.2, 64).astype(T)
    put('schlick/in', x)
    put('schlick/out', [MF.schlickFresnel(c) for c in x])
    e = rng.choice([1.0, 1.45, 1.33, 1.5], size=(96, 2)).astype(T)
    c = rng.uniform(0, 1, 96).astype(T)
    put('dielectric/in', np.column_stack([e, c]))
    put('dielectric/out', [MF.dielectricFresnel(a, b, cc) for (a, b), cc in zip(e, c)])
    ch = rng.uniform(0, 1, 64).astype(T)
    al = rng.uniform(0.001, 0.999, 64).astype(T)
    put('gtr/in', np.column_stack([ch, al]))
    put('gtr1/out', [MF.GTR1(a, b) for a, b in zip(ch, al)])
    put('gtr2/out', [MF.GTR2(a, b) for a, b in zip(ch, al)])
    put('smithggx/out', [MF.smithGGX(a, b) for a, b in zip(ch, al)])
    uv = rng.uniform(0, 1, (64, 2)).astype(T)
    al2 = np.concatenate([rng.uniform(0.001, 0.999, 48), rng.uniform(1.01, 1.5, 16)]).astype(T)   # alpha > 1: finite GTR1 samples
    put('sample_gtr/in', np.column_stack([uv, al2]))
    put('sample_gtr1/out', [arr(MF.sample_GTR1(u, v, a)) for (u, v), a in zip(uv, al2)])
    put('sample_gtr2/out', [arr(MF.sample_GTR2(u, v, a)) for (u, v), a in zip(uv, al2)])

    # ---------------------------------------------------------------- common.py
    n = unit(rng, 64).astype(T)
    v = rng.normal(size=(64, 3)).astype(T)
    put('tanspace/in', np.column_stack([n, v]))
    put('tanspace/out', [arr(C.tanspace(vec(a, T)) @ vec(b, T)) for a, b in zip(n, v)])
    hp = rng.uniform(-1, 1, (64, 2)).astype(T)
    hp[:, 1] = rng.uniform(0, 1, 64).astype(T)
    put('spherical/in', hp)
    put('spherical/out', [arr(C.spherical(h, p)) for h, p in hp])
    d = (unit(rng, 64) * rng.uniform(0.1, 3, (64, 1))).astype(T)
    put('dir2tex/in', d)
    put('dir2tex/out', [arr(C.dir2tex(vec(a, T))) for a in d])
    I = unit(rng, 96).astype(T)
    N = unit(rng, 96).astype(T)
    eta = rng.choice([1 / 1.45, 1.45, 1 / 1.5, 1.33, 1.0], 96).astype(T)
    put('reflect/in', np.column_stack([I, N]))
    put('reflect/out', [arr(C.reflect(vec(a, T), vec(b, T))) for a, b in zip(I, N)])
    put('refract/in', np.column_stack([I, N, eta]))
    r = [C.refract(vec(a, T), vec(b, T), e_) for a, b, e_ in zip(I, N, eta)]
    put('refract/out', [[float(h), *arr(t)] for h, t in r])
    assert 0 < sum(h for h, _ in r) < len(r), 'refract: both the refracting and the TIR branch must occur'

    # ---------------------------------------------------------------- geometries.py
    lo = rng.uniform(-2, 1, (128, 3))
    hi = lo + rng.uniform(0.05, 2, (128, 3))
    o = rng.uniform(-3, 3, (128, 3))
    dd = unit(rng, 128)
    dd[:24, 0] = 0.0                    # axis-parallel rays: the |d| < eps branch (geometries.py:33-35)
    dd[8:16, 1] = 1e-7
    o[:12] = (lo[:12] + hi[:12]) / 2    # some of them starting inside the slab
    bx = np.column_stack([lo, hi, o, dd]).astype(T)
    res = []
    for row in bx:
        h = GEO.Box(vec(row[0:3], T), vec(row[3:6], T)).intersect(GEO.Ray(vec(row[6:9], T), vec(row[9:12], T)))
        res.append([float(h.hit), float(h.near), float(h.far)])
    put('box/in', bx)
    put('box/out', res)
    assert 10 < sum(r_[0] for r_ in res) < 118

    tri = rng.uniform(-1, 1, (160, 9))
    tri[:8, 3:6] = tri[:8, 0:3] + 1e-4 * rng.normal(size=(8, 3))          # needle triangles
    tri[8:12, 6:9] = tri[8:12, 0:3] + 2 * (tri[8:12, 3:6] - tri[8:12, 0:3])  # degenerate: D = 0
    ro = rng.uniform(-2, 2, (160, 3))
    cen = tri.reshape(160, 3, 3).mean(axis=1)
    rd = cen - ro + 0.25 * rng.normal(size=(160, 3))                      # aimed near the triangle: hits and misses
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    rd[150:] = -rd[150:]                                                  # behind the origin: r <= 0
    fc = np.column_stack([tri, ro, rd]).astype(T)
    res, nrm, tex = [], [], []
    vn = unit(rng, 160 * 3).reshape(160, 9).astype(T)
    vt = rng.uniform(0, 1, (160, 6)).astype(T)
    for row, n9, t6 in zip(fc, vn, vt):
        f = GEO.Face(vec(row[0:3], T), vec(row[3:6], T), vec(row[6:9], T), vec(n9[0:3], T), vec(n9[3:6], T), vec(n9[6:9], T),
                     vec(t6[0:2], T), vec(t6[2:4], T), vec(t6[4:6], T), 0)
        h = f.intersect(GEO.Ray(vec(row[9:12], T), vec(row[12:15], T)))
        res.append([float(h.hit), float(h.depth), *arr(h.uv)])
        nrm.append(arr(f.normal(h)))
        tex.append(arr(f.texcoord(h)))
    put('face/in', fc)
    put('face/out', res)
    put('face/vn', vn)
    put('face/vt', vt)
    put('face/normal', nrm)
    put('face/texcoord', tex)
    assert 30 < sum(r_[0] for r_ in res) < 140

    sp = np.column_stack([rng.uniform(-1, 1, (96, 3)), rng.uniform(0.05, 1.5, 96), rng.uniform(-3, 3, (96, 3)), unit(rng, 96)])
    sp[:16, 4:7] = sp[:16, 0:3] + 0.3 * np.sqrt(sp[:16, 3:4]) * unit(rng, 16)     # origins inside the sphere
    sp = sp.astype(T)
    put('sphere/in', sp)
    put('sphere/out', [float(GEO.Sphere(vec(r_[0:3], T), r_[3]).intersect(GEO.Ray(vec(r_[4:7], T), vec(r_[7:10], T)))) for r_ in sp])
    ar = np.column_stack([rng.uniform(-1, 1, (96, 3)), rng.normal(size=(96, 3)), rng.normal(size=(96, 3)),
                          rng.uniform(-3, 3, (96, 3)), unit(rng, 96)]).astype(T)
    res = []
    for r_ in ar:
        h = GEO.Area(vec(r_[0:3], T), vec(r_[3:6], T), vec(r_[6:9], T)).intersect(GEO.Ray(vec(r_[9:12], T), vec(r_[12:15], T)))
        res.append([float(h.hit), float(h.depth), *arr(h.uv)])
    put('area/in', ar)
    put('area/out', res)

    # ---------------------------------------------------------------- disney.py
    names = sorted(MATERIALS)
    rows, brdf_out, bounce_in, bounce_out, branch = [], [], [], [], []
    trace = []
    orig_call = MAT.Choice.__call__

    def spy(self, r):                  # which way every Choice went (materials/__init__.py:37-48)
        ret = orig_call(self, r)
        trace.append(int(ret))
        return ret
    MAT.Choice.__call__ = spy
    try:
        for mi, name in enumerate(names):
            p = [T(x) for x in params14(name)]
            for k in range(40):
                nrm_ = unit(rng, 1)[0]
                ind = unit(rng, 1)[0]
                if np.dot(ind, nrm_) < 0:
                    ind = -ind                                   # indir on the normal's side (normal is flipped toward the ray)
                outd = unit(rng, 1)[0]
                if k % 4 != 0 and np.dot(outd, nrm_) < 0:
                    outd = -outd                                 # every fourth pair keeps a below-surface outdir
                sign = T(1.0) if k % 10 else T(-1.0)             # sign < 0 never occurs in path_trace: covered anyway
                m = DIS.Disney(vec(p[0:3], T), *p[3:])
                nv, iv, ov = vec(nrm_, T), vec(ind, T), vec(outd, T)
                rows.append([mi, *[float(x) for x in p], *arr(nv), float(sign), *arr(iv), *arr(ov)])
                brdf_out.append(arr(m.brdf(nv, sign, iv, ov)))
                sm = rng.uniform(0, 1, 3).astype(T)
                if k % 5 == 0:
                    sm[2] = T(rng.uniform(0, 0.12))              # small w: clearcoat / specular lobes
                del trace[:]
                b = m.bounce(nv, sign, iv, vec(sm, T))
                col = b.color if isinstance(b.color, ti.Matrix) else C.V3(b.color)     # a scalar assigned to a vec3 field broadcasts
                bounce_in.append([float(x) for x in sm])
                bounce_out.append([*arr(b.outdir), float(b.pdf), *arr(col)])
                branch.append(int(''.join(map(str, trace)), 2) + (1 << len(trace)))    # decisions as bits under a leading 1
    finally:
        MAT.Choice.__call__ = orig_call
    put('disney/in', rows)
    put('disney/brdf', brdf_out)
    put('disney/samp', bounce_in)
    put('disney/bounce', bounce_out)
    put('disney/branch', branch, np.int64)
    seen = set(branch)
    # 1: coat taken | 01: spec, then 1/0 transmission, then 1/0 reflect | 00: diffuse
    for want, what in ((0b11, 'clearcoat'), (0b100, 'diffuse'), (0b1010, 'specular reflection (no transmission)'),
                       (0b10111, 'transmission: reflect'), (0b10110, 'transmission: refract')):
        assert want in seen, f'{tag}: Choice branch "{what}" never taken; seen {sorted(map(bin, seen))}'

    # ---------------------------------------------------------------- path.py:11-15
    ab = np.concatenate([rng.uniform(0, 4, (48, 2)), [[0, 0], [0, 1], [1e-9, 2e7], [3e6, 1e-7]]]).astype(T)
    put('power/in', ab)
    put('power/out', [float(power_heuristic(a, b)) for a, b in ab])


def main():
    out = {}
    generate(np.float32, 'f32', out)
    generate(np.float64, 'f64', out)
    # ---------------------------------------------------------------- sampling/__init__.py:9-23 (integers: once)
    xs = np.concatenate([np.arange(0, 40), [511, 512, 2047, 65535, 1 << 20, (1 << 31) - 1], -np.arange(1, 12)]).astype(np.int64)
    wrap = lambda v: int(v) - (1 << 32) if int(v) >= 1 << 31 else int(v)      # Taichi's int(u32) is an i32 bit-cast  # noqa: E731
    out['int/wanghash/in'] = xs
    out['int/wanghash/out'] = np.array([wrap(SAMP.wanghash(int(x))) for x in xs], np.int64)
    ij = np.array([(i, j) for i in (0, 1, 2, 17, 255, 511, 2047) for j in (0, 1, 5, 300, 511, 2047)], np.int64)
    out['int/wanghash2/in'] = ij
    out['int/wanghash2/out'] = np.array([wrap(SAMP.wanghash2(int(i), int(j))) for i, j in ij], np.int64)
    out['material_names'] = np.array(sorted(MATERIALS))
    dst = os.path.join(HERE, 'reference_l1.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes,', len(out), 'arrays')


if __name__ == '__main__':
    main()
