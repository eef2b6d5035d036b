#!/usr/bin/env python3
'''
End-to-end cross-check vectors: the reference's OWN renderer source -- PathEngine._render / do_render /
path_trace (engine/path.py:18-93), LinearBVH build + intersect (tree/lbvh.py), GlobalStack (stack.py),
ModelPool / MaterialPool / ImagePool / LightPool / WorldLight / Camera / FilmTable / SobolSampler -- imported from
/root/reference and executed as plain Python on numpy scalars, with the `taichi` stand-in of
tests/golden/taichi_standin providing fields (numpy storage), i32 wrap-around and Matrix.  Small films of
three scenes are rendered with exams/benchmark.py's call sequence in single and double precision; the raw
film sums, the LBVH arrays and the Sobol state go to tests/golden/reference_path.npz, and
tests/test_reference_path_cpu.py holds the C oracle to them.

Build container only; takes a few minutes (the Sobol update is a Python loop over 21201 dimensions).

What is emulated rather than executed (all of it Taichi's compile-time machinery, none of it renderer math):
  * kernel-scope builtins: Taichi rewrites int / float / min / max inside kernels to element-wise casts and
    ti.min / ti.max; here those names are bound in the namespaces of the modules whose @ti.func bodies use them
    (ptina.common, ptina.sampling, ptina.tree.lbvh), after every module -- host code included -- has been imported
    (scalars behave exactly like the Python builtins; int() yields an i32 that wraps);
  * the `subscript` protocol of is_taichi_class objects (ModelPool()[i], FilmTable()[id, x, y]): mapped to
    __getitem__ / __setitem__;
  * ModelPool.from_numpy / ImagePool.from_numpy write field elements through `self[i][k] = ...` (an lvalue only
    inside Taichi): the vertex / mtlid / texel arrays are stored into the fields directly (image ids and base offsets
    come from the reference's own allocators);
  * pysobol (un-vendored dependency, absent): a module of that name serving the same public Joe-Kuo
    new-joe-kuo-6.21201 table from ptina_amd/data/joe_kuo_21201.npz in pysobol's flat [s, a, m_1..m_s] format;
  * the double-precision run gets the f32-rounded scene parameters the oracle's C API takes (camera matrix,
    material factors, lights), so that the two f64 evaluations start from identical numbers;
  * unset texture ids are -1 (DESIGN.md deviation Q6): WorldLight().set(fac, -1) and explicit 12-parameter
    materials with tex = -1, through the reference's own setters.
Taichi's arithmetic itself (fast-math, type inference of locals) is NOT reproduced: parity with real PTina
output stays formally unpinned; this pins the restatement's logic -- traversal order, sample consumption,
MIS bookkeeping, film accumulation -- to the reference's source.

usage: python3 tests/golden/make_reference_path_golden.py            (spawns one child per precision)
'''

import os
import subprocess
import sys
import time
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('PTINA_REFERENCE', '/root/reference')

CASES = {
    # name: (scene key, film nx, ny, spp)
    's34_default_light': ('s34', 14, 12, 3),
    's34_lobes_two_lights': ('lobes', 12, 10, 2),
    's34_textures_env': ('textured', 12, 10, 2),
}


def scene_of(key):
    '''-> (scene, lights or None for the default light, world (fac, tex))'''
    sys.path.insert(0, ROOT)
    from ptina_amd import scenes
    if key == 's34':
        return scenes.scene_s34(), None, ([0.1, 0.1, 0.1, 0.1], -1)
    if key == 'textured':
        # image.py:137-148 + common.py:183-192 (wrap-around bilinear), mtllib.py:30-38 (factor x texel),
        # light/world.py:22-29 + common.py:234-239 (equirect environment with the Blender axis swap)
        v, m, mats, _ = scenes.scene_s34()
        rng = np.random.default_rng(5)
        checker = np.ones((8, 8, 3), np.float32)
        checker[::2, 1::2] = 0.2
        checker[1::2, ::2] = 0.2
        mats = [list(x) for x in mats]
        mats[0][0] = ([1.0, 0.9, 0.8], 1)                  # walls: base colour x checker
        mats[3][0] = ([0.9, 0.9, 1.0], 1)
        mats[3][2] = (0.9, 2)                              # roughness x grey noise
        mats[4][1] = (0.8, 2)                              # metallic x grey noise
        images = [scenes.env_image(32, 16), checker, rng.uniform(0.3, 1.0, (5, 7)).astype(np.float32)]
        return (v, m, mats, images), None, ([1.0, 1.0, 1.0, 1.0], 0)
    # every Disney lobe on the two boxes + an area light and a point light (first-hit `break`, light index pick)
    v, m, mats, _ = scenes.scene_s34()
    mats = list(mats)
    mats[3] = scenes.material(basecolor=(0.9, 0.95, 1.0), roughness=0.25, transmission=0.8, ior=1.5, specular=0.5)
    mats[4] = scenes.material(basecolor=(0.7, 0.1, 0.1), roughness=0.5, clearcoat=1.0, clearcoatGloss=0.9, sheen=0.5,
                              subsurface=0.3, metallic=0.2)
    area = np.array([[1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 3.9], [0.0, -1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    point = np.eye(4)
    point[:3, 3] = (-1.2, 2.5, 1.0)
    lights = [(area, np.array([12.0, 11.0, 9.0]), 0.7, 'AREA'), (point, np.array([20.0, 20.0, 24.0]), 0.3, 'POINT')]
    return (v, m, mats, []), lights, ([0.1, 0.1, 0.1, 0.1], -1)


def child(prec):
    T = np.float64 if prec == 'f64' else np.float32
    sys.path.insert(0, os.path.join(HERE, 'taichi_standin'))
    sys.path.insert(0, REF)
    warnings.filterwarnings('ignore', category=RuntimeWarning)
    import taichi as ti
    assert 'taichi_standin' in ti.__file__
    ti.set_default_fp(T)

    # pysobol.data._sobol_data: flat [s, a, m_1 .. m_s] per dimension >= 1 (tools/encoding.py:47-48, sobol.py:49-53)
    z = np.load(os.path.join(ROOT, 'ptina_amd', 'data', 'joe_kuo_21201.npz'))
    flat = []
    for j in range(1, z['s'].shape[0]):
        s = int(z['s'][j])
        flat += [s, int(z['a'][j])] + [int(x) for x in z['m'][j][:s]]
    pysobol = types.ModuleType('pysobol')
    pysobol.data = types.ModuleType('pysobol.data')
    pysobol.data._sobol_data = flat
    sys.modules['pysobol'] = pysobol
    sys.modules['pysobol.data'] = pysobol.data

    import ptina.common as C
    from ptina.things import init_things, Stack, Camera, BVHTree, ImagePool, ModelPool, LightPool, WorldLight, \
        MaterialPool, FilmTable
    from ptina.engine.path import PathEngine
    from ptina.sampling.sobol import SobolSampler
    from ptina.image import Image
    import ptina.sampling as SAMP
    import ptina.tree.lbvh as LBVH
    import ptina.tools.matrix  # noqa: F401  (host code, imported lazily by Camera(): must star-import common BEFORE the names below exist)
    # kernel-scope builtins, bound only in the modules whose @ti.func bodies use them (host code keeps Python's):
    # common.py ifloor / iceil / clamp / bilerp, sampling wanghash*, lbvh getBoundingBox / genAABBSubstep
    C.int, C.float, C.min, C.max = ti.ti_int, ti.ti_float, ti.min, ti.max
    SAMP.int = ti.ti_int
    LBVH.min, LBVH.max = ti.min, ti.max
    assert SAMP.wanghash2(3, 5) == -1977258872 and isinstance(SAMP.wanghash2(3, 5), ti.I32)

    # the `subscript` protocol of is_taichi_class objects
    ModelPool.__getitem__ = lambda self, i: self.subscript(i)
    FilmTable.__getitem__ = lambda self, ix: self.subscript(*ix)
    FilmTable.__setitem__ = lambda self, ix, v: self.root.__setitem__((ix[0], ix[1] * self.ny + ix[2]), v)
    ImagePool.__getitem__ = lambda self, ix: self.subscript(*ix)
    Image.__getitem__ = lambda self, I: self.subscript(*I)

    out = {}
    t0 = time.time()
    init_things(max_faces=2**10, max_texels=2**10, max_materials=2**4, max_textures=2**2, max_lights=2**3,
                max_filmsize=2**10, max_filmpasses=3)
    eng = PathEngine()                        # SobolSampler(): vgrid + reset (64 skipped updates)
    sob = SobolSampler()
    print(prec, 'sobol ready after %.0f s, time =' % (time.time() - t0), sob.time[None], flush=True)
    out['sobol/time_after_reset'] = np.int64(sob.time[None])
    out['sobol/X_after_reset'] = sob.X.to_numpy().astype(np.int64)
    sobol_state = (sob.X.to_numpy(), sob.P.to_numpy(), int(sob.time[None]))

    for name, (key, nx, ny, spp) in CASES.items():
        scene, lights, world = scene_of(key)
        vertices, mtlids, materials, images = scene
        # rewind the sampler to its state after reset() (what a fresh process would have)
        sob.X.from_numpy(sobol_state[0])
        sob.P.from_numpy(sobol_state[1])
        sob.time[None] = sobol_state[2]

        FilmTable().set_size(nx, ny)
        n = mtlids.shape[0]
        ModelPool().vertices.from_numpy(np.asarray(vertices, np.float32).reshape(-1))    # ModelPool.from_numpy, model.py:54-60
        ModelPool().mtlids.from_numpy(np.asarray(mtlids, np.int32))
        ModelPool().nfaces[None] = n
        MaterialPool().load(materials)
        # ImagePool.load / load_one, image.py:69-96: same conversions, ids and base offsets from the reference's
        # allocators; the texels go into the field directly (from_numpy writes through lvalue subscripts)
        pool = ImagePool()
        pool.mman.reset()
        pool.idman.reset()
        for arr in images:
            arr = np.asarray(arr)
            if arr.dtype == np.uint8:
                arr = arr.astype(np.float32) / 255
            if arr.ndim == 2:
                arr = arr[:, :, None]
            if arr.shape[2] == 1:
                arr = np.stack([arr[:, :, 0]] * 3, axis=2)
            if arr.shape[2] == 3:
                arr = np.concatenate([arr, np.ones(arr.shape[:2] + (1,))], axis=2)
            iid = pool.new(arr.shape[0], arr.shape[1])
            base = int(pool.base[iid])
            pool.root.data[base:base + arr.shape[0] * arr.shape[1]] = arr.astype(np.float32).reshape(-1, 4)
        BVHTree().build()
        Camera().set_perspective(np.array([
            [1.73205081e+00, 0.00000000e+00, 0.00000000e+00, 1.01348227e-02],
            [0.00000000e+00, 1.73205081e+00, -1.73205081e-05, -3.36860025e+00],
            [0.00000000e+00, -1.00020002e-05, -1.00020002e+00, 5.27350023e+00],
            [0.00000000e+00, -1.00000000e-05, -1.00000000e+00, 5.37243564e+00],
        ]))                                   # exams/benchmark.py:18-23
        WorldLight().set(*world)
        if lights is not None:
            LightPool().clear()
            for l in lights:
                LightPool().add(*l)
        else:
            LightPool().clear()               # back to the default light of light/__init__.py:22-28
            LightPool().color[0] = [32, 32, 32]
            LightPool().pos[0] = [1, 2, 3]
            LightPool().size[0] = 0.5
            LightPool().type[0] = LightPool.TYPES['POINT']
            LightPool().count[None] = 1

        if prec == 'f64':
            # the oracle's C API takes f32 scene parameters in either build: give the double-precision run
            # exactly those values (real PTina stores all of them in f32 fields anyway)
            def f32_values(fld):
                fld.data[...] = fld.data.astype(np.float32).astype(np.float64)
            mp = MaterialPool()
            for pair in (mp.basecolor, mp.metallic, mp.roughness, mp.specular, mp.specularTint, mp.subsurface, mp.sheen,
                         mp.sheenTint, mp.clearcoat, mp.clearcoatGloss, mp.transmission, mp.ior):
                f32_values(pair.fac)
            for fld in (Camera()._V2W, Camera()._W2V, LightPool().color, LightPool().pos, LightPool().axes, LightPool().size,
                        WorldLight().fac):
                f32_values(fld)

        tree = BVHTree()
        out[f'{name}/tree/child'] = tree.child.to_numpy()[:n - 1].astype(np.int64)
        out[f'{name}/tree/leaf'] = tree.leaf.to_numpy()[:n].astype(np.int64)
        out[f'{name}/tree/mc'] = tree.mc.to_numpy()[:n].astype(np.int64)
        out[f'{name}/tree/bmin'] = tree.bmin.to_numpy()[:n - 1].astype(np.float64)
        out[f'{name}/tree/bmax'] = tree.bmax.to_numpy()[:n - 1].astype(np.float64)

        t1 = time.time()
        eng.render()                          # exams/benchmark.py:25-27: warm-up frame, (read back), clear
        FilmTable().clear()
        for _ in range(spp):                  # :29-33
            eng.render()
        film = FilmTable().root.to_numpy()[0, :nx * ny].astype(np.float64)
        assert np.all(film[:, 3] == spp)
        out[f'{name}/film'] = film
        # PreviewEngine.render, engine/preview.py:18-41: albedo -> pass 1, shading normal -> pass 2, two frames
        from ptina.engine.preview import PreviewEngine
        for _ in range(2):
            PreviewEngine().render()
        root = FilmTable().root.to_numpy()
        out[f'{name}/preview_albedo'] = root[1, :nx * ny].astype(np.float64)
        out[f'{name}/preview_normal'] = root[2, :nx * ny].astype(np.float64)
        out[f'{name}/size'] = np.array([nx, ny, spp], np.int64)
        out[f'{name}/sobol_time'] = np.int64(sob.time[None])
        print(prec, name, 'rendered in %.0f s; mean radiance' % (time.time() - t1), film[:, :3].mean() / spp, flush=True)

    # tree/lbvh.py:169-305 alone on the 978-triangle benchmark scene (build only: a render would take hours here)
    if prec == 'f32':
        from ptina_amd import scenes
        vertices, mtlids, _, _ = scenes.scene_s978()
        n = mtlids.shape[0]
        ModelPool().vertices.from_numpy(np.asarray(vertices, np.float32).reshape(-1))
        ModelPool().mtlids.from_numpy(np.asarray(mtlids, np.int32))
        ModelPool().nfaces[None] = n
        BVHTree().build()
        tree = BVHTree()
        out['s978/tree/child'] = tree.child.to_numpy()[:n - 1].astype(np.int64)
        out['s978/tree/leaf'] = tree.leaf.to_numpy()[:n].astype(np.int64)
        out['s978/tree/mc'] = tree.mc.to_numpy()[:n].astype(np.int64)
        out['s978/tree/bmin'] = tree.bmin.to_numpy()[:n - 1].astype(np.float64)
        out['s978/tree/bmax'] = tree.bmax.to_numpy()[:n - 1].astype(np.float64)
        assert len(set(out['s978/tree/mc'].tolist())) == n, 'S978 is expected to have distinct Morton codes'
    np.savez_compressed(os.path.join(HERE, f'_reference_path_{prec}.npz'), **out)


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1])
    merged = {}
    procs = [(p, subprocess.Popen([sys.executable, os.path.abspath(__file__), p])) for p in ('f32', 'f64')]
    for p, proc in procs:
        if proc.wait() != 0:
            raise SystemExit(f'{p} run failed')
    for p, _ in procs:
        f = os.path.join(HERE, f'_reference_path_{p}.npz')
        z = np.load(f)
        for k in z.files:
            merged[f'{p}/{k}'] = z[k]
        os.remove(f)
    dst = os.path.join(HERE, 'reference_path.npz')
    np.savez_compressed(dst, **merged)
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    main()
