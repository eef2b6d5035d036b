'''
CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares, fails loudly without a GPU, and the Python host layer packs inputs the way the
reference's pools do.  No compute calls.
'''

import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'miptina.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mpt_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from ptina_amd import _lib
    lib = _lib.load_library()
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f'libmiptina.so lacks {s}'
    assert sorted(_lib.SIGNATURES) == syms, 'ctypes table and header disagree'
    assert lib.mpt_version() >= 100


def test_library_exports_nothing_else_under_its_prefix():
    '''the ABI is the header: kernel launchers and helpers shared between the objects stay internal'''
    import shutil
    import subprocess
    from ptina_amd import _lib
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    if not os.path.exists(nm):
        pytest.skip('no nm')
    out = subprocess.run([nm, '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.split()[-1].startswith('mpt_'))
    assert exported == header_symbols()


def test_no_cpu_fallback():
    from ptina_amd import _lib
    lib = _lib.load_library()
    if lib.mpt_device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(RuntimeError, match='no HIP device'):
        _lib.Context()
    from ptina_amd import common
    common.reset_all()
    from ptina_amd.things import init_things
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        init_things()


def test_product_never_imports_the_oracle():
    bad = []
    for dp, dn, fn in os.walk(os.path.join(ROOT, 'ptina_amd')):
        for f in fn:
            if f.endswith(('.py', '.cpp', '.hip', '.h')):
                s = open(os.path.join(dp, f), errors='replace').read()
                if re.search(r'^\s*(import|from)\s+oracle\b', s, flags=re.M) or 'ptina_oracle' in s:
                    bad.append(f)
    assert not bad, bad


def test_sobol_vgrid_numpy_matches_golden_points():
    '''host-side calc_sobol_vgrid (numpy) -> Gray-code recurrence -> scipy's points'''
    from ptina_amd.sampling.sobol import calc_sobol_vgrid
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'sobol_points.npz'))
    V = calc_sobol_vgrid(2**20, 21201)
    assert V.shape == (21, 21201) and V.dtype == np.int64
    X = np.zeros(21201, np.int64)
    want = {int(k): p for k, p in zip(g['k'], g['P'])}
    for t in range(0, max(want)):
        c = 1
        v = t
        while v & 1:
            v >>= 1
            c += 1
        X ^= V[c]
        if t + 1 in want:
            assert np.array_equal((X / 2.0**32).astype(np.float32), want[t + 1])


def test_material_packing_follows_parameterpair_load():
    from ptina_amd.mtllib import MaterialPool, PARAMS
    pool = MaterialPool.__new__(MaterialPool)
    MaterialPool.__init__(pool, 4)
    assert PARAMS[0] == 'basecolor' and PARAMS[-1] == 'ior' and len(PARAMS) == 12
    pool.basecolor.load(1, [0.1, 0.2, 0.3], -1)           # 3-vector -> + [1.0]
    pool.metallic.load(1, None, 2)                        # None -> 1.0, broadcast to 4
    pool.roughness.load(1, np.float32(0.25), -1)          # 0-d array -> scalar
    pool.ior.load(1, np.array([1, 2, 3, 4.0]), 5)
    assert np.allclose(pool._fac[1, 0], [0.1, 0.2, 0.3, 1.0])
    assert np.allclose(pool._fac[1, 1], [1, 1, 1, 1]) and pool._tex[1, 1] == 2
    assert np.allclose(pool._fac[1, 2], [0.25] * 4)
    assert np.allclose(pool._fac[1, 11], [1, 2, 3, 4]) and pool._tex[1, 11] == 5
    assert np.all(pool._fac[0] == 0) and np.all(pool._tex[0] == -1)   # untouched: zero / none


def test_singleton_semantics():
    from ptina_amd.common import Singleton

    class Foo(metaclass=Singleton):
        def __init__(self, a=1):
            self.a = a
    assert Foo(5) is Foo(7) and Foo().a == 5


def test_matrix_helpers_reproduce_the_benchmark_camera():
    '''exams/benchmark.py:18-23 is perspective(fov=60) @ a view from ~(0,1.95,5.37)'''
    from ptina_amd.tools.matrix import perspective, lookat, ortho, frustum
    from ptina_amd.scenes import BENCH_CAMERA
    p = perspective(fov=60, aspect=1, near=0.05, far=500)
    assert abs(p[0, 0] - 1.73205081) < 1e-6 and abs(p[2, 2] + 1.00020002) < 1e-6
    view = np.linalg.inv(p) @ BENCH_CAMERA
    assert np.allclose(view[3], [0, 0, 0, 1], atol=1e-6)
    eye = np.linalg.inv(view)[:3, 3]
    assert np.allclose(eye, [-0.00585, 1.9449, 5.3724], atol=2e-3)
    assert np.allclose(lookat() @ np.array([0, 0, 3, 1.0]), [0, 0, 0, 1])
    assert np.allclose((ortho() @ np.array([1, -1, 0, 1.0]))[:2], [1, -1])
    assert frustum()[3, 2] == -1


def test_scenes_have_the_quoted_triangle_counts():
    from ptina_amd import scenes
    for name, n in (('s34', 34), ('s978', 978)):
        v, m, mats, imgs = scenes.get_scene(name)
        assert v.shape == (3 * n, 8) and v.dtype == np.float32 and m.shape == (n,)
        assert m.max() < len(mats) and all(len(x) == 12 for x in mats)
        nrm = np.linalg.norm(v[:, 3:6], axis=1)
        assert np.allclose(nrm, 1, atol=1e-5)


def test_ptina_alias_package_maps_onto_ptina_amd():
    '''`from ptina.things import *` etc. resolve to the ptina_amd modules (drop-in imports)'''
    import importlib
    import ptina_amd.filmtable
    import ptina_amd.engine.path
    import ptina_amd.tools.matrix
    assert importlib.import_module('ptina.filmtable') is ptina_amd.filmtable
    assert importlib.import_module('ptina.engine.path') is ptina_amd.engine.path
    assert importlib.import_module('ptina.tools.matrix') is ptina_amd.tools.matrix
    ns = {}
    exec('from ptina.things import *\nfrom ptina.engine.path import *', ns)
    for name in ('init_things', 'FilmTable', 'ModelPool', 'MaterialPool', 'ImagePool', 'BVHTree', 'Camera',
                 'LightPool', 'WorldLight', 'PathEngine', 'ti', 'np'):
        assert name in ns, name


def test_unit_kind_table_mirrors_the_header_enum():
    '''mpt_unit_eval's kinds: the ctypes-side table and the header's enum agree, and every kind cites a reference function'''
    from ptina_amd import _lib
    src = open(os.path.join(ROOT, 'include', 'miptina.h')).read()
    enum = dict((k.lower(), int(v)) for k, v in re.findall(r'MPT_UNIT_([A-Z0-9_]+) = (\d+)', src))
    n = enum.pop('kinds')
    assert n == len(enum) == len(_lib.UNIT_KINDS)
    assert {k: v[0] for k, v in _lib.UNIT_KINDS.items()} == enum
    for line in src.splitlines():
        if re.search(r'MPT_UNIT_[A-Z0-9_]+ = \d+,', line):
            assert re.search(r'\.py:\d+', line), f'no reference citation: {line.strip()}'


def test_rank_device_selection(monkeypatch):
    '''one rank per GPU: LOCAL_RANK indexes the node's devices, except under launchers that isolate one GPU per
    rank (then every rank sees a single device 0); MIPTINA_DEVICE overrides (round-2 ADVICE)'''
    from ptina_amd import _lib
    for k in ('MIPTINA_DEVICE', 'LOCAL_RANK', 'HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(k, raising=False)
    assert _lib.rank_device(8) == 0
    monkeypatch.setenv('LOCAL_RANK', '3')
    assert _lib.rank_device(8) == 3
    assert _lib.rank_device(1) == 3              # a one-GPU box: mpt_create refuses ("device out of range")
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '3')
    assert _lib.rank_device(1) == 0              # isolated: the rank's own GPU is its device 0
    assert _lib.rank_device(8) == 3
    monkeypatch.setenv('MIPTINA_DEVICE', '5')
    assert _lib.rank_device(1) == 5


def test_device_sah_workspace_holds_every_level_up_to_sah_max():
    '''round-3 ADVICE (high), restated for the round-6 pass: the chunk-bin workspace of the on-device SAH pass must hold every
    level it can reach -- a level of nseg segments (each of more than 1024 triangles) has at most n / chunk + nseg chunks of
    24 x bins words, bins falling from 1024 to 32 as the segments multiply.  The pure sizing rule (mpt_sah_workspace: no GPU)
    is checked for models up to sah_max = 2^22 faces, every segment count near the break points and a sweep up to the capacity.'''
    import ctypes as C
    from ptina_amd import _lib
    lib = _lib.load_library()
    out = (C.c_int64 * 4)()
    for n in (33, 1000, 60000, 1 << 20, 1_080_000, 1_100_000, 2_000_000, 2_600_000, 1 << 22):
        assert lib.mpt_sah_workspace(n, 1, out) == 0
        cap, ws = out[0], out[1]
        assert cap >= n // 1025 + 1 and out[3] == 1024
        probe = set(range(1, min(cap, 3000) + 1)) | {cap, max(cap - 1, 1)}
        for k in range(5, 21):                              # around every change of the bin count
            probe |= {x for x in ((1 << k) - 1, 1 << k, (1 << k) + 1) if 1 <= x <= cap}
        probe |= set(range(1, cap + 1, max(1, cap // 4000)))
        for nseg in sorted(probe):
            assert lib.mpt_sah_workspace(n, nseg, out) == 0
            assert out[2] <= ws, (n, nseg, out[2], ws)
            nb = out[3]
            ch = max(2048, min(16 * nb, (n // 256 + 255) // 256 * 256))       # positions per chunk when a level streams all n
            assert nb in (32, 64, 128, 256, 512, 1024) and out[2] == 24 * nb * (n // ch + nseg)
    assert lib.mpt_sah_workspace(0, 1, out) == 1 and lib.mpt_sah_workspace(10, -1, out) == 1
