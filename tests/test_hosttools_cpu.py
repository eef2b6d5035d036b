'''CPU tests of the host-side scene ingestion mirrors: OBJ, mesh composition, glTF,
the worker-thread proxy.  No GPU.'''

import base64
import os
import io
import json
import threading

import numpy as np
import pytest


def test_readobj_layout_and_triangulation():
    from ptina_amd.tools.readobj import readobj, writeobj
    src = b"""# quad + triangle + pentagon
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 2 0
vt 0 0
vt 1 1
vn 0 0 1
usemtl red
f 1/1/1 2/2/1 3/1/1 4/2/1
f 1//1 2//1 3//1
usemtl blue
f 1 2 3 4 5
"""
    obj = readobj(io.BytesIO(src))
    assert obj['v'].shape == (5, 3) and obj['vt'].shape == (2, 2) and obj['vn'].shape == (1, 3)
    f = obj['f']
    assert f.shape == (2 + 1 + 3, 3, 3) and f.dtype == np.int32
    assert f[0, :, 0].tolist() == [0, 1, 2] and f[1, :, 0].tolist() == [2, 3, 0]       # quad split
    assert f[2].tolist() == [[0, 0, 0], [1, 0, 0], [2, 0, 0]]                          # missing vt -> 0
    assert [t[:, 0].tolist() for t in f[3:]] == [[0, 1, 2], [0, 2, 3], [0, 3, 4]]      # fan
    assert obj['usemtl'] == [[0, b'red'], [3, b'blue']]
    v, tri = readobj(io.BytesIO(src), simple=True)
    assert tri.shape == (6, 3)
    out = io.StringIO()
    writeobj(out, obj)
    again = readobj(io.BytesIO(out.getvalue().encode()))
    assert np.array_equal(again['f'], obj['f']) and np.allclose(again['v'], obj['v'])


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_hosttools.npz')


def test_readobj_matches_reference_golden():
    '''outputs of the reference's own ptina.tools.readobj on the committed OBJ text
    (tests/golden/make_reference_golden.py ran the reference module in the build container)'''
    from ptina_amd.tools import readobj as R
    g = np.load(GOLDEN)
    text = g['obj_text'].tobytes()
    for tag, kw in (('default', {}), ('zxy', {'orient': 'zxy'}), ('flipped', {'orient': '-xZy'}), ('scaled', {'scale': 2.5}),
                    ('auto', {'scale': 'auto'}), ('nomtl', {'usemtl': False})):
        obj = R.readobj(io.BytesIO(text), **kw)
        for k in ('v', 'vt', 'vn', 'f'):
            want = g[f'readobj_{tag}_{k}']
            assert obj[k].shape == want.shape and obj[k].dtype == want.dtype, (tag, k)
            assert np.array_equal(obj[k], want) if k == 'f' else np.allclose(obj[k], want, rtol=1e-6, atol=1e-7), (tag, k)
        if tag != 'nomtl':
            assert [u[0] for u in obj['usemtl']] == g[f'readobj_{tag}_usemtl_start'].tolist()
            assert [bytes(u[1]) for u in obj['usemtl']] == g[f'readobj_{tag}_usemtl_name'].tolist()
        else:
            assert 'usemtl' not in obj
    v, tri = R.readobj(io.BytesIO(text), simple=True)
    assert np.array_equal(v, g['readobj_simple_v']) and np.array_equal(tri, g['readobj_simple_f'])
    obj = R.readobj(io.BytesIO(text))
    assert np.array_equal(R.objverts(obj), g['objverts'])
    assert np.array_equal(R.objnorms(obj), g['objnorms'])
    assert np.array_equal(R.objcoors(obj), g['objcoors'])
    assert np.array_equal(R.objmtlids(obj), g['objmtlids']) and R.objmtlids(obj).dtype == np.int32
    parts = R.objunpackmtls(obj)
    assert [bytes(k) for k in parts] == g['objunpackmtls_names'].tolist()
    for k, part in parts.items():
        assert np.array_equal(part['f'], g['objunpackmtls_f_' + k.decode()])
        assert part['v'] is obj['v']
    R.objmknorm(obj)
    assert np.allclose(obj['vn'], g['objmknorm_vn'], rtol=1e-6, atol=1e-7) and np.array_equal(obj['f'], g['objmknorm_f'])
    obj2 = R.readobj(io.BytesIO(g['obj2_text'].tobytes()))
    assert np.array_equal(R.objmtlids(obj2), g['obj2_mtlids'])


def test_compose_multiple_meshes_matches_reference_golden():
    from ptina_amd.multimesh import compose_multiple_meshes
    g = np.load(GOLDEN)
    prims = []
    for i in range(3):
        prims.append((g[f'compose_in{i}_p'], g[f'compose_in{i}_n'], g[f'compose_in{i}_t'], g[f'compose_in{i}_world'],
                      int(g[f'compose_in{i}_mtl']) if f'compose_in{i}_mtl' in g else None))
    verts, mtlids = compose_multiple_meshes(prims)
    assert verts.shape == g['compose_out0'].shape and verts.dtype == np.float64
    assert np.allclose(verts, g['compose_out0'], rtol=1e-13, atol=1e-14)
    assert np.array_equal(mtlids, g['compose_out1'])


def test_modelpool_dict_packing_matches_reference_indexing():
    '''ModelPool.load(dict): verts = v[f[:,:,0]], norms = vn[f[:,:,2]], coords = vt[f[:,:,1]]'''
    from ptina_amd.tools.readobj import readobj
    obj = readobj(io.BytesIO(b"v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0.25 0.75\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\n"))
    f = obj['f']
    arr = np.concatenate([obj['v'][f[:, :, 0]].reshape(-1, 3), obj['vn'][f[:, :, 2]].reshape(-1, 3),
                          obj['vt'][f[:, :, 1]].reshape(-1, 2)], axis=1)
    assert arr.shape == (3, 8)
    assert arr[1].tolist() == [1, 0, 0, 0, 0, 1, 0.25, 0.75]


def test_compose_multiple_meshes():
    from ptina_amd.multimesh import compose_multiple_meshes
    from ptina_amd.tools.matrix import translate, scale
    p = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], float)
    n = np.array([[[0, 0, 1]] * 3], float)
    t = np.array([[[0, 0], [1, 0], [0, 1]]], float)
    verts, mtl = compose_multiple_meshes([(p, n, t, translate([1, 2, 3]), 4),
                                          (p, n, None, scale([2, 2, 2]), None)])
    assert verts.shape == (6, 8) and verts.dtype == np.float64 and mtl.tolist() == [4, -1]
    assert verts[1, :3].tolist() == [2, 2, 3] and verts[1, 6:].tolist() == [1, 0]
    assert verts[4, :3].tolist() == [2, 0, 0] and verts[4, 3:6].tolist() == [0, 0, 1]     # normals re-normalised
    assert verts[4, 6:].tolist() == [0, 0]


def _tiny_gltf():
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32)
    nrm = np.array([[0, 0, 1]] * 4, np.float32)
    idx = np.array([0, 1, 2, 2, 1, 3], np.uint16)
    blob = pos.tobytes() + nrm.tobytes() + idx.tobytes()
    uri = 'data:application/octet-stream;base64,' + base64.b64encode(blob).decode()
    return {
        'asset': {'version': '2.0'}, 'scene': 0, 'scenes': [{'nodes': [0]}],
        'nodes': [{'mesh': 0, 'translation': [0, 0, -2], 'scale': [2, 1, 1]}],
        'meshes': [{'primitives': [{'attributes': {'POSITION': 0, 'NORMAL': 1}, 'indices': 2, 'material': 0}]}],
        'materials': [{'pbrMetallicRoughness': {'baseColorFactor': [0.8, 0.1, 0.1, 1], 'metallicFactor': 0.0,
                                                'roughnessFactor': 0.5}}],
        'buffers': [{'uri': uri, 'byteLength': len(blob)}],
        'bufferViews': [{'buffer': 0, 'byteOffset': 0, 'byteLength': 48},
                        {'buffer': 0, 'byteOffset': 48, 'byteLength': 48},
                        {'buffer': 0, 'byteOffset': 96, 'byteLength': 12}],
        'accessors': [{'bufferView': 0, 'componentType': 5126, 'count': 4, 'type': 'VEC3'},
                      {'bufferView': 1, 'componentType': 5126, 'count': 4, 'type': 'VEC3'},
                      {'bufferView': 2, 'componentType': 5123, 'count': 6, 'type': 'SCALAR'}],
    }


def test_readgltf_minimal(tmp_path):
    from ptina_amd.tools.readgltf import readgltf
    path = tmp_path / 'quad.gltf'
    path.write_text(json.dumps(_tiny_gltf()))
    vertices, mtlids, materials, images = readgltf(str(path))
    assert vertices.shape == (6, 8) and mtlids.tolist() == [0, 0] and images == []
    assert vertices[1, :3].tolist() == [2, 0, -2]                 # scale then translate
    assert np.allclose(vertices[:, 3:6], [0, 0, 1])
    (b, bt), (m, mt), (r, rt) = materials[0]
    assert b == [0.8, 0.1, 0.1, 1] and bt == -1 and (m, mt, r, rt) == (0.0, -1, 0.5, -1)
    # the 3-tuple is what MaterialPool.load zips against its 12 parameters
    assert len(materials[0]) == 3


def test_daemon_module_runs_on_one_thread():
    from ptina_amd.tools.mtworker import DaemonModule, OnDemandProxy

    class Fake:
        def __init__(self):
            self.ids = []

        def where(self):
            self.ids.append(threading.get_ident())
            return threading.get_ident()

        def boom(self):
            raise RuntimeError('x')
    fake = Fake()
    mod = OnDemandProxy(lambda: DaemonModule(lambda: fake))
    a, b = mod.where(), mod.where()
    assert a == b != threading.get_ident()
    assert mod.boom() is None                                      # swallowed and printed, like the reference


def test_benchmark_asset_round_trip(tmp_path):
    '''tools/make_assets.py writes the procedural scenes where PTina's scripts expect their glTF
    assets; reading them back yields the same triangles, grouped by material'''
    import importlib.util
    import os
    from ptina_amd import scenes
    from ptina_amd.tools.readgltf import readgltf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('make_assets', os.path.join(root, 'tools', 'make_assets.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(str(tmp_path))
    vertices, mtlids, materials, images = readgltf(str(tmp_path / 'monkey_cornell.gltf'))
    ref_v, ref_m, ref_mats, _ = scenes.scene_s978()
    assert vertices.shape == (3 * 978, 8) and images == []
    order = np.argsort(ref_m, kind='stable')
    assert np.array_equal(mtlids, ref_m[order])
    assert np.allclose(vertices.reshape(-1, 3, 8), ref_v.reshape(-1, 3, 8)[order], atol=1e-6)
    assert len(materials) == len(ref_mats) and all(len(m) == 3 for m in materials)
    assert np.allclose(materials[3][0][0][:3], ref_mats[3][0][0])


def test_readgltf_pinned_to_the_reference_loaders_steps():
    '''tools/readgltf.py against a committed hand-made scene (tests/golden/minimal_scene.gltf) and the outputs the
    reference's loader produces for it, derived step by step from ptina/tools/readgltf.py:15-240 by
    tests/golden/make_gltf_golden.py (gltflib is not installed, so the reference's loader itself cannot run): node
    transforms (scale, rotation, translation; parent @ child), indexed primitives, a primitive without uvs and
    material, the 3-of-12 material parameters of SURVEY Q7, the texture index handed on as the image id, images as
    [x][y][c]'''
    from ptina_amd.tools.readgltf import readgltf
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    z = np.load(os.path.join(here, 'gltf_expected.npz'))
    vertices, mtlids, materials, images = readgltf(os.path.join(here, 'minimal_scene.gltf'))
    assert vertices.dtype == np.float64 and vertices.shape == z['vertices'].shape == (9, 8)
    assert np.abs(vertices - z['vertices']).max() <= 1e-12
    assert np.array_equal(np.asarray(mtlids, np.int64), z['mtlids']) and z['mtlids'].tolist() == [1, 1, -1]
    assert len(materials) == 2 and all(len(m) == 3 for m in materials)          # base colour, metallic, roughness only
    for k, m in enumerate(materials):
        assert np.allclose(np.asarray(m[0][0], float), z['material_factors'][k, 0])
        assert float(m[1][0]) == z['material_factors'][k, 1, 0] and float(m[2][0]) == z['material_factors'][k, 2, 0]
        assert [int(m[0][1]), int(m[1][1]), int(m[2][1])] == z['material_textures'][k].tolist()
    assert z['material_textures'][1].tolist() == [1, -1, -1]                     # TEXTURE index 1, not its source (image 0)
    assert len(images) == 2
    for k in (0, 1):
        assert images[k].shape[:2] == (2, 3) and np.array_equal(images[k][..., :3], z[f'image{k}'])
    # and the loaded scene goes into the pools the way exams/benchmark.py does it (host packing only, no GPU)
    from ptina_amd.mtllib import MaterialPool
    pool = MaterialPool.__new__(MaterialPool)
    MaterialPool.__init__(pool, 4)
    for i, m in enumerate(materials):
        for pair, (fac, tex) in zip((pool.basecolor, pool.metallic, pool.roughness), m):
            pair.load(i, fac, tex)
    assert np.allclose(pool._fac[1, 0], [0.2, 0.4, 0.6, 1.0]) and pool._tex[1, 0] == 1
    assert np.all(pool._fac[1, 3:] == 0)                                          # the nine unset parameters stay zero (Q7)


def test_worker_functions_take_the_reference_parameter_names():
    '''round-5 ADVICE: the worker facade (reference worker.py:29-87) is called by keyword too -- `fast_export_image(pixels=...)`,
    `load_model(vertices=..., mtlids=...)`: the generated pass-throughs carry the reference's parameter names and defaults,
    reject what a plain def would reject, and hand the values on positionally'''
    import inspect
    from ptina_amd import worker, things
    want = {'set_size': '(nx, ny)', 'get_image': '(id=0)', 'fast_export_image': '(pixels, id=0)', 'clear_lights': '()',
            'set_world_light': '(fac, tex)', 'add_light': '(world, color, size, type)', 'load_model': '(vertices, mtlids)',
            'load_images': '(images)', 'load_materials': '(materials)', 'build_tree': '()', 'set_camera': '(pers)'}
    for name, sig in want.items():
        assert str(inspect.signature(getattr(worker, name))) == sig and getattr(worker, name).__name__ == name
    with pytest.raises(TypeError):
        worker.set_size(1)
    with pytest.raises(TypeError):
        worker.load_model(vertices=1, faces=2)
    seen = []

    class FakeFilm:
        def fast_export_image(self, out, id=0):
            seen.append((out, id))
    old = things.FilmTable
    things.FilmTable = FakeFilm
    try:
        worker.fast_export_image(pixels='P', id=2)
        worker.fast_export_image('Q')
    finally:
        things.FilmTable = old
    assert seen == [('P', 2), ('Q', 0)]
