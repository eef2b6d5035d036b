#!/usr/bin/env python3
'''
Write the procedural benchmark scenes as glTF files where PTina's scripts look for their assets
(assets/monkey_cornell.gltf, assets/cornell.gltf; the originals are git-ignored upstream), so that
`PYTHONPATH=<this repo> python <ptina>/exams/benchmark.py` runs unchanged against ptina_amd.
One mesh primitive per material; base colour / metallic / roughness go into pbrMetallicRoughness
(all that the reference's loader reads, tools/readgltf.py:116-134).
'''
import base64
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ptina_amd import scenes  # noqa: E402


def scene_to_gltf(scene):
    vertices, mtlids, materials, _ = scene
    v = np.asarray(vertices, np.float32).reshape(-1, 3, 8)
    blob = b''
    views, accessors, prims = [], [], []

    def add(arr, target_type):
        nonlocal blob
        arr = np.ascontiguousarray(arr)
        off = len(blob)
        blob += arr.tobytes()
        blob += b'\0' * (-len(blob) % 4)
        views.append({'buffer': 0, 'byteOffset': off, 'byteLength': arr.nbytes})
        acc = {'bufferView': len(views) - 1, 'componentType': 5126, 'count': int(arr.shape[0]), 'type': target_type}
        if target_type == 'VEC3' and arr.shape[1] == 3:
            acc['min'] = arr.min(axis=0).tolist()
            acc['max'] = arr.max(axis=0).tolist()
        accessors.append(acc)
        return len(accessors) - 1

    for m in sorted(set(int(x) for x in mtlids)):
        tri = v[mtlids == m].reshape(-1, 8)
        prims.append({'attributes': {'POSITION': add(tri[:, 0:3], 'VEC3'), 'NORMAL': add(tri[:, 3:6], 'VEC3'),
                                     'TEXCOORD_0': add(tri[:, 6:8], 'VEC2')},
                      'material': m, 'mode': 4})
    mats = []
    for mat in materials:
        base = list(mat[0][0]) if not np.isscalar(mat[0][0]) else [mat[0][0]] * 3
        mats.append({'pbrMetallicRoughness': {'baseColorFactor': [float(x) for x in (base + [1.0])[:4]],
                                              'metallicFactor': float(mat[1][0]), 'roughnessFactor': float(mat[2][0])}})
    return {'asset': {'version': '2.0', 'generator': 'ptina_amd tools/make_assets.py'},
            'scene': 0, 'scenes': [{'nodes': [0]}], 'nodes': [{'mesh': 0, 'name': 'scene'}],
            'meshes': [{'primitives': prims}], 'materials': mats,
            'buffers': [{'byteLength': len(blob),
                         'uri': 'data:application/octet-stream;base64,' + base64.b64encode(blob).decode()}],
            'bufferViews': views, 'accessors': accessors}


def main(outdir=None):
    outdir = outdir or os.path.join(ROOT, 'assets')
    os.makedirs(outdir, exist_ok=True)
    for fname, name in (('monkey_cornell.gltf', 's978'), ('cornell.gltf', 's34')):
        path = os.path.join(outdir, fname)
        with open(path, 'w') as fh:
            json.dump(scene_to_gltf(scenes.get_scene(name)), fh)
        print('wrote', path)


if __name__ == '__main__':
    main(*sys.argv[1:])
