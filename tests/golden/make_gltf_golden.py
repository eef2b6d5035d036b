#!/usr/bin/env python3
'''
Pins ptina_amd/tools/readgltf.py.  The reference's loader (ptina/tools/readgltf.py:15-240) needs gltflib, which is not
installed here, so it cannot be run; this script writes a small hand-made glTF scene (tests/golden/minimal_scene.gltf,
JSON with a base64 buffer) and derives what the reference's loader returns for it by restating that loader's steps
one by one, each citing the lines it follows, with plain struct / loops -- independent of ptina_amd's reader, which
uses numpy views and its own traversal.  The outputs go to tests/golden/gltf_expected.npz;
tests/test_hosttools_cpu.py::test_readgltf_pinned_to_the_reference_loaders_steps compares.

The scene exercises: a parent node with scale + rotation + translation and a child node with a translation
(readgltf.py:44-52,171-180: local = T @ R @ S, world = parent @ local), an indexed primitive with normals and
texture coordinates, a second primitive without TEXCOORD_0 and without a material (:203-207: uv 0, mtlid -1), two
materials of which one has a base-colour texture (:113-131: THREE (factor, texture) pairs per material -- base colour,
metallic, roughness; the other nine Disney parameters are never set; the texture id is the glTF TEXTURE
index taken as it is, :121-122), normals transformed by the world matrix itself and re-normalised (:213-214).
'''

import base64
import json
import math
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def scene():
    # mesh 0, primitive 0: a unit quad in the xy plane (4 vertices, 6 indices), normals +z, uvs = xy
    pos0 = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)]
    nrm0 = [(0, 0, 1)] * 4
    uv0 = [(0, 0), (1, 0), (1, 1), (0, 1)]
    idx0 = [0, 1, 2, 2, 3, 0]
    # primitive 1: one slanted triangle, no uvs, no material, 8-bit indices
    pos1 = [(0, 0, 1), (2, 0, 1), (0, 3, 2)]
    n = np.cross(np.subtract(pos1[1], pos1[0]), np.subtract(pos1[2], pos1[0]))
    n = (n / np.linalg.norm(n)).tolist()
    nrm1 = [tuple(n)] * 3
    idx1 = [0, 1, 2]
    blob = b''
    views, accessors = [], []

    def add(data, comp, typ, count):
        nonlocal blob
        while len(blob) % 4:
            blob += b'\0'
        views.append({'buffer': 0, 'byteOffset': len(blob), 'byteLength': len(data)})
        accessors.append({'bufferView': len(views) - 1, 'componentType': comp, 'type': typ, 'count': count})
        blob += data
        return len(accessors) - 1
    a_pos0 = add(b''.join(struct.pack('<3f', *p) for p in pos0), 5126, 'VEC3', 4)
    a_nrm0 = add(b''.join(struct.pack('<3f', *p) for p in nrm0), 5126, 'VEC3', 4)
    a_uv0 = add(b''.join(struct.pack('<2f', *p) for p in uv0), 5126, 'VEC2', 4)
    a_idx0 = add(struct.pack('<6H', *idx0), 5123, 'SCALAR', 6)
    a_pos1 = add(b''.join(struct.pack('<3f', *p) for p in pos1), 5126, 'VEC3', 3)
    a_nrm1 = add(b''.join(struct.pack('<3f', *p) for p in nrm1), 5126, 'VEC3', 3)
    a_idx1 = add(struct.pack('<3B', *idx1), 5121, 'SCALAR', 3)
    half = math.sqrt(0.5)
    return {
        'asset': {'version': '2.0', 'generator': 'tests/golden/make_gltf_golden.py (hand-made)'},
        'scene': 0,
        'scenes': [{'nodes': [0]}],
        'nodes': [
            {'name': 'parent', 'mesh': 0, 'children': [1], 'scale': [2.0, 1.0, 0.5], 'rotation': [0.0, 0.0, half, half],   # 90 deg about z
             'translation': [1.0, -2.0, 3.0]},
            {'name': 'child', 'mesh': 1, 'translation': [0.0, 0.0, -1.0]},
        ],
        'meshes': [
            {'primitives': [{'attributes': {'POSITION': a_pos0, 'NORMAL': a_nrm0, 'TEXCOORD_0': a_uv0}, 'indices': a_idx0, 'material': 1}]},
            {'primitives': [{'attributes': {'POSITION': a_pos1, 'NORMAL': a_nrm1}, 'indices': a_idx1}]},
        ],
        'materials': [
            {'pbrMetallicRoughness': {'baseColorFactor': [0.8, 0.05, 0.05, 1.0], 'metallicFactor': 0.0, 'roughnessFactor': 0.5}},
            {'pbrMetallicRoughness': {'baseColorFactor': [0.2, 0.4, 0.6, 1.0], 'metallicFactor': 0.25, 'roughnessFactor': 0.75,
                                      'baseColorTexture': {'index': 1}}},
        ],
        # texture 1 names image 0, but the reference hands the TEXTURE index on as the image id (readgltf.py:121-122): 1
        'textures': [{'source': 1}, {'source': 0}],
        'images': [{'uri': 'data:image/png;base64,' + base64.b64encode(png_2x3()).decode('ascii')},
                   {'uri': 'data:image/png;base64,' + base64.b64encode(png_2x3(flip=True)).decode('ascii')}],
        'buffers': [{'byteLength': len(blob), 'uri': 'data:application/octet-stream;base64,' + base64.b64encode(blob).decode('ascii')}],
        'bufferViews': views,
        'accessors': accessors,
    }


def png_2x3(flip=False):
    '''a 2 (wide) x 3 (high) RGB image, written with zlib only'''
    import zlib
    rows = [[(255, 0, 0), (0, 255, 0)], [(0, 0, 255), (255, 255, 0)], [(10, 20, 30), (40, 50, 60)]]
    if flip:
        rows = rows[::-1]
    raw = b''.join(b'\0' + bytes(c for px in r for c in px) for r in rows)

    def chunk(t, d):
        return struct.pack('>I', len(d)) + t + d + struct.pack('>I', zlib.crc32(t + d) & 0xffffffff)
    return b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', 2, 3, 8, 2, 0, 0, 0)) + chunk(b'IDAT', zlib.compress(raw)) + chunk(b'IEND', b'')


def expected(doc):
    '''the reference loader's steps on `doc`, restated with loops'''
    buf = base64.b64decode(doc['buffers'][0]['uri'].split('base64,')[1])                 # load_uri, readgltf.py:27-38
    views = [buf[v['byteOffset']:v['byteOffset'] + v['byteLength']] for v in doc['bufferViews']]   # :55-60
    fmt = {5120: 'b', 5121: 'B', 5122: 'h', 5123: 'H', 5125: 'I', 5126: 'f'}           # component_types, :68
    width = {'SCALAR': 1, 'VEC2': 2, 'VEC3': 3, 'VEC4': 4}

    def accessor(i):                                                                   # get_accessor_buffer, :67-86 (from the view's start)
        a = doc['accessors'][i]
        w = width[a['type']]
        vals = struct.unpack_from('<%d%s' % (a['count'] * w, fmt[a['componentType']]), views[a['bufferView']], 0)
        return [list(vals[k * w:(k + 1) * w]) for k in range(a['count'])] if w > 1 else list(vals)

    def matmul(A, B):
        return [[sum(A[i][k] * B[k][j] for k in range(4)) for j in range(4)] for i in range(4)]

    def local_matrix(node):                                                            # get_node_local_matrix, :44-52
        m = [[float(i == j) for j in range(4)] for i in range(4)]
        if 'scale' in node:                                                            # matrix.scale
            s = node['scale']
            m = matmul([[s[0], 0, 0, 0], [0, s[1], 0, 0], [0, 0, s[2], 0], [0, 0, 0, 1]], m)
        if 'rotation' in node:                                                         # matrix.quaternion, tools/matrix.py:78-89
            q = node['rotation']
            R = [[1.0 - 2 * (q[1] * q[1] + q[2] * q[2]), 2 * (q[0] * q[1] - q[3] * q[2]), 2 * (q[3] * q[1] + q[0] * q[2]), 0],
                 [2 * (q[0] * q[1] + q[3] * q[2]), 1.0 - 2 * (q[0] * q[0] + q[2] * q[2]), 2 * (q[1] * q[2] - q[3] * q[0]), 0],
                 [2 * (q[0] * q[2] - q[3] * q[1]), 2 * (q[1] * q[2] + q[3] * q[0]), 1.0 - 2 * (q[0] * q[0] + q[1] * q[1]), 0],
                 [0, 0, 0, 1]]
            m = matmul(R, m)
        if 'translation' in node:                                                      # matrix.translate
            t = node['translation']
            m = matmul([[1, 0, 0, t[0]], [0, 1, 0, t[1]], [0, 0, 1, t[2]], [0, 0, 0, 1]], m)
        return m

    materials = []                                                                     # process_material, :113-131
    for m in doc['materials']:
        pbr = m['pbrMetallicRoughness']
        bt = pbr['baseColorTexture']['index'] if 'baseColorTexture' in pbr else -1     # the TEXTURE index, as it is (:121-122)
        materials.append(((pbr.get('baseColorFactor'), bt), (pbr.get('metallicFactor'), -1), (pbr.get('roughnessFactor'), -1)))

    prims = []

    def process_node(ni, world):                                                       # process_node, :171-180
        node = doc['nodes'][ni]
        world = matmul(world, local_matrix(node))
        if 'mesh' in node:                                                             # process_mesh / process_primitive, :140-163
            for pr in doc['meshes'][node['mesh']]['primitives']:
                at = pr['attributes']
                prims.append((accessor(at['POSITION']), accessor(at['NORMAL']),
                              accessor(at['TEXCOORD_0']) if 'TEXCOORD_0' in at else None, world, accessor(pr['indices']), pr.get('material')))
        for ch in node.get('children', []):
            process_node(ch, world)
    for ni in doc['scenes'][doc['scene']]['nodes']:                                    # process_scene, :183-188
        process_node(ni, [[float(i == j) for j in range(4)] for i in range(4)])

    rows, mtlids = [], []
    for p, n, t, w, f, m in prims:                                                     # transform_primitive, :197-222
        if t is None:
            t = [[0.0, 0.0] for _ in p]
        if m is None:
            m = -1
        for i in f:
            P = [sum(w[r][c] * (list(p[i]) + [1.0])[c] for c in range(4)) for r in range(4)]      # np34(p, 1) @ w.T
            N = [sum(w[r][c] * (list(n[i]) + [0.0])[c] for c in range(4)) for r in range(3)]      # (np34(n, 0) @ w.T)[:, :3]
            ln = math.sqrt(sum(x * x for x in N))
            rows.append([P[0] / P[3], P[1] / P[3], P[2] / P[3], N[0] / ln, N[1] / ln, N[2] / ln, float(t[i][0]), float(t[i][1])])
        assert len(f) % 3 == 0
        mtlids += [m] * (len(f) // 3)
    # images: np.swapaxes(np.array(Image.open(...)), 0, 1), :89-103 -> [x][y][c]
    img = [[(255, 0, 0), (0, 0, 255), (10, 20, 30)], [(0, 255, 0), (255, 255, 0), (40, 50, 60)]]
    img1 = [col[::-1] for col in img]                                                   # the second image: rows reversed
    return np.array(rows, np.float64), np.array(mtlids, np.int64), materials, (np.array(img, np.uint8), np.array(img1, np.uint8))


def main():
    doc = scene()
    with open(os.path.join(HERE, 'minimal_scene.gltf'), 'w') as f:
        json.dump(doc, f, indent=1)
    v, m, mats, img = expected(doc)
    np.savez_compressed(os.path.join(HERE, 'gltf_expected.npz'), vertices=v, mtlids=m,
                        material_factors=np.array([[list(b[0]), [b2[0]] * 4, [b3[0]] * 4] for b, b2, b3 in mats], np.float64),
                        material_textures=np.array([[b[1], b2[1], b3[1]] for b, b2, b3 in mats], np.int64), image0=img[0], image1=img[1])
    print('wrote minimal_scene.gltf and gltf_expected.npz:', v.shape, m.tolist())


if __name__ == '__main__':
    main()
