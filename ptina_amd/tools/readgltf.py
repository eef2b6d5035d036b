'''
glTF 2.0 (.gltf JSON with external or base64 buffers, .glb) -> what the reference's loader
returns (tools/readgltf.py:15-240):  vertices [3n,8] f64, mtlids [n], materials as
((basecolor, tex), (metallic, tex), (roughness, tex)) 3-tuples, images as [x,y,c] arrays.
Written against the glTF specification with json/struct only (the reference uses gltflib).
'''

import base64
import io
import json
import os
import struct

import numpy as np

from . import matrix as mx

_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_WIDTH = {'SCALAR': 1, 'VEC2': 2, 'VEC3': 3, 'VEC4': 4, 'MAT2': 4, 'MAT3': 9, 'MAT4': 16}


def _quaternion(q):
    x, y, z, w = q
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (w * y + x * z)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return mx.affine(r, np.zeros(3))


def readgltf(path):
    with open(path, 'rb') as fh:
        blob = fh.read()
    glb_bin = None
    if blob[:4] == b'glTF':
        _, _, total = struct.unpack_from('<III', blob, 0)
        off = 12
        doc = None
        while off < total:
            clen, ctype = struct.unpack_from('<II', blob, off)
            chunk = blob[off + 8:off + 8 + clen]
            if ctype == 0x4E4F534A:
                doc = json.loads(chunk.decode('utf-8'))
            elif ctype == 0x004E4942:
                glb_bin = chunk
            off += 8 + clen
    else:
        doc = json.loads(blob.decode('utf-8'))
    base = os.path.dirname(os.path.abspath(path))

    def load_uri(uri):
        if uri.startswith('data:'):
            return base64.b64decode(uri[uri.index('base64,') + 7:])
        with open(uri if os.path.isabs(uri) else os.path.join(base, uri), 'rb') as fh:
            return fh.read()

    buffers = [glb_bin if 'uri' not in b else load_uri(b['uri']) for b in doc.get('buffers', [])]
    views = []
    for v in doc.get('bufferViews', []):
        o = v.get('byteOffset', 0)
        views.append((buffers[v['buffer']][o:o + v['byteLength']], v.get('byteStride', 0)))

    def accessor(i):
        a = doc['accessors'][i]
        data, stride = views[a['bufferView']]
        comp, width = np.dtype(_COMP[a['componentType']]), _WIDTH[a['type']]
        o, count = a.get('byteOffset', 0), a['count']
        item = comp.itemsize * width
        if stride and stride != item:
            rows = [np.frombuffer(data, comp, width, o + k * stride) for k in range(count)]
            arr = np.stack(rows)
        else:
            arr = np.frombuffer(data, comp, count * width, o).reshape(count, width)
        return arr[:, 0] if width == 1 else arr

    images = []
    for im in doc.get('images', []):
        raw = load_uri(im['uri']) if 'uri' in im else views[im['bufferView']][0]
        from PIL import Image
        images.append(np.swapaxes(np.array(Image.open(io.BytesIO(raw))), 0, 1))

    materials = []
    for m in doc.get('materials', []):
        pbr = m.get('pbrMetallicRoughness', {})
        if 'metallicRoughnessTexture' in pbr:
            raise AssertionError('metallicRoughness texture not supported')      # as the reference
        # the reference hands the glTF TEXTURE index on as the image id, without going through textures[i].source
        # (readgltf.py:121-122); exporters write texture i -> image i, so the two coincide on its assets.  Kept as it is.
        tex = pbr.get('baseColorTexture', {}).get('index', -1)
        materials.append(((pbr.get('baseColorFactor', [1.0, 1.0, 1.0, 1.0]), tex),
                          (pbr.get('metallicFactor', 1.0), -1), (pbr.get('roughnessFactor', 1.0), -1)))

    prims = []

    def visit(ni, world):
        node = doc['nodes'][ni]
        if 'matrix' in node:
            local = np.array(node['matrix'], float).reshape(4, 4).T
        else:
            local = np.eye(4)
            if 'scale' in node:
                local = mx.scale(node['scale']) @ local
            if 'rotation' in node:
                local = _quaternion(node['rotation']) @ local
            if 'translation' in node:
                local = mx.translate(node['translation']) @ local
        world = world @ local
        if 'mesh' in node:
            for pr in doc['meshes'][node['mesh']]['primitives']:
                if pr.get('mode', 4) != 4:
                    continue
                at = pr['attributes']
                pos = accessor(at['POSITION'])
                nrm = accessor(at['NORMAL']) if 'NORMAL' in at else None
                uv = accessor(at['TEXCOORD_0']) if 'TEXCOORD_0' in at else None
                idx = accessor(pr['indices']).astype(np.int64) if 'indices' in pr else np.arange(pos.shape[0])
                prims.append((pos, nrm, uv, world, idx, pr.get('material')))
        for ch in node.get('children', []):
            visit(ch, world)

    scene = doc['scenes'][doc.get('scene', 0)]
    for ni in scene['nodes']:
        visit(ni, np.eye(4))

    arrays, mtlids = [], []
    for pos, nrm, uv, world, idx, mtl in prims:
        assert nrm is not None, 'primitive without normals'
        p = pos.astype(np.float64)[idx]
        n = nrm.astype(np.float64)[idx]
        t = np.zeros((p.shape[0], 2)) if uv is None else uv.astype(np.float64)[idx]
        ph = np.concatenate([p, np.ones((p.shape[0], 1))], axis=1) @ world.T
        p = ph[:, :3] / ph[:, 3:4]
        n = n @ world[:3, :3].T
        n = n / np.linalg.norm(n, axis=1, keepdims=True)
        a = np.concatenate([p, n, t], axis=1)
        assert a.shape[0] % 3 == 0
        arrays.append(a)
        mtlids.append(np.full(a.shape[0] // 3, -1 if mtl is None else mtl))
    assert arrays, 'no triangle primitives'
    return np.concatenate(arrays), np.concatenate(mtlids), materials, images
