#!/usr/bin/env python3
'''Body of tests/test_parity_gpu.py::test_octant_ordered_8wide_tree_and_kernel, run as a program of its own because the 8-wide
octant-ordered tree and kernel live in an A/B build of the library (make -C ptina_amd/csrc oct -> libmiptina_oct.so, loaded through
MIPTINA_LIB) and a process binds one library.  Prints OCT-OK.  usage: oct_check_script.py <repo root>'''
import os
import sys

root = sys.argv[1]
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np  # noqa: E402
from ptina_amd import scenes  # noqa: E402
from helpers import setup_engine, FAST  # noqa: E402
import oracle as oracle_mod  # noqa: E402


def _engine(fresh, *a, **kw):
    return setup_engine(*a, **kw)


def main():

    import ctypes as C
    from helpers import assert_parity, setup_oracle
    from ptina_amd.things import FilmTable, BVHTree
    from ptina_amd.common import ctx, reset_all
    from ptina_amd._lib import fptr, iptr
    for name, kw, nx, ny, spp in (('c5', {'n': 60000}, 160, 128, 4), ('s978', {}, 96, 80, 8)):
        scene = scenes.get_scene(name, **kw)
        n = scene[1].shape[0]
        films = {}
        for w8 in (1, 0):
            reset_all()
            eng = _engine(None, scene, nx, ny, mode='fast', max_faces=max(n + 1, 1 << 21))
            c = ctx()
            c.set_option('lds', 0)
            c.set_option('wide8', w8)
            BVHTree().build()
            c.set_option('count', 1)
            c.call('mpt_reset_counters')
            eng.render(spp)
            cnt = c.counters()
            assert c.get_option('last_kernel') == (4 if w8 else 2)
            films[w8] = (FilmTable().get_image().copy(), cnt['n_node'] / cnt['rays'], cnt['n_tri'] / cnt['rays'])
            raw = FilmTable().get_raw().reshape(nx, ny, 4)
            assert np.all(raw[..., 3] == spp)
            if w8:
                nw = c.get_option('oct_nodes')
                assert nw > 0 and 1 <= c.get_option('oct_depth') <= 40
                rec = np.zeros((nw, 5, 4), np.float32)
                perm = np.zeros(n, np.int32)
                got = C.c_int(0)
                c.call('mpt_get_oct8', fptr(rec), iptr(perm), nw, C.byref(got))
                assert got.value == nw and np.array_equal(np.sort(perm), np.arange(n))
                words = rec.view(np.uint32)
                a_node, a_tri = words[:, 1, 2], words[:, 1, 3]
                imask, lmask = a_node >> 24, a_tri >> 24
                assert np.all(imask & lmask == 0)
                nint = np.array([bin(int(m)).count('1') for m in imask]); nleaf = np.array([bin(int(m)).count('1') for m in lmask])
                assert np.all(nint + nleaf >= 2) and np.all(nint + nleaf <= 8)
                # breadth-first numbering: children of node w start where those of node w - 1 end, triangles likewise
                assert np.array_equal(a_node & 0xffffff, 1 + np.concatenate([[0], np.cumsum(nint)[:-1]]))
                assert np.array_equal(a_tri & 0xffffff, np.concatenate([[0], np.cumsum(nleaf)[:-1]]))
                assert 1 + nint.sum() == nw and nleaf.sum() == n
                # quantised child boxes, decoded as the kernel does: they hold everything below them
                planes = words[:, 2:5, :].reshape(nw, 3, 4).copy()                   # [node][axis]{lo0-3, lo4-7, hi0-3, hi4-7}
                by = planes.view(np.uint8).reshape(nw, 3, 4, 4)
                lo_q = by[:, :, 0:2, :].reshape(nw, 3, 8).astype(np.float32); hi_q = by[:, :, 2:4, :].reshape(nw, 3, 8).astype(np.float32)
                origin = rec[:, 0, :3]; scale = np.stack([rec[:, 0, 3], rec[:, 1, 0], rec[:, 1, 1]], axis=1)
                lo = origin[:, :, None] + lo_q * scale[:, :, None]; hi = origin[:, :, None] + hi_q * scale[:, :, None]   # [node][axis][slot]
                used = ((imask | lmask)[:, None] >> np.arange(8)[None, :]) & 1
                assert np.all((lo_q[:, 0, :] == 255) & (hi_q[:, 0, :] == 0) | (used == 1))      # empty slots: inverted
                verts = scene[0].reshape(-1, 3, 8)[:, :, :3]
                # true boxes bottom-up: a node's box = union of its children's true boxes; leaves from the triangles (via perm -> slot -> face)
                tree = BVHTree().to_numpy()
                face_of_slot = tree['leaf']                                             # leaf slot -> face (lbvh.py leaf array)
                tlo = np.zeros((nw, 3)); thi = np.zeros((nw, 3))
                eps = 1e-4
                for w in range(nw - 1, -1, -1):
                    blo = np.full(3, np.inf); bhi = np.full(3, -np.inf)
                    ci = int(a_node[w] & 0xffffff); ti = int(a_tri[w] & 0xffffff)
                    for s_ in range(8):
                        if (int(imask[w]) >> s_) & 1:
                            clo, chi = tlo[ci], thi[ci]; ci += 1
                        elif (int(lmask[w]) >> s_) & 1:
                            tri = verts[face_of_slot[perm[ti]]]; ti += 1
                            clo, chi = tri.min(axis=0), tri.max(axis=0)
                        else:
                            continue
                        span = np.maximum(np.abs(clo), np.abs(chi)) * 1e-5 + eps * scale[w]
                        assert np.all(lo[w, :, s_] <= clo + span) and np.all(hi[w, :, s_] >= chi - span), (name, w, s_)
                        blo = np.minimum(blo, clo); bhi = np.maximum(bhi, chi)
                    tlo[w], thi[w] = blo, bhi
        reset_all()
        print(f'{name}: node visits per ray 8-wide {films[1][1]:.2f} / 4-wide {films[0][1]:.2f}, triangle tests {films[1][2]:.2f} / {films[0][2]:.2f}')
        assert films[1][1] < 0.8 * films[0][1]                       # fewer, wider steps
        assert_parity(films[1][0], films[0][0], *FAST, what=f'{name}: 8-wide octant-ordered vs 4-wide')
        if name == 's978':
            ref = setup_oracle(oracle_mod, scene, nx, ny)
            ref.render(spp)
            assert_parity(films[1][0], ref.get_image(), *FAST, what='s978 through the 8-wide kernel vs oracle')
    print('OCT-OK')


if __name__ == '__main__':
    main()
