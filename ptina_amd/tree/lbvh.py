'''
linear BVH (reference tree/lbvh.py).  build() = Morton codes -> keyed sort -> Karras
hierarchy -> boxes, packed into the traversal records (csrc/miptina.cpp mpt_build_tree);
intersect() (lbvh.py:314-347) is csrc/pt_device.h bvh_closest / bvh_occluded.
'''

from ..common import *                # noqa: F401,F403
from ..common import Singleton, register, ctx, np
from .._lib import fptr, iptr
import ctypes as C


class LinearBVH:
    def __init__(self, n=2**22):
        self.capacity = n

    def build(self):
        ctx().call('mpt_build_tree')

    def to_numpy(self):
        '''the reference-layout arrays (lbvh.py:48-56) of the built tree'''
        from ..model import ModelPool
        n = ModelPool().nfaces
        ni = max(n - 1, 1)
        child = np.zeros((ni, 2), np.int32)
        leaf = np.zeros(max(n, 1), np.int32)
        bmin = np.zeros((ni, 3), np.float32)
        bmax = np.zeros((ni, 3), np.float32)
        mc = np.zeros(max(n, 1), np.int32)
        depth = C.c_int32(0)
        ctx().call('mpt_get_tree', iptr(child), iptr(leaf), fptr(bmin), fptr(bmax), iptr(mc),
                   C.byref(depth))
        k = max(n - 1, 0)
        return dict(child=child[:k], leaf=leaf[:n], bmin=bmin[:k], bmax=bmax[:k], mc=mc[:n],
                    depth=depth.value)


@register
class BVHTree(LinearBVH, metaclass=Singleton):
    pass
