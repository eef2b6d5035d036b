'''
triangle mesh model storage (reference model.py).  The mesh is kept on the host until
BVHTree().build() packs it, leaf-ordered, into the device records (csrc/mpt_types.h).
'''

from .common import *                 # noqa: F401,F403
from .common import Singleton, register, ctx, np
from ._lib import fptr, iptr


@register
class ModelPool(metaclass=Singleton):
    def __init__(self, size=2**21):
        self.size = size
        self._nfaces = 0
        self._vertices = np.zeros((0, 8), np.float32)
        self._mtlids = np.zeros(0, np.int32)

    @property
    def nfaces(self):
        return self._nfaces

    def to_numpy(self, id=None):
        return self._vertices.copy(), self._mtlids.copy()

    def from_numpy(self, arr, mtlids):
        arr = np.ascontiguousarray(arr, np.float32)
        mtlids = np.ascontiguousarray(mtlids, np.int32)
        ctx().call('mpt_load_model', fptr(arr), iptr(mtlids), int(mtlids.shape[0]))
        self._vertices, self._mtlids = arr, mtlids
        self._nfaces = int(mtlids.shape[0])

    def load(self, arr, mtlids=None):
        '''reference model.py:62-86: [3n,8] array (pos3 nrm3 uv2), or an OBJ-style dict, or a path'''
        if isinstance(arr, str):
            from .tools.readobj import readobj
            arr = readobj(arr)

        if isinstance(arr, dict):
            f = arr['f']
            verts = arr['v'][f[:, :, 0]].reshape(f.shape[0] * 3, 3)
            norms = arr['vn'][f[:, :, 2]].reshape(f.shape[0] * 3, 3)
            coors = arr['vt'][f[:, :, 1]].reshape(f.shape[0] * 3, 2)
            arr = np.concatenate([verts, norms, coors], axis=1)

        arr = np.asarray(arr)
        if arr.dtype == np.float64:
            arr = arr.astype(np.float32)

        assert arr.shape[0] % 3 == 0
        if mtlids is None:
            mtlids = -np.ones(arr.shape[0] // 3, dtype=np.int32)
        else:
            mtlids = np.asarray(mtlids)
            assert mtlids.shape[0] == arr.shape[0] // 3
        assert mtlids.shape[0] < self.size, 'too many faces'

        self.from_numpy(arr, mtlids)
