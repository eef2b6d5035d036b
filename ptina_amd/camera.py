'''
camera that generates rays from a given 4x4 perspective matrix (reference camera.py).
The ray generation itself (camera.py:34-39) is csrc/pt_device.h camera_generate.
'''

from .common import *                 # noqa: F401,F403
from .common import Singleton, register, ctx, np
from ._lib import fptr


def _ortho_lookat_default():
    # Camera.__init__ default, reference camera.py:13-15: ortho() @ lookat() of tools/matrix.py
    from .tools.matrix import ortho, lookat
    return ortho() @ lookat()


@register
class Camera(metaclass=Singleton):
    def __init__(self):
        self._pers = None
        self.set_perspective(_ortho_lookat_default())

    def set_perspective(self, pers):
        '''reference camera.py:19-22: V2W = inv(pers) in f64, both stored as f32'''
        pers = np.asarray(pers, dtype=np.float64)
        assert pers.shape == (4, 4)
        invpers = np.linalg.inv(pers)
        self._V2W = np.ascontiguousarray(invpers, np.float32)
        self._W2V = np.ascontiguousarray(pers, np.float32)
        ctx().call('mpt_set_camera', fptr(self._V2W), fptr(self._W2V))

    @property
    def V2W(self):
        return self._V2W.copy()

    @property
    def W2V(self):
        return self._W2V.copy()
