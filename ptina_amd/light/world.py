'''
environment light (reference light/world.py); WorldLight.at (world.py:22-29) is
csrc/pt_device.h world_at.
'''

from ..common import *                # noqa: F401,F403
from ..common import Singleton, register, ctx, np
from .._lib import fptr


@register
class WorldLight(metaclass=Singleton):
    def __init__(self):
        # default factor 0.1 (world.py:14-16); texture "none" (deviation Q6: reference 0)
        self.fac = np.full(4, 0.1, np.float32)
        self.tex = -1

    def set(self, fac, tex):
        fac = np.asarray(fac, np.float32)
        if fac.ndim == 0:
            fac = np.full(4, float(fac), np.float32)
        if fac.shape[0] == 3:
            fac = np.concatenate([fac, [1.0]]).astype(np.float32)
        self.fac = np.ascontiguousarray(fac, np.float32)
        self.tex = int(tex)
        ctx().call('mpt_set_world_light', fptr(self.fac), self.tex)
