'''
point and area lights (reference light/__init__.py).  LightPool.hit/_sample
(light/__init__.py:51-121) are csrc/pt_device.h lights_hit / lights_sample.
'''

from ..common import *                # noqa: F401,F403
from ..common import Singleton, register, ctx, np
from .._lib import fptr, LIGHT_TYPES
import ctypes as C


@register
class LightPool(metaclass=Singleton):
    TYPES = dict(LIGHT_TYPES)

    def __init__(self, count=2**6):
        self.capacity = count
        # the context starts with the reference's default light:
        # POINT at (1,2,3), radius 0.5, colour 32 (light/__init__.py:22-28)
        self._count = 1

    @property
    def count(self):
        return self._count

    def clear(self):
        ctx().call('mpt_clear_lights')
        self._count = 0

    def add(self, world, color, size, type):
        '''reference light/__init__.py:34-49'''
        world = np.asarray(world, np.float64)
        pos = world @ np.array([0, 0, 0, 1])
        pos = pos[:3] / pos[3]
        axes = world[:3, :3]
        if type not in self.TYPES:
            raise KeyError(type)
        idx = C.c_int(-1)
        ctx().call('mpt_add_light', self.TYPES[type],
                   fptr(np.ascontiguousarray(color, np.float32)),
                   fptr(np.ascontiguousarray(pos, np.float32)),
                   fptr(np.ascontiguousarray(axes, np.float32)),
                   float(size), C.byref(idx))
        self._count = idx.value + 1
        return idx.value
