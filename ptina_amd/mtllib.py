'''
material table: 12 Disney parameters x (vec4 factor, texture id) per material
(reference mtllib.py).  MaterialPool.get (mtllib.py:79-95) is csrc/pt_device.h material_get.
'''

from .common import *                 # noqa: F401,F403
from .common import Singleton, register, ctx, np
from ._lib import fptr, iptr

PARAMS = ('basecolor', 'metallic', 'roughness', 'specular', 'specularTint', 'subsurface',
          'sheen', 'sheenTint', 'clearcoat', 'clearcoatGloss', 'transmission', 'ior')


class ParameterPair:
    '''one parameter column; a view into MaterialPool's host tables'''

    def __init__(self, pool, k):
        self.pool = pool
        self.k = k

    @property
    def fac(self):
        return self.pool._fac[:, self.k]

    @property
    def tex(self):
        return self.pool._tex[:, self.k]

    def load(self, i, fac, tex):
        '''reference mtllib.py:15-28'''
        if fac is None:
            fac = 1.0
        if isinstance(fac, np.ndarray):
            if len(fac.shape):
                fac = list(fac)
            else:
                fac = float(fac)
        if not isinstance(fac, (tuple, list)):
            fac = [fac, fac, fac, fac]
        if isinstance(fac, (tuple, list)) and len(fac) == 3:
            fac = list(fac) + [1.0]
        self.pool._fac[i, self.k] = fac
        self.pool._tex[i, self.k] = tex


@register
class MaterialPool(metaclass=Singleton):
    def __init__(self, count=2**6):
        self.capacity = count
        # Taichi fields start at zero (mtllib.py:12-13).  Documented deviation:
        # a never-loaded texture id is -1 ("none") here; the reference's 0 samples an image
        # that may not exist (`% 0`).
        self._fac = np.zeros((count, 12, 4), np.float32)
        self._tex = np.full((count, 12), -1, np.int32)
        self.count = 0
        for k, name in enumerate(PARAMS):
            setattr(self, name, ParameterPair(self, k))

    def load(self, materials):
        '''reference mtllib.py:58-77: each material = up to 12 (fac, tex) pairs in PARAMS
        order; missing trailing pairs keep their previous contents'''
        materials = list(materials)
        if len(materials) > self.capacity:
            raise RuntimeError(f'{len(materials)} materials exceed max_materials={self.capacity}')
        for i, material in enumerate(materials):
            params = [getattr(self, name) for name in PARAMS]
            for (fac, tex), param in zip(material, params):
                param.load(i, fac, tex)
        self.count = len(materials)
        ctx().call('mpt_load_materials', fptr(self._fac), iptr(self._tex), self.capacity)
