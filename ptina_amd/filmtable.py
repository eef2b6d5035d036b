'''
film that records the rendering result (reference filmtable.py): float4 per pixel and pass,
element x*ny + y, rgb sums + sample count in w.
'''

from .common import *                 # noqa: F401,F403
from .common import Singleton, register, ctx, np
from ._lib import fptr, host_array
import ctypes as C


@register
class FilmTable(metaclass=Singleton):
    def __init__(self, size=2**21, count=3):
        self.size = size
        self.count = count

    def _res(self):
        nx, ny = C.c_int(0), C.c_int(0)
        ctx().call('mpt_get_size', C.byref(nx), C.byref(ny))
        return nx.value, ny.value

    @property
    def nx(self):
        return self._res()[0]

    @property
    def ny(self):
        return self._res()[1]

    def set_size(self, nx, ny):
        ctx().call('mpt_set_size', int(nx), int(ny))

    def clear(self, id=0):
        '''zeroes every pass whatever `id` says, as the reference does (filmtable.py:44-45)'''
        ctx().call('mpt_clear', int(id))

    def get_image(self, id=0):
        '''reference filmtable.py:47-63: [nx, ny, 4] f32, rgb / w, w -> 1; empty -> (.9,.4,.9,0)'''
        nx, ny = self._res()
        arr = host_array((nx, ny, 4))          # a fresh array, as in the reference; page-locked -> one DMA
        ctx().call('mpt_get_image', int(id), fptr(arr))
        return arr

    def fast_export_image(self, out, id=0):
        '''reference filmtable.py:66-79: flat RGB f32 at (y * nx + x) * 3 into the caller's buffer'''
        nx, ny = self._res()
        assert out.dtype == np.float32 and out.size >= nx * ny * 3 and out.flags['C_CONTIGUOUS']
        ctx().call('mpt_fast_export_image', int(id), fptr(out))

    def get_raw(self, id=0):
        nx, ny = self._res()
        arr = np.empty((nx * ny, 4), np.float32)
        ctx().call('mpt_get_film_raw', int(id), fptr(arr))
        return arr
