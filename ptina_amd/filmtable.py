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
        self._next = None             # the array the next get_image(0) will return, already known to the library (_hint)
        self._size = None

    def _res(self):
        # (the size only changes through set_size below: remembered, so that render() / get_image() do not ask the library twice a step)
        if self._size is None:
            nx, ny = C.c_int(0), C.c_int(0)
            ctx().call('mpt_get_size', C.byref(nx), C.byref(ny))
            self._size = (nx.value, ny.value)
        return self._size

    @property
    def nx(self):
        return self._res()[0]

    @property
    def ny(self):
        return self._res()[1]

    def set_size(self, nx, ny):
        ctx().call('mpt_set_size', int(nx), int(ny))
        self._size = (int(nx), int(ny))

    def clear(self, id=0):
        '''zeroes every pass whatever `id` says, as the reference does (filmtable.py:44-45)'''
        ctx().call('mpt_clear', int(id))

    def _hint(self):
        '''called by PathEngine.render() before it enqueues frames: allocate the array the next get_image(0) will return and tell
        the library (mpt_hint_image), so that a render launch can write the resolved image while it drains.  The reference
        allocates that array inside get_image (filmtable.py:48); it is the same fresh array, made a little earlier'''
        shape = self._res() + (4,)
        if self._next is None or self._next.shape != shape:
            if self._next is not None:
                ctx().call('mpt_hint_image', 0, None)       # (waits for a launch that may be writing into the old one)
            self._next = host_array(shape)
            ctx().call('mpt_hint_image', 0, fptr(self._next))

    def get_image(self, id=0):
        '''reference filmtable.py:47-63: [nx, ny, 4] f32, rgb / w, w -> 1; empty -> (.9,.4,.9,0)'''
        nx, ny = self._res()
        arr = None
        if int(id) == 0 and self._next is not None:
            arr, self._next = self._next, None               # (the hint is spent by the call below)
            if arr.shape != (nx, ny, 4):
                ctx().call('mpt_hint_image', 0, None)
                arr = None
        if arr is None:
            arr = host_array((nx, ny, 4))      # a fresh array, as in the reference; page-locked -> one DMA
        try:
            ctx().call('mpt_get_image', int(id), fptr(arr))
        except Exception:
            # the call may have failed before the library let go of the hinted address (a flush that fails): `arr` is about to be
            # dropped and its page-locked buffer recycled for another array of the same size, so the library must forget it first
            # (mpt_hint_image(None) also waits for a launch that may still be writing into it)
            try:
                ctx().call('mpt_hint_image', 0, None)
            except Exception:
                pass
            raise
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
