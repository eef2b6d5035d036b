'''
texture atlas (reference image.py + allocator.py).  Texels live in one float4 device array;
Image.__call__ / bilerp (image.py:137-148, common.py:183-192) are csrc/pt_device.h image_sample.
'''

from .common import *                 # noqa: F401,F403
from .common import Singleton, register, ctx, np
from ._lib import fptr
import ctypes as C


@register
class ImagePool(metaclass=Singleton):
    def __init__(self, size=2**22, count=2**6):
        self.size = size
        self.count = count
        self._shapes = []

    def _prepare(self, arr):
        '''reference image.py:69-82'''
        if isinstance(arr, str):
            from PIL import Image as PILImage
            arr = np.swapaxes(np.array(PILImage.open(arr)), 0, 1)[:, ::-1]
        arr = np.asarray(arr)
        if arr.dtype == np.uint8:
            arr = arr.astype(np.float32) / 255
        nx, ny = arr.shape[0], arr.shape[1]
        if len(arr.shape) == 2:
            arr = arr[:, :, None]
        if arr.shape[2] == 1:
            arr = np.stack([arr[:, :, 0]] * 3, axis=2)
        if arr.shape[2] == 3:
            arr = np.concatenate([arr, np.ones((nx, ny, 1))], axis=2)
        return np.ascontiguousarray(arr, np.float32)

    def load_one(self, arr):
        arr = self._prepare(arr)
        id = C.c_int(-1)
        ctx().call('mpt_load_image', fptr(arr), arr.shape[0], arr.shape[1], C.byref(id))
        self._shapes.append(arr.shape[:2])
        return id.value

    def load(self, images):
        '''reference image.py:90-94: reset both allocators, then load in order'''
        ctx().call('mpt_reset_images')
        self._shapes = []
        for arr in images:
            self.load_one(arr)

    def nx(self, id):
        return self._shapes[id][0]

    def ny(self, id):
        return self._shapes[id][1]


class Image:
    def __init__(self, id):
        self.id = id

    @classmethod
    def load(cls, arr):
        return cls(ImagePool().load_one(arr))

    @property
    def nx(self):
        return ImagePool().nx(self.id)

    @property
    def ny(self):
        return ImagePool().ny(self.id)
