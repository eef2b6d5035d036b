'''
construction of all scene singletons with their capacities (reference things.py:12-28)
'''

from .common import *                 # noqa: F401,F403
from .stack import *                  # noqa: F401,F403
from .camera import *                 # noqa: F401,F403
from .tree import *                   # noqa: F401,F403
from .image import *                  # noqa: F401,F403
from .model import *                  # noqa: F401,F403
from .light import *                  # noqa: F401,F403
from .light.world import *            # noqa: F401,F403
from .mtllib import *                 # noqa: F401,F403
from .filmtable import *              # noqa: F401,F403
from . import _lib
from . import ti                      # noqa: F401  (scripts say ti.init(ti.cuda))


def init_things(
        max_faces=2**21,
        max_texels=2**22,
        max_materials=2**6,
        max_textures=2**6,
        max_lights=2**6,
        max_filmsize=2**21,
        max_filmpasses=3,
        device=None):
    if not _lib.have_context():
        kw = dict(max_faces=max_faces, max_texels=max_texels, max_materials=max_materials,
                  max_textures=max_textures, max_lights=max_lights, max_filmsize=max_filmsize,
                  max_filmpasses=max_filmpasses)
        if device is not None:
            kw['device'] = device
        _lib.get_context(**kw)
    Stack()
    Camera()
    BVHTree(max_faces)
    ImagePool(max_texels, max_textures)
    ModelPool(max_faces)
    LightPool(max_lights)
    WorldLight()
    MaterialPool(max_materials)
    FilmTable(max_filmsize, max_filmpasses)
