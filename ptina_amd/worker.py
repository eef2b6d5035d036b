'''
flat procedural API (reference worker.py:11-87), what the Blender add-on drives
'''

from .things import *                 # noqa: F401,F403
from .engine.path import PathEngine as DefaultEngine
from .engine.preview import PreviewEngine
from .common import ctx


def init():
    init_things()
    DefaultEngine()
    PreviewEngine()


def synchronize():
    ctx().call('mpt_synchronize')


def render(aa=True):
    DefaultEngine().render()


def render_preview(aa=True):
    PreviewEngine().render()


def set_size(nx, ny):
    FilmTable().set_size(nx, ny)


def get_size():
    return FilmTable().nx, FilmTable().ny


def clear(id=0):
    if hasattr(DefaultEngine(), 'reset'):
        DefaultEngine().reset()
    FilmTable().clear(id)


def set_mlt_param(lsp, sigma):
    '''Metropolis parameters: the MLT engine is outside this package's scope; accepted and
    ignored exactly as the reference does when the default engine is PathEngine
    (worker.py:45-49)'''
    if hasattr(DefaultEngine(), 'LSP'):
        DefaultEngine().LSP = lsp
    if hasattr(DefaultEngine(), 'Sigma'):
        DefaultEngine().Sigma = sigma


def get_image(id=0):
    return FilmTable().get_image(id)


def fast_export_image(pixels, id=0):
    FilmTable().fast_export_image(pixels, id)


def clear_lights():
    LightPool().clear()


def set_world_light(fac, tex):
    WorldLight().set(fac, tex)


def add_light(world, color, size, type):
    LightPool().add(world, color, size, type)


def load_model(vertices, mtlids):
    ModelPool().load(vertices, mtlids)


def load_images(images):
    ImagePool().load(images)


def load_materials(materials):
    MaterialPool().load(materials)


def build_tree():
    BVHTree().build()


def set_camera(pers):
    Camera().set_perspective(pers)
