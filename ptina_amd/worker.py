'''
The flat, procedural face of the renderer that the Blender add-on drives through its worker thread
(reference worker.py:11-87: same function names, same arguments, same effects).  Most entries hand their
arguments to one method of one singleton: those are generated from the table below; the few with
behaviour of their own are written out.
'''

from .things import *                 # noqa: F401,F403  (callers do `from ptina.worker import *` and expect the pools' names too)
from . import things as _things
from .engine.path import PathEngine as DefaultEngine
from .engine.preview import PreviewEngine
from .common import ctx

# worker function -> (singleton class, method): pure pass-throughs (reference worker.py:54-87)
_PASS_THROUGH = {
    'set_size': ('FilmTable', 'set_size'),
    'get_image': ('FilmTable', 'get_image'),
    'fast_export_image': ('FilmTable', 'fast_export_image'),
    'clear_lights': ('LightPool', 'clear'),
    'set_world_light': ('WorldLight', 'set'),
    'add_light': ('LightPool', 'add'),
    'load_model': ('ModelPool', 'load'),
    'load_images': ('ImagePool', 'load'),
    'load_materials': ('MaterialPool', 'load'),
    'build_tree': ('BVHTree', 'build'),
    'set_camera': ('Camera', 'set_perspective'),
}


def _pass_through(owner, method):
    def call(*args, **kwargs):
        return getattr(getattr(_things, owner)(), method)(*args, **kwargs)
    call.__doc__ = '%s().%s(...)' % (owner, method)
    return call


for _name, (_owner, _method) in _PASS_THROUGH.items():
    globals()[_name] = _pass_through(_owner, _method)
    globals()[_name].__name__ = _name


def init():
    '''every pool, then both engines (reference worker.py:11-14)'''
    init_things()
    for engine in (DefaultEngine, PreviewEngine):
        engine()


def synchronize():
    '''the reference forces a device sync by reading a field back (worker.py:17-18); here the C ABI has the call'''
    ctx().call('mpt_synchronize')


def render(aa=True):
    DefaultEngine().render()


def render_preview(aa=True):
    PreviewEngine().render()


def get_size():
    film = FilmTable()
    return film.nx, film.ny


def clear(id=0):
    '''an engine that keeps state of its own between frames (the reference's MLT engine) is reset with the film'''
    reset = getattr(DefaultEngine(), 'reset', None)
    if callable(reset):
        reset()
    FilmTable().clear(id)


def set_mlt_param(lsp, sigma):
    '''Metropolis parameters: the MLT engine is outside this package's scope, so with PathEngine as the default
    engine they find nothing to set -- accepted and ignored, as in the reference (worker.py:45-49)'''
    engine = DefaultEngine()
    for attr, value in (('LSP', lsp), ('Sigma', sigma)):
        if attr in dir(engine):
            setattr(engine, attr, value)
