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

# worker function -> (singleton class, method, the worker function's own parameters: name or (name, default)).  Pure
# pass-throughs (reference worker.py:29-30,50-87).  The parameter names are the worker API's -- callers may pass them by keyword
# (round-5 ADVICE: `fast_export_image(pixels=...)`, `load_model(vertices=...)`) -- and the values go on positionally, so the
# method behind may name its own parameters as it likes.
_PASS_THROUGH = {
    'set_size': ('FilmTable', 'set_size', ('nx', 'ny')),
    'get_image': ('FilmTable', 'get_image', (('id', 0),)),
    'fast_export_image': ('FilmTable', 'fast_export_image', ('pixels', ('id', 0))),
    'clear_lights': ('LightPool', 'clear', ()),
    'set_world_light': ('WorldLight', 'set', ('fac', 'tex')),
    'add_light': ('LightPool', 'add', ('world', 'color', 'size', 'type')),
    'load_model': ('ModelPool', 'load', ('vertices', 'mtlids')),
    'load_images': ('ImagePool', 'load', ('images',)),
    'load_materials': ('MaterialPool', 'load', ('materials',)),
    'build_tree': ('BVHTree', 'build', ()),
    'set_camera': ('Camera', 'set_perspective', ('pers',)),
}


def _pass_through(name, owner, method, params):
    import inspect
    P = inspect.Parameter
    sig = inspect.Signature([P(p, P.POSITIONAL_OR_KEYWORD) if isinstance(p, str) else P(p[0], P.POSITIONAL_OR_KEYWORD, default=p[1])
                             for p in params])

    def call(*args, **kwargs):
        bound = sig.bind(*args, **kwargs)           # (TypeError for a missing / unknown / doubled argument, like a plain def)
        bound.apply_defaults()
        return getattr(getattr(_things, owner)(), method)(*bound.args)
    call.__name__ = call.__qualname__ = name
    call.__signature__ = sig
    call.__doc__ = '%s().%s%s' % (owner, method, sig)
    return call


for _name, (_owner, _method, _params) in _PASS_THROUGH.items():
    globals()[_name] = _pass_through(_name, _owner, _method, _params)


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
