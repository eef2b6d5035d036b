'''
albedo & normal AOV integrator (reference engine/preview.py): primary hit only, albedo into
film pass 1, shading normal into pass 2.
'''

from . import *                       # noqa: F401,F403
from ..common import Singleton, register, ctx, np
from ..sampling.sobol import SobolSampler


@register
class PreviewEngine(metaclass=Singleton):
    def __init__(self):
        SobolSampler()

    def render(self, nframes=1):
        ctx().call('mpt_render_preview', int(nframes))
