'''
unidirectional path integrator (reference engine/path.py).  path_trace / do_render
(path.py:18-93) are the render megakernel in csrc/render_kernel.hip.
'''

from . import *                       # noqa: F401,F403
from ..common import Singleton, register, ctx, np
from ..sampling import *              # noqa: F401,F403
from ..sampling.sobol import *        # noqa: F401,F403
from ..sampling.sobol import SobolSampler


@register
class PathEngine(metaclass=Singleton):
    def __init__(self):
        SobolSampler()
        self._film_cls = None

    def render(self, nframes=1):
        '''reference path.py:75-77: one Sobol update + one sample per pixel, asynchronously.
        Consecutive calls are fused into one launch at the next read-back.'''
        if self._film_cls is None:
            from ..filmtable import FilmTable
            self._film_cls = FilmTable
        self._film_cls()._hint()          # (where get_image() will want the image: a launch may write it while it drains)
        ctx().call('mpt_render', int(nframes))
