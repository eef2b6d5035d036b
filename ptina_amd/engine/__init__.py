'''
integrators (reference engine/__init__.py:5-13 is the star-import hub every integrator pulls
its scene singletons from)
'''

from ..camera import *                # noqa: F401,F403
from ..model import *                 # noqa: F401,F403
from ..light import *                 # noqa: F401,F403
from ..light.world import *           # noqa: F401,F403
from ..filmtable import *             # noqa: F401,F403
from ..mtllib import *                # noqa: F401,F403
from ..stack import *                 # noqa: F401,F403
from ..tree import *                  # noqa: F401,F403
from ..image import *                 # noqa: F401,F403
