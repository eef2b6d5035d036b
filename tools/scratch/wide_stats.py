'''prints the 4-wide tree's size for the small scenes: how much LDS render_kernel_lds4 needs (nodes x MPT_LDS4_NODE_STRIDE, stack levels x 2 KiB)'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import helpers
from ptina_amd import scenes
from ptina_amd.common import ctx
for name in ('s34', 's978'):
    eng = helpers.setup_engine(scenes.get_scene(name), 64, 64)
    eng.render()
    c = ctx()
    print(name, {k: c.get_option(k) for k in ('wide_nodes', 'wide_stack', 'last_kernel')}, flush=True)
