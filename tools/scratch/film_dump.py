import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from ptina_amd import scenes
from ptina_amd.common import ctx, reset_all
from ptina_amd.things import FilmTable
from helpers import setup_engine
tag = sys.argv[1]
out = {}
for name, nx, ny, spp in (('s34', 64, 48, 2),):
    reset_all()
    eng = setup_engine(scenes.get_scene(name), nx, ny, mode='fast')
    c = ctx(); c.set_option('batch', 32); c.set_option('count', 1); c.call('mpt_reset_counters')
    eng.render(spp); c.call('mpt_flush')
    k = c.counters()
    out['%s_%d' % (name, nx)] = FilmTable().get_raw().copy()
    print(tag, name, nx, 'kernel', c.get_option('last_kernel'), 'node steps per ray %.4f tri tests per ray %.4f rays %d astray %d' % (k['n_node'] / k['rays'], k['n_tri'] / k['rays'], k['rays'], k['pl_taken']), flush=True)
np.savez(os.path.join(ROOT, 'gpurun_out', 'film_%s.npz' % tag), **out)
reset_all()
