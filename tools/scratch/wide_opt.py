import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from ptina_amd import scenes
from ptina_amd.common import ctx, reset_all
from ptina_amd.things import FilmTable, BVHTree
from helpers import setup_engine
films = {}
for name in ('s978', 's34'):
    for rep in range(2):
        for opt in (0, 1):
            reset_all()
            eng = setup_engine(scenes.get_scene(name), 512, 512, mode='fast')
            c = ctx(); film = FilmTable()
            c.set_option('batch', 32)
            c.set_option('wide_build', 0); c.set_option('wide_opt', opt)
            BVHTree().build()
            eng.render(); film.get_image(); film.clear()
            c.set_option('count', 1); c.call('mpt_reset_counters')
            eng.render(32); c.call('mpt_flush'); k = c.counters(); c.set_option('count', 0)
            film.clear()
            for _ in range(3):
                eng.render(32); film.get_image()
            c.call('mpt_synchronize'); c.kernel_time()
            K = 20
            t0 = time.perf_counter()
            for _ in range(K):
                eng.render(32); film.get_image()
            dt = (time.perf_counter() - t0) / K
            kms, nl = c.kernel_time()
            films[(name, opt)] = film.get_raw().copy()
            print('%s wide_opt %d: nodes %d stack %d, node steps per ray %.4f tri tests per ray %.4f | step %.4f ms kernel %.4f ms (kernel id %d)' % (
                name, opt, c.get_option('wide_nodes'), c.get_option('wide_stack'), k['n_node'] / k['rays'], k['n_tri'] / k['rays'], dt * 1e3, kms / nl, c.get_option('last_kernel')), flush=True)
    a, b = films[(name, 0)].view(np.uint32), films[(name, 1)].view(np.uint32)
    print('  %s films: %d of %d pixels differ' % (name, int((a != b).any(axis=1).sum()), a.shape[0]), flush=True)
reset_all()
