import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from ptina_amd import scenes
from ptina_amd.common import ctx, reset_all
from ptina_amd.things import FilmTable
from helpers import setup_engine
for stripes in (None, (16, 3, 8)):
    films = {}
    for rep in range(2):
        for dp in (0, 1):
            reset_all()
            eng = setup_engine(scenes.scene_s978(), 512, 512, mode='fast')
            c = ctx(); film = FilmTable()
            c.set_option('batch', 32)
            c.set_option('drain_pool', dp)
            if stripes: c.call('mpt_set_stripes', *stripes)
            eng.render(); film.get_image(); film.clear()
            for _ in range(3):
                eng.render(32); film.get_image()
            c.call('mpt_synchronize'); c.kernel_time()
            t0 = time.perf_counter()
            K = 30
            for _ in range(K):
                eng.render(32); img = film.get_image()
            dt = (time.perf_counter() - t0) / K
            kms, nl = c.kernel_time()
            films[dp] = film.get_raw().copy()
            print('stripes %s drain_pool %d: step %.4f ms, kernel %.4f ms' % (stripes, dp, dt * 1e3, kms / nl), flush=True)
    print('  films identical:', np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32)), flush=True)
reset_all()
