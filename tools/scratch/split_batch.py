import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from ptina_amd import scenes
from ptina_amd.common import ctx, reset_all
from ptina_amd.things import FilmTable
from helpers import setup_engine
ref = None
for batch in (32, 16, 8, 32, 16):
    reset_all()
    eng = setup_engine(scenes.scene_s978(), 512, 512, mode='fast')
    c = ctx(); film = FilmTable()
    c.set_option('batch', batch)
    eng.render(); film.get_image(); film.clear()
    for _ in range(3):
        eng.render(32); film.get_image()
    c.call('mpt_synchronize'); c.kernel_time()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        eng.render(32); img = film.get_image()
    dt = (time.perf_counter() - t0) / K
    kms, nl = c.kernel_time()
    raw = film.get_raw().copy()
    if ref is None: ref = raw
    print('batch %2d: step %.4f ms (%.1f Msamples/s), %d launches per step, kernel ms sum per step %.4f, film identical to batch 32: %s' % (batch, dt * 1e3, 512 * 512 * 32 / dt / 1e6, nl // K, kms / K, np.array_equal(raw.view(np.uint32), ref.view(np.uint32))), flush=True)
reset_all()
