#!/usr/bin/env python3
'''Diagnostic: the LDS-resident kernel over binary nodes (render_kernel_lds) against the one over 4-wide 8-bit nodes
(render_kernel_lds4, option lds_wide) on the benchmark scene: same film up to ties, work counters, launch times.'''
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
from ptina_amd import scenes, common  # noqa: E402
from ptina_amd.common import ctx  # noqa: E402
from ptina_amd.things import FilmTable  # noqa: E402
from helpers import setup_engine  # noqa: E402

n, spp = 512, 32
scene = scenes.scene_s978()
films = {}
only = [int(a) for a in sys.argv[1:]] or [0, 1]
for wide in only:
    common.reset_all()
    eng = setup_engine(scene, n, n, mode='fast')
    c = ctx()
    c.set_option('lds_wide', wide)
    c.set_option('count', 1)
    eng.render(spp)
    films[wide] = FilmTable().get_raw().copy()
    cnt = c.counters()
    print('lds_wide', wide, 'kernel', c.get_option('last_kernel'), 'wide nodes / depth', c.get_option('wide_nodes'), c.get_option('wide_depth'),
          'fast depth', c.get_option('fast_depth'), 'wide stack', c.get_option('wide_stack'),
          {k: round(cnt[k] / cnt['samples'], 3) for k in ('rays', 'n_node', 'n_box', 'n_tri', 'it_node', 'it_leaf', 'it_shade', 'it_new')}, flush=True)
    c.set_option('count', 0)
    FilmTable().clear()
    for rep in range(3):
        eng.render(spp)
    FilmTable().get_raw()
    c.kernel_time()
    t0 = time.perf_counter()
    for rep in range(20):
        eng.render(spp)
        FilmTable().get_image()
    dt = (time.perf_counter() - t0) / 20
    kms, nl = c.kernel_time()
    print('   step %.4f ms, kernel %.4f ms (%d launches)' % (dt * 1e3, kms / max(nl, 1), nl), flush=True)
if len(films) < 2:
    common.reset_all()
    sys.exit(0)
a, b = films[0], films[1]
d = np.abs(a - b)
rel = np.sqrt(((a[:, :3] - b[:, :3]) ** 2).mean() / (a[:, :3] ** 2).mean())
print('films: identical pixels %.4f %%, max abs diff %.3g, rel rmse %.3g' % (100.0 * (d.max(axis=1) == 0).mean(), d.max(), rel))
common.reset_all()
