#!/usr/bin/env python3
'''Diagnostic: step and kernel time of the whole benchmark film and of a 1/8 stripe share (one launch at a time, read back).'''
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from ptina_amd import scenes, common  # noqa: E402
from ptina_amd.common import ctx  # noqa: E402
from ptina_amd.things import FilmTable  # noqa: E402
from helpers import setup_engine  # noqa: E402

for parts in (1, 8):
    common.reset_all()
    eng = setup_engine(scenes.scene_s978(), 512, 512, mode='fast')
    c = ctx()
    c.set_option('batch', 32)
    if parts > 1:
        c.call('mpt_set_stripes', 16, 0, parts)
    for _ in range(3):
        eng.render(32)
        FilmTable().get_image()
    c.kernel_time()
    t0 = time.perf_counter()
    for _ in range(40):
        eng.render(32)
        FilmTable().get_image()
    dt = (time.perf_counter() - t0) / 40 * 1e3
    kms, nl = c.kernel_time()
    print('N=%d step %.4f ms kernel %.4f ms' % (parts, dt, kms / max(nl, 1)), flush=True)
common.reset_all()
