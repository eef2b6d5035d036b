#!/usr/bin/env python3
'''Diagnostic: a long run of launches -- steps that end with a read-back (each finalises its own tiles) and pipelined renders
(combine pass) in turn -- then the sample count of every pixel must be exactly the frames rendered, the film finite, and the
watchdog silent.  usage: soak.py [rounds]'''
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

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = setup_engine(scenes.scene_s978(), 512, 512, mode='fast')
c = ctx()
frames = 0
t0 = time.perf_counter()
for r in range(rounds):
    for _ in range(100):                 # steps with a read-back
        eng.render(32)
        FilmTable().get_image()
        frames += 32
    for _ in range(100):                 # pipelined
        eng.render(32)
        frames += 32
    eng.render(7)                        # a ragged batch
    frames += 7
    raw = FilmTable().get_raw()
    assert np.all(raw[:, 3] == frames), (r, frames, float(raw[:, 3].min()), float(raw[:, 3].max()))
    assert np.isfinite(raw).all()
    print('round %d: %d frames, %.1f s, mean radiance %.6f' % (r, frames, time.perf_counter() - t0, float(raw[:, :3].sum() / frames / raw.shape[0])), flush=True)
common.reset_all()
print('OK')
