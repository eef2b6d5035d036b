#!/usr/bin/env python3
'''Diagnostic: the data-comparing soak of the tail finalisation.  Every launch of a round finalises its own
tiles (film sum, resolve, image write-out by waves that have run out of work); after every round the RAW FILM must
be bit for bit the film the combine pass makes of the same Sobol index range (tests/helpers.py soak_finalisation).  Shapes: the
whole 512 x 512 film in 4-frame and 32-frame launches, every rank's 1/8 share as `bench.py --gpus 8` deals it (16-column stripes),
a ragged film; each quiet and with a stream of 1 GiB device-to-device copies beside the render.  Then the older checks: pipelined
launches (combine pass) and finalising ones in turn, exact sample counts, the watchdog silent.
usage: soak.py [launches, default 20000]'''
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
from helpers import setup_engine, soak_finalisation  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t0 = time.perf_counter()


def log(msg):
    print('[%7.1f s] %s' % (time.perf_counter() - t0, msg), flush=True)


shapes = []
for stress in (0, 24):
    shapes.append(dict(frames=4, per_round=50, stress_mb=1024, stress_copies=stress))
    shapes.append(dict(frames=32, per_round=10, stress_mb=1024, stress_copies=stress))
    for r in range(8):
        shapes.append(dict(frames=4, per_round=50, stripes=(16, r, 8), stress_mb=1024, stress_copies=stress))
    shapes.append(dict(nx=500, ny=310, frames=3, per_round=50, stress_mb=1024, stress_copies=stress))
    # the gather kernels' finalisation (the same scene forced off LDS: 4-wide 8-bit nodes; binary nodes), whole film and a 1/8 share
    shapes.append(dict(frames=4, per_round=50, stress_mb=1024, stress_copies=stress, opts=(('lds', 0),)))
    shapes.append(dict(frames=4, per_round=50, stripes=(16, 2, 8), stress_mb=1024, stress_copies=stress, opts=(('lds', 0), ('wide', 0))))
weight = sum(2.0 if 'stripes' not in s else 1.0 for s in shapes)
done = 0
for s in shapes:
    n = int(total * (2.0 if 'stripes' not in s else 1.0) / weight) + 1
    k = soak_finalisation(n, log=log, **s)
    done += k
    log('%d launches: %s -- bit-identical to the combine pass' % (k, s))
log('finalised launches compared bit for bit: %d' % done)

# pipelined and finalising launches in turn, a ragged batch, exact sample counts (round 4's soak)
eng = setup_engine(scenes.scene_s978(), 512, 512, mode='fast')
c = ctx()
frames = 0
for r in range(4):
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
    log('mixed round %d: %d frames, mean radiance %.6f' % (r, frames, float(raw[:, :3].sum() / frames / raw.shape[0])))
common.reset_all()
print('OK')
