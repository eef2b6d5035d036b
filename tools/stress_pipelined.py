#!/usr/bin/env python3
'''Diagnostic: the same pipelined render (2048 x 2048, render(256) = eight launches back to back; whole film or one stripe share)
repeated in fresh contexts must give the same raw film every time.  usage: stress_pipelined.py [rounds] [finalise] [stripes 0/1]'''
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
from ptina_amd import scenes, common  # noqa: E402
from ptina_amd.common import ctx  # noqa: E402
from ptina_amd.things import FilmTable  # noqa: E402
from helpers import setup_engine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
fin = int(sys.argv[2]) if len(sys.argv) > 2 else 1
stripes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n, spp = 2048, 256
scene = scenes.scene_s978()
ref = None
bad = 0
for i in range(rounds):
    common.reset_all()
    eng = setup_engine(scene, n, n, mode='fast', max_filmsize=n * n)
    c = ctx()
    c.set_option('finalise', fin)
    if stripes:
        c.call('mpt_set_stripes', 16, i % 8 if stripes == 2 else 3, 8)
    eng.render(spp)
    raw = FilmTable().get_raw().copy()
    if stripes == 2:
        continue
    if ref is None:
        ref = raw
        print('round 0: counted', float(raw[:, 3].max()), flush=True)
    else:
        d = (raw.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
        if d.any():
            bad += 1
            idx = np.flatnonzero(d)
            x, y = idx // n, idx % n
            print(f'round {i}: {len(idx)} pixels differ; x range {x.min()}..{x.max()}, y range {y.min()}..{y.max()}; '
                  f'w values {np.unique(raw[idx, 3])[:8]} vs {np.unique(ref[idx, 3])[:8]}; first {idx[:8]}', flush=True)
        else:
            print(f'round {i}: identical', flush=True)
common.reset_all()
print('BAD' if bad else 'OK', bad)
