#!/usr/bin/env python3
'''Diagnostic (a -DMPT_X_PAIRS=1 build, MIPTINA_LIB): how many lanes a step would have if a lane carried two paths.  Lanes i and
i + 32 of a wave stand for the two paths of one lane; per first step of a scheduling decision the build counts the ready lanes and
the PAIRS with at least one ready path (render_kernel.hip).  Prints both as fractions of the wave / of the 32 pairs.'''
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from ptina_amd import scenes, common  # noqa: E402
from ptina_amd.common import ctx  # noqa: E402
from helpers import setup_engine  # noqa: E402

eng = setup_engine(scenes.scene_s978(), 512, 512, mode='fast')
c = ctx()
c.set_option('batch', 32)
eng.render(1)
c.call('mpt_synchronize')
c.set_option('count', 1)
c.call('mpt_reset_counters')
eng.render(32)
c.call('mpt_synchronize')
k = c.counters()
out = {}
for name, pairs, lanes, steps in (('NODE', 'pl_local', 'pl_prim', 'pl_trips'), ('LEAF', 'pl_batches', 'pl_tidle', 'pl_taken'), ('SHADE', 'pl_batch_lanes', 'pl_sidle', 'it_shade')):
    n = max(k[steps], 1)
    out[name] = {'decisions': k[steps], 'ready_lanes_of_64': round(k[lanes] / n, 2), 'ready_pairs_of_32': round(k[pairs] / n, 2),
                 'lane_fraction': round(k[lanes] / n / 64, 3), 'pair_fraction': round(k[pairs] / n / 32, 3),
                 'gain': round((k[pairs] / 32) / (k[lanes] / 64), 3)}
print('pairs', json.dumps(out))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'pairs.json'), 'w'), indent=1)
common.reset_all()
