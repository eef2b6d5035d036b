#!/usr/bin/env python3
'''writes tests/golden/oracle_s34_24x24x4.npz: the CPU oracle's film for S34, 24x24, after
PTina's benchmark sequence (one warm-up frame + clear, then 4 frames).  A regression pin for
the oracle and a small fixture the GPU path is compared with.  NOT output of real PTina
(Taichi is not installable here): parity with the reference stays unpinned.'''
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import oracle                      # noqa: E402
from ptina_amd import scenes       # noqa: E402
from helpers import setup_oracle   # noqa: E402

o = setup_oracle(oracle, scenes.scene_s34(), 24, 24)
o.render(1)
o.clear()
o.render(4)
out = os.path.join(ROOT, 'tests', 'golden', 'oracle_s34_24x24x4.npz')
np.savez_compressed(out, film=o.get_film_raw(), time=o.sobol_state()[0], image=o.get_image())
print('wrote', out)
