#!/usr/bin/env python3
'''GPU-box stress: the 40 seeded random scenes of tests/test_parity_gpu.py::test_forty_random_scenes_at_the_stated_bounds
(tests/helpers.py stress_scene / stress_compare: 1..900 triangles beside the walls, random opaque materials, lights, film sizes,
spp and batch sizes) through the strict build and the four production kernels, each production film against the strict film at the
STATED fast bounds with explicit firefly accounting; prints one line per scene and the count of scenes that fail.
usage: tools/stress_random_scenes.py [first_seed [last_seed]]'''
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import stress_compare   # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
last = int(sys.argv[2]) if len(sys.argv) > 2 else 139
bad = 0
for seed in range(first, last + 1):
    ok, flies, msgs = stress_compare(seed, log=lambda m: print(m, flush=True))
    bad += 0 if ok else 1
print('bad', bad)
sys.exit(1 if bad else 0)
