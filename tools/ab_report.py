#!/usr/bin/env python3
'''one line per gpurun_out/ab_*.log / bench*.log: the numbers an A/B of library builds is read by'''
import glob
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'ab_*.log')) + glob.glob(os.path.join(ROOT, 'gpurun_out', 'bench_quick.log')))
for f in names:
    if not os.path.exists(f):
        f = os.path.join(ROOT, 'gpurun_out', f + '.log')
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d['counters_per_sample']
        print(f"{os.path.basename(f):28s} value {d['value']:8.1f}  ms/step {d['ms_per_step']:.4f}  resolve-only {d['value_resolve_only']:8.1f}  "
              f"kernel {d['roofline']['avg_kernel_ms']:.4f} ms  per 64 samples: node {c['it_node']*64:.1f} leaf {c['it_leaf']*64:.1f} "
              f"shade {c['it_shade']*64:.2f} new {c['it_new']*64:.2f}  lanes/node-step {c['n_node']/max(c['it_node'],1e-9):.1f} "
              f"lanes/shade {c['n_shade']/max(c['it_shade'],1e-9):.1f}")
    except Exception as e:
        print(os.path.basename(f), 'ERR', e, open(f).read()[-300:] if os.path.exists(f) else '')
