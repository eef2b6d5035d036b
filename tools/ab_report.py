#!/usr/bin/env python3
'''one line per gpurun_out/ab_*.log (tools/gpu_round.sh ablibs): value, step, kernel, resolve-only'''
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', (sys.argv[1] if len(sys.argv) > 1 else 'ab_') + '*.log'))):
    for l in open(f):
        if l.startswith('{'):
            d = json.loads(l)
            print('%-28s value %8.1f  step %.4f ms  kernel %.4f ms  step-kernel %.3f  resolve-only %8.1f' % (
                os.path.basename(f)[:-4], d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'],
                d['ms_per_step'] - d['roofline']['avg_kernel_ms'], d['value_resolve_only']))
