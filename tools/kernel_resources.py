#!/usr/bin/env python3
'''Prints VGPRs / AGPRs / SGPRs / scratch / occupancy / LDS of every kernel of the production render object
(the compiler's own -Rpass-analysis=kernel-resource-usage remarks).
usage: tools/kernel_resources.py [extra hipcc flags]'''
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'ptina_amd', 'csrc')
cmd = ['/opt/rocm/bin/hipcc', '-c', '-O3', '-fPIC', '-std=c++17', '--offload-arch=gfx950', '-DMPT_STRICT=0',
       '-ffp-contract=fast', '-fno-slp-vectorize', '-Rpass-analysis=kernel-resource-usage', *sys.argv[1:],
       'render_kernel.hip', '-o', '/tmp/_kres.o']
out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r'remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|'
                  r'LDS Size \[bytes/block\]):\s*(\S+)', line)
    if not m:
        if 'error' in line:
            print(line)
        continue
    k, v = m.groups()
    if k == 'Function Name':
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k.split()[0]] = v
for name, r in rows.items():
    import shutil
    filt = shutil.which('c++filt')
    nm = (subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else '') or name
    nm = nm.replace('(MptRenderParams)', '').replace('void ', '')
    print('%-44s VGPR %4s AGPR %3s SGPR %4s scratch %4s occ %2s lds %s' % (
        nm[:44], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('TotalSGPRs', '?'), r.get('ScratchSize', '?'),
        r.get('Occupancy', '?'), r.get('LDS', '?')))
