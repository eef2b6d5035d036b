#!/usr/bin/env python3
'''copy the rocprofv3 summaries of the last gpurun into profiles/ (tracked), named per round'''
import csv
import collections
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    fs = sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', pattern)), key=os.path.getmtime)
    return fs[-1] if fs else None


f = newest('prof/*/*kernel_stats.csv')
if f:
    shutil.copy(f, os.path.join(dst, f'{tag}_kernel_stats.csv'))
    print('kernel stats ->', f'{tag}_kernel_stats.csv')

f = newest('prof/*/*kernel_trace.csv')
if f:
    rows = [r for r in csv.DictReader(open(f)) if 'render_kernel' in r['Kernel_Name']]
    with open(os.path.join(dst, f'{tag}_render_dispatches.csv'), 'w') as fh:
        fh.write('dispatch,kernel,duration_us,start_ns,end_ns,grid,workgroup,lds_bytes,vgprs,sgprs,scratch\n')
        durs = []
        for r in rows:
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            durs.append(dur)
            fh.write(f"{r['Dispatch_Id']},\"{r['Kernel_Name']}\",{dur:.1f},{r['Start_Timestamp']},{r['End_Timestamp']},"
                     f"{r['Grid_Size_X']},{r['Workgroup_Size_X']},{r['LDS_Block_Size']},{r['VGPR_Count']},{r['SGPR_Count']},{r['Scratch_Size']}\n")
    med = sorted(durs)[len(durs) // 2] if durs else 0
    full = [d for d in durs if 0.7 * med < d < 1.3 * med]
    print('render dispatches ->', f'{tag}_render_dispatches.csv', '| launches within 30% of the median:', len(full),
          'avg ms:', round(sum(full) / max(len(full), 1) / 1e3, 4))

summary = {}
for d in ('pmc1', 'pmc2', 'pmc3', 'pmc4', 'pmc5'):
    f = newest(f'{d}/*/*counter_collection.csv')
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        if 'render_kernel' not in k:
            continue
        for cn, vals in v.items():
            vals = sorted(vals)
            summary.setdefault(k, {})[cn] = {'median': vals[len(vals) // 2], 'n': len(vals)}
if summary:
    with open(os.path.join(dst, f'{tag}_pmc_summary.json'), 'w') as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)
    print('pmc summary ->', f'{tag}_pmc_summary.json')
# big scenes (`tools/gpu_round.sh pmcbig1..4` = `run_configs.py C4 C5` under --pmc): the production kernel
# of the big scenes (4-wide gather, render_kernel_wide<false, true>: 8-bit child boxes) is dispatched as [C4 1-spp, C4 x6 (32 frames of 1024^2), C5 1-spp, C5 x3 (16 frames of 1024^2)]
big = {}
for d in ('pmcbig1', 'pmcbig2', 'pmcbig3', 'pmcbig4', 'pmcbig5'):
    f = newest(f'{d}/*/*counter_collection.csv')
    if not f:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if 'render_kernel_wide<false' in r['Kernel_Name'] or 'render_kernel_fast<32, false>' in r['Kernel_Name']:
            per[r['Counter_Name']][int(r['Dispatch_Id'])] += float(r['Counter_Value'])
    for cn, byd in per.items():
        vals = [byd[k] for k in sorted(byd)]
        if len(vals) != 11:
            continue
        for name, sl in (('C4 99k tris, 32 frames x 1024x1024 per launch', vals[1:7]),
                         ('C5 1M tris, 16 frames x 1024x1024 per launch', vals[8:11])):
            sl = sorted(sl)
            big.setdefault(name, {})[cn] = sl[len(sl) // 2]
if big:
    with open(os.path.join(dst, f'{tag}_pmc_big_summary.json'), 'w') as fh:
        json.dump(big, fh, indent=1, sort_keys=True)
    print('big-scene pmc summary ->', f'{tag}_pmc_big_summary.json')
# the same scenes through the 8-wide octant-ordered kernel (option wide8: `tools/gpu_round.sh pmcoct1 pmcoct3 pmcoct4`), same dispatch pattern
octs = {}
for d in ('pmcoct1', 'pmcoct3', 'pmcoct4'):
    f = newest(f'{d}/*/*counter_collection.csv')
    if not f:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if 'render_kernel_oct<false' in r['Kernel_Name']:
            per[r['Counter_Name']][int(r['Dispatch_Id'])] += float(r['Counter_Value'])
    for cn, byd in per.items():
        vals = [byd[k] for k in sorted(byd)]
        if len(vals) != 11:
            continue
        for name, sl in (('C4 99k tris, 32 frames x 1024x1024 per launch', vals[1:7]),
                         ('C5 1M tris, 16 frames x 1024x1024 per launch', vals[8:11])):
            sl = sorted(sl)
            octs.setdefault(name, {})[cn] = sl[len(sl) // 2]
if octs:
    with open(os.path.join(dst, f'{tag}_pmc_oct_summary.json'), 'w') as fh:
        json.dump(octs, fh, indent=1, sort_keys=True)
    print('8-wide kernel pmc summary ->', f'{tag}_pmc_oct_summary.json')
for name in ('bench.log',):
    src = os.path.join(ROOT, 'gpurun_out', name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(dst, f'{tag}_{name}'))
# one tracked file per diagnostic of tools/gpu_diag.py (shares_sync, stamps_<scene>, timeline, fixedcost, ...): nothing overwrites another
for src in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'diag_*.json'))):
    key = os.path.basename(src)[len('diag_'):-len('.json')]
    if key.startswith(('parity_', 'timing_', 'sched_')):
        continue
    shutil.copy(src, os.path.join(dst, f'{tag}_{key}.json'))
    print('diagnostic ->', f'{tag}_{key}.json')

# tools/run_configs.py's table of every single-GPU configuration (a run restricted to some of them writes configs_subset.json)
f = os.path.join(ROOT, 'gpurun_out', 'configs.json')
if os.path.exists(f):
    try:
        if len(json.load(open(f))) >= 5:
            shutil.copy(f, os.path.join(dst, f'{tag}_configs.json'))
            print('configs ->', f'{tag}_configs.json')
    except Exception as e:
        print('configs.json not copied:', e)

# (VERDICT r04 asked for the share measurements under this name)
f = os.path.join(dst, f'{tag}_shares_sync.json')
if os.path.exists(f):
    shutil.copy(f, os.path.join(dst, f'{tag}_shares.json'))

