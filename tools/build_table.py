#!/usr/bin/env python3
'''The per-kernel table of mpt_build_tree: launches, device time, modelled bytes, GB/s, fraction of the
8 TB/s HBM peak -- from a `rocprofv3 --kernel-trace --stats` summary of tools/build_profile.py and its JSON.
usage: tools/build_table.py KERNEL_STATS.csv BUILD_PROFILE.json [OUT.json]     (prints a markdown table)

The bytes are a MODEL of what each kernel has to move (inputs read once, outputs written once, per triangle / node / streamed
position), not counters: n = triangles, nw = 4-wide nodes, elems / part_words = positions streamed / chunk-bin words written
over the binned levels of the SAH pass (the library counts them: options sah_kelems, sah_part_kwords).'''
import csv
import json
import sys

PEAK = 8000.0       # GB/s, /opt/skills/guides/MI355X_MICROARCH.md


def model(n, nw, st):
    elems, pw = st.get('sah_kelems', 0) * 1000, st.get('sah_part_kwords', 0) * 1000
    return [
        # (row, match substrings, bytes)
        ('centroid_bounds_kernel', ['centroid_bounds_kernel'], n * (96 + 12)),
        ('morton_keys_kernel', ['morton_keys_kernel'], n * (12 + 8)),
        ('rocPRIM sort of the Morton keys', ['radix_sort', 'merge_sort', 'radix_merge'], n * 16 * 11),
        ('hierarchy_kernel', ['hierarchy_kernel'], n * 40),
        ('fit_boxes_kernel', ['fit_boxes_kernel'], n * 190),
        ('pack_nodes_kernel', ['pack_nodes_kernel'], n * 248),
        ('pack_tris_kernel', ['pack_tris_kernel'], n * 232),
        ('derive_tfast_kernel', ['derive_tfast_kernel'], n * 112),
        ('sb_prims_kernel', ['sb_prims_kernel'], n * 132),
        ('sb_bin_kernel', ['sb_bin_kernel'], elems * 32 + pw * 4),
        ('sb_reduce_kernel', ['sb_reduce_kernel'], pw * 4),
        ('sb_choose_kernel', ['sb_choose_kernel'], pw * 4),
        ('sb_plan_kernel', ['sb_plan_kernel'], st.get('sah_segments', 0) * 200),
        ('sb_scatter_kernel', ['sb_scatter_kernel'], elems * 64),
        ('sb_finish_kernel', ['sb_finish_kernel'], n * 96),
        # the round-5 pass (profiles/r06_build_kernel_stats_*_before.csv): every level streamed all n positions
        ('sb_bounds_kernel (r05)', ['sb_bounds_kernel'], None),
        ('sb_small_kernel (r05)', ['sb_small_kernel'], None),
        ('sb_choose_wave_kernel (r05)', ['sb_choose_wave_kernel'], None),
        ('sb_reset / pred / newseg / pack (r05)', ['sb_reset_kernel', 'sb_pred_kernel', 'sb_newseg_kernel', 'sb_pack_kernel'], None),
        ('wb_area_kernel', ['wb_area_kernel'], n * 64),
        ('wb_expand_kernel', ['wb_expand_kernel'], nw * 384),
        ('wb_totals_kernel', ['wb_totals_kernel'], nw * 4 // 256 * 8 + 64),
        ('wb_link_kernel', ['wb_link_kernel'], nw * 48),
        ('rocPRIM scans', ['scan'], None),
        ('copies / fills', ['__amd_rocclr'], None),
    ]


def main():
    stats, prof = sys.argv[1], json.load(open(sys.argv[2]))
    runs = prof['runs']
    builds = len(runs) + 1                     # + the untimed first build of tools/build_profile.py
    n, nw = prof['ntri'], runs[-1].get('wide_nodes', 0)
    rows = list(csv.DictReader(open(stats)))
    used, out = set(), []
    for name, keys, nbytes in model(n, nw, runs[-1]):
        calls, ns = 0, 0
        for i, r in enumerate(rows):
            if i in used or not any(k in r['Name'] for k in keys):
                continue
            used.add(i)
            calls += int(r['Calls'])
            ns += int(r['TotalDurationNs'])
        if calls == 0:
            continue
        us = ns / builds / 1e3
        row = {'kernel': name, 'launches_per_build': round(calls / builds, 1), 'us_per_build': round(us, 1)}
        if nbytes:
            row['model_bytes'] = int(nbytes)
            row['GBs'] = round(nbytes / (us * 1e-6) / 1e9, 1)
            row['frac_of_hbm_peak'] = round(row['GBs'] / PEAK, 4)
        out.append(row)
    rest = sum(int(r['TotalDurationNs']) for i, r in enumerate(rows) if i not in used) / builds / 1e3
    if rest > 0:
        out.append({'kernel': 'other (context set-up of the first build)', 'launches_per_build': None, 'us_per_build': round(rest, 1)})
    total = sum(r['us_per_build'] for r in out)
    walls = [r['wall_ms'] for r in runs]
    res = {'scene': prof['scene'], 'ntri': n, 'wide_nodes': nw, 'builds_in_trace': builds, 'wall_ms_per_build': min(walls),
           'device_us_per_build': round(total, 1), 'phases_ms': {k: v for k, v in runs[-1].items() if k.endswith('_ms')},
           'sah': {k: v for k, v in runs[-1].items() if k.startswith('sah_')}, 'kernels': out}
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], 'w'), indent=1)
    print(f"{prof['scene']}: {n} triangles, wall {min(walls):.2f} ms per build, kernels {total / 1e3:.2f} ms")
    print('| kernel | launches | us | model MB | GB/s | of 8 TB/s |')
    print('|---|---|---|---|---|---|')
    for r in out:
        mb = f"{r['model_bytes'] / 1e6:.1f}" if 'model_bytes' in r else '-'
        print(f"| {r['kernel']} | {r['launches_per_build']} | {r['us_per_build']} | {mb} | {r.get('GBs', '-')} | {r.get('frac_of_hbm_peak', '-')} |")


if __name__ == '__main__':
    main()
