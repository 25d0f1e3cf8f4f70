#!/usr/bin/env python3
"""Condenses the outputs of tools/profile_round.sh into the files kept under profiles/:
    <tag>_kernel_stats_bench_steps100.csv, <tag>_kernel_stats_bench_ring_steps100.csv   rocprofv3 --stats kernel tables
    <tag>_kernel_stats_k1_batched_{pool8,distinct64,ring64}.csv, <tag>_kernel_stats_nuscenes_scene.csv
    <tag>_pmc_traffic.json, <tag>_pmc_traffic_{ring,config4,nusc}.json   FETCH_SIZE / WRITE_SIZE in KB per launch
    <tag>_k1_batched_summary.txt / .json       batched K1: kernel times + traffic of both pools, physical bytes and rates
FETCH_SIZE is stored as collected; readers double it for wide streaming reads on gfx950 (MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def short(name):
    name = name.split('(')[0]
    return name[5:].strip() if name.startswith('void ') else name.strip()


def counters(dirname, counter):
    per = defaultdict(list)
    for path in glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') == counter:
                    per[short(row['Kernel_Name'])].append(float(row['Counter_Value']))
    return per


def keep(k):
    return k.startswith(('bev_', 'k1_', 'k1n_', 'k0n_', 'k2_', 'k3_', 'dedup'))


def traffic(out, suffix):
    res = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        for k, v in counters(os.path.join(out, f'{counter}_{suffix}'), counter).items():
            if not keep(k):
                continue
            tail = v[len(v) // 2:]                 # second half of the launches = steady state
            res.setdefault(k, {})[counter + '_KB'] = round(sum(tail) / len(tail), 1)
            res[k]['launches_averaged'] = len(tail)
    return res


def stats(out, sub):
    rows = {}
    for f in glob.glob(os.path.join(out, sub, '**', '*kernel_stats.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[short(r['Name'])] = {'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) / 1e3}
    return rows


def copy_stats(out, sub, dst):
    f = glob.glob(os.path.join(out, sub, '**', '*kernel_stats.csv'), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(out, dst))


def main():
    out, tag = sys.argv[1], sys.argv[2]
    note = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), KB per launch, averaged over '
            'the second half of the launches; FETCH_SIZE must be doubled for streaming reads on gfx950 (MI355X_MICROARCH.md)')
    copy_stats(out, 'stats', f'{tag}_kernel_stats_bench_steps100.csv')
    copy_stats(out, 'stats_ring', f'{tag}_kernel_stats_bench_ring_steps100.csv')
    copy_stats(out, 'stats_k1_8', f'{tag}_kernel_stats_k1_batched_pool8.csv')
    copy_stats(out, 'stats_k1_64', f'{tag}_kernel_stats_k1_batched_distinct64.csv')
    copy_stats(out, 'stats_k1_ring', f'{tag}_kernel_stats_k1_batched_ring64.csv')
    copy_stats(out, 'stats_nusc', f'{tag}_kernel_stats_nuscenes_scene.csv')
    copy_stats(out, 'stats_nusc_ring', f'{tag}_kernel_stats_nuscenes_scene_sweep_order.csv')
    for suffix, name in (('head', ''), ('ring', '_ring'), ('config4', '_config4'), ('nusc', '_nusc'), ('nusc_ring', '_nusc_ring')):
        json.dump({'note': note, 'kernels': traffic(out, suffix)}, open(os.path.join(out, f'{tag}_pmc_traffic{name}.json'), 'w'), indent=1)
    # batched K1
    summ = {'note': note + '; FETCH_SIZE / WRITE_SIZE are KiB (x 1024 bytes) everywhere: physical_MB = (2 FETCH_KB + WRITE_KB) x 1024 / 1e6 per call, as bench.py:pmc_traffic computes it; GBps_physical = physical bytes / kernel time'}
    lines = []
    for pool, sub in ((8, 'stats_k1_8'), (64, 'stats_k1_64')):
        st, tr = stats(out, sub), traffic(out, f'k1_{pool}')
        blk = {}
        for k in st:
            if not k.startswith('k1_'):
                continue
            t = tr.get(k, {})
            phys = (2.0 * t.get('FETCH_SIZE_KB', 0.0) + t.get('WRITE_SIZE_KB', 0.0)) * 1024.0
            blk[k] = {'avg_us': round(st[k]['avg_us'], 2), 'calls': st[k]['calls'], **t, 'physical_MB': round(phys / 1e6, 1),
                      'GBps_physical': round(phys / st[k]['avg_us'] / 1e3, 0) if st[k]['avg_us'] else None}
        total_us = sum(v['avg_us'] for v in blk.values())
        total_phys = sum(v['physical_MB'] for v in blk.values())
        try:
            bench = json.load(open(os.path.join(out, f'k1_{pool}.json')))
        except Exception:
            bench = {}
        summ[f'pool{pool}'] = {'kernels': blk, 'sum_us': round(total_us, 2), 'physical_MB': round(total_phys, 1),
                               'GBps_physical': round(total_phys * 1e3 / total_us, 0) if total_us else None,
                               'frac_physical_of_8TBps': round(total_phys * 1e3 / total_us / 8000.0, 3) if total_us else None,
                               'bench': {k: bench.get(k) for k in ('us_per_call_wall_back_to_back', 'us_per_call_hip_events',
                                                                   'alg_bytes', 'frac', 'kept')}}
        lines.append(f'== pool {pool} ({"8 distinct frames repeated" if pool == 8 else "64 distinct frames"})')
        for k, v in blk.items():
            lines.append(f'  {k[:44]:44s} avg {v["avg_us"]:7.2f} us  fetch {v.get("FETCH_SIZE_KB", 0):10.1f} KB  write '
                         f'{v.get("WRITE_SIZE_KB", 0):10.1f} KB  physical {v["physical_MB"]:7.1f} MB = {v["GBps_physical"]} GB/s')
        s = summ[f'pool{pool}']
        lines.append(f'  sum {s["sum_us"]} us, physical {s["physical_MB"]} MB = {s["GBps_physical"]} GB/s = '
                     f'{s["frac_physical_of_8TBps"]} of 8 TB/s;  bench: {s["bench"]}')
    json.dump(summ, open(os.path.join(out, f'{tag}_k1_batched_summary.json'), 'w'), indent=1)
    open(os.path.join(out, f'{tag}_k1_batched_summary.txt'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
