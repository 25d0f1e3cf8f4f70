#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/profile_bench.sh into
    <tag>_kernel_stats.csv   (copy of the --stats kernel table)
    <tag>_pmc_traffic.json   (FETCH_SIZE / WRITE_SIZE in KB per launch, averaged over the steady-state launches)
FETCH_SIZE is reported as collected; readers double it for streaming reads on gfx950 (MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def short(name):
    name = name.split('(')[0]
    for pre in ('void ', ):
        if name.startswith(pre):
            name = name[len(pre):]
    return name.strip()


def counters(dirname, counter):
    per = defaultdict(list)
    for path in glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') == counter:
                    per[short(row['Kernel_Name'])].append(float(row['Counter_Value']))
    return per


def all_counters(dirname):
    per = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                per[short(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
    return per


def main_sq(out, tag):
    res = {}
    for d in ('sq1', 'sq2'):
        for k, counters in all_counters(os.path.join(out, d)).items():
            if not (k.startswith('bev_') or k.startswith('k1_') or k.startswith('k2_')):
                continue
            for name, v in counters.items():
                tail = v[len(v) // 2:]
                res.setdefault(k, {})[name] = round(sum(tail) / len(tail))
    json.dump({'note': 'rocprofv3 --pmc SQ_* (two passes, --kernel-trace only), per launch, bench.py steady state',
               'kernels': res}, open(os.path.join(out, f'{tag}_pmc_sq_per_kernel_avg.json'), 'w'), indent=1)
    for k, c in res.items():
        if 'SQ_WAVE_CYCLES' in c and 'SQ_ACTIVE_INST_ANY' in c:
            print(k, 'issue share of wave cycles: %.2f' % (c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']),
                  'VALU insts per wave: %.0f' % (c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_WAVES', 1), 1)))


def main():
    if sys.argv[1] == '--sq':
        return main_sq(sys.argv[2], sys.argv[3])
    out, tag = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(out, 'stats', '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, f'{tag}_kernel_stats_bench_steps100.csv'))
    stats = glob.glob(os.path.join(out, 'stats_ring', '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, f'{tag}_kernel_stats_bench_ring_steps100.csv'))
    res = {}
    for counter, d in (('FETCH_SIZE', 'fetch'), ('WRITE_SIZE', 'write')):
        for k, v in counters(os.path.join(out, d), counter).items():
            if not (k.startswith('bev_') or k.startswith('k1_') or k.startswith('k2_') or k.startswith('dedup')):
                continue
            tail = v[len(v) // 2:]                 # second half of the launches = steady state (window full)
            res.setdefault(k, {})[counter + '_KB'] = round(sum(tail) / len(tail), 1)
            res[k]['launches_averaged'] = len(tail)
    json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), KB per launch, '
                       'bench.py steady state; FETCH_SIZE must be doubled for streaming reads on gfx950 '
                       '(MI355X_MICROARCH.md, HBM)', 'kernels': res},
              open(os.path.join(out, f'{tag}_pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
