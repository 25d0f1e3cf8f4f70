#!/bin/bash
# Round 5: L1 / L2 counters of the batched K1 (front + append) on 64 distinct uniform frames and on 64 ring-model frames.
# One --pmc pass per counter (more of one block per pass than the hardware holds makes rocprofv3 abort), kernel trace only.
#   tools/experiments/k1_pmc_r5.sh   -> gpurun_out/k1_pmc_r5/{summary.json, *.log}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/k1_pmc_r5
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
COUNTERS=${COUNTERS:-"TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum FETCH_SIZE WRITE_SIZE TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"}
for KIND in uniform ring; do
  ARGS="k1 64"; [ $KIND = ring ] && ARGS="k1 64 ring"
  for C in $COUNTERS; do
    if ! grep -qw "$C" $O/counters_list.txt; then echo "$KIND $C: not in rocprofv3 -L" >> $O/missing.txt; continue; fi
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C -d $O/${KIND}_$C -o run --output-format csv -- python3 $R/tools/experiments/pass_only.py $ARGS > $O/${KIND}_$C.log 2>&1 || echo "$KIND $C: rocprofv3 failed" >> $O/missing.txt
  done
  python3 $R/tools/experiments/k1_lines.py $KIND 64 > $O/lines_$KIND.json 2> $O/lines_$KIND.err
done
python3 - $O <<'PY'
import sys, csv, glob, json, os, collections
O = sys.argv[1]
out = {}
for kind in ('uniform', 'ring'):
    blk = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(O, kind + '_*'))):
        if not os.path.isdir(d):
            continue
        acc = collections.defaultdict(list)
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                name = r['Kernel_Name'].split('(')[0].replace('void ', '')
                if name.startswith('k1_'):
                    acc[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
        for (name, c), v in acc.items():
            tail = v[len(v) // 2:]
            blk[name][c] = sum(tail) / len(tail)
            blk[name]['launches_averaged'] = len(tail)
    try:
        lines = json.load(open(os.path.join(O, 'lines_%s.json' % kind)))
    except Exception as e:
        lines = {'error': repr(e)}
    out[kind] = {'kernels': blk, 'host_side_line_count': lines}
json.dump(out, open(os.path.join(O, 'summary.json'), 'w'), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
cat $O/missing.txt 2>/dev/null
