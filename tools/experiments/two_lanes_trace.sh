# round 5: kernel trace of two lanes (threads) on one GPU -> how much do the two streams' kernels overlap?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/two_lanes
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in 1 2; do
  rocprofv3 --kernel-trace -d $O/t$L -o run --output-format csv -- python3 $R/tools/experiments/two_lanes_trace.py 40 $L > $O/run$L.log 2>&1
  tail -1 $O/run$L.log
done
python3 - $O <<'PY'
import sys, csv, glob, collections
O = sys.argv[1]
for L in (1, 2):
    rows = []
    for f in glob.glob('%s/t%d/**/*kernel_trace.csv' % (O, L), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows = [r for r in rows if r['Kernel_Name'].startswith(('bev_', 'void bev_', 'void k1_', 'k1_'))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # the steady-state run = the last 40 steps of every queue: take the last 3 * 40 * L kernels
    rows = rows[-3 * 40 * L:]
    qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else ('Stream_Id' if 'Stream_Id' in rows[0] else None)
    iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get(qkey, '0') if qkey else '0', r['Kernel_Name'].split('(')[0]) for r in rows]
    span = max(e for _, e, _, _ in iv) - min(s for s, _, _, _ in iv)
    busy_sum = sum(e - s for s, e, _, _ in iv)
    # union of the intervals
    union, cur_s, cur_e = 0, None, None
    for s, e, _, _ in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    per_kernel = collections.defaultdict(list)
    for s, e, q, n in iv:
        per_kernel[n].append((e - s) / 1e3)
    print('lanes %d: %d kernels on queues %s; span %.1f us, sum of kernel durations %.1f us, union (GPU busy) %.1f us, overlapped %.1f us = %.1f %% of the sum'
          % (L, len(iv), sorted(set(q for _, _, q, _ in iv)), span / 1e3, busy_sum / 1e3, union / 1e3, (busy_sum - union) / 1e3, 100.0 * (busy_sum - union) / busy_sum))
    for n, v in per_kernel.items():
        print('   %-44s n %4d  mean %.1f us' % (n[:44], len(v), sum(v) / len(v)))
PY
