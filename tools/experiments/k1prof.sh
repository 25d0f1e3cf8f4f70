# rocprofv3 kernel stats + HBM traffic counters of the batched K1 launch (pool 8 and pool 64)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for POOL in 8 64; do
  O=$R/gpurun_out/k1prof_$POOL
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O/stats -o run --output-format csv -- python3 $R/tools/experiments/k1_batched.py $POOL 64 20 > $O/stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o run --output-format csv -- python3 $R/tools/experiments/k1_batched.py $POOL 64 5 > $O/fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o run --output-format csv -- python3 $R/tools/experiments/k1_batched.py $POOL 64 5 > $O/write.log 2>&1
  echo "== pool $POOL"; grep "cfg=" $O/stats.log
  python3 - $O <<'PY'
import sys, csv, glob, collections
O = sys.argv[1]
for f in glob.glob(O + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:60], r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 2))
for kind in ('fetch', 'write'):
    for f in glob.glob(O + '/' + kind + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
        for k, v in acc.items():
            if 'k1_' in k:
                print(kind, k, 'KB/launch', round(sum(v[len(v)//2:]) / max(1, len(v[len(v)//2:])), 1))
PY
done
