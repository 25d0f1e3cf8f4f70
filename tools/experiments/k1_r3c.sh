# round 3: which part of k1 front bounds the 64-distinct-frame batch?  rocprofv3 kernel times per experiment bit
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for EXP in 0 8 16 24 40 56; do
  O=$R/gpurun_out/k1exp_$EXP
  rm -rf $O; mkdir -p $O
  PCA_K1_EXP=$EXP rocprofv3 --kernel-trace --stats -d $O -o run --output-format csv -- python3 $R/tools/experiments/k1_batched.py 64 64 10 > $O/log.txt 2>&1
  echo "== EXP=$EXP $(grep cfg= $O/log.txt)"
  python3 - $O <<'PY'
import sys, csv, glob
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:3]:
        print('   ', r['Name'][:50], r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 2))
PY
done
