R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/nusc_prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o run --output-format csv -- python3 $R/tools/experiments/nusc_scene.py > $O/log.txt 2>&1
python3 - $O <<'PY'
import sys, csv, glob
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:22]:
        print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), 'avg_us', round(float(r['AverageNs']) / 1e3, 1), 'total_ms', round(float(r['TotalDurationNs']) / 1e6, 2))
PY
