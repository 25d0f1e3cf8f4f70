R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for SUB in 64 32 16; do
  O=$R/gpurun_out/k1prof_sub$SUB
  rm -rf $O; mkdir -p $O
  PCA_K1_SUB=$SUB rocprofv3 --kernel-trace -d $O/stats -o run --output-format csv -- python3 $R/tools/experiments/k1_batched.py 8 64 3 > $O/stats.log 2>&1
  echo "== sub $SUB"; grep "cfg=" $O/stats.log
  python3 - $O <<'PY'
import sys, csv, glob
O = sys.argv[1]
for f in glob.glob(O + '/stats/**/*kernel_trace.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    k1 = [r for r in rows if 'k1_' in r['Kernel_Name']]
    last = k1[-12:]
    t0 = int(last[0]['Start_Timestamp'])
    for r in last:
        print(r['Kernel_Name'][:30], 'grid', r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), 'start', (int(r['Start_Timestamp']) - t0) / 1e3, 'dur_us', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
PY
done
