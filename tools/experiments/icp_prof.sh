# kernel trace of the device ICP (per-launch durations in launch order)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/icp_prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/t -o run --output-format csv -- python3 $R/tools/experiments/icp_time.py > $O/run.log 2>&1
tail -8 $O/run.log
python3 - $O <<'PY'
import sys, csv, glob
rows = []
for f in glob.glob(sys.argv[1] + '/t/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if r['Kernel_Name'].startswith(('icp_', 'void icp_'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // 4                      # 4 registrations
last = rows[-n:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    print('%-40s start %9.1f us  dur %8.1f us' % (r['Kernel_Name'][:40], (int(r['Start_Timestamp']) - t0) / 1e3,
                                                  (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
