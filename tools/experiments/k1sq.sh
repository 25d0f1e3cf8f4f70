# SQ counters of the batched K1 launch (two --pmc passes, kernel trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
POOL=${1:-8}
O=$R/gpurun_out/k1sq_$POOL
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    -d $O/sq1 -o run --output-format csv -- python3 $R/tools/experiments/k1_batched.py $POOL 64 5 > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d $O/sq2 -o run --output-format csv -- python3 $R/tools/experiments/k1_batched.py $POOL 64 5 > $O/sq2.log 2>&1
python3 - $O <<'PY'
import sys, csv, glob, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/sq*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k1_' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    m = {n: sum(v[len(v)//2:]) / len(v[len(v)//2:]) for n, v in c.items()}
    print(k)
    for n in sorted(m):
        print('   %-24s %14.0f   per wave %10.1f' % (n, m[n], m[n] / max(m.get('SQ_WAVES', 1), 1)))
PY
