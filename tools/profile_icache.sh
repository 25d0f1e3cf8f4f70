#!/bin/bash
# Runs on the GPU box (gpurun): instruction-fetch / issue-class counters per kernel of bench.py in two --pmc passes (kernel
# trace only) -> gpurun_out/prof_<tag>/<tag>_pmc_sq_per_kernel_avg.json
#   tools/profile_icache.sh <tag>
set -e
TAG=${1:-r03_ic}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES \
    -d $OUT/sq1 -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ring --no-extras "$@" > /dev/null 2> $OUT/sq1.err
echo "ic pass 1 done"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM \
    -d $OUT/sq2 -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ring --no-extras "$@" > /dev/null 2> $OUT/sq2.err
echo "ic pass 2 done"
python3 $R/tools/pmc_summary.py --sq $OUT $TAG
python3 - $OUT/${TAG}_pmc_sq_per_kernel_avg.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))['kernels']
for k, c in d.items():
    print(k, {n: v for n, v in c.items()})
PY
