#!/bin/bash
# Runs on the GPU box (gpurun): SQ counters per kernel of bench.py in two --pmc passes (kernel trace only) ->
# gpurun_out/prof_<tag>/<tag>_pmc_sq_per_kernel_avg.json (per-launch averages over the steady-state launches).
#   tools/profile_sq.sh <tag> [extra bench.py flags, e.g. --scene ring]
set -e
TAG=${1:-r01_x}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    -d $OUT/sq1 -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ring --no-extras "$@" > /dev/null 2> $OUT/sq1.err
echo "sq pass 1 done"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d $OUT/sq2 -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ring --no-extras "$@" > /dev/null 2> $OUT/sq2.err
echo "sq pass 2 done"
python3 $R/tools/pmc_summary.py --sq $OUT $TAG
