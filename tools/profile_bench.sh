#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel statistics + HBM traffic counters of bench.py, summaries into
# gpurun_out/ (copy what is to be kept into profiles/).  PMC passes are separate runs with --kernel-trace only.
#   tools/profile_bench.sh <tag>
set -e
TAG=${1:-r01_x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-ring --no-extras > $OUT/bench.json 2> $OUT/stats.err
echo "stats done"
rocprofv3 --kernel-trace --stats -d $OUT/stats_ring -o run --output-format csv -- python3 $R/bench.py --scene ring --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_ring.json 2> $OUT/stats_ring.err
echo "ring stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ring --no-extras > /dev/null 2> $OUT/fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o run --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ring --no-extras > /dev/null 2> $OUT/write.err
echo "write done"
python3 $R/tools/pmc_summary.py $OUT $TAG
