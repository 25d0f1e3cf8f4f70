#!/bin/bash
# Runs on the GPU box (gpurun): the round's profile set -- rocprofv3 kernel statistics of the headline (uniform and ring
# scene) and FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, kernel trace only) of the headline, the ring scene, the
# giant-window config, the NuScenes scene and the batched K1 (pool 8 and 64 distinct frames).  Summaries into
# gpurun_out/prof_<tag>/ (copy what is to be kept into profiles/).
#   tools/profile_round.sh <tag>
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --warmup 5 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- $B --steps 100 --no-ring > $OUT/bench.json 2> $OUT/stats.err; echo "stats done"
rocprofv3 --kernel-trace --stats -d $OUT/stats_ring -o run --output-format csv -- $B --steps 100 --scene ring > $OUT/bench_ring.json 2> $OUT/stats_ring.err; echo "ring stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${C}_head -o run --output-format csv -- $B --steps 20 --no-ring > /dev/null 2> $OUT/${C}_head.err
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${C}_ring -o run --output-format csv -- $B --steps 20 --scene ring > /dev/null 2> $OUT/${C}_ring.err
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${C}_config4 -o run --output-format csv -- python3 $R/tools/experiments/pass_only.py config4 > /dev/null 2> $OUT/${C}_config4.err
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${C}_nusc -o run --output-format csv -- python3 $R/tools/experiments/pass_only.py nuscenes_scene > /dev/null 2> $OUT/${C}_nusc.err
  rocprofv3 --kernel-trace --pmc $C -d $OUT/${C}_nusc_ring -o run --output-format csv -- python3 $R/tools/experiments/pass_only.py nuscenes_scene_sweep > /dev/null 2> $OUT/${C}_nusc_ring.err
  for POOL in 8 64; do
    rocprofv3 --kernel-trace --pmc $C -d $OUT/${C}_k1_$POOL -o run --output-format csv -- python3 $R/tools/experiments/pass_only.py k1 $POOL > /dev/null 2> $OUT/${C}_k1_$POOL.err
  done
  echo "$C done"
done
for POOL in 8 64; do
  rocprofv3 --kernel-trace --stats -d $OUT/stats_k1_$POOL -o run --output-format csv -- python3 $R/tools/experiments/pass_only.py k1 $POOL > $OUT/k1_$POOL.json 2> $OUT/stats_k1_$POOL.err
done
rocprofv3 --kernel-trace --stats -d $OUT/stats_k1_ring -o run --output-format csv -- python3 $R/tools/experiments/pass_only.py k1 64 ring > $OUT/k1_ring.json 2> $OUT/stats_k1_ring.err
rocprofv3 --kernel-trace --stats -d $OUT/stats_nusc -o run --output-format csv -- python3 $R/tools/experiments/pass_only.py nuscenes_scene > $OUT/nusc.json 2> $OUT/stats_nusc.err
rocprofv3 --kernel-trace --stats -d $OUT/stats_nusc_ring -o run --output-format csv -- python3 $R/tools/experiments/pass_only.py nuscenes_scene_sweep > $OUT/nusc_ring.json 2> $OUT/stats_nusc_ring.err
echo "k1 / nusc stats done"
python3 $R/tools/pmc_round.py $OUT $TAG
