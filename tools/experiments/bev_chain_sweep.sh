#!/bin/bash
# Owed re-transforms before the raster writes the coordinates back (PCA_BEV_CHAIN): headline, ring model, config 4.
set -e
mkdir -p gpurun_out
for c in 4 3 2 1 4; do
  for scene in uniform ring; do
  PCA_BEV_CHAIN=$c python bench.py --steps 200 --no-extras --no-cpu-baseline --no-ring --scene $scene > gpurun_out/chain_$c.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/chain_$c.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$scene', 'chain=$c', round(d['value'],1), round(d['ms_per_step']*1e3,2), round(r['avg_launch_us'],2), {k:round(v['avg_us'],1) for k,v in r['kernels'].items()})
PY
  done
  PCA_BEV_CHAIN=$c python tools/experiments/pass_only.py config4 > gpurun_out/chain4_$c.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/chain4_$c.json').read().strip().splitlines()[-1])
print('config4', 'chain=$c', round(d['ms_per_step'],3), {k:round(v,1) for k,v in d['kernels_avg_us'].items()})
PY
done
