#!/bin/bash
# Frames that cannot reach the view left out of level 1 (PCA_BEV_CULL; host-side hull) against binning the whole window.
set -e
mkdir -p gpurun_out
for f in 1 0 1 0; do
  for scene in uniform ring; do
  PCA_BEV_CULL=$f python bench.py --steps 200 --no-extras --no-cpu-baseline --no-ring --scene $scene > gpurun_out/cull_$f.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/cull_$f.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$scene', 'cull=$f', round(d['value'],1), round(d['ms_per_step']*1e3,2), round(r['avg_launch_us'],2), {k:round(v['avg_us'],1) for k,v in r['kernels'].items()})
PY
  done
done
