#!/bin/bash
# Level 1's workgroup count (PCA_BEV_G, the cap on the number of pieces a window is cut into) on the headline and the ring model.
set -e
mkdir -p gpurun_out
for scene in uniform ring; do
for g in ${GS:-512 640 768 896 980 512}; do
  PCA_BEV_G=$g python bench.py --steps 200 --no-extras --no-cpu-baseline --no-ring --scene $scene > gpurun_out/g_$g.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/g_$g.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$scene', 'G=$g', round(d['value'],1), round(d['ms_per_step']*1e3,2), round(r['avg_launch_us'],2), {k:round(v['avg_us'],1) for k,v in r['kernels'].items()})
PY
done
done
