#!/bin/bash
# K1 of a step riding in the raster's first kernel (pca_k1_defer) against K1 as its own launch: headline and ring model.
set -e
mkdir -p gpurun_out
for f in 1 0 1 0; do
  echo "== PCA_FUSE_K1=$f uniform"
  PCA_FUSE_K1=$f python bench.py --steps 200 --no-extras --no-cpu-baseline --no-ring > gpurun_out/fuse_$f.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/fuse_$f.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], {k:round(v['avg_us'],1) for k,v in r['kernels'].items()})
PY
  echo "== PCA_FUSE_K1=$f ring"
  PCA_FUSE_K1=$f python bench.py --steps 200 --no-extras --no-cpu-baseline --no-ring --scene ring > gpurun_out/fuse_ring_$f.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/fuse_ring_$f.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], {k:round(v['avg_us'],1) for k,v in r['kernels'].items()})
PY
done
