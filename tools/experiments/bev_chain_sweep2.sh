#!/bin/bash
# Owed re-transforms before the raster writes the coordinates back (PCA_BEV_CHAIN) with up to 8 allowed: headline, ring model.
set -e
for c in 4 6 8 5 4 8; do
  for scene in uniform ring; do
  PCA_BEV_CHAIN=$c python bench.py --steps 240 --no-extras --no-cpu-baseline --no-ring --scene $scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$scene chain=$c', round(d['value'],1), round(d['ms_per_step']*1e3,2), round(r['avg_launch_us'],2), {k:round(x['avg_us'],1) for k,x in r['kernels'].items()})"
  done
done
