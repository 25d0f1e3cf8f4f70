#!/bin/bash
# Level 1's register path at 12 (shipped) against 16 points per thread (libpca_hip_rp16.so built with -DBIN_REG_P=16): headline, ring.
set -e
cp pc-accumulation-lib_amd/pca_amd/libpca_hip.so /tmp/lib12.so; cp pc-accumulation-lib_amd/pca_amd/libpca_hip_rp16.so /tmp/lib16.so
for v in 12 16 12 16; do
  cp /tmp/lib$v.so pc-accumulation-lib_amd/pca_amd/libpca_hip.so
  for scene in uniform ring; do
  python bench.py --steps 200 --no-extras --no-cpu-baseline --no-ring --scene $scene 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$scene REG_P=$v', round(d['value'],1), round(d['ms_per_step']*1e3,2), round(r['avg_launch_us'],2), {k:round(x['avg_us'],1) for k,x in r['kernels'].items()})"
  done
done
cp /tmp/lib12.so pc-accumulation-lib_amd/pca_amd/libpca_hip.so
