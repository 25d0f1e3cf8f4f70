#!/bin/bash
# ring-model / uniform step as a function of the light/heavy tile threshold
for hm in 4096 3072 2560; do
  echo "== PCA_BEV_HEAVY_MIN=$hm"
  PCA_BEV_HEAVY_MIN=$hm timeout -k 10 200 python bench.py --steps 50 --no-cpu-baseline > gpurun_out/hm.log 2>&1
  tail -1 gpurun_out/hm.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['ring_model']; print(d['ms_per_step'], {k: round(v['avg_us'],1) for k,v in d['roofline']['kernels'].items() if 'cells' in k}, '| ring', r['ms_per_step'], {k: round(v,1) for k,v in r['kernels_avg_us'].items() if 'cells' in k})"
done
