#!/bin/bash
for g in 256 384 512 768 1024; do
  PCA_BEV_G=$g PCA_BEV_CHUNK=1024 timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-ring > gpurun_out/u.log 2>&1
  tail -1 gpurun_out/u.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('G', $g, d['ms_per_step'], d['roofline']['avg_launch_us'], {k: round(v['avg_us'],1) for k,v in d['roofline']['kernels'].items()})"
done
