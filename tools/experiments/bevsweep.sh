for G in 128 256 512 1024; do
  PCA_BEV_G=$G PCA_BEV_CHUNK=4096 timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bev_$G.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/bev_$G.json')); print($G, round(d['value'],1), {k:round(v['avg_us'],1) for k,v in d['roofline']['kernels'].items()})"
done
