for MODE in persistent legacy; do
  if [ $MODE = legacy ]; then export PCA_K1_LEGACY=1; else unset PCA_K1_LEGACY; fi
  timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/k1_$MODE.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/k1_$MODE.json')); print('$MODE', round(d['value'],1), round(d['roofline']['kernels']['kitti_project_sample_filter']['avg_us'],1), round(d['roofline']['k1_batched']['avg_launch_us'],1), round(d['roofline']['k1_batched']['frac'],3))"
done
