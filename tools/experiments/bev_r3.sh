# round 3: rasteriser after a change -- BEV parity tests, then the headline with per-kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "bev or chain or stress" 2>&1 | tail -3
python bench.py --no-extras --no-cpu-baseline --steps 100 > gpurun_out/bev_r3.json 2> gpurun_out/bev_r3.err
python - gpurun_out/bev_r3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d['roofline']['kernels']
print('value %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']), {n: round(v['avg_us'], 1) for n, v in k.items()}, 'unit %.1f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'])
PY
