# BASELINE configs[3] at one tenth and the ring model with the library as built, after the kernel parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "bev or chain or stress or giant or heavy or config4" 2>&1 | tail -2
python - <<PY
import sys, os, json, builtins
sys.path.insert(0, os.getcwd())
import bench
rp = builtins.print
builtins.print = lambda *a, **k: None
r = bench.config4_pass()
g = bench.ring_model_pass(20)
builtins.print = rp
print('config4 ms/step %.3f' % r['ms_per_step'], {k: round(v, 1) for k, v in r['kernels_avg_us'].items()}, 'frac %.3f' % r['roofline_bev_unit']['frac'])
print('ring ms/step %.4f' % g['ms_per_step'], {k: round(v, 1) for k, v in g['kernels_avg_us'].items()})
PY
