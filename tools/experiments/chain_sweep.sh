# owed-chain length sweep on the headline step
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for K in 1 2 3 4; do
  PCA_BEV_CHAIN=$K python bench.py --no-extras --no-cpu-baseline --steps 100 > gpurun_out/gs.json 2> gpurun_out/gs.err
  python - $K <<'PY'
import json, sys
d = json.load(open('gpurun_out/gs.json'))
k = d['roofline']['kernels']
print('chain', sys.argv[1], 'value %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']), {n: round(v['avg_us'], 1) for n, v in k.items()}, 'unit %.1f' % d['roofline']['avg_launch_us'])
PY
done
