# level 1 walking the window forwards and backwards on alternate calls (what it read last is what it reads first): env A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do for A in 0 1; do
  PCA_BEV_ALTERNATE=$A python bench.py --no-extras --no-cpu-baseline --steps 100 > gpurun_out/gs.json 2> gpurun_out/gs.err
  python - $A <<'PY'
import json, sys
d = json.load(open('gpurun_out/gs.json'))
k = d['roofline']['kernels']
print('alternate', sys.argv[1], 'value %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']), {n: round(v['avg_us'], 1) for n, v in k.items()}, 'unit %.1f' % d['roofline']['avg_launch_us'])
PY
done; done
