# round 3: what bounds bev_tile_bin?  PCA_BEV_DBG bits: 64 no pass-B stores, 128 no colour / intensity gathers, 256 no owed chain, 512 nothing kept
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for D in 0 64 128 192 256 512 768; do
  PCA_BEV_DBG=$D python bench.py --no-extras --no-cpu-baseline --steps 100 > gpurun_out/binexp_$D.json 2> gpurun_out/binexp.err
  python - gpurun_out/binexp_$D.json $D <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d['roofline']['kernels']
print('DBG', sys.argv[2], 'value %.0f' % d['value'], {n: round(v['avg_us'], 1) for n, v in k.items()})
PY
done
