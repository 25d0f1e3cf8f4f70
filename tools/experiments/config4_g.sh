# round 3: BASELINE configs[3] at one tenth -- level-1 group count / chunk size for giant windows
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for CFG in "512 8192" "256 8192" "128 8192" "1024 8192" "512 12288"; do
  set -- $CFG
  PCA_BEV_G=$1 PCA_BEV_CHUNK=$2 python - <<PY
import sys, os, json, builtins
sys.path.insert(0, os.getcwd())
import bench
rp = builtins.print
builtins.print = lambda *a, **k: None
r = bench.config4_pass()
builtins.print = rp
print('G=$1 chunk=$2', 'ms/step %.3f' % r['ms_per_step'], {k: round(v, 1) for k, v in r['kernels_avg_us'].items()}, 'frac %.3f' % r['roofline_bev_unit']['frac'])
PY
done
