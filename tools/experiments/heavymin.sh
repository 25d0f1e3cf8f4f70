# ring-model step vs the record count above which a tile goes to the heavy kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
for HM in 1536 2048 2560 3072 4096; do
  PCA_BEV_HEAVY_MIN=$HM timeout -k 10 200 python $R/bench.py --no-extras --no-cpu-baseline --scene ring --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('heavy_min $HM ms_per_step %.4f' % d['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['roofline']['kernels'].items()})"
done
