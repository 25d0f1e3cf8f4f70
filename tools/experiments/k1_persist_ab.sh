# round 5: persistent SPLIT front kernel (next tile's points prefetched) against one tile per workgroup
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for KIND in "k1 64" "k1 64 ring" "k1 8" "k1 128"; do
  for V in "0 4" "1 2" "1 3" "1 4"; do set -- $V
    echo "$KIND persist=$1 per_cu=$2: $(PCA_K1_PERSIST=$1 PCA_K1_PERSIST_PER_CU=$2 python tools/experiments/pass_only.py $KIND 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wall us %.2f  events us %.2f  frac %.4f' % (d['us_per_call_wall_back_to_back'], d['us_per_call_hip_events'], d['frac']))")"
  done
done
