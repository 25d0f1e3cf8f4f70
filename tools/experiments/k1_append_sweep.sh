# round 5: k1_append variants (tiles per workgroup, non-temporal accesses) on 64 ring-model / 64 uniform frames
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for KIND in ring uniform; do
  ARGS="k1 64"; [ $KIND = ring ] && ARGS="k1 64 ring"
  for V in 1 1,nt 4 4,nt 8 8,nt; do
    echo "$KIND PCA_K1_APPEND=$V: $(PCA_K1_APPEND=$V python tools/experiments/pass_only.py $ARGS | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wall us %.2f  events us %.2f  frac %.4f  frac_events %.4f  kept %d' % (d['us_per_call_wall_back_to_back'], d['us_per_call_hip_events'], d['frac'], d['frac_on_hip_event_time'], d['kept']))")"
  done
done
