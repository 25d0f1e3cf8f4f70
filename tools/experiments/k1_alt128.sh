# round 5: is the gain of the non-temporal append real for inputs that come from HBM?  64 frames cycled (257 MB of inputs: the
# Infinity Cache keeps part of them between calls) against two alternating batches of 64 different frames (514 MB).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for N in 64 128; do for KIND in uniform ring; do for A in 1,plain 1,nt; do
  ARGS="k1 $N"; [ $KIND = ring ] && ARGS="k1 $N ring"
  echo "$KIND distinct=$N append=$A: $(PCA_K1_APPEND=$A python tools/experiments/pass_only.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wall us %.2f  events us %.2f  frac %.4f  frac_events %.4f' % (d['us_per_call_wall_back_to_back'], d['us_per_call_hip_events'], d['frac'], d['frac_on_hip_event_time']))")"
done; done; done
