# round 5: pieces ordered by half of the tile (PCA_BEV_SPLIT=1, default) against round 4's form (0): config 4 and the ring model
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for S in 0 1; do
  echo "config4 PCA_BEV_SPLIT=$S: $(PCA_BEV_SPLIT=$S python tools/experiments/pass_only.py config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step %.3f  unit us %.1f  frac %.4f  kernels %s' % (d['ms_per_step'], d['roofline_bev_unit']['us_sum_of_kernels'], d['roofline_bev_unit']['frac'], {k: round(v,1) for k,v in d['kernels_avg_us'].items()}))")"
  echo "ring    PCA_BEV_SPLIT=$S: $(PCA_BEV_SPLIT=$S python tools/experiments/pass_only.py ring 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step %.4f  Mpts/s %.0f  kernels %s' % (d['ms_per_step'], d['Mpoints_per_s'], {k: round(v,1) for k,v in d['kernels_avg_us'].items()}))")"
done; done
