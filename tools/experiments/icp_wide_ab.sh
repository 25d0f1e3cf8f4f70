# round 5: a wave with few searching queries works them off with all 64 lanes (PCA_ICP_WIDE_MAX = how few; 0 = never)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for W in 0 1 2 4 8; do
  echo "PCA_ICP_WIDE_MAX=$W: $(PCA_ICP_WIDE_MAX=$W python tools/experiments/icp_time.py 2>&1 | grep '^ms' | tail -2 | tr '\n' ' ')"
done; done
PCA_ICP_WIDE_MAX=${BEST:-2} bash tools/experiments/icp_prof.sh 2>&1 | grep -E "icp_match|icp_solve" | tail -20
