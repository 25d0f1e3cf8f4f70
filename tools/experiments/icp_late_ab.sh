# round 5: late passes of the device ICP through icp_match_late (one thread per query, searches by the wave's groups) from pass N on
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for L in 99 4 3 5 2; do
  echo "PCA_ICP_LATE_FROM=$L: $(PCA_ICP_LATE_FROM=$L python tools/experiments/icp_time.py 2>&1 | grep '^ms' | tail -2 | tr '\n' ' ')"
done; done
PCA_ICP_LATE_FROM=4 bash tools/experiments/icp_prof.sh 2>&1 | grep -vE "^E2026|^W2026" | tail -30
