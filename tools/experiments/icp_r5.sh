# round 5: device ICP -- wall time per registration, kernel trace, and the ablations of icp_normals
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== plain"; python tools/experiments/icp_time.py 2>&1 | grep -E "^ms" 
echo "== no eigen decomposition (PCA_ICP_DBG=2)"; PCA_ICP_DBG=2 python tools/experiments/icp_time.py 2>&1 | grep -E "^ms" | tail -1
echo "== no covariance pass (PCA_ICP_DBG=4)"; PCA_ICP_DBG=4 python tools/experiments/icp_time.py 2>&1 | grep -E "^ms" | tail -1
echo "== trace"; bash tools/experiments/icp_prof.sh 2>&1 | grep -vE "^E2026|^W2026" 
echo "== trace, no eigen"; PCA_ICP_DBG=2 bash tools/experiments/icp_prof.sh 2>&1 | grep -E "icp_normals" | tail -1
echo "== trace, no covariance pass"; PCA_ICP_DBG=4 bash tools/experiments/icp_prof.sh 2>&1 | grep -E "icp_normals" | tail -1
