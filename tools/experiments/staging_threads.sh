# round 5: the unchanged driver's step against the size of the staging pool and the store form
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
nproc
for rep in 1 2; do for T in 3 7 11 15; do for NT in 1 0; do
  echo "threads=$T nt=$NT: $(PCA_STAGING_THREADS=$T PCA_STAGING_NT=$NT python tools/experiments/pcie_step.py 30 2>/dev/null | tail -1)"
done; done; done
