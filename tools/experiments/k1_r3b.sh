# round 3: per-phase stamps of the K1 batch forms on 64 distinct frames
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for MODE in split linked; do
  for POOL in 8 64; do
    echo "=== MODE=$MODE POOL=$POOL"
    PCA_K1_STAMPS=1 PCA_K1_MODE=$MODE PCA_K1_PF=0 timeout -k 10 120 python tools/experiments/k1_batched.py $POOL 64 5 2>&1 | tail -16
  done
done
