# ablations of the fused K1 (results invalid with SKIP): 1 = no look-back, 4 = no stores, STATIC = no ticket
for CFG in 512x8 512x4 256x4; do
  for POOL in 8 64; do
    for V in "PCA_K1_SKIP=0" "PCA_K1_SKIP=1" "PCA_K1_SKIP=1 PCA_K1_STATIC=1" "PCA_K1_SKIP=5" "PCA_K1_SKIP=5 PCA_K1_STATIC=1"; do
      echo -n "$V  "; env $V PCA_K1_CFG=$CFG timeout -k 10 120 python tools/experiments/k1_batched.py $POOL 64 20 2>&1 | tail -1
    done
  done
done
