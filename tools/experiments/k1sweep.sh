# K1 split form on the batched launch: tile shape; pool 8 = images shared by 8 frames, pool 64 = all distinct
for CFG in 512x4 256x4 1024x4; do
  for POOL in 8 64; do
    PCA_K1_CFG=$CFG timeout -k 10 120 python tools/experiments/k1_batched.py $POOL 64 20 2>&1 | tail -1
  done
done
