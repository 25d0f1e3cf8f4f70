# round 3: K1 SPLIT after the latency work -- correctness first, then timing (pool 8 = inputs cached, 64 = distinct) and stamps
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "k1_ and not k1n" 2>&1 | tail -2
for POOL in 8 64; do
  timeout -k 10 120 python tools/experiments/k1_batched.py $POOL 64 20 2>&1 | tail -1
done
for CFG in 256x4 1024x4; do
  PCA_K1_CFG=$CFG timeout -k 10 120 python tools/experiments/k1_batched.py 64 64 20 2>&1 | tail -1
done
PCA_K1_STAMPS=1 timeout -k 10 120 python tools/experiments/k1_batched.py 64 64 5 2>&1 | tail -16
