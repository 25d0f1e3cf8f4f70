# round 3: K1 with the dependent colour gather: tests, pool 8 / 64, one-frame FUSED time; then the nt-gather build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "k1_ and not k1n" 2>&1 | tail -2
for POOL in 8 64; do timeout -k 10 120 python tools/experiments/k1_batched.py $POOL 64 20 2>&1 | tail -1; done
timeout -k 10 120 python tools/experiments/k1_batched.py 8 1 50 2>&1 | tail -1
cp pc-accumulation-lib_amd/pca_amd/libpca_hip.so /tmp/std.so
cp tools/experiments/libpca_hip_nt.bin pc-accumulation-lib_amd/pca_amd/libpca_hip.so
echo "== nt gathers"
for POOL in 8 64; do timeout -k 10 120 python tools/experiments/k1_batched.py $POOL 64 20 2>&1 | tail -1; done
cp /tmp/std.so pc-accumulation-lib_amd/pca_amd/libpca_hip.so
