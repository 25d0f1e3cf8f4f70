# level 1 with only the key in registers (BIN_LEAN) at two workgroups per CU: parity first, then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp pc-accumulation-lib_amd/pca_amd/libpca_hip.so /tmp/std.so
cp tools/experiments/libpca_l228.bin pc-accumulation-lib_amd/pca_amd/libpca_hip.so
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "bev or chain or stress" 2>&1 | tail -3
cp /tmp/std.so pc-accumulation-lib_amd/pca_amd/libpca_hip.so
VARIANTS="old l228 l218 l224" bash tools/experiments/ab.sh
