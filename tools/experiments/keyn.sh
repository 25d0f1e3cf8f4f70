# level 1 with one view branch per batch of points instead of one per point: parity first, then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp pc-accumulation-lib_amd/pca_amd/libpca_hip.so /tmp/std.so
cp tools/experiments/libpca_new.bin pc-accumulation-lib_amd/pca_amd/libpca_hip.so
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "bev or chain or stress" 2>&1 | tail -2
cp /tmp/std.so pc-accumulation-lib_amd/pca_amd/libpca_hip.so
VARIANTS="old new" bash tools/experiments/ab.sh
VARIANTS="old new" bash tools/experiments/ab.sh
