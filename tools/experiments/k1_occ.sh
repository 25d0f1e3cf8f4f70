# K1 batched front kernel against the tiles resident per CU (LDS padding): tools/experiments/libpca_pad{0,16000,40000}.bin = 4 / 3 / 2 tiles per CU
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp pc-accumulation-lib_amd/pca_amd/libpca_hip.so /tmp/std.so
for v in pad0 pad16000 pad40000 pad0; do
  cp tools/experiments/libpca_$v.bin pc-accumulation-lib_amd/pca_amd/libpca_hip.so
  echo "$v: $(python tools/experiments/k1_batched.py 64 64 20 2>&1 | grep cfg=)"
done
cp /tmp/std.so pc-accumulation-lib_amd/pca_amd/libpca_hip.so
