# ICP timing of several builds of the library on one box: VARIANTS="icp5 icp8" tools/experiments/icp_ab.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp pc-accumulation-lib_amd/pca_amd/libpca_hip.so /tmp/std.so
for rep in 1 2; do for v in ${VARIANTS:-icp5 icp8}; do
  cp tools/experiments/libpca_$v.bin pc-accumulation-lib_amd/pca_amd/libpca_hip.so
  echo "$v: $(python tools/experiments/icp_time.py 2>&1 | grep '^ms' | tail -1)  | no skip: $(PCA_ICP_NO_SKIP=1 python tools/experiments/icp_time.py 2>&1 | grep '^ms' | tail -1)"
done; done
cp /tmp/std.so pc-accumulation-lib_amd/pca_amd/libpca_hip.so
