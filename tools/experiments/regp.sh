# round 3: level 1 with more points per thread in registers and fewer, longer workgroups (one round on 256 CUs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp pc-accumulation-lib_amd/pca_amd/libpca_hip.so /tmp/std.so
run() {  # lib G chunk
  cp $1 pc-accumulation-lib_amd/pca_amd/libpca_hip.so
  PCA_BEV_G=$2 PCA_BEV_CHUNK=$3 python bench.py --no-extras --no-cpu-baseline --steps 100 > gpurun_out/gs.json 2> gpurun_out/gs.err
  python - "$1" $2 $3 <<'PY'
import json, sys
d = json.load(open('gpurun_out/gs.json'))
k = d['roofline']['kernels']
print(sys.argv[1].split('/')[-1], 'G', sys.argv[2], 'chunk', sys.argv[3], 'value %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']), {n: round(v['avg_us'], 1) for n, v in k.items()})
PY
}
run /tmp/std.so 512 8192
run tools/experiments/libpca_regp16.bin 320 12288
run tools/experiments/libpca_regp16.bin 384 12288
run tools/experiments/libpca_regp20.bin 256 16384
run tools/experiments/libpca_regp20.bin 320 16384
cp tools/experiments/libpca_regp20.bin pc-accumulation-lib_amd/pca_amd/libpca_hip.so
PCA_BEV_G=256 PCA_BEV_CHUNK=16384 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "bev or chain" 2>&1 | tail -2
cp /tmp/std.so pc-accumulation-lib_amd/pca_amd/libpca_hip.so
