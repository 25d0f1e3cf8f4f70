# A/B of two builds of the library on ONE box, alternating: tools/experiments/libpca_old.bin vs libpca_new.bin
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp pc-accumulation-lib_amd/pca_amd/libpca_hip.so /tmp/std.so
run() {
  cp $1 pc-accumulation-lib_amd/pca_amd/libpca_hip.so
  python bench.py --no-extras --no-cpu-baseline --steps 100 > gpurun_out/gs.json 2> gpurun_out/gs.err
  python - "$1" <<'PY'
import json, sys
d = json.load(open('gpurun_out/gs.json'))
k = d['roofline']['kernels']
print(sys.argv[1].split('/')[-1], 'value %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']), {n: round(v['avg_us'], 1) for n, v in k.items()}, 'unit %.1f' % d['roofline']['avg_launch_us'])
PY
}
for rep in 1 2; do for v in ${VARIANTS:-old new}; do run tools/experiments/libpca_$v.bin; done; done
cp /tmp/std.so pc-accumulation-lib_amd/pca_amd/libpca_hip.so
