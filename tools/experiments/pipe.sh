# round 3: level 1 register path with the next batch's loads in flight (two register sets), batches of 2 / 3 / 4 / 6 points
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
run /tmp/std.so
for r in 2 3 4 6; do run tools/experiments/libpca_pipe$r.bin; done
cp /tmp/std.so pc-accumulation-lib_amd/pca_amd/libpca_hip.so
