# round 4: level-1 workgroup count again, now that the counter tables are XCD-major (round 3: 640 / 768 / 1024 chunks of 1024
# threads 72 / 75 / 88 us against 61 at 512 -- part of that "fixed cost per workgroup" was the partial-line table writes)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for CFG in "512 8192" "640 8192" "768 6144" "1016 5120" "384 12288" "512 8192"; do
  set -- $CFG
  PCA_BEV_G=$1 PCA_BEV_CHUNK=$2 python bench.py --no-extras --no-cpu-baseline --steps 100 > gpurun_out/gs.json 2> gpurun_out/gs.err
  python - $1 $2 <<'PY'
import json, sys
d = json.load(open('gpurun_out/gs.json'))
k = d['roofline']['kernels']
print('G', sys.argv[1], 'chunk', sys.argv[2], 'value %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']), {n: round(v['avg_us'], 1) for n, v in k.items()}, flush=True)
PY
done
