# round 3: level-1 workgroup count on the headline window (5.17 M points): more, shorter workgroups even out the tail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for CFG in "512 8192" "640 8192" "768 6144" "1024 5120" "384 12288"; do
  set -- $CFG
  PCA_BEV_G=$1 PCA_BEV_CHUNK=$2 python bench.py --no-extras --no-cpu-baseline --steps 100 > gpurun_out/gs.json 2> gpurun_out/gs.err
  python - $1 $2 <<'PY'
import json, sys
d = json.load(open('gpurun_out/gs.json'))
k = d['roofline']['kernels']
print('G', sys.argv[1], 'chunk', sys.argv[2], 'value %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']), {n: round(v['avg_us'], 1) for n, v in k.items()})
PY
done
