# A/B of environment switches of ONE build on one box, alternating:  VAR=PCA_BEV_CULL VALUES="0 1" tools/experiments/ab_env.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
VAR=${VAR:-PCA_BEV_CULL}
for rep in 1 2; do for v in ${VALUES:-0 1}; do
  env $VAR=$v python bench.py --no-extras --no-cpu-baseline --steps ${STEPS:-100} ${BENCH_ARGS:-} > gpurun_out/ab_$VAR$v.json 2> gpurun_out/ab_$VAR$v.err || { tail -5 gpurun_out/ab_$VAR$v.err; exit 1; }
  python - "$VAR=$v" gpurun_out/ab_$VAR$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d['roofline']['kernels']
print(sys.argv[1], 'value %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']), {n: round(v['avg_us'], 1) for n, v in k.items()}, 'unit %.1f' % d['roofline']['avg_launch_us'], flush=True)
PY
done; done
