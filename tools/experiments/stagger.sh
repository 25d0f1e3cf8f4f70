# bev_tile_cells: start offset between the workgroups of a CU (PCA_BEV_STAGGER) on the headline, the ring model and the NuScenes scene
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for v in ${VALUES:-0 3}; do
  PCA_BEV_STAGGER=$v python bench.py --no-extras --no-cpu-baseline --steps 100 > gpurun_out/sg.json 2> gpurun_out/sg.err
  PCA_BEV_STAGGER=$v python bench.py --no-extras --no-cpu-baseline --steps 100 --scene ring > gpurun_out/sg_ring.json 2>> gpurun_out/sg.err
  PCA_BEV_STAGGER=$v python tools/experiments/pass_only.py nuscenes_scene > gpurun_out/sg_nusc.json 2>> gpurun_out/sg.err
  python - $v <<'PY'
import json, sys
d = json.load(open('gpurun_out/sg.json')); r = json.load(open('gpurun_out/sg_ring.json'))
k = d['roofline']['kernels']; kr = r['roofline']['kernels']
try:
    n = json.load(open('gpurun_out/sg_nusc.json'))
    ns = 'nusc batched %.2f ms bev_many %.0f us' % (n['batched']['ms_per_scene'], n['bev_many']['us_per_call_hip_events'])
except Exception as e:
    ns = 'nusc ? %s' % e
print('stagger', sys.argv[1], 'value %.0f cells %.1f unit %.1f | ring value %.0f cells %.1f heavy %.1f |' % (
    d['value'], k['bev_cells']['avg_us'], d['roofline']['avg_launch_us'], r['value'], kr['bev_cells']['avg_us'], kr.get('bev_cells_heavy', {}).get('avg_us', 0)), ns, flush=True)
PY
done; done
