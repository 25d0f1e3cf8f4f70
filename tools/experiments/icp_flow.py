"""bench.py's kitti_icp_flow block on its own (integrate with the device ICP from host arrays + one BEV per frame).
env: PCA_ICP_STREAM=0 -> the registration on the caller's stream"""
import builtins, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
rp, builtins.print = builtins.print, (lambda *a, **k: None)
out = bench.kitti_icp_flow_pass(60)
builtins.print = rp
print(json.dumps({k: out[k] for k in ('ms_per_step', 'Mpoints_per_s', 'icp_ms_per_registration', 'icp_share_of_step', 'metres_per_frame_recovered_mean')}))
