"""The NuScenes scene block of bench.py on its own (batched / stepwise / PCIe-inclusive scene wall time, batched K1n, bev_many)."""
import json, os, sys, builtins
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
rp = builtins.print
builtins.print = lambda *a, **k: None
out = bench.nuscenes_scene_pass()
builtins.print = rp
print(json.dumps(out, indent=1))
