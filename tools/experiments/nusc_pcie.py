"""The NuScenes scene from HOST arrays (bench.py nuscenes_scene_pass, pcie form only), for A/B of the upload path."""
import builtins, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
rp = builtins.print
builtins.print = lambda *a, **k: None
out = bench.nuscenes_scene_pass(reps=2, forms=('pcie', ))
builtins.print = rp
print('pcie scene: %.1f ms' % out['pcie']['ms_per_scene'])
