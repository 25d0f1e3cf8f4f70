"""cProfile of bench.nuscenes_scene_pass (the batched / stepwise / pcie forms of one NuScenes-shaped scene)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import builtins
import bench
rp = builtins.print
builtins.print = lambda *a, **k: None
bench.nuscenes_scene_pass(40, 2)
pr = cProfile.Profile()
pr.enable()
r = bench.nuscenes_scene_pass(40, 5)
pr.disable()
builtins.print = rp
print({k: v for k, v in r.items() if not isinstance(v, dict)})
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
print(s.getvalue())
