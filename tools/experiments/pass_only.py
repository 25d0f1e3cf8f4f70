"""One side pass of bench.py on its own (for the profiler): pass_only.py config4 | nuscenes_scene | ring | k1 <pool>."""
import builtins, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
rp = builtins.print
builtins.print = lambda *a, **k: None
what = sys.argv[1]
if what == 'config4':
    out = bench.config4_pass()
elif what == 'nuscenes_scene':
    out = bench.nuscenes_scene_pass(reps=3)
elif what == 'nuscenes_scene_sweep':
    out = bench.nuscenes_scene_pass(reps=3, order='sweep')
elif what == 'ring':
    out = bench.ring_model_pass(50)
elif what == 'k1':
    n = int(sys.argv[2])
    frame_fn = bench.ring_frame if len(sys.argv) > 3 and sys.argv[3] == 'ring' else bench.synth_frame
    out = bench.k1_batched_pass(bench.device_pool(frame_fn, 7, n), n)          # n = 128: two alternating batches of 64
else:
    raise SystemExit('unknown pass')
builtins.print = rp
print(json.dumps(out))
