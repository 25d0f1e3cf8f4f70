"""Host time of the benchmark step by function (perf_counter wrappers at class / module level; inclusive times)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import torch  # noqa: E402
import builtins  # noqa: E402

acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
real_print = builtins.print
builtins.print = lambda *a, **k: None
st = bench.Stepper(acc, pool)
st.fill()
out = torch.empty((200, 21, bench.PX, bench.PX), dtype=torch.float16, device='cuda')
for k in range(20):
    st.step(out[k])
torch.cuda.synchronize()
spent = {}


def timed(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or name

    def w(*a, **kw):
        t = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            d = spent.setdefault(label, [0.0, 0])
            d[0] += time.perf_counter() - t
            d[1] += 1
    setattr(owner, name, w)


import sem_pc_accum, kitti360_sem_pc_accum  # noqa: E402
from pca_amd import host_logic as hl, device_store  # noqa: E402
from bev_generator import bev_generator as bg, sem_bev  # noqa: E402
K = kitti360_sem_pc_accum.Kitti360SemanticPointCloudAccumulator
S = sem_pc_accum.SemanticPointCloudAccumulator
for owner, names in ((K, ['integrate', '_frame_tensors']), (S, ["update_poses", "update_sem_pcs", "remove_observations", "_window_inputs", "_after_integrate", "get_incremental_path_dists", "_run_bev"]),
                     (device_store.DeviceStore, ['append_kitti', 'bev', 'retransform', 'evict', 'max_window_points', 'c_store']),
                     (bg.BEVGenerator, ['generate', '_raster_params', 'rasterise']), (sem_bev.SemBEVGenerator, ['generate_bev']),
                     (hl, ['transform_ego_split', 'incremental_path_dists', 'heading_rot_ang', 'rotation_matrix_3d', 'pose_dist']),
                     (bench, ['present_index'])):
    for n in names:
        if hasattr(owner, n):
            timed(owner, n, owner.__name__.split('.')[-1] + '.' + n)
ctx = __import__('pca_amd._lib', fromlist=['x']).Context.get()


class LibProxy:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, k):
        f = getattr(self._lib, k)

        def w(*a):
            t = time.perf_counter()
            r = f(*a)
            d = spent.setdefault('C:' + k, [0.0, 0])
            d[0] += time.perf_counter() - t
            d[1] += 1
            return r
        return w


ctx.lib = LibProxy(ctx.lib)
N = 300
t0 = time.perf_counter()
for k in range(N):
    st.step(out[k % 200])
t1 = time.perf_counter()
torch.cuda.synchronize()
builtins.print = real_print
print('enqueue %.1f us/step with timers' % (1e6 * (t1 - t0) / N))
for k, v in sorted(spent.items(), key=lambda kv: -kv[1][0]):
    print('  %-44s %6.1f us/step (%d calls)' % (k, 1e6 * v[0] / N, v[1]))
