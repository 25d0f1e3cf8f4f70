import os, sys, ctypes as C, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, numpy as np, torch
from pca_amd.icp import GpuIcp
from pca_amd import _lib
warnings.simplefilter('ignore')
A, B = bench.ring_frame(0, 3)[0], bench.ring_frame(0, 4)[0]
icp = GpuIcp()
r = icp.register(GpuIcp.to_device(A), GpuIcp.to_device(B), 1e3, np.eye(4))
torch.cuda.synchronize()
buf = np.zeros((131072, 4), np.uint64)
_lib.Context.get().lib.pca_debug_icp_stamps(buf.ctypes.data_as(C.c_void_p))
n = len(B)
ua, ub = buf[:n, 0] / 100.0, buf[:n, 1] / 100.0
rng = np.linalg.norm(B[:, :3], axis=1)
fine, coarse, off = buf[:n, 2] & 255, buf[:n, 2] >> 8, buf[:n, 3]
print('pass A: median %.1f us p90 %.1f p99 %.1f max %.1f | pass B: median %.1f p90 %.1f p99 %.1f max %.1f' % (
    np.median(ua), np.percentile(ua, 90), np.percentile(ua, 99), ua.max(), np.median(ub), np.percentile(ub, 90), np.percentile(ub, 99), ub.max()))
for lo, hi in ((0, 5), (5, 10), (10, 20), (20, 40), (40, 79), (79, 200)):
    m = (rng >= lo) & (rng < hi)
    if m.any():
        print('  range %3d-%3d m: %6d points, pass A median %.1f p99 %.1f, pass B median %.1f, fine rings %.1f coarse rings %.1f offers median %d p99 %d' % (
            lo, hi, m.sum(), np.median(ua[m]), np.percentile(ua[m], 99), np.median(ub[m]), fine[m].mean(), coarse[m].mean(), np.median(off[m]), np.percentile(off[m], 99)))
