import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'pc-accumulation-lib_amd'))
import numpy as np, torch, bench
from pca_amd.icp import GpuIcp
from pca_amd import _lib
a = GpuIcp.to_device(bench.ring_frame(0, 3)[0]); b = GpuIcp.to_device(bench.ring_frame(0, 4)[0])
icp = GpuIcp()
r = icp.register(a, b, 1e3, np.eye(4))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): r = icp.register(a, b, 1e3, np.eye(4))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print('120k-point ring frames: %.2f ms per registration, %d iterations, t = %s, fitness %.3f rmse %.3f' % (dt * 1e3, r.iterations, r.transformation[:3, 3], r.fitness, r.inlier_rmse))
ctx = _lib.Context.get(); ctx.profile(1); icp.register(a, b, 1e3, np.eye(4)); print(ctx.profile_read()['icp']); ctx.profile(0)
