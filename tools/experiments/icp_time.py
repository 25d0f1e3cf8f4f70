"""Device ICP on two consecutive ring-model frames: wall time per registration (for a kernel trace run under rocprofv3)."""
import sys
import time
import warnings

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from pca_amd.icp import GpuIcp  # noqa: E402

warnings.simplefilter('ignore')
a, b = GpuIcp.to_device(bench.ring_frame(0, 3)[0]), GpuIcp.to_device(bench.ring_frame(0, 4)[0])
icp = GpuIcp()
icp.register(a, b, 1e3, np.eye(4))
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    r = icp.register(a, b, 1e3, np.eye(4))
    print('ms %.3f iterations %d fitness %.6f rmse %.6f' % (1e3 * (time.perf_counter() - t0), r.iterations, r.fitness, r.inlier_rmse))
print(np.array2string(r.transformation, precision=6))
