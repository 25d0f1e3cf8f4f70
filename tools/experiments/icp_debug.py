import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'pc-accumulation-lib_amd'); sys.path.insert(0, 'tests')
import numpy as np
import test_gpu_icp as t
from pca_amd.icp import GpuIcp
for nb, naz, mr in ((32, 900, 30.0), (64, 700, 18.0), (96, 500, 14.0)):
    prev = t.sweep(0.0, 0.0, 0.0, 5, n_beams=nb, n_az=naz, max_range=mr)
    new = t.sweep(0.9, 0.03, 0.008, 6, n_beams=nb, n_az=naz, max_range=mr)
    res = GpuIcp().register(GpuIcp.to_device(prev), GpuIcp.to_device(new), 1e3, np.eye(4))
    T_ref, rmse_ref, it = t.icp_model(prev, new)
    print(nb, naz, mr, len(prev), 'gpu', res.transformation[:3, 3], res.inlier_rmse, res.iterations, '| model', T_ref[:3, 3], rmse_ref, it,
          '| diff', np.linalg.norm(res.transformation[:3, 3] - T_ref[:3, 3]))
