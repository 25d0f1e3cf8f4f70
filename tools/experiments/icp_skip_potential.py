"""How much of a warm ICP pass could be skipped?  Per iteration of the device point-to-plane ICP (two consecutive ring-model
sweeps): how far the source points move, and for what fraction the nearest target provably stays the nearest
(|movement| < (d2 - d1) / 2, d1 / d2 = distance to the nearest / second-nearest target before the move)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import numpy as np
import torch
from scipy.spatial import cKDTree
from pca_amd.icp import GpuIcp
warnings.simplefilter('ignore')
for pair in ((3, 4), (40, 41)):
    A, B = bench.ring_frame(0, pair[0])[0], bench.ring_frame(0, pair[1])[0]
    a, b = GpuIcp.to_device(A), GpuIcp.to_device(B)
    tree = cKDTree(B[:, :3].astype(np.float64))
    Ts = [np.eye(4)]
    for k in range(1, 14):
        icp = GpuIcp()
        icp.max_iteration = k
        r = icp.register(a, b, 1e3, np.eye(4))
        Ts.append(r.transformation)
        if r.iterations < k:
            break
    P = np.c_[A[:, :3].astype(np.float64), np.ones(len(A))]
    print('pair', pair, 'iterations', r.iterations)
    for k in range(1, len(Ts)):
        q0, q1 = (P @ Ts[k - 1].T)[:, :3], (P @ Ts[k].T)[:, :3]
        d, _ = tree.query(q0, k=2)
        slack = (d[:, 1] - d[:, 0]) / 2
        mv = np.linalg.norm(q1 - q0, axis=1)
        print('  update %d: movement median %.2e max %.2e | slack median %.2e | nearest provably unchanged for %.1f %%' % (
            k, np.median(mv), mv.max(), np.median(slack), 100 * (slack > mv).mean()), flush=True)
