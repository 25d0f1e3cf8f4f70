import sys, time, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'pc-accumulation-lib_amd'))
import numpy as np, torch, bench
from pca_amd.icp import GpuIcp
def run(name, a, b):
    a, b = GpuIcp.to_device(a), GpuIcp.to_device(b)
    icp = GpuIcp()
    r = icp.register(a, b, 1e3, np.eye(4))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = icp.register(a, b, 1e3, np.eye(4))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print('%-28s n=%6d/%6d  %.2f ms  %d iterations' % (name, a.shape[0], b.shape[0], dt * 1e3, r.iterations))
A, B = bench.ring_frame(0, 3)[0], bench.ring_frame(0, 4)[0]
run('all points', A, B)
ra, rb = np.linalg.norm(A[:, :3], axis=1), np.linalg.norm(B[:, :3], axis=1)
run('range < 79 m', A[ra < 79], B[rb < 79])
run('range < 40 m', A[ra < 40], B[rb < 40])
run('range > 10 m and < 79', A[(ra > 10) & (ra < 79)], B[(rb > 10) & (rb < 79)])
