"""Randomised check of the device ICP's searches: fitness / rmse of the returned pose against exact nearest neighbours (k-d tree)
for random street scenes (tests/test_gpu_icp.py::sweep) with added far returns, random motions, cloud sizes and update counts.
usage: icp_fuzz.py <first seed> <last seed>"""
import os, sys, warnings
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'pc-accumulation-lib_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from scipy.spatial import cKDTree
import test_gpu_icp as t
from pca_amd.icp import GpuIcp
warnings.simplefilter('ignore')
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    n_beams, n_az = int(rng.choice([8, 16, 32, 64])), int(rng.choice([180, 450, 900, 1800]))
    max_range = float(rng.choice([20.0, 45.0, 90.0]))
    dx, dy, dyaw = rng.uniform(-1.5, 1.5), rng.uniform(-0.3, 0.3), rng.uniform(-0.04, 0.04)
    a = t.sweep(0.0, 0.0, 0.0, 1000 + seed, n_beams, n_az, max_range)
    b = t.sweep(dx, dy, dyaw, 2000 + seed, n_beams, n_az, max_range)
    n_far = int(rng.integers(0, 300))
    if n_far:
        ang = rng.uniform(-np.pi, np.pi, n_far)
        rad = rng.uniform(30.0, 120.0, n_far)
        far = np.stack([rad * np.cos(ang), rad * np.sin(ang), rng.uniform(-3.0, 6.0, n_far), np.zeros(n_far)], 1).astype(np.float32)
        step = rng.normal(size=(n_far, 3))
        step *= (rng.uniform(0.0, 5.0, n_far) / np.linalg.norm(step, axis=1))[:, None]
        twin = far.copy()
        twin[:, :3] += step.astype(np.float32)
        a = np.concatenate([a, far]).astype(np.float32)
        b = np.concatenate([b, twin]).astype(np.float32)
    if rng.random() < 0.3:
        sh = np.float32(rng.uniform(-6.0, 6.0))
        a[:, 2] += sh; b[:, 2] += sh
    cap = float(rng.choice([0.5, 1.5, 3.9]))
    updates = int(rng.choice([1, 2, 3, 5, 30]))
    icp = GpuIcp()
    icp.max_iteration = updates
    r = icp.register(GpuIcp.to_device(a), GpuIcp.to_device(b), cap, np.eye(4))
    T = r.transformation
    q = a[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    inside = (np.abs(q[:, 0]) < 127.0) & (np.abs(q[:, 1]) < 127.0) & (np.abs(q[:, 2]) < 15.0)
    tgt = b[:, :3].astype(np.float64)
    tin = (np.abs(tgt[:, 0]) < 128.0) & (np.abs(tgt[:, 1]) < 128.0) & (tgt[:, 2] >= -16.0) & (tgt[:, 2] < 16.0)
    d, _ = cKDTree(tgt[tin]).query(q)
    inl = (d < cap) & inside
    ok = inside.all() and abs(r.fitness - inl.mean()) < 1e-12 and (inl.sum() == 0 or abs(r.inlier_rmse - np.sqrt((d[inl] ** 2).mean())) < 1e-9)
    if not ok:
        bad += 1
        print('SEED', seed, 'FAILED: beams %d az %d n %d far %d cap %.1f updates %d (%d done) fitness %.9f / %.9f rmse %.9f / %.9f inside %s' % (
            n_beams, n_az, len(a), n_far, cap, updates, r.iterations, r.fitness, inl.mean(), r.inlier_rmse,
            np.sqrt((d[inl] ** 2).mean()) if inl.any() else 0.0, inside.all()), flush=True)
    if seed % 20 == 0:
        print('seed', seed, flush=True)
print('seeds %d..%d: %d failures' % (lo, hi, bad))
