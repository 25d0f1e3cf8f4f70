"""Randomised parity run of the round-5 launch structure: the KITTI drop-in with K1 riding in the raster and the frames out of
view left out (the defaults) against the same accumulator with both off -- planes, polylines, evictions sample for sample, stored
rows at the end.  Random paths (yaw / pitch / step), frame sizes, view sizes, grid sizes, horizons, sample positions, skipped
rasters, host and device inputs, camera frames and per-point labels.   usage: flow_fuzz.py [seconds=240] [first_seed=0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import builtins
import numpy as np
import torch
import sem_pc_accum
from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
from pca_amd import host_logic as hl
from pca_amd.device_store import DeviceStore

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
FILTERS = [10, 11, 12, 16, 18, 255]
SEM_IDXS = {'road': 0, 'car': 13, 'truck': 14, 'bus': 15, 'motorcycle': 17}
cam_to_velo = np.array([[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418],
                        [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
                        [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824], [0, 0, 0, 1]])
rp, builtins.print = builtins.print, (lambda *a, **k: None)
t_end = time.time() + budget
runs = samples = left_out = rode = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(8, 120)), int(rng.integers(8, 400))
    f = rng.uniform(20, 200)
    P = np.array([[f, 0, W / 2, 0], [0, f, H / 2, 0], [0, 0, 1, 0]]) @ np.linalg.inv(cam_to_velo)
    use_gt = bool(rng.integers(0, 2))
    horizon = float(rng.uniform(15, 90))
    view, px = float(rng.uniform(8, 70)), int(rng.choice([32, 64, 96, 128, 256]))
    spread = float(rng.uniform(5, 45))
    DeviceStore.BOX_EVERY = int(rng.integers(1, 9))
    pool = []
    for _ in range(4):
        n = int(rng.choice([1, 7, 300, 3000, 9000, 20000]))
        pc = np.stack([rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), rng.uniform(-2, 3, n), rng.uniform(0, 1, n)],
                      1).astype(np.float32)
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        sem = rng.integers(0, 19, (H, W)).astype(np.uint8)
        pool.append((pc, img, sem, rng.integers(0, 19, (n, 1)).astype(np.int64)))
    by_id = {id(p[1]): p[2] for p in pool}

    class Model:
        def pred(self, rgb):
            return by_id[id(rgb)][None, None]
    sem_pc_accum.SemSegONNX = lambda path: Model()
    steps = int(rng.integers(30, 140))
    Ts = []
    for k in range(steps):
        yaw, pitch = rng.normal(0, 0.04), rng.normal(0, 0.004)
        R = hl.rotation_matrix_3d(yaw)
        Rp = np.array([[np.cos(pitch), 0, np.sin(pitch)], [0, 1, 0], [-np.sin(pitch), 0, np.cos(pitch)]])
        T = np.eye(4)
        T[:3, :3] = Rp @ R
        T[:3, 3] = [-rng.uniform(0.2, 2.5), rng.normal(0, 0.03), rng.normal(0, 0.003)]
        Ts.append(T)
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': P}
    bev = dict(type='sem', view_size=view, pixel_size=px, max_trans_radius=0., zoom_thresh=0., do_warp=False, int_scaler=20.,
               int_sep_scaler=20., int_mid_threshold=0.5, height_filter=None if rng.random() < 0.5 else float(rng.uniform(0, 3)))
    accs = []
    for new in (True, False):
        acc = Kitti360SemanticPointCloudAccumulator(horizon, calib, 1e3, 'm', FILTERS, SEM_IDXS, use_gt, dict(bev))
        acc._store_args = dict(capacity=1 << int(rng.integers(16, 21)), max_frames=int(rng.integers(8, 200)))
        it = iter(Ts)
        acc.pose_provider = lambda pc, it=it: next(it)
        acc._defer_k1 = new
        accs.append(acc)
    accs[0].store.cull, accs[1].store.cull = True, False
    if os.environ.get('FLOW_FUZZ_GENERAL_BEV'):               # integrate() through the one-call path, generate_bev() through the general one
        for acc in accs:
            acc._fast_bev_ok = lambda idx: False
    plan = [(int(rng.integers(0, 4)), rng.random() < 0.75, rng.random(), rng.random() < 0.5) for _ in range(steps)]
    # now and then the stream changes its camera (another calibration: the frames taken before lose their cone) or switches
    # between the camera's class map and per-point labels (frames with and without a cone in one window)
    switch_cam = int(rng.integers(5, steps)) if rng.random() < 0.3 else -1
    switch_gt = int(rng.integers(5, steps)) if rng.random() < 0.3 else -1
    for k, (fi, raster, where, host) in enumerate(plan):
        if k == switch_cam:
            f2 = rng.uniform(20, 200)
            P2 = np.array([[f2, 0, W / 2, 0], [0, f2, H / 2, 0], [0, 0, 1, 0]]) @ np.linalg.inv(cam_to_velo)
            for acc in accs:
                acc.P_velo_frame = P2
        if k == switch_gt:
            use_gt = not use_gt
            for acc in accs:
                acc.use_gt_sem = use_gt
                if getattr(acc, 'semseg_model', None) is None:
                    acc.semseg_model = Model()
        pc, img, sem, gt = pool[fi]
        pin = pc if host else torch.from_numpy(pc).cuda()
        obs = (img, pin, gt if use_gt else None)
        outs = []
        for acc in accs:
            ev = acc.integrate([obs])
            n = len(acc.poses)
            out = None
            if raster and n >= 3:
                pidx = 1 + int(where * (n - 2))
                out = acc.generate_bev(pidx, 1, gen_future=True)[0]
            outs.append((ev, out))
        assert outs[0][0] == outs[1][0], (seed, k, 'evicted')
        if outs[0][1] is not None:
            samples += 1
            a, b = outs[0][1], outs[1][1]
            for key in a.keys():
                if key.startswith('trajs_'):
                    assert all(np.array_equal(x, y) for x, y in zip(a[key], b[key])), (seed, k, key)
                else:
                    assert np.array_equal(a[key].view(np.uint16), b[key].view(np.uint16)), (seed, k, key)
    ra, rb = (np.concatenate(acc.sem_pcs) if len(acc.poses) else np.zeros((0, 10)) for acc in accs)
    assert np.array_equal(ra, rb), (seed, 'rows')
    for acc in accs:
        acc.store.check_status()
    left_out += accs[0].store.hints_taken
    assert accs[1].store.hints_taken == 0
    runs += 1
    seed += 1
builtins.print = rp
print('flow fuzz: %d sequences (seeds up to %d), %d samples compared bit for bit, frames left out in %d of them; 0 failures'
      % (runs, seed - 1, samples, left_out))
