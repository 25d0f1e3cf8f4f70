#!/usr/bin/env python3
"""Headline benchmark: Mpoints/s projected+accumulated and BEV frames/s @256x256 (BASELINE.json).

Workload at every N (BASELINE.json configs[1], SURVEY.md 8d "Config 2"): KITTI-360-shaped synthetic frames (120 000 points
f32, 376x1408 RGB + semseg), 200 m accumulation horizon at 1 m / frame (~200 live frames, ~5 M stored points), one
256x256 x 21-plane BEV per integrated frame; every rank runs its own sequence (weak scaling, no data-path collective).

One STEP = integrate one frame (K1 fused project / sample / filter / append + the re-transform of every stored point +
horizon eviction) + generate one BEV sample (bin, scan, scatter, per-cell reduce).  Inputs (point clouds, images, semseg
maps) are resident in HBM before the timed region; BEV tensors stay in HBM on the rank that made them.  The K-step timed
region is repeated -- at least 5 times and until 0.25 s have been timed in all, the first repetition dropped -- and
`value` / `ms_per_step` are the MEDIAN repetition (min / max beside it): `--steps 20` and `--steps 200` give the same value.

Beside `value` the JSON line carries (rank 0): per-kernel HIP-event times and the roofline of the dominant unit; the
batched K1 (the north-star kernel) on 64 frames per call; the same step on ring-model (skewed) frames; BASELINE configs[2]
(NuScenes kernels at full size), configs[3] (1 M points / frame, 512^2 grid, scaled frame count) and configs[4] (the nine
KITTI-360 sequences cut into warm-up-prefixed chunks over the ranks -- a STRONG-scaling job whose wall time is reported
at every N); the PCIe-inclusive step; the CPU baselines (C port and numpy-shaped restatement of the reference).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus 8                      # starts its own ranks (torch.distributed.run, RCCL); so does the driver's
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --workload config5 --gpus N   # the sharded nine-sequence job as the headline value (strong scaling)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'pc-accumulation-lib_amd')
sys.path.insert(0, ROOT)
sys.path.insert(0, PKG)

import numpy as np  # noqa: E402

N_PTS, IMG_H, IMG_W = 120_000, 376, 1408
HORIZON_M, BEV_HORIZON_M, VIEW_M, PX = 200.0, 80, 80, 256
FILTERS = [10, 11, 12, 16, 18, 255]
SEM_IDXS = {'road': 0, 'car': 13, 'truck': 14, 'bus': 15, 'motorcycle': 17}
POOL = 8                                  # distinct synthetic frames cycled through by the step loop
REPEATS = 5                               # repetitions of the K-step timed region, at least (median reported)
MIN_TIMED_S = 0.25                        # ... and as many as it takes to have timed this many seconds in all
MAX_REPEATS = 2000
GATHER_CHUNK = 16                         # multi-GPU: BEV samples per asynchronous gather to rank 0
HBM_PEAK_GBS = 8000.0                     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PROFILE_TAG = 'r05'                       # profiles/<tag>_pmc_traffic*.json: rocprofv3 PMC passes of THESE kernels
KITTI360_LENGTHS = [11270, 14384, 730, 11440, 6610, 9578, 2960, 13855, 3540]   # run_kitti360_bev_gen.py:172-173, end - start

CAM_TO_VELO = np.array([[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418],
                        [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
                        [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824], [0, 0, 0, 1]])
P_RECT = np.array([[552.554261, 0, 682.049453, 0], [0, 552.554261, 238.769549, 0], [0, 0, 1, 0]])
P_VELO_FRAME = P_RECT @ np.linalg.inv(CAM_TO_VELO)


def t_new_prev():
    """1 m / frame on a gentle curve: Rz(-0.002) . trans(-1, 0, 0)   (SURVEY.md 8d)."""
    c, s = np.cos(-0.002), np.sin(-0.002)
    R = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.]])
    Tt = np.eye(4)
    Tt[0, 3] = -1.0
    return R @ Tt


def synth_frame(seq, frame):
    """K-shape frame, seed = 1000*sequence + frame."""
    rng = np.random.default_rng(1000 * seq + frame)
    pc = np.stack([rng.uniform(-60, 60, N_PTS), rng.uniform(-60, 60, N_PTS), rng.uniform(-2, 3, N_PTS),
                   rng.uniform(0, 1, N_PTS)], 1).astype(np.float32)
    img = rng.integers(0, 256, (IMG_H, IMG_W, 3), dtype=np.uint8)
    sem = rng.integers(0, 19, (IMG_H, IMG_W)).astype(np.uint8)
    sem[rng.random((IMG_H, IMG_W)) < 0.01] = 255
    return pc, img, sem


N_BEAMS, N_AZ = 64, 1875                  # ring model: 64 x 1875 = 120 000 returns per revolution
SENSOR_H = 1.73


def ring_frame(seq, frame):
    """Ring-model frame: a 64-beam spinning lidar over a ground plane with boxes (parked cars,
    facades) -- point density falls off as 1/r^2, so the cells next to the driven path hold hundreds to
    thousands of points.  This is the contention / load-imbalance case; the uniform K-shape frame is the
    throughput case.  The street scene slides by 1 m per frame."""
    rng = np.random.default_rng(1000 * seq + frame + 500_000)
    az = np.repeat(np.linspace(-np.pi, np.pi, N_AZ, endpoint=False)[None], N_BEAMS, 0).ravel()
    el = np.repeat(np.deg2rad(np.linspace(2.0, -24.8, N_BEAMS))[:, None], N_AZ, 1).ravel()
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], 1)
    with np.errstate(divide='ignore', invalid='ignore'):
        t = np.where(d[:, 2] < 0, -SENSOR_H / d[:, 2], np.inf)
    kind = np.zeros(len(t), np.int8)                                 # 0 ground, 1 car, 2 facade
    shift = float(frame % 25)
    boxes = []
    for j in range(-4, 5):
        x0 = 25.0 * j - shift
        boxes += [(x0, 12.0, 9.0, 3.0, 8.0, 2), (x0 + 12.5, -12.0, 9.0, 3.0, 8.0, 2)]          # facades
        boxes += [(x0 + 5.0, 4.6, 2.2, 0.9, 1.5, 1), (x0 + 17.0, -4.6, 2.2, 0.9, 1.5, 1)]        # parked cars
    for cx, cy, hx, hy, h, kd in boxes:
        lo = np.array([cx - hx, cy - hy, -SENSOR_H]); hi = np.array([cx + hx, cy + hy, -SENSOR_H + h])
        with np.errstate(divide='ignore', invalid='ignore'):
            t1 = lo / d; t2 = hi / d
        tn = np.nanmax(np.minimum(t1, t2), 1); tf = np.nanmin(np.maximum(t1, t2), 1)
        hit = (tn <= tf) & (tn > 0) & (tn < t)
        t = np.where(hit, tn, t); kind[hit] = kd
    miss = ~np.isfinite(t) | (t > 80.0)
    t = np.where(miss, 80.0, t) + rng.normal(0, 0.02, len(t))
    xyz = d * t[:, None]
    inten = np.where(kind == 0, 0.3, 0.5) + rng.normal(0, 0.03, len(t))
    lane = (kind == 0) & ((np.abs(np.abs(xyz[:, 1]) - 1.75) < 0.1))
    inten = np.clip(np.where(lane, 0.9, inten), 0, 1)
    pc = np.concatenate([xyz, inten[:, None]], 1).astype(np.float32)
    # image-space semantics: road trapezoid / sidewalk below the horizon row, facades + vegetation + sky above, cars
    v, u = np.mgrid[0:IMG_H, 0:IMG_W]
    sem = np.full((IMG_H, IMG_W), 2, np.uint8)
    sem[(v < 120)] = 10
    sem[(v >= 120) & (v < 200) & ((u // 64) % 3 == 0)] = 8
    below = v > 238
    half = 60 + (v - 238) * 6.0
    sem[below] = 1
    sem[below & (np.abs(u - 682) < half)] = 0
    for cu in (250, 1100):
        sem[250:300, cu:cu + 120] = 13
    palette = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    img = (palette[sem].astype(np.int16) + rng.integers(-8, 9, (IMG_H, IMG_W, 3))).clip(0, 255).astype(np.uint8)
    return pc, img, sem


class ResidentSemSeg:
    """Stand-in for the external ONNX CNN (not part of the hot path): hands back the semseg map that is
    already resident in HBM for the image it is asked about."""

    def __init__(self):
        self.by_ptr = {}

    def pred(self, rgb):
        return self.by_ptr[rgb.data_ptr()][None, None]


def pmc_traffic(suffix='', prefixes=('bev_tile', )):
    """HBM bytes per launch of the named kernels from the committed rocprofv3 PMC passes of THESE kernels
    (profiles/<tag>_pmc_traffic<suffix>.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes with
    --kernel-trace only; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot run the profiler
    itself: (bytes, source) or (None, None) when the file is not there."""
    path = os.path.join(ROOT, 'profiles', PROFILE_TAG + '_pmc_traffic' + suffix + '.json')
    if not os.path.exists(path):
        return None, None
    pmc = json.load(open(path))['kernels']
    names = [k for k in pmc if k.startswith(tuple(prefixes))]
    # a kernel that only runs while the window fills (the heavy-tile kernel on uniform data: launched until no heavy tile
    # was seen for 64 calls, never in steady state) is not part of a steady-state launch's traffic
    most = max((pmc[k].get('launches_averaged', 0) for k in names), default=0)
    names = [k for k in names if pmc[k].get('launches_averaged', 0) * 2 >= most]
    if not names:
        return None, None
    total = sum(2.0 * pmc[k].get('FETCH_SIZE_KB', 0.0) + pmc[k].get('WRITE_SIZE_KB', 0.0) for k in names) * 1024.0
    return total, 'profiles/%s_pmc_traffic%s.json (2*FETCH_SIZE + WRITE_SIZE, summed over %s)' % (PROFILE_TAG, suffix, ', '.join(sorted(names)))


def present_index(acc):
    """Sample trigger of run_kitti360_bev_gen.py:218-230: first pose more than 80 m of path behind the newest, with 80 m
    of path ahead of it (conditions 1 and 2; one sample per frame: no minimum spacing).  One call into the pose track."""
    return acc._track.trigger(BEV_HORIZON_M, 0, 0.0)


def device_pool(frame_fn, seq, n, model=None):
    import torch
    pool = []
    for k in range(n):
        pc, img, sem = frame_fn(seq, k)
        f = (torch.from_numpy(img).cuda(), torch.from_numpy(pc).cuda(), torch.from_numpy(sem).cuda())
        if model is not None:
            model.by_ptr[f[0].data_ptr()] = f[2]
        pool.append(f)
    return pool


def new_accumulator(model, T=None, bev_px=PX, view=VIEW_M):
    """Drop-in KITTI accumulator over frames resident in HBM (poses: the synthetic curve)."""
    import sem_pc_accum
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    sem_pc_accum.SemSegONNX = lambda path: model
    bev_params = dict(type='sem', view_size=view, pixel_size=bev_px, max_trans_radius=0., zoom_thresh=0., do_warp=False,
                      int_scaler=20., int_sep_scaler=20., int_mid_threshold=0.5, height_filter=None)
    calib = {'h_velo_cam': np.linalg.inv(CAM_TO_VELO), 'p_cam_frame': P_RECT, 'p_velo_frame': P_VELO_FRAME}
    acc = Kitti360SemanticPointCloudAccumulator(HORIZON_M, calib, 1e3, 'resident', FILTERS, SEM_IDXS, False, bev_params)
    acc._store_args = dict(capacity=1 << 26, max_frames=1 << 14)
    T = t_new_prev() if T is None else T
    acc.pose_provider = lambda pc: T
    return acc


def make_accumulator(frame_fn, seq):
    model = ResidentSemSeg()
    pool = device_pool(frame_fn, seq, POOL, model)
    return new_accumulator(model), pool, model


class Stepper:
    """The benchmark step on one accumulator."""

    def __init__(self, acc, pool):
        self.acc, self.pool, self.n = acc, pool, 0

    def integrate(self):
        rgb, pc, _ = self.pool[self.n % len(self.pool)]
        self.n += 1
        self.acc.integrate([(rgb, pc, None)])

    def step(self, out=None):
        self.integrate()
        idx = present_index(self.acc)
        if idx is None:
            return None
        return self.acc.generate_bev_device(idx, out=out)

    def fill(self):
        while present_index(self.acc) is None or len(self.acc.poses) < 195:
            self.integrate()


# --------------------------------------------------------------------------------------------------------------------
#  side measurements (rank 0, after the headline; reported beside, never as, `value`)
# --------------------------------------------------------------------------------------------------------------------
def ring_model_pass(steps):
    """The same step on ring-model frames: steady-state time per step and per-kernel HIP-event times."""
    import torch
    from pca_amd import _lib
    acc, pool, _ = make_accumulator(ring_frame, 0)
    st = Stepper(acc, pool)
    st.fill()
    out = torch.empty((21, PX, PX), dtype=torch.float16, device='cuda')
    for _ in range(5):
        st.step(out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step(out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx = _lib.Context.get()
    ctx.profile(True)
    for _ in range(steps):
        st.step(out)
    prof = ctx.profile_read()
    ctx.profile(False)
    acc.store.check_status()
    stored = int(acc.store.offsets()[-1] - acc.store.offsets()[0])
    occ = out[14].float()                                  # road plane of 'full': != 0.5 where a cell was observed
    return {'Mpoints_per_s': N_PTS * steps / dt / 1e6, 'bev_frames_per_s': steps / dt, 'ms_per_step': 1e3 * dt / steps,
            'steps': steps, 'stored_pts': stored, 'live_frames': acc.store.n_frames,
            'observed_cells': int((occ != 0.5).sum().item()),
            'kernels_avg_us': {k: 1e3 * v[0] / v[1] for k, v in prof.items() if v[1]},
            'note': '64 beams x 1875 azimuths over a ground plane with boxes; density ~1/r^2 (contention case)'}


def k1_batched_pass(pool_frames, n_distinct, batch=64, reps=20):
    """K1 alone on `batch` frames per call (7.68 M points): the shape in which the fused project+sample+filter kernel is
    throughput- rather than launch-latency-bound.  Wall clock over `reps` back-to-back C calls on prebuilt descriptors
    (the GPU stays busy: launch gaps of the unit are included, host preparation is not), HIP events beside it.
    n_distinct > batch: the calls ALTERNATE between n_distinct / batch different batches, so that no call finds its inputs where
    the call before left them (two batches of 64 frames = 514 MB of inputs: more than the 256 MB Infinity Cache holds)."""
    import torch
    from pca_amd import _lib
    from pca_amd.device_store import DeviceStore
    ctx = _lib.Context.get()
    tmp = DeviceStore(capacity=batch * N_PTS, max_frames=batch + 1)
    n_sets = max(n_distinct // batch, 1)
    sets = []
    for i in range(n_sets):
        fr = [dict(pts=pool_frames[(i * batch + k) % n_distinct][1], rgb=pool_frames[(i * batch + k) % n_distinct][0],
                   sem=pool_frames[(i * batch + k) % n_distinct][2]) for k in range(batch)]
        sets.append((fr, DeviceStore.kitti_descs(fr)))
    frames, descs = sets[0]
    calls = [0]

    def call():
        fr, ds = sets[calls[0] % n_sets]
        calls[0] += 1
        tmp.clear()
        tmp.append_kitti(fr, P_VELO_FRAME, IMG_H, IMG_W, FILTERS, descs=ds)
    m_proj = kept = 0
    for fr, ds in sets:
        tmp.clear()
        tmp.append_kitti(fr, P_VELO_FRAME, IMG_H, IMG_W, [], descs=ds)      # no class filter: the in-frustum count M_proj
        m_proj += int(tmp.offsets()[-1])
        tmp.clear()
        tmp.append_kitti(fr, P_VELO_FRAME, IMG_H, IMG_W, FILTERS, descs=ds)
        kept += int(tmp.offsets()[-1])
    m_proj, kept = m_proj / n_sets, kept / n_sets              # per call (mean over the alternating batches)
    for _ in range(3 * n_sets):
        call()
    torch.cuda.synchronize()
    ctx.profile(True)
    for _ in range(5 * n_sets):
        call()
    ev = ctx.profile_read()['kitti_project_sample_filter']
    ctx.profile(False)
    # back-to-back: frame_off[0] stays 0, every call rewrites the same slots -- no host work between calls but ctypes
    import ctypes as C
    lib = ctx.lib
    st = tmp.c_store()
    Pc, fm = _lib.f64_array(P_VELO_FRAME, 12), _lib.class_mask(FILTERS)
    tmp.clear()
    torch.cuda.synchronize()
    reps = reps * n_sets
    t0 = time.perf_counter()
    for r in range(reps):
        ctx.check(lib.pca_kitti_project_sample_filter(ctx.h, sets[r % n_sets][1], batch, Pc, IMG_H, IMG_W, fm, C.byref(st),
                                                      tmp.frame_off.data_ptr(), 0, ctx.stream()))
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / reps
    tmp.check_status()
    alg = 16.0 * N_PTS * batch + 4.0 * m_proj + 40.0 * kept
    in_mb = n_distinct * (N_PTS * 16 + IMG_H * IMG_W * 4) / 1e6
    return {'frames_per_call': batch, 'distinct_frames': n_distinct, 'alternating_batches': n_sets, 'points_per_call': N_PTS * batch,
            'kept': kept, 'in_frustum': m_proj,
            'us_per_call_wall_back_to_back': us, 'us_per_call_hip_events': 1e3 * ev[0] / ev[1],
            'alg_bytes': alg, 'GBps': alg / us / 1e3, 'frac': alg / us / 1e3 / HBM_PEAK_GBS,
            'frac_on_hip_event_time': alg / (1e3 * ev[0] / ev[1]) / 1e3 / HBM_PEAK_GBS,
            'frac_note': '`frac` is on the back-to-back wall time (launch gaps in, per-kernel events out); the HIP-event pair '
                         'around the two kernels of one call gives frac_on_hip_event_time',
            'Mpoints_per_s': N_PTS * batch / us,
            'input_MB': in_mb,
            'inputs_cache_resident': bool(in_mb < 64e6 / 1e6),
            'inputs_fit_infinity_cache': bool(in_mb + 100.0 < 256 * 1.048576),
            'cache_note': 'the calls cycle through %.0f MB of inputs and write ~95 MB (staging + store) per call; the Infinity Cache '
                          'holds 256 MiB: below ~170 MB of inputs a call finds part of them on-die' % in_mb}


def nuscenes_pass(frames=40, reps=10):
    """BASELINE configs[2]: NuScenes kernels at full size -- 34 720 points, 6 x 900x1600 images,
    40 frames per scene, 256^2 BEV at view 51.2 m with height filter 3 m.  K0n / K1n / K3 / BEV times (HIP events)."""
    import torch
    from pca_amd import _lib
    from pca_amd.device_store import DeviceStore, make_bev_params
    import ctypes as C
    ctx = _lib.Context.get()
    lib = ctx.lib
    n, ncam, H, W = 34_720, 6, 900, 1600
    g = torch.Generator(device='cuda').manual_seed(3)
    imgs = torch.randint(0, 256, (ncam, H, W, 3), device='cuda', dtype=torch.uint8, generator=g)
    sems = torch.randint(0, 19, (ncam, H, W), device='cuda', dtype=torch.uint8, generator=g)
    rng = np.random.default_rng(3)
    pc = np.stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-2, 4, n),
                   rng.integers(0, 256, n).astype(float), rng.uniform(1.01, W - 1.01, n), rng.uniform(1.01, H - 1.01, n),
                   rng.integers(-1, 5, n).astype(float)], 1)
    cam = rng.integers(-1, ncam, n)
    pc_d, cam_d = torch.from_numpy(pc).cuda(), torch.from_numpy(cam).cuda()
    st = DeviceStore(capacity=(frames + 2) * n, max_frames=frames + 4, intensity_div255=True)
    filters = [10, 11, 12, 16, 18]

    def T_of(k):
        a = 0.002 * k
        T = np.eye(4)
        T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        T[0, 3] = 1.0 * k
        return T
    for k in range(frames):
        st.append_nusc(pc_d, cam_d, imgs, sems, T_of(k), filters)
    st.mark_dynamic([(f, 0) for f in range(min(frames, 16))])
    prm = make_bev_params(T_of(frames // 2)[:3, 3], np.eye(3), 0., 0., 51.2, 256, 3., 1., 30., 0.12, 0, [13, 14, 15, 17], True)
    out = torch.empty((21, 256, 256), dtype=torch.float16, device='cuda')
    st.bev(frames // 2, prm, out16=out)
    # K0n inputs: lidar points + 2 + 6 rigid transforms, intrinsics
    pts = torch.from_numpy(np.ascontiguousarray(pc[:, :3])).cuda()
    Tl, Tg = _lib.f64_array(np.eye(4), 16), _lib.f64_array(T_of(3), 16)
    Tc = _lib.f64_array(np.stack([np.linalg.inv(T_of(3)) for _ in range(ncam)]), 16 * ncam)
    K = np.array([[1266., 0, 816.], [0, 1266., 491.], [0, 0, 1.]])
    Kc, whc = _lib.f64_array(np.stack([K] * ncam), 9 * ncam), _lib.f64_array(np.tile([W, H], ncam).astype(float), 2 * ncam)
    ego = torch.empty((n, 3), dtype=torch.float64, device='cuda')
    uv = torch.empty((n, 2), dtype=torch.float64, device='cuda')
    ci = torch.empty(n, dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    ctx.profile(True)
    for r in range(reps):
        st.evict(1)
        st.append_nusc(pc_d, cam_d, imgs, sems, T_of(frames + r), filters)
        st.mark_dynamic([(f, 0) for f in range(16)])
        ctx.check(lib.pca_nusc_project_cams(ctx.h, pts.data_ptr(), n, Tl, Tg, Tc, Kc, whc, ncam, ego.data_ptr(),
                                            uv.data_ptr(), ci.data_ptr(), ctx.stream()))
        st.bev(frames // 2, prm, out16=out)
    prof = {k: 1e3 * v[0] / v[1] for k, v in ctx.profile_read().items() if v[1]}
    ctx.profile(False)
    st.check_status()
    sizes = st.sizes()
    m_kept = float(sizes[-1])
    m_on_cam = float((cam >= 0).sum())
    alg_k1n = 56.0 * n + 4.0 * m_on_cam + 40.0 * m_kept
    alg_k0n = 24.0 * n + 48.0 * n
    stored = int(sizes.sum())
    bev_us = sum(v for k, v in prof.items() if k.startswith('bev_'))
    alg_bev = 40.0 * stored + 21.0 * 256 * 256 * 4.0
    return {'workload': 'NuScenes-shape: %d pts x 7 f64, 6 x 900x1600 images, %d frames (%d stored pts), 256^2 BEV, view '
                        '51.2 m, height filter 3 m, (1, 30, 0.12)' % (n, frames, stored),
            'kernels_avg_us': prof,
            'k1n': {'us': prof.get('nusc_sample_filter_transform'), 'alg_bytes': alg_k1n,
                    'frac': alg_k1n / prof['nusc_sample_filter_transform'] / 1e3 / HBM_PEAK_GBS},
            'k0n': {'us': prof.get('nusc_project_cams'), 'alg_bytes': alg_k0n,
                    'frac': alg_k0n / prof['nusc_project_cams'] / 1e3 / HBM_PEAK_GBS},
            'k3_us': prof.get('mark_dynamic'),
            'bev_unit': {'us_sum_of_kernels': bev_us, 'alg_bytes': alg_bev, 'frac': alg_bev / bev_us / 1e3 / HBM_PEAK_GBS},
            'note': 'single small launches: latency-, not bandwidth-bound at 35 k points per frame'}


def nusc_sweep_rows(k, n_az=1085, n_beam=32, ncam=6, H=900, W=1600):
    """One NuScenes-shaped observation in the ORDER a spinning 32-beam lidar delivers it (nuscenes_obs_dataloader.py:162-202
    of the reference keeps the sweep's order): azimuth steps in turn, the 32 beams of a step together; range from a ground
    plane 1.84 m below the sensor, walls at 30 m above the horizon.  Six cameras of 60 degrees tile the circle; a point's
    pixel follows from its azimuth inside its camera's sector (u) and its elevation (v) -- so neighbours in the array are
    neighbours in the image, unlike SURVEY 8d's uniformly drawn pixel coordinates.  Returns ((n,7) rows, (n,) camera index)."""
    rng = np.random.default_rng(4000 + k)
    az = np.repeat(2.0 * np.pi * (np.arange(n_az) + 0.37 * (k % 3)) / n_az, n_beam)
    el = np.tile(np.deg2rad(-30.67 + (41.33 / (n_beam - 1)) * np.arange(n_beam)), n_az)
    el = el + rng.normal(0.0, 1e-4, el.shape)
    r_ground = 1.84 / np.maximum(np.sin(-el), 1e-3)
    r = np.where(el < -0.03, np.minimum(r_ground, 60.0), 30.0) * (1.0 + 0.01 * rng.standard_normal(el.shape))
    x, y, z = r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)
    sector = 2.0 * np.pi / ncam
    cam = np.floor(az / sector).astype(np.int64) % ncam
    rel = az - (cam + 0.5) * sector                        # angle off the camera's axis, |rel| <= 30 degrees
    f = (W / 2.0) / np.tan(sector / 2.0)
    u = W / 2.0 + f * np.tan(rel)
    v = H / 2.0 - f * np.tan(el) / np.cos(rel)
    on = (u > 1.01) & (u < W - 1.01) & (v > 1.01) & (v < H - 1.01)
    cam = np.where(on, cam, -1)
    u, v = np.clip(u, 1.01, W - 1.01), np.clip(v, 1.01, H - 1.01)
    n = az.shape[0]
    rows = np.stack([x, y, z, rng.integers(0, 256, n).astype(float), u, v, rng.integers(-1, 5, n).astype(float)], 1)
    return rows, cam


def nuscenes_scene_pass(frames=40, reps=5, order='uniform', forms=None):
    """BASELINE configs[2] end to end through the drop-in NuScenesOracleSemanticPointCloudAccumulator: one synthetic scene
    (40 frames x 34 720 points, 6 x 900x1600 images, two GT instances, one of them moving) is
    integrated and swept for BEV samples as run_nuscenes_bev_gen.py:234-271 does it (conditions 1-3 with a 10 m horizon,
    1 m spacing).  Three forms, wall clock per scene:
      batched   device-resident observations, integrate_many (ONE K1n front + append launch, one K3 launch) and
                generate_bev_many (ONE launch of each raster kernel for all samples), planes stay in HBM until awaited;
      stepwise  the same observations through integrate() per frame and generate_bev() per sample (the unchanged driver's
                call sequence on device inputs);
      pcie      host arrays in (6 x 4.3 MB images + points per frame), host fp16 dicts out, every sample awaited."""
    import torch
    import sem_pc_accum
    from nuscenes_oracle_sem_pc_accum import NuScenesOracleSemanticPointCloudAccumulator
    from pca_amd import _lib
    from pca_amd.ingest import DeviceImages
    n, ncam, H, W = 34_720, 6, 900, 1600
    gdev = torch.Generator(device='cuda').manual_seed(33)
    stacks = [(torch.randint(0, 256, (ncam, H, W, 3), device='cuda', dtype=torch.uint8, generator=gdev),
               torch.randint(0, 19, (ncam, H, W), device='cuda', dtype=torch.uint8, generator=gdev)) for _ in range(4)]
    host_stacks = [(i.cpu().numpy(), s.cpu().numpy()) for i, s in stacks]
    rng = np.random.default_rng(33)
    filters = [10, 11, 12, 16, 18]

    class Model:                                             # stand-in for the CNN: the class map resident where the image is
        def __init__(self):
            self.by_id = {}

        def pred(self, rgb):
            return self.by_id[id(rgb)][None, None]
    model = Model()
    dev_obs, host_obs = [], []
    for k in range(frames):
        if order == 'sweep':
            pc, cam = nusc_sweep_rows(k)
        else:
            pc = np.stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-2, 4, n),
                           rng.integers(0, 256, n).astype(float), rng.uniform(1.01, W - 1.01, n), rng.uniform(1.01, H - 1.01, n),
                           rng.integers(-1, 5, n).astype(float)], 1)
            cam = rng.integers(-1, ncam, n)
        a = 0.002 * k
        T = np.eye(4)
        T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        T[:3, 3] = [1000.0 + 1.0 * k, 500.0, 0.0]
        meta = dict(ego_at_lidar_ts=T, ego_global_x=T[0, 3], ego_global_y=T[1, 3], inst_tokens=['car_a', 'car_b'],
                    inst_cls=[0, 0], inst_center=[np.array([1010.0 + 0.5 * k, 505.0, 0.5]), np.array([1020.0, 495.0, 0.5])])
        imgs_d, sems_d = stacks[k % 4]
        handles = [object() for _ in range(ncam)]            # what the model is asked about, camera by camera
        for j, h in enumerate(handles):
            model.by_id[id(h)] = sems_d[j]
        dev_obs.append([dict(meta, images=DeviceImages(handles, imgs_d), pc=torch.from_numpy(pc).cuda(),
                             pc_cam_idx=torch.from_numpy(cam).cuda())])
        imgs_h, sems_h = host_stacks[k % 4]
        himgs = [imgs_h[j] for j in range(ncam)]
        for j, im in enumerate(himgs):
            model.by_id[id(im)] = sems_h[j]
        host_obs.append([dict(meta, images=himgs, pc=pc, pc_cam_idx=cam)])
    sem_pc_accum.SemSegONNX = lambda path: model
    bev_params = dict(type='sem', view_size=51.2, pixel_size=256, max_trans_radius=0., zoom_thresh=0., do_warp=False,
                      int_scaler=1., int_sep_scaler=30., int_mid_threshold=0.12, height_filter=3.)

    def new_acc():
        acc = NuScenesOracleSemanticPointCloudAccumulator('resident', filters, SEM_IDXS, False, dict(bev_params), 'synthetic',
                                                          False, None)
        acc._store_args = dict(intensity_div255=True, capacity=(frames + 2) * n, max_frames=frames + 8)
        return acc

    def sample_idxs(acc, horizon=10.0, spacing=1.0):        # the driver's sweep, run_nuscenes_bev_gen.py:242-262
        d = acc.get_incremental_path_dists()
        out, prev = [], 0
        for idx in range(len(acc.poses) - 1):
            if d[idx] < horizon or d[-1] - d[idx] < horizon:
                continue
            if acc.dist(acc.get_pose(prev), acc.get_pose(idx)) < spacing:
                continue
            prev = idx
            out.append(idx)
        return out

    def batched():
        acc = new_acc()
        acc.integrate_many(dev_obs)
        bevs = acc.generate_bev_many(sample_idxs(acc), gen_future=True)
        assert bevs[-1]['rgb_full'].shape == (3, 256, 256)               # awaits the one copy of all samples
        return acc, len(bevs)

    def stepwise(obs):
        acc = new_acc()
        for o in obs:
            acc.integrate(o)
        bevs = [acc.generate_bev(idx, 1, gen_future=True)[0] for idx in sample_idxs(acc)]
        for b in bevs:
            assert b['rgb_full'].shape == (3, 256, 256)
        return acc, len(bevs)
    out = {'workload': '%d frames x %d points x 7 f64, 6 x %dx%d images per frame, 256^2 BEV at view 51.2 m, height filter 3 m; '
                       'sample sweep with a 10 m horizon and 1 m spacing' % (frames, n, H, W)}
    ctx = _lib.Context.get()
    all_forms = (('batched', batched), ('stepwise', lambda: stepwise(dev_obs)), ('pcie', lambda: stepwise(host_obs)))
    want = forms if forms is not None else (('batched', 'stepwise', 'pcie') if order == 'uniform' else ('batched', ))
    for name, fn in [f for f in all_forms if f[0] in want]:
        fn()                                                             # warm-up (allocations, pinned blocks)
        torch.cuda.synchronize()
        times = []
        for _ in range(reps if name != 'pcie' else 2):
            t0 = time.perf_counter()
            acc, n_bev = fn()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        dt = float(np.median(times))
        out[name] = {'ms_per_scene': 1e3 * dt, 'Mpoints_per_s': frames * n / dt / 1e6, 'bev_frames_per_s': n_bev / dt,
                     'bev_samples': n_bev}
    if 'batched' not in want:
        return out
    # the batched K1n unit by HIP events (front + append) against its algorithmic bytes
    acc = new_acc()
    acc.integrate_many(dev_obs)
    torch.cuda.synchronize()
    ctx.profile(True)
    for _ in range(reps):
        acc = new_acc()
        acc.integrate_many(dev_obs)
    ev = ctx.profile_read()['nusc_sample_filter_transform']
    ctx.profile(False)
    sizes = acc.store.sizes()
    m_on_cam = sum(int((o[0]['pc_cam_idx'] >= 0).sum().item()) for o in dev_obs)
    alg = 56.0 * n * frames + 4.0 * m_on_cam + 40.0 * float(sizes.sum())
    us = 1e3 * ev[0] / ev[1]
    out['k1n_batched'] = {'frames_per_call': frames, 'us_per_call_hip_events': us, 'alg_bytes': alg, 'GBps': alg / us / 1e3,
                          'frac': alg / us / 1e3 / HBM_PEAK_GBS, 'kept': int(sizes.sum())}
    ctx.profile(2)
    acc.generate_bev_many(sample_idxs(acc), gen_future=True)
    n_bev = len(sample_idxs(acc))
    for _ in range(reps):
        acc.generate_bev_many(sample_idxs(acc), gen_future=True)
    unit = ctx.profile_read()['bev_unit']
    ctx.profile(False)
    stored = int(sizes.sum())
    alg_bev = n_bev * (40.0 * stored + 21.0 * 256 * 256 * 4.0)
    us = 1e3 * unit[0] / unit[1]
    out['bev_many'] = {'samples_per_call': n_bev, 'us_per_call_hip_events': us, 'us_per_sample': us / max(n_bev, 1),
                       'alg_bytes': alg_bev, 'frac': alg_bev / us / 1e3 / HBM_PEAK_GBS}
    return out


def config4_pass(frames=100, n=1_000_000, px=512, view=160.0):
    """BASELINE configs[3] with the frame count scaled to the bench's time budget: `frames` x 1 M
    points, every point kept, 512^2 grid, view 160 m.  One steady-state step = evict + owed re-transform fused into the
    BEV + append of 1 M points + BEV over the whole window.  (Full size, 1000 frames = 1e9 points = 37 GB:
    tests/test_gpu_kernels.py and tools/experiments/config4_time.py.)"""
    import torch as T
    from pca_amd import _lib
    from pca_amd.device_store import DeviceStore, make_bev_params
    st = DeviceStore(capacity=(frames + 40) * n, max_frames=frames + 64)     # room for every step below: no slide
    g = T.Generator(device='cuda').manual_seed(4)
    classes = T.tensor([0, 1, 2, 8, 9, 13, 14], device='cuda', dtype=T.uint8)
    P = np.eye(4)[:3]

    def frame():
        pts = T.empty((n, 4), device='cuda', dtype=T.float32)
        pts[:, :2] = (T.rand((n, 2), device='cuda', generator=g) * 160 - 80).float()
        pts[:, 2] = (T.rand(n, device='cuda', generator=g) * 5 - 2).float()
        pts[:, 3] = T.rand(n, device='cuda', generator=g).float()
        return dict(pts=pts.contiguous(), sem_gt=classes[T.randint(0, 7, (n, ), device='cuda', generator=g)])
    Tm = np.eye(4)
    Tm[0, 3] = -0.03125
    pool = [frame() for _ in range(4)]
    for k in range(frames):
        if k:
            st.retransform(Tm)
        st.append_kitti([pool[k % 4]], P, 1, 1, [255])
    prm = make_bev_params((0.25, -0.5, 0.0), np.eye(3), 0., 0., view, px, None, 20., 20., 0.5, 0, [13, 14, 15, 17], False)
    out = T.empty((21, px, px), dtype=T.float16, device='cuda')
    ctx = _lib.Context.get()

    def step(k):
        st.evict(1)
        st.retransform(Tm, defer=True)
        st.append_kitti([pool[k % 4]], P, 1, 1, [255])
        st.bev(st.n_frames // 2, prm, out16=out)
    for k in range(3):
        step(k)
    T.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for k in range(reps):
        step(k)
    T.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ctx.profile(1)
    for k in range(5):
        step(k)
    prof = {k: 1e3 * v[0] / v[1] for k, v in ctx.profile_read().items() if v[1]}
    ctx.profile(0)
    st.check_status()
    stored = int(st.offsets()[-1] - st.offsets()[0])
    alg = 40.0 * stored + 21 * px * px * 4 + 48.0 * (stored - n)
    unit = sum(v for k, v in prof.items() if k.startswith('bev_'))
    return {'workload': '%d frames x %d points (all kept), %d^2 grid, view %.0f m; %d stored points = %.1f GB'
                        % (frames, n, px, view, stored, stored * 37 / 1e9),
            'ms_per_step': 1e3 * dt, 'Mpoints_per_s': n / dt / 1e6, 'bev_frames_per_s': 1 / dt,
            'kernels_avg_us': prof,
            'roofline_bev_unit': {'us_sum_of_kernels': unit, 'alg_bytes': alg, 'GBps': alg / unit / 1e3,
                                  'frac': alg / unit / 1e3 / HBM_PEAK_GBS}}


def extras_pass(acc):
    """Timings of the opt-in / next-row pieces on the benchmark's own data: voxel de-duplication of the full window and
    the device ICP on two frames."""
    import torch
    from pca_amd.icp import GpuIcp
    import warnings
    out = {}
    st = acc.store
    before = int(st.offsets()[-1] - st.offsets()[0])
    st.voxel_dedup(0.1)                                   # warm-up (allocates the table)
    torch.cuda.synchronize()
    mid = int(st.offsets()[-1] - st.offsets()[0])
    t0 = time.perf_counter()
    st.voxel_dedup(0.1)
    torch.cuda.synchronize()
    out['voxel_dedup'] = {'voxel_m': 0.1, 'points_before': before, 'points_after': mid,
                          'ms_second_pass_over_%d_pts' % mid: 1e3 * (time.perf_counter() - t0)}
    a, b = GpuIcp.to_device(ring_frame(0, 3)[0]), GpuIcp.to_device(ring_frame(0, 4)[0])
    icp = GpuIcp()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        icp.register(a, b, 1e3, np.eye(4))
        t0 = time.perf_counter()
        r = icp.register(a, b, 1e3, np.eye(4))
    out['device_icp'] = {'ms_per_registration': 1e3 * (time.perf_counter() - t0), 'points': N_PTS,
                         'iterations': r.iterations, 'scene': 'two consecutive ring-model frames'}
    return out


def pcie_inclusive_pass(acc, pool, n_steps):
    """The step as the UNCHANGED drivers run it: observations start as host arrays (PIL-like image, (N,4) f32 points) and
    every BEV sample ends as a host fp16 dict -- per step 4 MB H2D + 2.75 MB D2H.  Two forms:
      plain     integrate(host arrays) + generate_bev(), every sample's planes awaited before the next step;
      deferred  the same calls, but a sample's planes (they leave on a side stream into pinned memory: LazyBev) are only
                touched one step later -- what the driver gets when it hands the dict to write_compressed_pickle, whose
                background writer collects it then (no disk, no gzip in this number).
    Both are bound by the host: Python + ctypes through the drop-in classes, plus the staging copies (pca_host_stage_h2d:
    pinned copies on the library's thread pool).  One warming repetition, then the median of three."""
    import torch
    host_pool = [(f[0].cpu().numpy(), f[1].cpu().numpy(), f[2].cpu().numpy()) for f in pool]
    cur = {'k': 0}

    class HostSemSeg:                       # stand-in for the CNN: hands back the host map of the current frame
        def pred(self, rgb):
            return host_pool[cur['k'] % len(pool)][2][None, None]
    model = acc.semseg_model
    acc.semseg_model = HostSemSeg()
    out = {'H2D_MB_per_step': (N_PTS * 16 + IMG_H * IMG_W * 4) / 1e6, 'D2H_MB_per_step': 21 * PX * PX * 2 / 1e6}
    for name in ('plain', 'deferred'):
        times = []
        for rep in range(4):                # the first repetition warms the pinned blocks of the copies; median of the rest
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            parked = None
            for k in range(n_steps):
                rgb_h, pc_h, _ = host_pool[k % len(host_pool)]
                cur['k'] = k
                acc.integrate([(rgb_h, pc_h, None)])
                bev = acc.generate_bev(present_index(acc), 1, gen_future=True)[0]
                if name == 'deferred':
                    bev, parked = parked, bev
                if bev is not None:
                    assert bev['rgb_full'].shape == (3, PX, PX)            # waits for that sample's planes
            if parked is not None:
                assert parked['rgb_full'].shape == (3, PX, PX)
            torch.cuda.synchronize()
            if rep > 0:
                times.append(time.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]
        out[name] = {'Mpoints_per_s': N_PTS * n_steps / dt / 1e6, 'bev_frames_per_s': n_steps / dt, 'ms_per_step': 1e3 * dt / n_steps,
                     'ms_per_step_repeats': [round(1e3 * t / n_steps, 4) for t in times]}
    acc.semseg_model = model
    out.update(Mpoints_per_s=out['deferred']['Mpoints_per_s'], bev_frames_per_s=out['deferred']['bev_frames_per_s'],
               ms_per_step=out['deferred']['ms_per_step'], steps=n_steps,
               note='headline of this block = the deferred form; host-bound (Python), not PCIe-bound')
    return out


def two_sequences_pass(steps, n_lanes=2, reps=9):
    """TWO (n_lanes) independent sequences on the one GPU, each with its own pca_ctx and stream (pca_amd._lib.Lane), each
    driven by its own host thread (ctypes releases the GIL inside a library call): the headline step, unchanged, on every
    lane at once.  What it measures is the capacity a single sequence leaves idle -- its kernels fill the step's time but
    end in tails (level 1: two rounds of one workgroup per CU; the tile kernel: as long as its densest tile) that another
    sequence's workgroups can fill.  Independent sequences are what the reference's driver loops over
    (run_kitti360_bev_gen.py:161-173).  Aggregate = all lanes' points over the time from a common start until the LAST lane
    has finished (device included); median of `reps` repetitions of `steps` steps per lane."""
    import threading
    import torch
    from pca_amd import _lib
    lanes = [_lib.Lane() for _ in range(n_lanes)]
    steppers, outs = [], []
    for k, lane in enumerate(lanes):                          # set-up one lane after the other, on this thread
        with lane:
            acc, pool, _ = make_accumulator(synth_frame, 20 + k)
            st = Stepper(acc, pool)
            st.fill()
            out = torch.empty((steps, 21, PX, PX), dtype=torch.float16, device='cuda')
            for i in range(3):
                st.step(out[i % steps])
            lane.synchronize()
            steppers.append(st)
            outs.append(out)
    torch.cuda.synchronize()
    start = threading.Barrier(n_lanes + 1)
    done = threading.Barrier(n_lanes + 1)
    lane_s = [[0.0] * (reps + 1) for _ in range(n_lanes)]
    errors = []

    def run(k):
        try:
            with lanes[k]:
                for r in range(reps + 1):
                    start.wait()
                    t0 = time.perf_counter()
                    for i in range(steps):
                        steppers[k].step(outs[k][i])
                    lanes[k].synchronize()
                    lane_s[k][r] = time.perf_counter() - t0
                    done.wait()
        except Exception as e:                                 # noqa: BLE001  (reported by the main thread)
            errors.append(repr(e))
            start.abort()
            done.abort()
    threads = [threading.Thread(target=run, args=(k, ), daemon=True) for k in range(n_lanes)]
    for t in threads:
        t.start()
    wall = []
    try:
        for r in range(reps + 1):
            start.wait()
            t0 = time.perf_counter()
            done.wait()
            wall.append(time.perf_counter() - t0)
    except threading.BrokenBarrierError:
        pass
    for t in threads:
        t.join(timeout=60)
    if errors:
        return {'error': errors[0][:300]}
    for k, lane in enumerate(lanes):
        with lane:
            steppers[k].acc.store.check_status()
    wall = wall[1:]                                            # the first repetition warms up
    dt = float(np.median(wall))
    per_lane = [float(np.median(ls[1:])) for ls in lane_s]
    # the same on ONE lane, measured the same way right here (same box, same thermal state): the ratio's denominator
    with lanes[0]:
        one = []
        for r in range(reps + 1):
            lanes[0].synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                steppers[0].step(outs[0][i])
            lanes[0].synchronize()
            one.append(time.perf_counter() - t0)
        steppers[0].acc.store.check_status()
    one_dt = float(np.median(one[1:]))
    checks = [int(o.view(torch.int16).to(torch.int64).sum().item()) for o in outs]
    return {'lanes': n_lanes, 'driven_by': 'one host thread per lane (own pca_ctx + stream each)', 'steps_per_lane': steps,
            'repetitions': reps, 'Mpoints_per_s': n_lanes * N_PTS * steps / dt / 1e6, 'bev_frames_per_s': n_lanes * steps / dt,
            'ms_per_step_per_sequence': [1e3 * t / steps for t in per_lane], 'ms_per_step_aggregate': 1e3 * dt / steps / n_lanes,
            'one_lane_alone': {'Mpoints_per_s': N_PTS * steps / one_dt / 1e6, 'ms_per_step': 1e3 * one_dt / steps},
            'aggregate_over_one_lane': one_dt * n_lanes / dt, 'plane_checksums': checks}


def two_ranks_one_gpu_pass(steps, warmup):
    """The same question answered with two PROCESSES (no GIL between the sequences): bench.py itself as two gloo ranks that
    share the one GPU (the rehearsal path of tests/test_gpu_bench.py), each rank its own sequence; `value` of that run is the
    aggregate over both.  This process sleeps meanwhile (its memory stays allocated, its streams are idle)."""
    env = dict(os.environ, PCA_BENCH_BACKEND='gloo', PCA_BENCH_CHILD='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '2', '--steps', str(steps), '--warmup', str(warmup),
           '--no-extras', '--no-cpu-baseline']
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        return {'error': 'timeout'}
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{') and '"metric"' in ln]
    if p.returncode != 0 or not line:
        return {'error': (p.stderr or p.stdout)[-300:]}
    d = json.loads(line[-1])
    return {'ranks': 2, 'backend': 'gloo (two processes on ONE GPU; no data-path collective)', 'Mpoints_per_s': d['value'],
            'bev_frames_per_s': d['bev_frames_per_s'], 'ms_per_step_per_sequence': d['ms_per_step'],
            'repeats': d.get('repeats', {}).get('n'), 'wall_s_of_the_child_run': time.perf_counter() - t0}


def kitti_icp_flow_pass(steps=30):
    """The KITTI flow as the reference's own driver runs it (run_kitti360_bev_gen.py:146-155 passes no poses: every frame goes
    through ICP, kitti360_sem_pc_accum.py:115-127): integrate() with the device ICP as pose source (PCA_POSE_PROVIDER=gpu_icp:
    previous sweep -> new sweep, point-to-plane) on CONSECUTIVE ring-model frames from HOST arrays, one BEV per frame, planes
    back on the host one step later (LazyBev).  The pose chain is sequential, so this flow cannot be chunk-sharded; its step
    is bound by the registration.  The street of the ring model repeats every 25 m, so 25 frames cycle into a 1 m / frame drive."""
    import torch
    host_pool = [ring_frame(3, k) for k in range(25)]
    cur = {'k': 0}

    class HostSemSeg:
        def pred(self, rgb):
            return host_pool[cur['k'] % 25][2][None, None]
    acc = new_accumulator(HostSemSeg())
    icp_s = [0.0, 0]
    provider = acc._gpu_icp_pose

    def timed_pose(pc):
        t0 = time.perf_counter()
        T = provider(pc)
        icp_s[0] += time.perf_counter() - t0
        icp_s[1] += 1
        return T
    acc.pose_provider = timed_pose

    def step(k):
        pc, img, _ = host_pool[k % 25]
        cur['k'] = k
        acc.integrate([(img, pc, None)])
        idx = present_index(acc)
        return None if idx is None else acc.generate_bev(idx, 1, gen_future=True)[0]
    k = 0
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')                        # (the 4 m search cap of the device ICP: said once per process)
        while present_index(acc) is None or len(acc.poses) < 150:
            step(k)
            k += 1
            if k > 400:
                break
        filled = k
        for _ in range(3):
            step(k)
            k += 1
        torch.cuda.synchronize()
        icp_s[0], icp_s[1] = 0.0, 0
        Ts = []
        t0 = time.perf_counter()
        parked = None
        for _ in range(steps):
            bev = step(k)
            k += 1
            bev, parked = parked, bev
            if bev is not None:
                assert bev['rgb_full'].shape == (3, PX, PX)
            Ts.append(acc.T_prev_origin.copy())
        if parked is not None:
            assert parked['rgb_full'].shape == (3, PX, PX)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    acc.store.check_status()
    # the drive the registrations recovered: the scene slides by 1 m per frame along x
    adv = [float(np.linalg.norm((np.linalg.inv(Ts[i]) @ Ts[i + 1])[:3, 3])) for i in range(len(Ts) - 1)]
    return {'workload': 'integrate() with PCA_POSE_PROVIDER=gpu_icp on consecutive ring-model frames (120 k points, host arrays) '
                        '+ one 256^2 BEV per frame, planes to the host one step later; %d live frames' % acc.store.n_frames,
            'steps': steps, 'frames_to_fill': filled, 'ms_per_step': 1e3 * dt / steps, 'Mpoints_per_s': N_PTS * steps / dt / 1e6,
            'bev_frames_per_s': steps / dt, 'icp_ms_per_registration': 1e3 * icp_s[0] / max(icp_s[1], 1),
            'icp_share_of_step': icp_s[0] / dt, 'metres_per_frame_recovered_mean': float(np.mean(adv)) if adv else None,
            'metres_per_frame_recovered_minmax': [float(np.min(adv)), float(np.max(adv))] if adv else None,
            'note': 'sequential pose chain: not chunk-shardable; whole sequences still shard over the GPUs',
            'note_recovered_motion': 'the scene slides 1 m per frame; point-to-plane ICP recovers ~0.85 m of it on THIS synthetic scene, device '
                                     'and CPU alike (the same algorithm with exact, uncapped k-d-tree neighbours -- tests/test_gpu_icp.py:icp_model -- '
                                     'gives 0.838 m on frames 3 -> 4): the ground rings travel with the sensor and pull towards zero motion.  '
                                     'Known motions on a ray-cast scene with world-fixed structure are recovered to 3 cm (tests/test_gpu_icp.py)'}


# --------------------------------------------------------------------------------------------------------------------
#  BASELINE configs[4]: the nine KITTI-360 sequences sharded over the ranks (strong scaling)
# --------------------------------------------------------------------------------------------------------------------
def config5_pass(rank, world, scale, barrier, allmax, coll_dev, dist, lanes_per_gpu=1):
    """Every rank plans the whole job (host replay of the sample trigger, identical everywhere), runs its own chunks --
    warm-up prefix through integrate_many, then integrate + BEV at the sample jobs -- and keeps its BEVs in HBM; the last
    GATHER_CHUNK samples of every rank go to rank 0 as a content check (checksums compared).  Timed: barrier .. barrier
    around the compute of all ranks."""
    import torch
    from pca_amd import sharded_run as sr
    from pca_amd import shard
    lengths = [max(int(round(n * scale)), 2) for n in KITTI360_LENGTHS]
    T = t_new_prev()
    seq_Ts = [np.tile(T, (n, 1, 1)) for n in lengths]
    t0 = time.perf_counter()
    jobs, loads, samples = sr.plan(seq_Ts, world, HORIZON_M, float(BEV_HORIZON_M), 1.0)
    plan_s = time.perf_counter() - t0
    model = ResidentSemSeg()
    pools = {}

    def pool_of(seq):
        if seq not in pools:
            pools[seq] = device_pool(synth_frame, 100 + seq, POOL, model)
        return pools[seq]
    for j in jobs[rank]:
        pool_of(j.seq)
    # (every lane keeps its last GATHER_CHUNK samples; lane 0's are what the content check sends to rank 0)
    n_lanes = max(1, min(int(lanes_per_gpu), len(jobs[rank]))) if jobs[rank] else 1
    rings = [torch.empty((GATHER_CHUNK, 21, PX, PX), dtype=torch.float16, device='cuda') for _ in range(n_lanes)]
    ring = rings[0]
    counts = [0] * n_lanes
    torch.cuda.synchronize()

    def run_job(job, lane):
        acc = new_accumulator(model, T)
        pool = pool_of(job.seq)

        def get_obs(f):
            rgb, pc, _ = pool[f % POOL]
            return [(rgb, pc, None)]

        def on_sample(f, present_idx):
            acc.generate_bev_device(present_idx, out=rings[lane][counts[lane] % GATHER_CHUNK])
            counts[lane] += 1
        sr.run_chunk(acc, get_obs, job, on_sample, warm_batch=64)
        acc.store.check_status()

    def run():
        sr.run_on_lanes(jobs[rank], run_job, n_lanes)
    import builtins
    real_print, builtins.print = builtins.print, (lambda *a, **k: None)
    try:
        barrier()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        mine_s = time.perf_counter() - t0                  # this rank's own wall time (an imbalance shows here)
        barrier()
        elapsed = allmax(time.perf_counter() - t0)
    finally:
        builtins.print = real_print
    total_frames, total_samples = sum(lengths), sum(len(s) for s in samples)
    my_frames = sum(j.cost for j in jobs[rank])
    out = {'workload': 'nine KITTI-360 sequence lengths x %.3g = %s frames of 120k points, sample trigger of the driver '
                       '(80 m / 80 m / 1 m), cut into chunks with a warm-up prefix' % (scale, lengths),
           'scaling': 'strong', 'n_gpus': world, 'frames': total_frames, 'bev_samples': total_samples,
           'seconds': elapsed, 'Mpoints_per_s': total_frames * N_PTS / elapsed / 1e6,
           'bev_frames_per_s': total_samples / elapsed,
           'frames_incl_warmup_per_rank': loads, 'plan_seconds_host': plan_s, 'lanes_per_gpu': n_lanes,
           'ideal_speedup_of_this_plan': float(total_frames) / max(loads),
           'peak_hbm_GB_torch_allocator': torch.cuda.max_memory_allocated() / 1e9}
    # content check: the last chunk of every rank -> rank 0, checksums compared.  A failure of the check's own collectives
    # is reported in the block, it does not take the measurement above down with it.
    mine = ring.view(torch.int16).to(torch.int64).sum().reshape(1)
    use_dist = world > 1 or dist.is_initialized()
    if use_dist:
        ok = None
        try:
            sums = [torch.zeros(1, dtype=torch.int64, device=coll_dev) for _ in range(world)]
            dist.all_gather(sums, mine.to(coll_dev))
            t0 = time.perf_counter()
            got = shard.gather_to_rank0(ring if coll_dev == 'cuda' else ring.cpu(), sizes=[GATHER_CHUNK] * world)
            barrier()
            tg = time.perf_counter() - t0
            if rank == 0:
                ok = all(int(g.view(torch.int16).to(torch.int64).sum().item()) == int(s.item()) for g, s in zip(got, sums))
                mb = (world - 1) * GATHER_CHUNK * 21 * PX * PX * 2 / 1e6
                out['gather_check'] = {'samples_per_rank': GATHER_CHUNK, 'MB_into_rank0': mb, 'ms': 1e3 * tg,
                                       'GBps': mb / 1e3 / tg, 'checksums_match': bool(ok), 'backend': dist.get_backend(),
                                       'on_device_tensors': coll_dev == 'cuda'}
        except RuntimeError as e:                                   # torch.distributed / RCCL errors
            out['gather_check'] = {'error': repr(e)[:300]}
        assert ok is not False, 'gathered BEV tensors differ from what the ranks produced'
    out['frames_this_rank_incl_warmup'] = my_frames
    # per-rank wall times of the compute (before the closing barrier): the plan's imbalance as it was measured
    if use_dist:
        try:
            ts = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
            dist.all_gather(ts, torch.tensor([mine_s], dtype=torch.float64, device=coll_dev))
            out['seconds_per_rank'] = [float(t.item()) for t in ts]
        except RuntimeError as e:
            out['seconds_per_rank'] = {'error': repr(e)[:200]}
    else:
        out['seconds_per_rank'] = [mine_s]
    return out


def config5_headline(c5):
    """What a reader of SCALE_rNN.json wants from the strong-scaling block without opening it: how well the plan COULD
    scale, and how uneven the ranks' own wall times were."""
    out = {'config5_ideal_speedup_of_this_plan': c5.get('ideal_speedup_of_this_plan')}
    spr = c5.get('seconds_per_rank')
    if isinstance(spr, list) and spr:
        out['config5_seconds_per_rank_min'] = float(min(spr))
        out['config5_seconds_per_rank_max'] = float(max(spr))
    return out


# --------------------------------------------------------------------------------------------------------------------
#  CPU baselines
# --------------------------------------------------------------------------------------------------------------------
def cpu_baseline(steps=20):
    """(1) the C oracle (scalar port of the reference algorithm, 1 core) on a bounded sample of the same workload: fill the
    200-frame window, then time `steps` full steps; (2) the numpy-shaped restatement (oracle/numpy_shape.py: the
    reference's own algorithmic shape, all host cores through BLAS where numpy uses it) on the same window."""
    from oracle import numpy_shape as ns
    from oracle import oracle as orc
    from pca_amd import host_logic as hl
    T = t_new_prev()
    frames = [synth_frame(0, k) for k in range(2)]
    st = orc.Store(220 * 40000)
    track = hl.PoseTrack()
    sizes = []
    lo = 0
    last = {}

    def step(k, do_bev):
        nonlocal lo, sizes
        pc, img, sem = frames[k % 2]
        if track.poses:
            track.apply_transform(T)
            orc.retransform(st, T, lo, st.n)
        sizes.append(orc.kitti_project_sample_filter(st, pc, P_VELO_FRAME, img, sem, None, IMG_H, IMG_W, FILTERS))
        track.append([0., 0., 0.])
        if len(track.poses) > 1:
            ev = track.evict_beyond(HORIZON_M, track.push_segment())
            lo += int(np.sum(sizes[:ev]))
            sizes = sizes[ev:]
        if do_bev:
            d = hl.incremental_path_dists(track.seg_array())
            idx = int(((d - BEV_HORIZON_M) > 0).argmax())
            origin = np.array(track.poses[idx])
            ego = np.array(track.poses[:idx]) - origin
            R = hl.rotation_matrix_3d(hl.heading_rot_ang(ego))
            prm = orc.make_bev_params(origin, R, 0., 0., VIEW_M, PX, None, 20., 20., 0.5, 0, [13, 14, 15, 17], False)
            sub = orc.Store(1)
            for name in ('x', 'y', 'z', 'intensity', 'rgbs', 'inst', 'dyn'):
                setattr(sub, name, getattr(st, name)[lo:st.n])
            sub.n = sub.cap = st.n - lo
            orc.bev(sub, int(np.sum(sizes[:idx])), prm)
            last.update(idx=idx, origin=origin, R=R)

    for k in range(205):
        step(k, False)
    t0 = time.perf_counter()
    for k in range(steps):
        step(205 + k, True)
    dt = time.perf_counter() - t0
    out = {'value': N_PTS * steps / dt / 1e6, 'unit': 'Mpoints/s', 'bev_frames_per_s': steps / dt, 'cores': 1,
           'kind': 'port',
           'sample': f'oracle/pca_oracle.c (scalar C restatement of the reference algorithm, ONE core; not the '
                     f'numpy-shaped form, which is reported under numpy_shape), 205-frame window fill untimed, then {steps} '
                     f'full steps (retransform ~{st.n - lo} stored pts + integrate 120k pts + one 256x256 BEV) in {dt:.1f} s',
           'host_cpus': os.cpu_count()}
    # ---- numpy-shaped: same window as a list of (M,10) frames -------------------------------------------------------
    try:
        from threadpoolctl import threadpool_info
        blas = [{'api': p.get('internal_api'), 'threads': p.get('num_threads')} for p in threadpool_info()]
    except Exception:
        blas = None
    rows = st.rows(lo, st.n)
    edges = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    window = [rows[a:b].copy() for a, b in zip(edges[:-1], edges[1:])]
    pc, img, sem = frames[0]
    n_steps = 2
    t0 = time.perf_counter()
    for _ in range(n_steps):
        ns.retransform(window, T)
        window.append(ns.integrate_frame(pc.astype(np.float64), P_VELO_FRAME, img, sem, FILTERS))
        window.pop(0)
        ns.bev(window, last['idx'], last['origin'], last['R'], VIEW_M, PX, 0, [13, 14, 15, 17], (20., 20., 0.5))
    dtn = (time.perf_counter() - t0) / n_steps
    # the reference's own loop shape for the two per-cell reductions, on every 8th point of the window (bounded sample)
    sub = np.concatenate(window)[::8].copy()
    sub[:, :3] -= last['origin']
    g = ns._prep(sub, last['R'], VIEW_M, PX)
    t0 = time.perf_counter()
    ns._min_z_loops(g, PX)
    ns._median_loops(g, PX, 4)
    dtl = time.perf_counter() - t0
    out['numpy_shape'] = {
        'value': N_PTS / dtn / 1e6, 'unit': 'Mpoints/s', 'bev_frames_per_s': 1.0 / dtn, 'seconds_per_step': dtn,
        'cores': os.cpu_count(), 'blas': blas, 'kind': 'port',
        'sample': f'oracle/numpy_shape.py, vectorised reductions (np.lexsort medians / minima): {n_steps} full steps on the '
                  f'filled {len(window)}-frame window',
        'reference_loop_shape': {
            'seconds_measured': dtl, 'points': int(g.shape[0]),
            'what': 'the per-point min-z loop and the per-cell np.median loop of ONE colour channel of ONE point set, on '
                    'every 8th in-view point of the window',
            'seconds_per_bev_extrapolated': dtl * 8 * (1 + 3 * 3) / 2.0,
            'note': 'x8 points, (1 min-z + 3 channels) x 3 sets relative to the (1 + 1) x 1 measured'}}
    return out


# --------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """--gpus N > 1 without a launcher: start N ranks (torch.distributed.run, one per GPU) BEFORE this process touches
    the GPU, relay rank 0's JSON line, exit with the job's status."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line)
    if p.returncode != 0 or line is None:
        sys.stderr.write(p.stdout[-4000:])
        sys.exit(p.returncode or 1)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', choices=['config2', 'config5'], default='config2',
                    help='config2 = the per-frame step on one sequence per rank (headline, weak scaling); config5 = the nine '
                         'KITTI-360 sequences sharded over the ranks as the headline value (strong scaling)')
    ap.add_argument('--config5-scale', type=float, default=0.1, help='fraction of the nine sequence lengths (1 = 74 367 frames)')
    ap.add_argument('--lanes', type=int, default=int(os.environ.get('PCA_LANES', '1')),
                    help='config 5: chunks of a rank run on this many lanes (host thread + pca_ctx + stream each) at once')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--scene', choices=['uniform', 'ring'], default='uniform',
                    help='uniform = SURVEY 8d K-shape frame (headline); ring = 64-beam ring model')
    ap.add_argument('--no-ring', action='store_true', help='skip the additional ring-model pass')
    ap.add_argument('--no-extras', action='store_true', help='skip the side measurements (K1 batches, NuScenes, config 4/5, ...)')
    ap.add_argument('--extras', choices=['all', 'config5', 'none'], default='all',
                    help='config5 = of the side measurements only the sharded nine-sequence job (same process group as the headline)')
    ap.add_argument('--gather', action='store_true',
                    help='multi-GPU: also stream every finished BEV tensor to rank 0 inside the timed region')
    args = ap.parse_args()
    if args.extras == 'none':
        args.no_extras = True
    only_c5 = args.extras == 'config5' and not args.no_extras
    if args.gpus > 1 and 'RANK' not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs (no CPU fallback)'
    # PCA_BENCH_BACKEND=gloo + fewer GPUs than ranks: rehearsal of the multi-rank control flow on a one-GPU box
    backend = os.environ.get('PCA_BENCH_BACKEND', 'nccl')
    n_dev = torch.cuda.device_count()
    if backend == 'nccl' and world > 1 and int(os.environ.get('LOCAL_WORLD_SIZE', world)) > n_dev:
        # RCCL wants one device per rank; two ranks on one card die inside the first collective with a duplicate-device error.
        # Said here, in one line, before anything is initialised (the gloo rehearsal shares cards on purpose: modulo below)
        sys.stderr.write(f'bench.py: {os.environ.get("LOCAL_WORLD_SIZE", world)} local ranks but {n_dev} visible GPU(s): the nccl (RCCL) '
                         f'backend needs one GPU per rank (rehearse the control flow on fewer cards with PCA_BENCH_BACKEND=gloo)\n')
        sys.exit(2)
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    # under a launcher (RANK set) the process group is initialised even for ONE rank: `torchrun --nproc-per-node 1 bench.py`
    # takes RCCL, the gathers of device tensors and the checksum check through the real transport on a one-GPU box (the
    # driver's own N = 1 run starts bench.py without a launcher: no process group, nothing changes there)
    use_dist = world > 1 or ('RANK' in os.environ and os.environ.get('PCA_BENCH_DIST_AT_1', '1') != '0')
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            try:
                dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))   # RCCL on ROCm
            except Exception as e:                            # (no retry, no re-exec: the GPU is initialised by now)
                sys.stderr.write(f'bench.py: rank {rank}: init_process_group(nccl) on cuda:{dev_index} failed: {e!r}\n'
                                 f'  MASTER_ADDR={os.environ.get("MASTER_ADDR")} MASTER_PORT={os.environ.get("MASTER_PORT")} '
                                 f'WORLD_SIZE={world} HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}\n'
                                 f'  (rehearse the control flow without RCCL: PCA_BENCH_BACKEND=gloo)\n')
                sys.exit(3)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    coll_dev = 'cuda' if backend == 'nccl' else 'cpu'

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    from pca_amd import _lib
    frame_fn = ring_frame if args.scene == 'ring' else synth_frame
    import builtins
    real_print = builtins.print
    quiet = lambda *a, **k: None                      # the accumulator prints one line per frame, as the reference

    if args.workload == 'config5':
        c5 = config5_pass(rank, world, args.config5_scale, barrier, allmax, coll_dev, dist, args.lanes)
        if rank == 0:
            out = {'metric': 'Mpoints/s projected+accumulated and BEV frames/s @256x256; 1/2/4/8 GPU',
                   'value': c5['Mpoints_per_s'], 'unit': 'Mpoints/s', 'bev_frames_per_s': c5['bev_frames_per_s'],
                   'n_gpus': world, 'steps': c5['frames'], 'warmup': 0, 'ms_per_step': 1e3 * c5['seconds'] / c5['frames'],
                   'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
                   'config': {'workload': c5['workload'], 'points_per_frame': N_PTS, 'image': [IMG_H, IMG_W], 'bev_px': PX},
                   'config5': c5, 'roofline': None, 'cpu_baseline': None}
            out.update(config5_headline(c5))
            print(json.dumps(out))
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- inputs resident in HBM: every rank works on its own sequence (scene shard) ----
    acc, pool, model = make_accumulator(frame_fn, rank)
    builtins.print = quiet
    st = Stepper(acc, pool)
    # ---- untimed: fill the accumulation window, then W warm-up steps ----
    st.fill()
    for _ in range(args.warmup):
        st.step()
    acc.store.check_status()
    bev_buf = torch.empty((args.steps, 21, PX, PX), dtype=torch.float16, device='cuda')
    # Every rank produces the BEVs of its own sequence and keeps them (a per-rank writer stores them, as the
    # reference's one-file-per-sample output allows): the step is the same at every N and the timed region holds no
    # data-path collective.  --gather additionally streams the finished tensors to rank 0 INSIDE the timed region, in
    # chunks of GATHER_CHUNK samples with async_op=True (RCCL's own stream: one chunk travels while the next is
    # computed).  Without it one chunk is gathered after the timed region as a check of the RCCL path.
    chunks = [(lo, min(lo + GATHER_CHUNK, args.steps)) for lo in range(0, args.steps, GATHER_CHUNK)]

    def recv_bufs(lo, hi):
        return [torch.empty((hi - lo, 21, PX, PX), dtype=torch.float16, device=coll_dev) for _ in range(world)] \
            if rank == 0 else None

    def chunk_of(lo, hi):
        return bev_buf[lo:hi] if backend == 'nccl' else bev_buf[lo:hi].cpu()
    gathered = [recv_bufs(lo, hi) for lo, hi in chunks] if (use_dist and args.gather) else None

    # ---- timed region: exactly K steps (barrier + synchronize on both sides, max over ranks).  The region is REPEATED:
    #      at least REPEATS times and until MIN_TIMED_S seconds have been timed in all, whatever --steps says (a 20-step
    #      region is 2.4 ms: one hiccup of the host would be the whole measurement); the first repetition only warms up
    #      and is dropped; `value` / `ms_per_step` are the median over the others ----
    times = []
    rep, n_reps = 0, REPEATS + 1
    while rep < n_reps:
        barrier()
        t0 = time.perf_counter()
        pending = []
        ci = 0
        for k in range(args.steps):
            st.step(bev_buf[k])
            if gathered is not None and k + 1 == chunks[ci][1]:
                pending.append(dist.gather(chunk_of(*chunks[ci]), gathered[ci], dst=0, async_op=True))
                ci += 1
        for h in pending:
            h.wait()
        barrier()
        times.append(allmax(time.perf_counter() - t0))
        rep += 1
        if rep == 1:                                   # every rank sees the same (all-reduced) time: same decision everywhere
            n_reps = min(max(REPEATS + 1, int(np.ceil(MIN_TIMED_S / max(times[0], 1e-6))) + 1), MAX_REPEATS)
    first_repeat_s, times = times[0], times[1:]
    elapsed = float(np.median(times))
    gather_check = None
    if use_dist:
        # the last chunk of every rank -> rank 0 (untimed unless --gather already did it): checksums + link rate.  A failure
        # of the check's own collectives is reported in the block, it does not take the measurement above down with it.
        ok = None
        try:
            lo, hi = chunks[-1]
            mine = bev_buf[lo:hi].view(torch.int16).to(torch.int64).sum().reshape(1).to(coll_dev)
            sums = [torch.zeros(1, dtype=torch.int64, device=coll_dev) for _ in range(world)]
            dist.all_gather(sums, mine)
            last = gathered[-1] if gathered is not None else recv_bufs(lo, hi)
            barrier()
            tg = time.perf_counter()
            dist.gather(chunk_of(lo, hi), last, dst=0)
            barrier()
            tg = time.perf_counter() - tg
            if rank == 0:
                ok = all(int(g.view(torch.int16).to(torch.int64).sum().item()) == int(s.item()) for g, s in zip(last, sums))
                mb = (world - 1) * (hi - lo) * 21 * PX * PX * 2 / 1e6
                gather_check = {'in_timed_region': gathered is not None, 'samples_per_rank': hi - lo, 'MB_into_rank0': mb,
                                'ms': 1e3 * tg, 'GBps': mb / 1e3 / tg, 'checksums_match': bool(ok), 'backend': backend}
        except RuntimeError as e:                                   # torch.distributed / RCCL errors
            gather_check = {'error': repr(e)[:300]}
        assert ok is not False, 'gathered BEV tensors differ from what the ranks produced'
    acc.store.check_status()
    stored = int(acc.store.offsets()[-1] - acc.store.offsets()[0])
    n_live = acc.store.n_frames
    sizes = acc.store.sizes()

    # ---- second identical pass with per-kernel HIP events (on the launch stream, inside the library) ----
    ctx = _lib.Context.get()
    ctx.profile(True)
    for k in range(args.steps):
        st.step(bev_buf[k])
    prof = ctx.profile_read()
    ctx.profile(False)
    # ---- third pass: the BEV unit (its kernels back to back) bracketed by ONE event pair per call, so that the
    #      unit's duration carries its launch gaps but not the per-kernel events of the pass above ----
    ctx.profile(2)
    for k in range(args.steps):
        st.step(bev_buf[k])
    unit = ctx.profile_read()['bev_unit']
    ctx.profile(False)
    builtins.print = real_print

    # ---- the step as an unchanged driver runs it (host arrays in, host dict out): right after the headline, on its
    #      accumulator, before the passes that churn through tens of GB ----
    pcie = two_seq = None
    if rank == 0 and world == 1 and not args.no_extras and not only_c5:
        builtins.print = quiet
        pcie = pcie_inclusive_pass(acc, pool, min(args.steps, 30))
        # two independent sequences on this GPU (threads with a lane each; then two processes): right after the headline,
        # before the passes that churn through tens of GB
        two_seq = two_sequences_pass(max(args.steps, 20))
        two_seq['single_sequence_headline_Mpoints_per_s'] = N_PTS * args.steps / elapsed / 1e6
        two_seq['aggregate_over_headline'] = two_seq.get('Mpoints_per_s', 0.0) / (N_PTS * args.steps / elapsed / 1e6)
        builtins.print = real_print
        two_seq['two_processes'] = two_ranks_one_gpu_pass(max(args.steps, 20), args.warmup)
        if 'Mpoints_per_s' in two_seq['two_processes']:
            two_seq['two_processes']['aggregate_over_headline'] = two_seq['two_processes']['Mpoints_per_s'] / (N_PTS * args.steps / elapsed / 1e6)

    # ---- BASELINE configs[4] at this N: every rank takes part (strong-scaling job, reported beside `value`) ----
    c5 = None
    if not args.no_extras:
        c5 = config5_pass(rank, world, args.config5_scale, barrier, allmax, coll_dev, dist, args.lanes)
        if args.lanes == 1 and world == 1 and rank == 0 and not only_c5:
            # the same job with two chunks at a time on the one GPU (two lanes): reported beside, adopted as the default if it pays
            c5b = config5_pass(rank, world, args.config5_scale, barrier, allmax, coll_dev, dist, 2)
            c5['two_lanes'] = {k: c5b[k] for k in ('seconds', 'Mpoints_per_s', 'bev_frames_per_s', 'lanes_per_gpu')}
            c5['two_lanes']['speedup_over_one_lane'] = c5['seconds'] / c5b['seconds']

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    side = {}
    if world == 1 and not args.no_extras and not only_c5:
        builtins.print = quiet
        side['pcie_inclusive'] = pcie
        side['two_sequences'] = two_seq
        side['k1_batched'] = k1_batched_pass(pool, POOL)
        side['k1_batched']['note'] = ('%d distinct frames repeated: the 32 MB of inputs stay in L2 / Infinity Cache between calls -- '
                                      'NOT an HBM figure; k1_batched_distinct is the one to compare with the 0.5 target' % POOL)
        big = device_pool(synth_frame, 7, 128)
        side['k1_batched_distinct'] = k1_batched_pass(big[:64], 64)
        # two alternating batches of 64 different frames: no call finds its inputs where the call before left them
        side['k1_batched_distinct128'] = k1_batched_pass(big, 128)
        del big
        ringpool = device_pool(ring_frame, 7, 64)          # 64 distinct ring-model frames: a real sweep's point ORDER
        side['k1_batched_ring'] = k1_batched_pass(ringpool, 64)
        side['k1_batched_ring']['note'] = ('ring-model frames (SURVEY 8d): consecutive points are neighbours in azimuth, so the '
                                           'lanes of a gather share image lines; the uniform K-shape frame scatters them')
        del ringpool
        side['nuscenes'] = nuscenes_pass()
        side['nuscenes']['scene'] = nuscenes_scene_pass()
        sweep = nuscenes_scene_pass(order='sweep')             # the same scene with the points in a real sweep's order
        side['nuscenes']['scene']['k1n_batched_ring'] = dict(sweep['k1n_batched'], ms_per_scene_batched=sweep['batched']['ms_per_scene'],
                                                             note='32 beams x 1085 azimuth steps in sweep order, pixels from '
                                                                  'azimuth / elevation: neighbours in the array are neighbours in the image')
        side['config4'] = config4_pass()
        side['kitti_icp_flow'] = kitti_icp_flow_pass()
        if not args.no_ring and args.scene == 'uniform':
            side['ring_model'] = ring_model_pass(min(args.steps, 50))
        side['extras'] = extras_pass(acc)                  # mutates the store: last use of `acc`
        builtins.print = real_print

    kern = {k: {'ms_total': v[0], 'launches': v[1], 'avg_us': 1e3 * v[0] / v[1]} for k, v in prof.items() if v[1]}
    # Algorithmic bytes per launch (SURVEY.md 8d / DESIGN.md):
    #   K2 retransform: 48 B per stored point;  K1: 16 N + 4 M_proj + 40 M_kept;
    #   BEV (its kernels as one unit): 40 B per window point + 21 px^2 4 B.
    # In steady state the owed re-transform of a step is applied by the BEV's first pass (it reads every
    # coordinate anyway): the BEV unit then also does K2's work, so its algorithmic bytes include K2's.
    m_kept = float(np.mean(sizes))
    # in-frustum points per frame, COUNTED: the pool's frames through K1 without a class filter
    from pca_amd.device_store import DeviceStore
    cnt = DeviceStore(capacity=len(pool) * N_PTS, max_frames=len(pool) + 1)
    cnt.append_kitti([dict(pts=f[1], rgb=f[0], sem=f[2]) for f in pool], P_VELO_FRAME, IMG_H, IMG_W, [])
    m_proj = float(np.mean(cnt.sizes()))
    del cnt
    k2_fused = 'retransform' not in kern
    # K1 of the step riding in the raster's first kernel (pca_k1_defer, the drop-in's default): no launch of its own; the unit
    # then does K1's work too, minus the 40 B per kept point that the raster no longer reads back from the store
    k1_fused = 'kitti_project_sample_filter' not in kern
    alg_k1 = 16.0 * N_PTS + 4.0 * m_proj + 40.0 * m_kept
    alg = {
        'bev': 40.0 * stored + 21.0 * PX * PX * 4.0 + (48.0 * (stored - sizes[-1]) if k2_fused else 0.0)
               + (alg_k1 - 40.0 * m_kept if k1_fused else 0.0),
    }
    bev_us = 1e3 * unit[0] / unit[1]                    # one event pair around the unit (see above)
    bev_names = ('bev_bin', 'bev_cells', 'bev_cells_heavy')
    bev_us_sum = sum(kern[k]['avg_us'] for k in bev_names if k in kern)
    units = {'bev': bev_us}
    if not k1_fused:
        alg['kitti_project_sample_filter'] = alg_k1
        units['kitti_project_sample_filter'] = kern['kitti_project_sample_filter']['avg_us']
    if not k2_fused:
        alg['retransform'] = 48.0 * stored
        units['retransform'] = kern['retransform']['avg_us']
    dominant = max(units, key=lambda k: units[k])
    achieved = alg[dominant] / (units[dominant] * 1e-6) / 1e9
    # HBM traffic per launch from the committed rocprofv3 PMC passes of THESE kernels (FETCH_SIZE / WRITE_SIZE in
    # separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); bench.py cannot run the profiler
    traffic, traffic_src = pmc_traffic('') if dominant == 'bev' else (None, None)
    roofline = {'bound': 'hbm', 'kernel': dominant, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                'algorithmic_bytes_per_launch': alg[dominant], 'avg_launch_us': units[dominant],
                'retransform_fused_into_bev': k2_fused, 'k1_fused_into_bev': k1_fused,
                'view_culling': ('on: in calls that write nothing back (3 of 4) level 1 does not read the frames that cannot reach '
                                 'the view; the algorithmic bytes still count every window point (SURVEY 8d), the physical ones '
                                 '(traffic) are what is moved') if os.environ.get('PCA_BEV_CULL', '1') != '0' else 'off',
                'bev_unit_sum_of_per_kernel_events_us': bev_us_sum,
                'all': {k: {'avg_us': units[k], 'alg_bytes': alg[k],
                            'GBps': alg[k] / (units[k] * 1e-6) / 1e9,
                            'frac': alg[k] / (units[k] * 1e-6) / 1e9 / HBM_PEAK_GBS} for k in units},
                'kernels': kern}
    if traffic:
        roofline['frac_physical'] = traffic / (units[dominant] * 1e-6) / 1e9 / HBM_PEAK_GBS
    k1_pmc_path = os.path.join(ROOT, 'profiles', PROFILE_TAG + '_k1_batched_summary.json')
    k1_pmc = json.load(open(k1_pmc_path)) if os.path.exists(k1_pmc_path) else {}
    for k, pool in (('k1_batched', 'pool8'), ('k1_batched_distinct', 'pool64'), ('k1_batched_distinct128', None),
                    ('k1_batched_ring', None)):
        if k in side:
            roofline[k] = side.pop(k)
            phys = k1_pmc.get(pool, {}).get('physical_MB') if pool else None
            if phys:                                       # PMC bytes of the two kernels / the wall time of the call
                roofline[k]['traffic'] = phys * 1e6
                roofline[k]['frac_physical'] = phys * 1e6 / (roofline[k]['us_per_call_wall_back_to_back'] * 1e-6) / 1e9 / HBM_PEAK_GBS
                roofline[k]['traffic_source'] = 'profiles/%s_k1_batched_summary.json' % PROFILE_TAG
    for blk, suffix, prefixes in (('ring_model', '_ring', ('bev_tile', )), ('config4', '_config4', ('bev_tile', )),
                                  ('nuscenes', '_nusc', ('bev_tile', 'k1n_'))):
        if blk in side:
            t, src = pmc_traffic(suffix, prefixes)
            side[blk]['traffic'] = t
            side[blk]['traffic_source'] = src

    out = {
        'metric': 'Mpoints/s projected+accumulated and BEV frames/s @256x256; 1/2/4/8 GPU',
        'value': world * N_PTS * args.steps / elapsed / 1e6,
        'unit': 'Mpoints/s',
        'bev_frames_per_s': world * args.steps / elapsed,
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps,
        'repeats': {'n': len(times), 'statistic': 'median (the first, warming, repetition dropped)',
                    'timed_seconds_total': float(sum(times)), 'ms_per_step_first_dropped': 1e3 * first_repeat_s / args.steps,
                    'ms_per_step_min': 1e3 * min(times) / args.steps,
                    'ms_per_step_max': 1e3 * max(times) / args.steps,
                    'value_min': world * N_PTS * args.steps / max(times) / 1e6,
                    'value_max': world * N_PTS * args.steps / min(times) / 1e6},
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': 'KITTI-360 single forward cam, 120k pts/frame, 200 m horizon (~%d live frames, '
                               '%d stored pts), one 256x256x21 BEV per integrated frame' % (n_live, stored),
                   'points_per_frame': N_PTS, 'image': [IMG_H, IMG_W], 'bev_px': PX, 'view_m': VIEW_M,
                   'sharding': 'one independent sequence per GPU, no data-path collective'
                               + ('; BEV tensors streamed to rank 0 (RCCL, overlapped)' if args.gather else '')},
        'roofline': roofline,
    }
    if c5 is not None:
        out['config5'] = c5
        out.update(config5_headline(c5))
    if gather_check is not None:
        out['gather_check'] = gather_check
    out.update(side)
    if not args.no_cpu_baseline and world == 1:           # reported at N = 1 only
        out['cpu_baseline'] = cpu_baseline()
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
