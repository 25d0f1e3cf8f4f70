#!/usr/bin/env python3
"""Headline benchmark: Mpoints/s projected+accumulated and BEV frames/s @256x256 (BASELINE.json).

Workload (BASELINE.json configs[1], SURVEY.md 8d "Config 2"): KITTI-360-shaped synthetic frames
(120 000 points f32, 376x1408 RGB + semseg), 200 m accumulation horizon at 1 m / frame (~200 live
frames, ~5 M stored points), one 256x256 x 21-plane BEV per integrated frame.

One STEP = integrate one frame  (K2 re-transform of every stored point + K1 fused project / sample /
filter / append + horizon eviction)  +  generate one BEV sample (bin, scan, scatter, per-cell reduce).
Inputs (point clouds, images, semseg maps) are resident in HBM before the timed region; BEV tensors
stay in HBM on the rank that made them (multi-GPU: no data-path collective; --gather streams them to rank 0 over RCCL
inside the timed region, chunk by chunk, overlapped with compute).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
POOL = 8                                  # distinct synthetic frames cycled through
GATHER_CHUNK = 16                         # multi-GPU: BEV samples per asynchronous gather to rank 0
HBM_PEAK_GBS = 8000.0                     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

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


def present_index(acc):
    """Sample trigger of run_kitti360_bev_gen.py:218-230: first pose more than 80 m of path behind the newest."""
    if len(acc.poses) < 2:
        return None
    d = acc.get_incremental_path_dists()
    if d[-1] < BEV_HORIZON_M:
        return None
    idx = int(((d - BEV_HORIZON_M) > 0).argmax())
    if d[-1] - d[idx] < BEV_HORIZON_M:
        return None
    return idx


def make_accumulator(frame_fn, seq):
    """Drop-in accumulator over a pool of synthetic frames resident in HBM."""
    import torch
    import sem_pc_accum
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    model = ResidentSemSeg()
    sem_pc_accum.SemSegONNX = lambda path: model
    bev_params = dict(type='sem', view_size=VIEW_M, pixel_size=PX, max_trans_radius=0., zoom_thresh=0., do_warp=False,
                      int_scaler=20., int_sep_scaler=20., int_mid_threshold=0.5, height_filter=None)
    calib = {'h_velo_cam': np.linalg.inv(CAM_TO_VELO), 'p_cam_frame': P_RECT, 'p_velo_frame': P_VELO_FRAME}
    acc = Kitti360SemanticPointCloudAccumulator(HORIZON_M, calib, 1e3, 'resident', FILTERS, SEM_IDXS, False, bev_params)
    acc._store_args = dict(capacity=1 << 26, max_frames=1 << 14)
    T = t_new_prev()
    acc.pose_provider = lambda pc: T
    pool = []
    for k in range(POOL):
        pc, img, sem = frame_fn(seq, k)
        f = (torch.from_numpy(img).cuda(), torch.from_numpy(pc).cuda(), torch.from_numpy(sem).cuda())
        model.by_ptr[f[0].data_ptr()] = f[2]
        pool.append(f)
    return acc, pool, model


def ring_model_pass(steps, frame_fn=None):
    """The same step on ring-model frames (rank 0 only, after the headline measurement): steady-state time per
    step and per-kernel HIP-event times.  Reported beside the headline number, never as `value`."""
    import torch
    from pca_amd import _lib
    acc, pool, _ = make_accumulator(frame_fn or ring_frame, 0)
    n = [0]

    def step(out=None):
        rgb, pc, _ = pool[n[0] % POOL]
        n[0] += 1
        acc.integrate([(rgb, pc, None)])
        idx = present_index(acc)
        if idx is None:
            return
        pcs, trajs = acc._window_inputs(idx, True)
        acc.sem_bev_generator.generate(pcs, trajs, device_only=True, out=out)

    while present_index(acc) is None or len(acc.poses) < 195:
        step()
    out = torch.empty((21, PX, PX), dtype=torch.float16, device='cuda')
    for _ in range(5):
        step(out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx = _lib.Context.get()
    ctx.profile(True)
    for _ in range(steps):
        step(out)
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


def extras_pass(acc):
    """Timings of the opt-in / next-row pieces on the benchmark's own data (rank 0, after the headline measurement;
    reported beside, never as, `value`): voxel de-duplication of the full window and the device ICP on two frames."""
    import torch
    from pca_amd.icp import GpuIcp
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
    icp.register(a, b, 1e3, np.eye(4))
    t0 = time.perf_counter()
    r = icp.register(a, b, 1e3, np.eye(4))
    out['device_icp'] = {'ms_per_registration': 1e3 * (time.perf_counter() - t0), 'points': N_PTS,
                         'iterations': r.iterations, 'scene': 'two consecutive ring-model frames'}
    return out


def cpu_baseline(steps=20):
    """Oracle (scalar C port of the reference algorithm, 1 core) on a bounded sample of the same workload:
    fill the 200-frame window, then time `steps` full steps (re-transform + integrate + BEV)."""
    from oracle import oracle as orc
    from pca_amd import host_logic as hl
    T = t_new_prev()
    frames = [synth_frame(0, k) for k in range(2)]
    st = orc.Store(220 * 40000)
    track = hl.PoseTrack()
    sizes = []
    lo = 0

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

    for k in range(205):
        step(k, False)
    t0 = time.perf_counter()
    for k in range(steps):
        step(205 + k, True)
    dt = time.perf_counter() - t0
    return {'value': N_PTS * steps / dt / 1e6, 'unit': 'Mpoints/s', 'bev_frames_per_s': steps / dt, 'cores': 1,
            'kind': 'port',
            'sample': f'oracle/pca_oracle.c (scalar C port), 205-frame window fill untimed, then {steps} full steps '
                      f'(retransform ~{st.n - lo} stored pts + integrate 120k pts + one 256x256 BEV) in {dt:.1f} s',
            'host_cpus': os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--scene', choices=['uniform', 'ring'], default='uniform',
                    help='uniform = SURVEY 8d K-shape frame (headline); ring = 64-beam ring model')
    ap.add_argument('--no-ring', action='store_true', help='skip the additional ring-model pass')
    ap.add_argument('--gather', action='store_true',
                    help='multi-GPU: also stream every finished BEV tensor to rank 0 inside the timed region')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs (no CPU fallback)'
    # PCA_BENCH_BACKEND=gloo + fewer GPUs than ranks: rehearsal of the multi-rank control flow on a one-GPU box
    backend = os.environ.get('PCA_BENCH_BACKEND', 'nccl')
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))   # RCCL on ROCm
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from pca_amd import _lib
    frame_fn = ring_frame if args.scene == 'ring' else synth_frame
    # ---- inputs resident in HBM: every rank works on its own sequence (scene shard) ----
    acc, pool, model = make_accumulator(frame_fn, rank)

    import builtins
    real_print = builtins.print
    builtins.print = lambda *a, **k: None           # the accumulator prints one line per frame, as the reference

    frame_no = [0]

    def step(bev_out=None, slot=0):
        rgb, pc, _ = pool[frame_no[0] % POOL]
        frame_no[0] += 1
        acc.integrate([(rgb, pc, None)])
        idx = present_index(acc)
        if idx is None:
            return None
        pcs, trajs = acc._window_inputs(idx, True)
        gen = acc.sem_bev_generator
        return gen.generate(pcs, trajs, device_only=True, out=None if bev_out is None else bev_out[slot])

    # ---- untimed: fill the accumulation window, then W warm-up steps ----
    while present_index(acc) is None or len(acc.poses) < 195:
        rgb, pc, _ = pool[frame_no[0] % POOL]
        frame_no[0] += 1
        acc.integrate([(rgb, pc, None)])
    for _ in range(args.warmup):
        step()
    acc.store.check_status()
    bev_buf = torch.empty((args.steps, 21, PX, PX), dtype=torch.float16, device='cuda')
    # Every rank produces the BEVs of its own sequences and keeps them (a per-rank writer stores them, as the
    # reference's one-file-per-sample output allows): the step is the same at every N and the timed region holds no
    # data-path collective.  --gather additionally streams the finished tensors to rank 0 INSIDE the timed region, in
    # chunks of GATHER_CHUNK samples with async_op=True (RCCL's own stream: one chunk travels while the next is
    # computed).  Without it one chunk is gathered after the timed region as a check of the RCCL path.
    coll_dev = 'cuda' if backend == 'nccl' else 'cpu'
    chunks = [(lo, min(lo + GATHER_CHUNK, args.steps)) for lo in range(0, args.steps, GATHER_CHUNK)]

    def recv_bufs(lo, hi):
        return [torch.empty((hi - lo, 21, PX, PX), dtype=torch.float16, device=coll_dev) for _ in range(world)] \
            if rank == 0 else None

    def chunk_of(lo, hi):
        return bev_buf[lo:hi] if backend == 'nccl' else bev_buf[lo:hi].cpu()
    gathered = [recv_bufs(lo, hi) for lo, hi in chunks] if (world > 1 and args.gather) else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps ----
    barrier()
    t0 = time.perf_counter()
    pending = []
    ci = 0
    for k in range(args.steps):
        step(bev_buf, k)
        if gathered is not None and k + 1 == chunks[ci][1]:
            pending.append(dist.gather(chunk_of(*chunks[ci]), gathered[ci], dst=0, async_op=True))
            ci += 1
    for h in pending:
        h.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    gather_check = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # the last chunk of every rank -> rank 0 (untimed unless --gather already did it): content check + link rate
        lo, hi = chunks[-1]
        last = gathered[-1] if gathered is not None else recv_bufs(lo, hi)
        barrier()
        tg = time.perf_counter()
        if gathered is None:
            dist.gather(chunk_of(lo, hi), last, dst=0)
        barrier()
        tg = time.perf_counter() - tg
        if rank == 0:                                 # every rank's last BEV arrived and is a plausible probability map
            for g in last:
                assert float(g[-1, 0].float().min()) > 0.0 and float(g[-1, 0].float().max()) < 1.0
            mb = (world - 1) * (hi - lo) * 21 * PX * PX * 2 / 1e6
            gather_check = {'in_timed_region': gathered is not None, 'samples_per_rank': hi - lo, 'MB_into_rank0': mb,
                            'ms': None if gathered is not None else 1e3 * tg,
                            'GBps': None if gathered is not None else mb / 1e3 / tg}
    acc.store.check_status()
    stored = int(acc.store.offsets()[-1] - acc.store.offsets()[0])
    n_live = acc.store.n_frames
    sizes = acc.store.sizes()

    # ---- second identical pass with per-kernel HIP events (on the launch stream, inside the library) ----
    ctx = _lib.Context.get()
    ctx.profile(True)
    for k in range(args.steps):
        step(bev_buf, k)
    prof = ctx.profile_read()
    ctx.profile(False)
    # ---- third pass: the BEV unit (its five kernels back to back) bracketed by ONE event pair per call, so that the
    #      unit's duration carries its launch gaps but not the per-kernel events of the pass above ----
    ctx.profile(2)
    for k in range(args.steps):
        step(bev_buf, k)
    unit = ctx.profile_read()['bev_unit']
    ctx.profile(False)

    # ---- PCIe-inclusive variant of the step (host numpy inputs as the unchanged drivers pass them, BEV dict of
    #      host fp16 arrays out): reported beside, never as, `value` ----
    host_pool = [(f[0].cpu().numpy(), f[1].cpu().numpy(), f[2].cpu().numpy()) for f in pool]

    class HostSemSeg:
        def pred(self, rgb):
            return host_sem[id(rgb)][None, None]
    host_sem = {id(h[0]): h[2] for h in host_pool}
    acc.semseg_model = HostSemSeg()
    n_host = min(args.steps, 30)
    torch.cuda.synchronize()
    th0 = time.perf_counter()
    for k in range(n_host):
        rgb_h, pc_h, _ = host_pool[k % POOL]
        acc.integrate([(rgb_h, pc_h, None)])
        idx = present_index(acc)
        bevs = acc.generate_bev(idx, 1, gen_future=True)
    torch.cuda.synchronize()
    host_elapsed = time.perf_counter() - th0
    assert bevs[0]['rgb_full'].shape == (3, PX, PX)
    acc.semseg_model = model

    # ---- K1 alone, batched: 64 frames (7.68 M points) per launch -- the shape in which the fused
    #      project+sample+filter kernel is throughput- rather than launch-latency-bound ----
    from pca_amd.device_store import DeviceStore
    k1_batch = 64
    tmp = DeviceStore(capacity=k1_batch * N_PTS, max_frames=k1_batch + 1)
    frames = [dict(pts=pool[k % POOL][1], rgb=pool[k % POOL][0], sem=pool[k % POOL][2]) for k in range(k1_batch)]
    tmp.append_kitti(frames, P_VELO_FRAME, IMG_H, IMG_W, FILTERS)           # warm-up
    torch.cuda.synchronize()
    ctx.profile(True)
    for _ in range(5):
        tmp.clear()
        tmp.append_kitti(frames, P_VELO_FRAME, IMG_H, IMG_W, FILTERS)
    k1b = ctx.profile_read()['kitti_project_sample_filter']
    ctx.profile(False)
    k1b_kept = int(tmp.offsets()[-1])
    del tmp
    ring, extras = None, None
    if rank == 0 and world == 1 and not args.no_ring and args.scene == 'uniform':
        ring = ring_model_pass(min(args.steps, 50))
        extras = extras_pass(acc)                          # mutates the store: last use of `acc`
    builtins.print = real_print

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    kern = {k: {'ms_total': v[0], 'launches': v[1], 'avg_us': 1e3 * v[0] / v[1]} for k, v in prof.items() if v[1]}
    # Algorithmic bytes per launch (SURVEY.md 8d / DESIGN.md):
    #   K2 retransform: 48 B per stored point;  K1: 16 N + 4 M_proj + 40 M_kept;
    #   BEV (hist+scan+scatter+cells as one unit): 40 B per window point + 21 px^2 4 B.
    # In steady state the owed re-transform of a step is applied by the BEV's first pass (it reads every
    # coordinate anyway): the BEV unit then also does K2's work, so its algorithmic bytes include K2's.
    m_kept = float(np.mean(sizes))
    m_proj = m_kept * 19.0 / 14.0 / 0.99               # 14 of 19 uniform classes survive, 1 % 'ignore'
    k2_fused = 'retransform' not in kern
    alg = {
        'kitti_project_sample_filter': 16.0 * N_PTS + 4.0 * m_proj + 40.0 * m_kept,
        'bev': 40.0 * stored + 21.0 * PX * PX * 4.0 + (48.0 * (stored - sizes[-1]) if k2_fused else 0.0),
    }
    bev_us = 1e3 * unit[0] / unit[1]                    # one event pair around the unit (see above)
    bev_us_sum = sum(kern[k]['avg_us'] for k in ('bev_bin', 'bev_scan', 'bev_scatter', 'bev_cells', 'bev_cells_heavy')
                     if k in kern)
    units = {'kitti_project_sample_filter': kern['kitti_project_sample_filter']['avg_us'], 'bev': bev_us}
    if not k2_fused:
        alg['retransform'] = 48.0 * stored
        units['retransform'] = kern['retransform']['avg_us']
    dominant = max(units, key=lambda k: units[k])
    achieved = alg[dominant] / (units[dominant] * 1e-6) / 1e9
    # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate passes,
    # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); bench.py cannot run the profiler itself
    traffic, traffic_src = None, None
    pmc_path = os.path.join(ROOT, 'profiles', 'r01_e_pmc_traffic.json')
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))['kernels']
        names = {'bev': ('bev_tile_hist', 'bev_tile_scan', 'bev_tile_scatter<false>', 'bev_tile_cells<false>',
                         'bev_tile_cells_heavy<false>')}
        if dominant in names and all(k in pmc for k in names[dominant]):
            traffic = sum(2.0 * pmc[k]['FETCH_SIZE_KB'] + pmc[k]['WRITE_SIZE_KB'] for k in names[dominant]) * 1024.0
            traffic_src = 'profiles/r01_e_pmc_traffic.json (2*FETCH_SIZE + WRITE_SIZE, summed over the unit\'s kernels)'
    roofline = {'bound': 'hbm', 'kernel': dominant, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                'algorithmic_bytes_per_launch': alg[dominant], 'avg_launch_us': units[dominant],
                'retransform_fused_into_bev': k2_fused, 'bev_unit_sum_of_per_kernel_events_us': bev_us_sum,
                'all': {k: {'avg_us': units[k], 'alg_bytes': alg[k],
                            'GBps': alg[k] / (units[k] * 1e-6) / 1e9,
                            'frac': alg[k] / (units[k] * 1e-6) / 1e9 / HBM_PEAK_GBS} for k in units},
                'kernels': kern}
    k1b_us = 1e3 * k1b[0] / k1b[1]
    k1b_bytes = 16.0 * N_PTS * k1_batch + 4.0 * (k1b_kept * 19.0 / 14.0 / 0.99) + 40.0 * k1b_kept
    roofline['k1_batched'] = {'frames_per_launch': k1_batch, 'points_per_launch': N_PTS * k1_batch, 'kept': k1b_kept,
                              'avg_launch_us': k1b_us, 'alg_bytes': k1b_bytes, 'GBps': k1b_bytes / (k1b_us * 1e-6) / 1e9,
                              'frac': k1b_bytes / (k1b_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                              'Mpoints_per_s': N_PTS * k1_batch / k1b_us}

    out = {
        'metric': 'Mpoints/s projected+accumulated and BEV frames/s @256x256; 1/2/4/8 GPU',
        'value': world * N_PTS * args.steps / elapsed / 1e6,
        'unit': 'Mpoints/s',
        'bev_frames_per_s': world * args.steps / elapsed,
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps,
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
        'pcie_inclusive': {'Mpoints_per_s': N_PTS * n_host / host_elapsed / 1e6, 'bev_frames_per_s': n_host / host_elapsed,
                           'ms_per_step': 1e3 * host_elapsed / n_host, 'steps': n_host,
                           'note': 'host numpy inputs (4 MB H2D per frame) and host fp16 BEV dict (2.75 MB D2H) per step'},
    }
    if gather_check is not None:
        out['gather_check'] = gather_check
    if ring is not None:
        out['ring_model'] = ring
        out['extras'] = extras
    if not args.no_cpu_baseline and world == 1:           # reported at N = 1 only
        out['cpu_baseline'] = cpu_baseline()
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
