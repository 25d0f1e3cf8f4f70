"""Synthetic sequence shared by the sharded-run tests (CPU planning tests and the GPU equality tests)."""
import numpy as np

H, W, N_PTS = 48, 80, 1500
FILTERS = [10, 11, 12, 16, 18, 255]
SEM_IDXS = {'road': 0, 'car': 13, 'truck': 14, 'bus': 15, 'motorcycle': 17}
BEV = dict(type='sem', view_size=24, pixel_size=32, max_trans_radius=0., zoom_thresh=0., do_warp=False,
           int_scaler=20., int_sep_scaler=20., int_mid_threshold=0.5, height_filter=None)
ACCUM_H, BEV_H, SPACING = 60., 20., 1.
CAM_TO_VELO = np.array([[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418],
                        [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
                        [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824], [0, 0, 0, 1]])
P_VELO = np.array([[40., 0, W / 2, 0], [0, 40., H / 2, 0], [0, 0, 1, 0]]) @ np.linalg.inv(CAM_TO_VELO)


def transforms(n, seed=3):
    """T_new_prev per frame: ~1 m steps of varying length on a wandering curve (no two path sums tie)."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        a = rng.normal(-0.003, 0.004)
        c, s = np.cos(a), np.sin(a)
        step = rng.uniform(0.7, 1.3)
        out.append(np.array([[c, -s, 0, -step], [s, c, 0, rng.normal(0, 0.01)], [0, 0, 1, 0], [0, 0, 0, 1.]]))
    return np.stack(out)


def observation(seq, frame):
    """The observations list integrate() takes, GT semantics (class per point), seeded per (sequence, frame)."""
    from PIL import Image
    rng = np.random.default_rng(100_000 * seq + frame)
    pc = np.stack([rng.uniform(-18, 18, N_PTS), rng.uniform(-18, 18, N_PTS), rng.uniform(-2, 2, N_PTS),
                   rng.uniform(0, 1, N_PTS)], 1).astype(np.float32)
    sem = rng.choice([0, 0, 1, 2, 8, 13, 14, 10], (N_PTS, 1))
    img = Image.fromarray(np.zeros((H, W, 3), np.uint8))
    return [(img, pc, sem)]


def make_accumulator(Ts):
    """Fresh drop-in accumulator whose pose source serves Ts[frame] for the frames it is handed, in order."""
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': P_VELO}
    acc = Kitti360SemanticPointCloudAccumulator(ACCUM_H, calib, 1e3, None, FILTERS, SEM_IDXS, True, dict(BEV))
    acc._store_args = dict(capacity=1 << 18, max_frames=512)
    return acc


def run_job(job, Ts, seq=0, warm_batch=16):
    """Runs a ChunkJob; returns {frame: planes float16 [21,px,px] (host)}."""
    from pca_amd import sharded_run as sr
    acc = make_accumulator(Ts)
    cursor = [job.warm_start]

    def provider(pc):
        T = Ts[cursor[0]]
        cursor[0] += 1
        return T
    acc.pose_provider = provider
    out = {}

    def on_sample(f, present_idx):
        bev = acc.generate_bev(present_idx, 1, gen_future=True)[0]
        out[f] = np.concatenate([np.concatenate([bev[f'road_{s}'][None], bev[f'intensity_{s}'][None], bev[f'rgb_{s}'],
                                                 bev[f'dynamic_{s}'][None], bev[f'elevation_{s}'][None]])
                                 for s in ('present', 'future', 'full')])
    sr.run_chunk(acc, lambda f: observation(seq, f), job, on_sample, warm_batch)
    acc.store.check_status()
    return out
