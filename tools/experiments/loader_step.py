"""The loader-driven step on a full-size synthetic KITTI-360 tree (120 k points, 376x1408 PNG, GT labels): ms per frame
through the plain loader and through PrefetchingLoader with 1 / 4 / 8 reader threads (integrate only, no BEV)."""
import os
import sys
import tempfile
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'pc-accumulation-lib_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import builtins  # noqa: E402
from fake_kitti import SEQ, write_tree  # noqa: E402
from datasets.kitti360_utils import get_camera_intrinsics, get_transf_matrices  # noqa: E402
from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator  # noqa: E402
from obs_dataloaders.kitti360_obs_dataloader import Kitti360Dataloader  # noqa: E402
from pca_amd.ingest import PrefetchingLoader  # noqa: E402

N = 48
root = os.path.join(tempfile.mkdtemp(), 'KITTI-360')
write_tree(root, first_idx=0, n_frames=N, n_pts=120000, H=376, W=1408)
_, h_velo_cam = get_transf_matrices(root)
p_cam = get_camera_intrinsics(root)
calib = {'h_velo_cam': h_velo_cam, 'p_cam_frame': p_cam, 'p_velo_frame': np.matmul(p_cam, h_velo_cam)}
Ts = np.load(os.path.join(root, 'T_new_prev.npy'))
real_print = builtins.print
FILTERS = [10, 11, 12, 13, 14, 15, 16, 18]
SEM_IDXS = {'road': 0, 'car': 13, 'truck': 14, 'bus': 15, 'motorcycle': 17}
BEV = dict(type='sem', view_size=80, pixel_size=256, max_trans_radius=0., zoom_thresh=0., do_warp=False, int_scaler=20.,
           int_sep_scaler=20., int_mid_threshold=0.5, height_filter=None)


def run(loader, label):
    acc = Kitti360SemanticPointCloudAccumulator(200., calib, 1e3, 'none', FILTERS, SEM_IDXS, True, BEV)
    it = iter(Ts)
    acc.pose_provider = lambda pc: next(it)
    builtins.print = lambda *a, **k: None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for obs in loader:
        acc.integrate(obs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    builtins.print = real_print
    print('%-34s %.2f ms per frame' % (label, 1e3 * dt / N), flush=True)


mk = lambda: Kitti360Dataloader(root, 1, [SEQ], [0], [N])
run(mk(), 'plain loader (warm-up)')
run(mk(), 'plain loader')
for images in ('1', '0'):
    os.environ['PCA_INGEST_IMAGES'] = images
    os.environ['PCA_INGEST_THREAD'] = '0'
    run(PrefetchingLoader(mk(), depth=8), 'Prefetching, images=%s, inline reads' % images)
    os.environ['PCA_INGEST_THREAD'] = '1'
    for th, depth in ((1, 8), (4, 8), (8, 8), (16, 24), (32, 48)):
        os.environ['PCA_INGEST_THREADS'] = str(th)
        run(PrefetchingLoader(mk(), depth=depth), 'Prefetching, images=%s, %d reader threads, depth %d' % (images, th, depth))
    os.environ.pop('PCA_INGEST_THREADS', None)
    run(PrefetchingLoader(mk(), depth=48), 'Prefetching, images=%s, default threads, depth 48' % images)
