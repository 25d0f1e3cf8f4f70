"""Writes a tiny synthetic KITTI-360 directory tree (paths per the reference's Kitti360Dataloader and
kitti360_utils: velodyne .bin, rectified PNG, per-point label .bin, calibration text files)."""
import os

import numpy as np
from PIL import Image

SEQ = '2013_05_28_drive_0000_sync'
CAM_TO_VELO = np.array([[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418],
                        [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
                        [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824]])


def t_new_prev(k):
    a = -0.004
    c, s = np.cos(a), np.sin(a)
    T = np.array([[c, -s, 0, -1.0], [s, c, 0, 0.01 * (k % 3)], [0, 0, 1, 0], [0, 0, 0, 1.]])
    return T


def write_tree(root, first_idx, n_frames, n_pts=3000, H=64, W=96, seed=5):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, 'calibration'), exist_ok=True)
    with open(os.path.join(root, 'calibration', 'calib_cam_to_velo.txt'), 'w') as f:
        f.write(' '.join(repr(float(v)) for v in CAM_TO_VELO.ravel()) + '\n')
    P = np.array([[40., 0, W / 2, 0], [0, 40., H / 2, 0], [0, 0, 1, 0]])
    with open(os.path.join(root, 'calibration', 'perspective.txt'), 'w') as f:
        f.write('calib_time: 00-xxx-0000 00:00:00\n')
        f.write('P_rect_00: ' + ' '.join(repr(float(v)) for v in P.ravel()) + '\n')
        f.write('P_rect_01: ' + ' '.join(repr(float(v)) for v in P.ravel()) + '\n')
    dirs = [os.path.join(root, 'data_3d_raw', SEQ, 'velodyne_points', 'data'),
            os.path.join(root, 'data_2d_raw', SEQ, 'image_00', 'data_rect'),
            os.path.join(root, 'data_3d_semantics', 'raw', SEQ, 'labels')]
    for d in dirs:
        os.makedirs(d, exist_ok=True)
    frames = []
    for k in range(n_frames):
        name = f'{first_idx + k:010d}'
        pc = np.stack([rng.uniform(-25, 25, n_pts), rng.uniform(-25, 25, n_pts), rng.uniform(-2, 3, n_pts),
                       rng.uniform(0, 1, n_pts)], 1).astype(np.float32)
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        lab = rng.choice([7, 8, 11, 21, 26, 27, 23, 24, 6], n_pts).astype(np.int16)      # KITTI-360 label ids
        pc.tofile(os.path.join(dirs[0], name + '.bin'))
        Image.fromarray(img).save(os.path.join(dirs[1], name + '.png'))
        lab.tofile(os.path.join(dirs[2], name + '.bin'))
        frames.append((pc, img, lab))
    Ts = np.stack([t_new_prev(k) for k in range(n_frames)])
    np.save(os.path.join(root, 'T_new_prev.npy'), Ts)
    return frames, Ts, P
