"""KITTI-360 file readers and calibration parsing (host glue; reference: datasets/kitti360_utils.py)."""
import os

import numpy as np


def read_pc_bin_file(path):
    """velodyne .bin -> (N,4) float32 [x,y,z,intensity]."""
    return np.fromfile(path, dtype=np.float32).reshape((-1, 4))


def read_sem_gt_bin_file(path):
    """per-point int16 labels -> (N,1), or None if the file does not exist."""
    if not os.path.isfile(path):
        return None
    return np.fromfile(path, dtype=np.int16)[:, None]


def conv_semantic_ids(sem_gt: np.array, idx2idx: dict):
    """Remaps label ids in place, one (old -> new) pair after the other IN DICT ORDER (a later pair sees
    the result of an earlier one, exactly as the reference's sequential masking does)."""
    for old_idx, new_idx in idx2idx.items():
        sem_gt[sem_gt[:, 0] == old_idx] = new_idx
    return sem_gt


def filter_semseg_pc(pc, filters):
    return pc[~np.isin(pc[:, -1], list(filters))]


def extract_seseg_pc(pc, filter):
    return pc[pc[:, -1] == filter]


def get_transf_matrices(kitti360_path: str):
    """(H_cam_velo, H_velo_cam): camera->velodyne 4x4 from calib_cam_to_velo.txt and its inverse."""
    flat = np.genfromtxt(os.path.join(kitti360_path, 'calibration', 'calib_cam_to_velo.txt'), delimiter=" ")
    H_cam_velo = np.concatenate((flat.reshape((3, 4)), np.array([0, 0, 0, 1]).reshape(1, 4)), axis=0)
    return H_cam_velo, np.linalg.inv(H_cam_velo)


def get_camera_intrinsics(kitti360_path: str):
    """3x4 rectified projection matrix P_rect_00 from calibration/perspective.txt."""
    with open(os.path.join(kitti360_path, 'calibration', 'perspective.txt'), 'r') as f:
        for line in f:
            key, _, rest = line.partition(':')
            if key == 'P_rect_00':
                return np.array(rest.split(), dtype=float).reshape((3, 4))
    raise Exception('Did not find \'P_rect_00\' entry in calibration file.')
