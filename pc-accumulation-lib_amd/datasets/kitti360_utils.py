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


def _calibration(root, name):
    return os.path.join(root, 'calibration', name)


def get_transf_matrices(kitti360_path: str):
    """(camera -> velodyne, velodyne -> camera) as 4x4 matrices; the file holds the 12 numbers of the top 3x4 block."""
    top = np.genfromtxt(_calibration(kitti360_path, 'calib_cam_to_velo.txt'), delimiter=" ").reshape((3, 4))
    cam_to_velo = np.vstack([top, [[0, 0, 0, 1]]]).astype(float)
    return cam_to_velo, np.linalg.inv(cam_to_velo)


def get_camera_intrinsics(kitti360_path: str):
    """The 3x4 rectified projection of camera 00 ('P_rect_00: <12 numbers>' in perspective.txt)."""
    wanted = 'P_rect_00'
    for line in open(_calibration(kitti360_path, 'perspective.txt')):
        key, _, numbers = line.partition(':')
        if key == wanted:
            return np.array(numbers.split(), dtype=float).reshape((3, 4))
    raise Exception(f"Did not find '{wanted}' entry in calibration file.")
