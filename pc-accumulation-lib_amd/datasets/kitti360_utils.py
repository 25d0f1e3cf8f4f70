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


_LUT_CACHE = {}


def conv_semantic_ids(sem_gt: np.array, idx2idx: dict):
    """Remaps label ids in place, one (old -> new) pair after the other IN DICT ORDER (a later pair sees
    the result of an earlier one, exactly as the reference's sequential masking does).
    For the usual input -- an (N,1) integer array -- the 46 passes are composed into one lookup table (same result,
    1.5 ms -> 0.1 ms per 120 k-point frame); anything else takes the sequential form."""
    if sem_gt.ndim == 2 and sem_gt.shape[1] == 1 and sem_gt.dtype.kind in 'iu' and len(idx2idx) > 0 and \
            all(np.iinfo(sem_gt.dtype).min <= int(v) <= np.iinfo(sem_gt.dtype).max for v in idx2idx.values()):
        key = tuple(idx2idx.items())
        hit = _LUT_CACHE.get(key)
        if hit is None:
            keys = [int(k) for k in idx2idx] + [int(v) for v in idx2idx.values()]
            lo, hi = min(keys), max(keys)
            table = np.arange(lo, hi + 1, dtype=np.int64)
            for old_idx, new_idx in idx2idx.items():
                table[table == old_idx] = new_idx
            if len(_LUT_CACHE) > 16:
                _LUT_CACHE.clear()
            hit = _LUT_CACHE[key] = (table, lo)
        table, lo = hit
        col = sem_gt[:, 0].astype(np.int64)
        k = col - lo
        inside = (k >= 0) & (k < table.size)
        col[inside] = table[k[inside]]
        sem_gt[:, 0] = col                                   # numpy casts back to the array's dtype as the masked assignment does
        return sem_gt
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
