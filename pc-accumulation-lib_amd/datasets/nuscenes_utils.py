"""NuScenes helpers (reference: datasets/nuscenes_utils.py).

Hot-path pieces and where they run:
  homo_transform (:46-60)           host numpy here for tiny inputs (instance centres, lane poses); per-point
                                     transforms run in the K1n / K0n device kernels
  pts_feat_from_img (:181-214)      'nearest' is fused into K1n on the device; the numpy form here serves
                                     direct callers.  'bilinear' (2-D feature maps only, as in the reference) runs
                                     on the device: pca_sample_bilinear; the same weights drive the opt-in
                                     sample_mode='bilinear' of K1 / K1n
  NuScenesCamera.project_pts3d      device kernel K0n via ``project_to_cameras`` (all cameras in one launch)
nuscenes-devkit / pyquaternion are imported lazily: everything that does not touch the dataset works
without them.
"""
import os.path as osp
from abc import ABC

import numpy as np


def homo_transform(tf, points):
    """(N,3) -> (N,3): rows of (tf @ [p;1])[:3]."""
    assert tf.shape == (4, 4), f"{tf.shape} is not (4, 4)"
    assert points.shape == (points.shape[0], 3), f"{points.shape} is not (N, 3)"
    homo = np.concatenate([points, np.ones((points.shape[0], 1))], axis=1)
    return (tf @ homo.T)[:3, :].T


def pts_feat_from_img(pts_uv, img, method='bilinear'):
    """Samples img (H,W[,C]) at float pixel coordinates (N,2) = (u along x, v along y)."""
    assert isinstance(img, np.ndarray), f"{type(img)} is not supported"
    assert method in ('bilinear', 'nearest'), f"{method} is not supported"
    img_wh = np.array([img.shape[1], img.shape[0]], dtype=float)
    assert np.all((pts_uv > 1) & (pts_uv < img_wh - 1)), "pts_uv must be all inside image"
    if method == 'nearest':
        uv = np.round(pts_uv).astype(int)
        return img[uv[:, 1], uv[:, 0]]
    if img.ndim != 2:
        # the reference's branch multiplies (N,) weights with (N,C) features: it only broadcasts for 2-D maps
        raise ValueError(f'operands could not be broadcast together with shapes ({pts_uv.shape[0]},) {img[0, [0]].shape}')
    return sample_bilinear_device(pts_uv, img)


def sample_bilinear_device(pts_uv, feat_map):
    """(N,2) float pixel coordinates, (H,W) map -> (N,) f64 bilinear samples on the device (pca_sample_bilinear: floor /
    ceil neighbours, un-fused f64 weights, the fourth as 1 - (sum of the others) -- the reference's arithmetic)."""
    import torch
    from pca_amd import _lib
    ctx = _lib.Context.get()
    dev = torch.device('cuda', ctx.device_index)
    m = torch.from_numpy(np.ascontiguousarray(feat_map, dtype=np.float64)).to(dev)
    uv = torch.from_numpy(np.ascontiguousarray(pts_uv, dtype=np.float64)).to(dev)
    out = torch.empty(uv.shape[0], dtype=torch.float64, device=dev)
    ctx.check(ctx.lib.pca_sample_bilinear(ctx.h, m.data_ptr(), int(m.shape[0]), int(m.shape[1]), uv.data_ptr(),
                                          int(uv.shape[0]), out.data_ptr(), ctx.stream()))
    if ctx.status() & _lib.STATUS_UV_OUT_OF_IMAGE:
        raise AssertionError('pts_uv must be all inside image')
    return out.cpu().numpy()


def project_to_cameras(pc_lidar, ego_from_lidar, glob_from_ego, cams_glob_from_self, cams_K, cams_wh):
    """lidar -> ego -> global -> every camera, pinhole projection, LAST camera wins (device kernel K0n;
    reference: obs_dataloaders/nuscenes_obs_dataloader.py:162-202).
    Returns host arrays (pc_in_ego (N,3), pc_uv (N,2), pc_cam_idx (N,) int64)."""
    import ctypes as C

    import torch
    from pca_amd import _lib
    ctx = _lib.Context.get()
    dev = torch.device('cuda', ctx.device_index)
    pc = torch.from_numpy(np.ascontiguousarray(pc_lidar[:, :3], dtype=np.float64)).to(dev)
    n = pc.shape[0]
    ncam = len(cams_glob_from_self)
    cam_from_glob = np.stack([np.linalg.inv(T) for T in cams_glob_from_self])
    ego = torch.empty((n, 3), dtype=torch.float64, device=dev)
    uv = torch.empty((n, 2), dtype=torch.float64, device=dev)
    cam = torch.empty((n, ), dtype=torch.int64, device=dev)
    ctx.check(ctx.lib.pca_nusc_project_cams(
        ctx.h, pc.data_ptr(), n, _lib.f64_array(ego_from_lidar, 16), _lib.f64_array(glob_from_ego, 16),
        _lib.f64_array(cam_from_glob, 16 * ncam), _lib.f64_array(np.stack(cams_K), 9 * ncam),
        _lib.f64_array(np.stack(cams_wh), 2 * ncam), ncam, ego.data_ptr(), uv.data_ptr(), cam.data_ptr(),
        ctx.stream()))
    return ego.cpu().numpy(), uv.cpu().numpy(), cam.cpu().numpy()


class NuScenesSensor(ABC):
    """Pose chain of one sample_data record: ego_from_self, glob_from_ego, glob_from_self (4x4)."""

    def __init__(self, nusc, record):
        from nuscenes.utils.geometry_utils import transform_matrix
        from pyquaternion import Quaternion
        self.token = record['token']
        self.channel = record['channel']
        cs = nusc.get('calibrated_sensor', record['calibrated_sensor_token'])
        self.ego_from_self = transform_matrix(cs['translation'], Quaternion(cs['rotation']))
        ego = nusc.get('ego_pose', record['ego_pose_token'])
        self.glob_from_ego = transform_matrix(ego['translation'], Quaternion(ego['rotation']))
        self.glob_from_self = self.glob_from_ego @ self.ego_from_self
        self.img = None
        self.img_hw = None
        self.cam_K = None
        self.pc = None


class NuScenesCamera(NuScenesSensor):

    def __init__(self, nusc, record):
        from PIL import Image
        super().__init__(nusc, record)
        self.img_wh = np.array([record['width'], record['height']], dtype=float)
        self.img = Image.open(osp.join(nusc.dataroot, record['filename']))
        cs = nusc.get('calibrated_sensor', record['calibrated_sensor_token'])
        self.cam_K = np.array(cs['camera_intrinsic'])

    def project_pts3d(self, pc, depth_thres=1e-3):
        """(N,3) camera-frame points -> (uv (N,2) with -10 for invalid depth, mask_in_img (N,)).
        Host numpy form for direct callers; the per-frame 6-camera loop uses ``project_to_cameras``."""
        valid = pc[:, 2] > depth_thres
        out = np.zeros((pc.shape[0], 2), dtype=float) - 10
        out[valid] = view_points(pc[valid].T, self.cam_K, normalize=True)[:2, :].T
        inside = np.all((out > 1) & (out < self.img_wh - 1), axis=1)
        return out, inside & valid


def view_points(points, view, normalize):
    """nuscenes-devkit's published pinhole helper, restated: pad `view` into a 4x4, multiply the
    homogeneous (4,N) points, keep 3 rows, optionally divide by the depth row."""
    viewpad = np.eye(4)
    viewpad[:view.shape[0], :view.shape[1]] = view
    n = points.shape[1]
    out = np.dot(viewpad, np.concatenate((points, np.ones((1, n)))))[:3, :]
    if normalize:
        out = out / out[2:3, :].repeat(3, 0).reshape(3, n)
    return out


class NuScenesLidar(NuScenesSensor):

    def __init__(self, nusc, lidar_record):
        super().__init__(nusc, lidar_record)

    @staticmethod
    def get_pointcloud(nusc, sample_record, num_sweeps=None):
        from nuscenes.utils.data_classes import LidarPointCloud
        if num_sweeps is not None:
            assert sample_record is not None and num_sweeps <= 10
            pc, times = LidarPointCloud.from_file_multisweep(nusc, sample_record, 'LIDAR_TOP', 'LIDAR_TOP',
                                                             nsweeps=num_sweeps)
            return np.vstack([pc.points[:4, :], times]).T
        rec = nusc.get('sample_data', sample_record['data']['LIDAR_TOP'])
        return LidarPointCloud.from_file(osp.join(nusc.dataroot, rec['filename'])).points[:4, :].T
