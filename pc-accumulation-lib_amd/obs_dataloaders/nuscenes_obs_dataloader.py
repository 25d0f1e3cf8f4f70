"""NuScenes observation loader (reference: obs_dataloaders/nuscenes_obs_dataloader.py).

Same class, constructor and observation dict as the reference, so that `run_nuscenes_bev_gen.py` imports and runs
unchanged.  What runs where:

  * lidar -> ego -> global -> 6 cameras, pinhole projection, "last camera wins" (reference :162-202): ONE device launch
    (`datasets.nuscenes_utils.project_to_cameras`, kernel K0n) instead of 2 + 6 numpy transforms and 6 projections;
  * merging the lidar sweeps of a sample and labelling points with their GT boxes (the reference's
    `inst_centric_get_sweeps`, datasets/nuscenes_utils.py:332-531) is done by `NuScenesDataloader.sweep_provider`, a
    callable with that function's signature returning its dict; the default is this package's own
    `datasets.nuscenes_sweeps.inst_centric_get_sweeps`.

nuscenes-devkit / pyquaternion are only imported when a real dataset is walked.
"""
import numpy as np

from datasets.nuscenes_utils import NuScenesCamera, NuScenesLidar, project_to_cameras
from obs_dataloaders.obs_dataloader import ObservationDataloader

class NuScenesDataloader(ObservationDataloader):
    sweep_provider = None       # callable(nusc, sample_token, **cfg) -> dict (points, instances_token, ...)

    def __init__(self, nusc, scene_ids=None, batch_size=1, num_sweeps=5):
        super().__init__(None, batch_size)
        self.nusc = nusc
        self.num_sweeps = num_sweeps
        self.cam_channels = ['CAM_FRONT', 'CAM_FRONT_LEFT', 'CAM_FRONT_RIGHT', 'CAM_BACK', 'CAM_BACK_LEFT',
                             'CAM_BACK_RIGHT']
        if scene_ids is None:
            scene_ids = range(self.nusc.scene)     # as the reference (:31-32): raises TypeError for a list of scenes
        self.sample_tokens = []
        for scene_idx in scene_ids:
            token = self.nusc.scene[scene_idx]['first_sample_token']
            while token != '':
                self.sample_tokens.append(token)
                token = self.nusc.get('sample', token)['next']
        # columns of the sweep matrix
        self.int_idx, self.sweep_idx, self.inst_idx, self.cls_idx = 3, 5, 6, 7
        big = 1000
        self.pc_range = [-big, -big, -big, big, big, big]

    def __len__(self):
        return len(self.sample_tokens)

    # ---- pieces a test (or another dataset layout) can replace ------------------------------------------------
    def _sweeps(self, sample_token, cfg):
        if self.sweep_provider is not None:
            return self.sweep_provider(self.nusc, sample_token, **cfg)
        from datasets.nuscenes_sweeps import inst_centric_get_sweeps
        return inst_centric_get_sweeps(self.nusc, sample_token, **cfg)

    def _lidar(self, sample):
        return NuScenesLidar(self.nusc, self.nusc.get('sample_data', sample['data']['LIDAR_TOP']))

    def _cameras(self, sample):
        return [NuScenesCamera(self.nusc, self.nusc.get('sample_data', sample['data'][ch])) for ch in self.cam_channels]

    # ---- one observation -------------------------------------------------------------------------------------
    def read_host(self, idx):
        """Everything of one observation that needs no GPU: sample records, the merged sweeps, sensor poses and intrinsics,
        the camera images (as the sensor objects hold them).  `finish_obs` turns it into the reference's observation; the
        ingest pipeline (pca_amd.ingest.NuScenesPrefetchingLoader) runs this part on reader threads, ahead of the GPU."""
        sample_token = self.sample_tokens[idx]
        sample = self.nusc.get('sample', sample_token)
        obs = {'meta': {'sample_token': sample_token, 'scene_token': sample['scene_token'],
                        'cam_channels': self.cam_channels}}
        cfg = {
            'n_sweeps': self.num_sweeps,
            'center_radius': 2.0,
            'in_box_tolerance': 5e-2,
            'return_instances_last_box': True,
            'point_cloud_range': self.pc_range,
            'detection_classes': ('car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'motorcycle', 'bicycle',
                                  'pedestrian'),
            'map_point_feat2idx': {'sweep_idx': self.sweep_idx, 'inst_idx': self.inst_idx, 'cls_idx': self.cls_idx},
        }
        out = self._sweeps(sample_token, cfg)
        lidar = self._lidar(sample)
        cameras = self._cameras(sample)
        obs['ego_at_lidar_ts'] = lidar.glob_from_ego
        obs['images'] = [cam.img for cam in cameras]
        obs['inst_tokens'] = out['instances_token']
        obs['inst_cls'] = [int(cls.item()) for cls in out['instances_name']]
        obs['inst_center'] = out['instances_center']
        sd = self.nusc.get('sample_data', sample['data']['LIDAR_TOP'])
        x, y, _ = self.nusc.get('ego_pose', sd['ego_pose_token'])['translation']
        obs['ego_global_x'] = x
        obs['ego_global_y'] = y
        geom = dict(pc=np.asarray(out['points']),               # lidar frame, (N, >= 7) sweep matrix
                    ego_from_lidar=lidar.ego_from_self, glob_from_ego=lidar.glob_from_ego,
                    cams_glob_from_self=[cam.glob_from_self for cam in cameras], cams_K=[cam.cam_K for cam in cameras],
                    cams_wh=[cam.img_wh for cam in cameras])
        return obs, geom

    def finish_obs(self, obs, geom):
        """Projection onto the cameras (K0n on the device) and the (N,7) point rows, as host arrays."""
        pc = geom['pc']
        pc_in_ego, pc_uv, pc_cam_idx = project_to_cameras(pc[:, :3], geom['ego_from_lidar'], geom['glob_from_ego'],
                                                          geom['cams_glob_from_self'], geom['cams_K'], geom['cams_wh'])
        obs['pc_cam_idx'] = pc_cam_idx
        obs['pc'] = np.concatenate([pc_in_ego, pc[:, self.int_idx:self.int_idx + 1], pc_uv,
                                    pc[:, self.inst_idx:self.inst_idx + 1]], axis=1)
        return obs

    def read_obs(self, idx):
        """dict with the reference's keys: images, pc (N,7) [x,y,z (ego), intensity, u, v, instance idx], pc_cam_idx
        (N,), ego_at_lidar_ts (4,4), meta, inst_tokens, inst_cls, inst_center, ego_global_x, ego_global_y."""
        return self.finish_obs(*self.read_host(idx))
