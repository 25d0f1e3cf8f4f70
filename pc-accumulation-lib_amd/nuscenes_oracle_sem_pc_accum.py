"""NuScenes accumulator with oracle (ground-truth) ego poses -- drop-in for the reference's
``nuscenes_oracle_sem_pc_accum.NuScenesOracleSemanticPointCloudAccumulator``.

Per frame (reference :139-270): sample rgb + semseg of the camera each point was projected on, drop
invalid / filtered points, transform ego -> world (K1n, one fused device kernel), run the GT-box tracker
on the host and flag points of moving instances -- retroactively in all stored frames -- with K3.
"""
import numpy as np

from pca_amd.tracker import InstanceTracker
from sem_pc_accum import SemanticPointCloudAccumulator


def homo_transform(tf, points):
    from datasets.nuscenes_utils import homo_transform as _h
    return _h(tf, points)


class NuScenesOracleSemanticPointCloudAccumulator(SemanticPointCloudAccumulator):

    def __init__(self,
                 semseg_onnx_path=None,
                 semseg_filters=None,
                 sem_idxs=None,
                 use_gt_sem=None,
                 bev_params=None,
                 loc=None,
                 get_gt_lanes=False,
                 dataroot=None):
        super().__init__(None, None, semseg_onnx_path, semseg_filters, sem_idxs, use_gt_sem, bev_params)
        if use_gt_sem:
            raise NotImplementedError()
        self.ts = 0
        self.xyz_idx, self.int_idx, self.rgb_idx, self.sem_idx, self.inst_idx, self.dyn_idx = 0, 3, 4, 7, 8, 9
        self._store_args = dict(intensity_div255=True, capacity=1 << 22)
        self.T_global_world = None      # 'global' (map) -> 'world' (first ego frame), set by the first frame
        self.ego_pose_z = 1.            # lift the ego pose from the road surface

        self._tracker = InstanceTracker(track_classes=[0, 1, 2, 3, 5], trans_thresh=1.0)
        self.track_inst_clss = self._tracker.track_classes
        self.dyn_obj_trans_thresh = self._tracker.trans_thresh

        self.map = loc
        self.ego_global_xs = []
        self.ego_global_ys = []
        self.get_gt_lanes = get_gt_lanes
        if self.get_gt_lanes:
            from datasets.nuscenes_lanemap import get_centerlines
            self.gt_lane_poses = get_centerlines(dataroot, loc)

    # tracker state under the reference's attribute names
    @property
    def instances(self):
        return self._tracker.observations

    @property
    def dyn_instances(self):
        return self._tracker.dynamic

    @property
    def token2idx(self):
        return [dict(t, ts=k) for k, t in enumerate(self._tracker.index_at)]

    # ---- integrate -----------------------------------------------------------------------------
    def integrate(self, observations: list):
        obs = observations[0]
        T_ego_global = obs['ego_at_lidar_ts']
        if self.T_global_world is None:
            self.T_global_world = np.linalg.inv(T_ego_global)
            if self.get_gt_lanes:
                self.gt_lane_poses = [homo_transform(self.T_global_world, lane) for lane in self.gt_lane_poses]

        pose, semsegs = self._append_frame(obs['images'], obs['pc'], obs['pc_cam_idx'], T_ego_global,
                                           self.ego_pose_z)
        self._track.append(pose)
        self.rgbs.append(obs['images'])
        self.semsegs.append(semsegs)
        self.ego_global_xs.append(obs['ego_global_x'])
        self.ego_global_ys.append(obs['ego_global_y'])

        # fake detector + tracker on GT boxes: host decides, device flags
        centers = [homo_transform(self.T_global_world, np.expand_dims(c, 0))[0] for c in obs['inst_center']]
        marks = self._tracker.observe(self.ts, obs['inst_tokens'], obs['inst_cls'], centers)
        self.store.mark_dynamic(marks)

        if len(self._track) > 1:
            path_length = self._track.push_segment()
        else:
            path_length = 0
        print(f'    ts {self.ts} | #pc {self.store.n_frames} |', f'path length {path_length:.2f}')
        self.ts += 1
        self._after_integrate()

    def _ego_world(self, T_ego_global, ego_pose_z):
        T_ego_world = self.T_global_world @ T_ego_global
        pose = T_ego_world[:3, -1].tolist()
        pose[2] += ego_pose_z
        return T_ego_world, pose

    def _frame_inputs(self, rgbs, pc, pc_cam_idx):
        """Device tensors of one observation: (pc (n,7) f64, cam_idx (n,) i64, imgs (ncam,H,W,3) u8, sems (ncam,H,W) u8) and
        the semseg maps as the model returned them.  Tensors that are on the device already (pca_amd.ingest) pass through."""
        import torch
        dev = self.store.device
        semsegs = [self.semseg_model.pred(rgb)[0, 0] for rgb in rgbs]
        dev_sems = [getattr(m, 'dev', m) for m in semsegs]          # utils.onnx_utils.DeviceMap: already in HBM

        if getattr(self, '_uploader', None) is None:
            from pca_amd.ingest import PinnedUploader
            self._uploader = PinnedUploader(dev)      # host arrays leave through reused pinned blocks, asynchronously

        def up(kind, a, np_dtype, t_dtype):
            if isinstance(a, torch.Tensor):
                return a.to(device=dev, dtype=t_dtype).contiguous()
            return self._uploader(kind, np.ascontiguousarray(a, dtype=np_dtype))

        stack = getattr(rgbs, 'dev', rgbs)                          # pca_amd.ingest.DeviceImages: the six images as one tensor
        if isinstance(stack, torch.Tensor):
            imgs = up('imgs', stack, np.uint8, torch.uint8)
        else:
            imgs = self._uploader.upload_stack('imgs', [np.asarray(rgb, dtype=np.uint8) for rgb in rgbs])
        if isinstance(dev_sems[0], torch.Tensor):
            sems = self._as_one_stack(dev_sems)              # a model that ran the six cameras as one batch: no copy
            if sems is None or sems.device != dev or sems.dtype != torch.uint8:
                sems = torch.stack([s.to(device=dev, dtype=torch.uint8) for s in dev_sems]).contiguous()
        else:
            sems = self._uploader.upload_stack('sems', [np.asarray(m, dtype=np.uint8) for m in semsegs])
        return (up('pc', pc, np.float64, torch.float64), up('cam', pc_cam_idx, np.int64, torch.int64), imgs, sems, semsegs)

    @staticmethod
    def _as_one_stack(maps):
        """The tensor [k, H, W] the k maps are consecutive slices of, if there is one (else None)."""
        base = maps[0]._base
        if base is None or base.dim() != maps[0].dim() + 1 or base.shape[0] != len(maps) or not base.is_contiguous():
            return None
        step = maps[0].numel() * maps[0].element_size()
        p0 = base.data_ptr()
        for j, m in enumerate(maps):
            if m._base is not base or tuple(m.shape) != tuple(base.shape[1:]) or not m.is_contiguous() or m.data_ptr() != p0 + j * step:
                return None
        return base

    def _append_frame(self, rgbs, pc, pc_cam_idx, T_ego_global, ego_pose_z):
        T_ego_world, pose = self._ego_world(T_ego_global, ego_pose_z)
        pc_d, cam_d, imgs, sems, semsegs = self._frame_inputs(rgbs, pc, pc_cam_idx)
        self.store.append_nusc(pc_d, cam_d, imgs, sems, T_ego_world, self.semseg_filters, sample_mode=self.sample_mode)
        return pose, semsegs

    def integrate_many(self, batch: list):
        """Extension (no reference counterpart): integrates a list of observation lists -- what integrate() would be handed
        call by call; the reference's driver integrates a whole scene before its first BEV (run_nuscenes_bev_gen.py:236-237)
        -- with ONE batched K1n call (front + append launch over the tiles of all frames) and ONE K3 launch for all the
        tracker's marks, instead of a latency-bound launch pair per frame.  Stored points, poses, segment distances, tracker
        state and dynamic flags are those of the call-by-call form, bit for bit."""
        if self.voxel_dedup:                           # the opt-in de-duplication runs between frames: keep that order
            for observations in batch:
                self.integrate(observations)
            return
        # a prefetching loader reuses its buffers every RING batches: refuse BEFORE anything of the accumulator has changed
        # (poses, tracker, images, ts), so that a refused call leaves it as it was
        from pca_amd.ingest import check_ring_lifetime
        check_ring_lifetime([observations[0].get(k) for observations in batch for k in ('pc', 'pc_cam_idx', 'images')],
                            len(batch))
        frames, marks = [], []
        for observations in batch:
            obs = observations[0]
            T_ego_global = obs['ego_at_lidar_ts']
            if self.T_global_world is None:
                self.T_global_world = np.linalg.inv(T_ego_global)
                if self.get_gt_lanes:
                    self.gt_lane_poses = [homo_transform(self.T_global_world, lane) for lane in self.gt_lane_poses]
            T_ego_world, pose = self._ego_world(T_ego_global, self.ego_pose_z)
            pc_d, cam_d, imgs, sems, semsegs = self._frame_inputs(obs['images'], obs['pc'], obs['pc_cam_idx'])
            frames.append(dict(pc=pc_d, cam_idx=cam_d, imgs=imgs, sems=sems, T=T_ego_world))
            self._track.append(pose)
            self.rgbs.append(obs['images'])
            self.semsegs.append(semsegs)
            self.ego_global_xs.append(obs['ego_global_x'])
            self.ego_global_ys.append(obs['ego_global_y'])
            centers = [homo_transform(self.T_global_world, np.expand_dims(c, 0))[0] for c in obs['inst_center']]
            marks += list(self._tracker.observe(self.ts, obs['inst_tokens'], obs['inst_cls'], centers))
            path_length = self._track.push_segment() if len(self._track) > 1 else 0
            print(f'    ts {self.ts} | #pc {self.store.n_frames + len(frames)} |', f'path length {path_length:.2f}')
            self.ts += 1
            self._integrated += 1
        self.store.append_nusc_many(frames, self.semseg_filters, sample_mode=self.sample_mode)
        self.store.mark_dynamic(marks)                 # flags only ever go from 0 to 1: the order of the marks is immaterial

    def obs2sem_vec_space(self, rgbs, pc, pc_cam_idx, T_ego_global, ego_pose_z: float = 0) -> tuple:
        """Host-array form: ((M,10) rows, pose, semsegs).  Raises AssertionError like the reference if a
        point assigned to a camera has pixel coordinates outside the open box (1, wh-1)."""
        from pca_amd.device_store import DeviceStore
        main, self._store = self._store, DeviceStore(capacity=max(len(pc), 1), max_frames=2, intensity_div255=True)
        try:
            pose, semsegs = self._append_frame(rgbs, pc, pc_cam_idx, T_ego_global, ego_pose_z)
            self._store.check_status()
            rows = self._store.rows(0)
        finally:
            self._store = main
        return rows, pose, semsegs

    # ---- trajectories of other agents (host) ---------------------------------------------------
    def get_split_dyn_obj_trajs(self, split_idx: int, skip_ego_traj: bool = True):
        return self._tracker.split_trajectories(split_idx)

    def get_dyn_obj_trajs(self, ts_start: int = 0, ts_end: int = None, skip_ego_traj: bool = True):
        out = self._tracker.trajectories(ts_start, ts_end)
        if not skip_ego_traj:
            out.append(self._track.as_array().tolist())
        return out

    # ---- BEV -----------------------------------------------------------------------------------
    def _window_inputs_for(self, present_idx, gen_future):
        others = self.get_split_dyn_obj_trajs(present_idx)
        lanes = self.gt_lane_poses if self.get_gt_lanes else None
        return self._window_inputs(present_idx, gen_future, others, lanes)

    def generate_bev(self, present_idx: int = None, bev_num: int = 1, gen_future: bool = False):
        pcs, trajs = self._window_inputs_for(present_idx, gen_future)
        self._check_status_once()
        return self._run_bev(pcs, trajs, bev_num)

    def generate_bev_many(self, present_idxs, gen_future: bool = True):
        self._check_status_once()
        return super().generate_bev_many(present_idxs, gen_future)

    def _check_status_once(self):
        """The reference asserts on a pixel coordinate outside the image inside integrate(); here the kernels raise a status
        bit that is read -- one stream synchronisation -- before the first BEV after new frames, not before every sample."""
        if getattr(self, '_status_checked_at', -1) != self.ts:
            self.store.check_status()
            self._status_checked_at = self.ts

    @staticmethod
    def get_tf_pose(inst_tf: np.array) -> np.array:
        return inst_tf[:3, -1]

    @staticmethod
    def get_obj_inst_poses_ts(inst_obs: list) -> tuple:
        poses, tss = zip(*inst_obs)
        return poses, tss

    @staticmethod
    def cal_pose_change(pose_0: np.array, pose_1: np.array) -> float:
        return np.linalg.norm(pose_1 - pose_0)
