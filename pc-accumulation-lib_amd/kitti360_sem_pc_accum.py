"""KITTI-360 accumulator -- drop-in for the reference's
``kitti360_sem_pc_accum.Kitti360SemanticPointCloudAccumulator``.

Per frame (reference kitti360_sem_pc_accum.py:41-88): estimate T_new_prev, re-express every stored point
and pose in the new ego frame (K2, device), append the new frame's projected / sampled / class-filtered
points (K1, device, fused), evict frames beyond the path horizon.  Nothing is read back from the GPU.

Pose source: the reference calls Open3D point-to-plane ICP (an external C++ library that only FEEDS the
hot path a 4x4 matrix).  ``pose_provider`` is the hook for that input: a callable ``pc (N,4) -> T_new_prev``.
The default provider issues the same Open3D call; without ``open3d`` the first integrate() raises, unless a pose
source is named (PCA_KITTI_T_FILE, PCA_POSE_PROVIDER=<module>:<callable>, or PCA_POSE_PROVIDER=gpu_icp for the device
ICP of pca_amd/icp.py, which is NOT the reference's Open3D registration: parity with it is unpinned).
"""
import numpy as np

from sem_pc_accum import SemanticPointCloudAccumulator


class Kitti360SemanticPointCloudAccumulator(SemanticPointCloudAccumulator):

    def __init__(self, horizon_dist: float, calib_params: dict, icp_threshold: float, semseg_onnx_path: str,
                 semseg_filters: list, sem_idxs: dict, use_gt_sem: bool, bev_params: dict):
        super().__init__(horizon_dist, icp_threshold, semseg_onnx_path, semseg_filters, sem_idxs, use_gt_sem,
                         bev_params)
        self.H_velo_cam = calib_params['h_velo_cam']
        self.P_cam_frame = calib_params['p_cam_frame']
        self.P_velo_frame = calib_params['p_velo_frame']
        self._gpu_icp = None
        self._uploader = None            # pinned staging of host-array observations (pca_amd.ingest.PinnedUploader)
        self._prev_sweep = None
        self.pose_provider = self._default_pose_provider()

    # ---- pose input ----------------------------------------------------------------------------
    def _default_pose_provider(self):
        """Pose source without touching the driver: PCA_KITTI_T_FILE=<.npy of (F,4,4) T_new_prev, one per frame>,
        PCA_POSE_PROVIDER=<module>:<callable(pc) -> 4x4>, or PCA_POSE_PROVIDER=gpu_icp / open3d.  The default is the
        reference's Open3D ICP call; the device ICP is opt-in only (its poses differ from Open3D's)."""
        import os
        path = os.environ.get('PCA_KITTI_T_FILE')
        if path:
            Ts = iter(np.load(path))
            return lambda pc: next(Ts)
        spec = os.environ.get('PCA_POSE_PROVIDER')
        if spec == 'gpu_icp':
            return self._gpu_icp_pose
        if spec == 'open3d':
            return self._icp_pose
        if spec:
            import importlib
            mod, _, fn = spec.partition(':')
            return getattr(importlib.import_module(mod), fn)
        return self._icp_pose

    def _gpu_icp_pose(self, pc):
        """kitti360_sem_pc_accum.py:115-127 on the device: previous sweep -> new sweep, point-to-plane."""
        from pca_amd.icp import GpuIcp
        if self._gpu_icp is None:
            self._gpu_icp = GpuIcp()
        new = GpuIcp.to_device(pc)
        if self._prev_sweep is None:
            self._prev_sweep = new
        reg = self._gpu_icp.register(self._prev_sweep, new, self.icp_threshold, self.icp_trans_init)
        self._prev_sweep = new
        return reg.transformation

    def _icp_pose(self, pc):
        try:
            import open3d as o3d
        except ImportError as e:
            raise ImportError('the KITTI-360 flow takes its poses from Open3D ICP, as the reference does, and open3d is '
                              'not importable; name another pose source: accumulator.pose_provider = callable, '
                              'PCA_KITTI_T_FILE=<poses.npy>, PCA_POSE_PROVIDER=<module>:<callable> or '
                              'PCA_POSE_PROVIDER=gpu_icp (device ICP, not bit-compatible with Open3D)') from e
        pcd_new = self.pc2pcd(pc)
        if self.pcd_prev is None:
            self.pcd_prev = pcd_new
        reg = o3d.pipelines.registration.registration_icp(
            self.pcd_prev, pcd_new, self.icp_threshold, self.icp_trans_init,
            o3d.pipelines.registration.TransformationEstimationPointToPlane())
        self.pcd_prev = pcd_new
        return reg.transformation

    # ---- device inputs -------------------------------------------------------------------------
    def _frame_tensors(self, rgb, pc, sem_gt):
        """Uploads one observation (or passes cuda tensors through).  Returns (frame dict, semseg, H, W).  Host arrays go
        through the pinned staging of pca_amd.ingest.PinnedUploader: all of an observation's arrays in ONE call, their
        host-side copies side by side."""
        import torch
        dev = self.store.device
        dtype_np = {torch.float32: np.float32, torch.uint8: np.uint8}
        want, host = {}, []                     # name -> device tensor / position in `host`

        def up(name, a, dtype):
            a = getattr(a, 'dev', a)             # pca_amd.ingest.DeviceImage
            if isinstance(a, torch.Tensor):
                if a.device != dev or a.dtype != dtype or not a.is_contiguous():
                    a = a.to(device=dev, dtype=dtype).contiguous()
                want[name] = a
            else:
                want[name] = len(host)
                host.append((name, np.ascontiguousarray(a, dtype=dtype_np[dtype])))

        up('pts', pc, torch.float32)
        semseg = None
        if sem_gt is None:
            # (np.asarray: a PIL image is converted as the reference's np.array(rgb) does, an ndarray is not copied again)
            img = rgb if isinstance(rgb, torch.Tensor) or hasattr(rgb, 'dev') else np.asarray(rgb)
            up('rgb', img, torch.uint8)
            # a model that works on the device gets the uploaded image: one H2D serves the CNN and K1, and its class map
            # (utils.onnx_utils.DeviceMap) goes to K1 without ever visiting the host
            if getattr(self.semseg_model, 'accepts_device', False):
                if not isinstance(want['rgb'], torch.Tensor):
                    want['rgb'] = self._upload(host[want['rgb']:want['rgb'] + 1])[0]
                    host.pop()
                semseg = self.semseg_model.pred(want['rgb'])[0, 0]
            else:
                semseg = self.semseg_model.pred(rgb)[0, 0]
            up('sem', semseg, torch.uint8)
        else:
            sg = sem_gt if isinstance(sem_gt, torch.Tensor) else np.asarray(sem_gt)[:, -1]
            up('sem_gt', sg, torch.uint8)       # trainIds 0..18 and 255
        if host:
            for (name, _), t in zip(host, self._upload(host)):
                want[name] = t
        frame = want
        H, W = tuple(frame['sem'].shape) if sem_gt is None else (1, 1)
        return frame, semseg, H, W

    def _upload(self, items):
        if self._uploader is None:
            from pca_amd.ingest import PinnedUploader
            self._uploader = PinnedUploader(self.store.device)
        return self._uploader.upload_many(items)

    # ---- integrate -----------------------------------------------------------------------------
    def integrate(self, observations: list):
        rgb, pc, sem_gt = observations[0]
        if not self.use_gt_sem:
            sem_gt = None
        T_new_prev = np.asarray(self.pose_provider(pc), dtype=np.float64)
        self.T_prev_origin = np.matmul(self.T_prev_origin, T_new_prev)
        frame, semseg, H, W = self._frame_tensors(rgb, pc, sem_gt)

        if len(self._track) > 0:          # move everything stored so far into the new ego frame (K2, owed to the next reader)
            self.update_sem_pcs(T_new_prev)
        self.store.append_kitti([frame], self.P_velo_frame, H, W, self.semseg_filters, sample_mode=self.sample_mode)
        # the pose bookkeeping of the frame in one call: update_poses, the new pose [0,0,0], the newest path segment and
        # the horizon eviction (sem_pc_accum.py:156-228; kitti360_sem_pc_accum.py:60-88)
        idx, path_length = self._track.step(T_new_prev, self.horizon_dist)
        self.rgbs.append(rgb)
        self.semsegs.append(semseg)
        if idx:
            self.store.evict(idx)
            self.rgbs = self.rgbs[idx:]
            self.semsegs = self.semsegs[idx:]
        if path_length is not None:
            print(f'    #pc {self.store.n_frames} |', f'path length {path_length:.2f}')
        self._after_integrate()
        return idx

    def integrate_many(self, batch: list, max_frames_per_launch: int = 64):
        """Extension (no reference counterpart): integrates a list of observation lists -- what integrate() would be
        handed call by call -- with ONE fused K1 call per <= max_frames_per_launch frames and one re-transform pass for
        the frames stored before, instead of a K1 + K2 pair per frame.  Stored points, poses, segment distances and
        evictions are those of the call-by-call form, bit for bit (the transforms a point owes are applied in the same
        order with the same roundings).  Returns the list of integrate()'s return values.  The sharded runner uses it for
        the warm-up prefix of a chunk, where no BEV sample is taken between frames."""
        if self.voxel_dedup:                           # the opt-in de-duplication runs between frames: keep that order
            return [self.integrate(obs) for obs in batch]
        out = []
        for b0 in range(0, len(batch), max_frames_per_launch):
            out += self._integrate_batch(batch[b0:b0 + max_frames_per_launch])
        return out

    def _integrate_batch(self, batch):
        # a prefetching loader reuses its buffers every few batches: refuse BEFORE anything of the accumulator has changed
        # (poses, images, the pose provider's position), so that a refused call leaves it as it was
        from pca_amd.ingest import check_ring_lifetime
        check_ring_lifetime([item for observations in batch for item in observations[0]], len(batch))
        frames, Ts, shape = [], [], None
        for observations in batch:
            rgb, pc, sem_gt = observations[0]
            if not self.use_gt_sem:
                sem_gt = None
            T_new_prev = np.asarray(self.pose_provider(pc), dtype=np.float64)
            self.T_prev_origin = np.matmul(self.T_prev_origin, T_new_prev)
            frame, semseg, H, W = self._frame_tensors(rgb, pc, sem_gt)
            if shape not in (None, (H, W)):
                raise ValueError('integrate_many: all images of a batch must have one size')
            shape = (H, W)
            frames.append(frame)
            Ts.append(T_new_prev)
            self.rgbs.append(rgb)
            self.semsegs.append(semseg)
        self.store.flush_pending()
        self.store.append_kitti(frames, self.P_velo_frame, shape[0], shape[1], self.semseg_filters,
                                sample_mode=self.sample_mode)
        self.store.retransform_batch(np.stack(Ts), len(frames))
        removed, total = [], 0
        for T_new_prev in Ts:                           # host bookkeeping, frame by frame as integrate() does
            idx, path_length = self._track.step(T_new_prev, self.horizon_dist)
            if path_length is not None:
                print(f'    #pc {len(self._track)} |', f'path length {path_length:.2f}')
            removed.append(idx)
            total += idx
            self._integrated += 1
        self.store.poll_status()
        if total:
            self.store.evict(total)
            self.rgbs = self.rgbs[total:]
            self.semsegs = self.semsegs[total:]
        return removed

    def obs2sem_vec_space(self, rgb, pc, sem_gt=None) -> tuple:
        """Host-array form of one observation: ((M,10) rows, pose, semseg, T_new_prev).  integrate() does
        not go through here (it keeps the rows on the device)."""
        from pca_amd.device_store import DeviceStore
        T_new_prev = np.asarray(self.pose_provider(pc), dtype=np.float64)
        self.T_prev_origin = np.matmul(self.T_prev_origin, T_new_prev)
        main, self._store = self._store, DeviceStore(capacity=max(len(pc), 1), max_frames=2)
        try:
            frame, semseg, H, W = self._frame_tensors(rgb, pc, sem_gt)
            self._store.append_kitti([frame], self.P_velo_frame, H, W, self.semseg_filters)
            rows = self._store.rows(0)
        finally:
            self._store = main
        return rows, [0., 0., 0.], semseg, T_new_prev

    # ---- BEV -----------------------------------------------------------------------------------
    def generate_bev(self, present_idx: int = None, bev_num: int = 1, gen_future: bool = False):
        pcs, trajs = self._window_inputs(present_idx, gen_future)
        return self._run_bev(pcs, trajs, bev_num)
