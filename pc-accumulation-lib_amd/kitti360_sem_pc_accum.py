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
import os

import numpy as np

from sem_pc_accum import SemanticPointCloudAccumulator

_MODS = None


def _mods():
    """What the per-frame calls need from modules that are imported on first use (an `import` statement inside a function that
    runs ten thousand times a second is a dictionary lookup and a rebinding each: ~4 us of the step, all of them together)."""
    global _MODS
    if _MODS is None:
        import ctypes

        import torch
        from bev_generator.sem_bev import LazyBev, SemBEVGenerator, _PendingCopy
        from pca_amd import host_logic
        from pca_amd._lib import PcaKittiObs
        _MODS = (ctypes, torch, LazyBev, _PendingCopy, SemBEVGenerator, host_logic, PcaKittiObs)
    return _MODS


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
        self._cobs = None                # the observation as the library takes it (reused)
        self._ga = None                  # argument block of pca_kitti_generate_bev_v, kept between calls
        self._ga_traj = None
        self._fast = os.environ.get('PCA_FAST_CALLS', '1') != '0'      # one library call per driver call (0: the general path)
        # K1 of integrate() left for the generate_bev() that follows it (it rides in the raster's first kernel: pca_k1_defer);
        # 0: K1 runs inside integrate()
        self._defer_k1 = self._fast and os.environ.get('PCA_FUSE_K1', '1') != '0'
        self._prev_sweep = None
        self._sweep_dev = None           # (host sweep, its device copy) of the frame the device ICP has just registered
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
        import torch
        from pca_amd.icp import GpuIcp
        if self._gpu_icp is None:
            self._gpu_icp = GpuIcp()
        # ONE upload serves the registration and K1: an (N,4) f32 host sweep goes through the pinned staging (asynchronous copy,
        # no pageable .to(device) the host would sit in) and integrate() hands the device tensor on to K1
        if not isinstance(pc, torch.Tensor) and getattr(pc, 'dtype', None) == np.float32 and pc.ndim == 2 and pc.shape[1] == 4:
            new = self._upload([('icp_pts', pc)])[0]
            self._sweep_dev = (pc, new)
        else:
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
    def _obs_pointers(self, rgb, pc, sem_gt):
        """The observation as pca_kitti_integrate takes it: (PcaKittiObs, semseg, H, W, what must stay alive), host arrays
        named in the struct's host_mask (the library stages them: one pinned block, ONE H2D copy), device tensors passed as
        they are.  None if this observation needs the general path (a model that wants the image on the device first)."""
        _, torch, _, _, _, _, PcaKittiObs = _mods()
        dev = self.store.device
        obs = self._cobs
        if obs is None:
            obs = self._cobs = PcaKittiObs()
        keep, mask = [], 0

        def put(a, np_dtype, t_dtype, bit):
            nonlocal mask
            if not isinstance(a, torch.Tensor):
                # device wrappers, WITHOUT running their lazy properties: a device tensor found in the object's own
                # attributes is used as it is -- utils.onnx_utils.DeviceMap keeps its class map under 'dev' (going through
                # np.asarray would copy it to the host and stage it back up, frame after frame), pca_amd.ingest.DeviceImage
                # its upload under '_dev'; a DeviceImage whose copy is not there yet ('_dev' is None: asking for `.dev` would
                # start a synchronous pageable upload only to be thrown away) hands over its host array with the others
                own = getattr(a, '__dict__', {})
                d = own.get('dev') if isinstance(own.get('dev'), torch.Tensor) else own.get('_dev')
                if isinstance(d, torch.Tensor):
                    a = d
                elif '_host' in own and '_dev' in own:
                    a = a.host
            if isinstance(a, torch.Tensor):
                if a.device != dev or a.dtype != t_dtype or not a.is_contiguous():
                    a = a.to(device=dev, dtype=t_dtype).contiguous()
                keep.append(a)
                return a.data_ptr(), tuple(a.shape)
            a = np.ascontiguousarray(a, dtype=np_dtype)
            keep.append(a)
            mask |= bit
            return a.ctypes.data, a.shape

        obs.pts, shape = put(pc, np.float32, torch.float32, 1)
        if len(shape) != 2 or shape[1] != 4:
            raise ValueError('pc must be (N, 4): x, y, z, intensity')
        obs.n = int(shape[0])
        semseg = None
        if sem_gt is None:
            if getattr(self.semseg_model, 'accepts_device', False) and not isinstance(rgb, torch.Tensor) \
                    and getattr(rgb, '__dict__', {}).get('_dev') is None:
                return None
            obs.rgb, ishape = put(rgb, np.uint8, torch.uint8, 2)
            # a model that works on the device gets the device image (its class map goes to K1 without visiting the host)
            src = keep[-1] if getattr(self.semseg_model, 'accepts_device', False) else rgb
            semseg = self.semseg_model.pred(src)[0, 0]
            obs.sem, sshape = put(semseg, np.uint8, torch.uint8, 4)
            if len(ishape) != 3 or ishape[2] != 3 or tuple(ishape[:2]) != tuple(sshape):
                raise ValueError(f'image {tuple(ishape)} and class map {tuple(sshape)} do not fit together')
            obs.sem_gt = None
            H, W = int(sshape[0]), int(sshape[1])
        else:
            sg = sem_gt if isinstance(sem_gt, torch.Tensor) else np.asarray(sem_gt)[:, -1]
            obs.sem_gt, gshape = put(sg, np.uint8, torch.uint8, 8)         # trainIds 0..18 and 255
            if tuple(gshape) != (obs.n, ):
                raise ValueError('sem_gt must hold one label per point')
            obs.rgb = obs.sem = None
            H, W = 1, 1
        obs.host_mask = mask
        return obs, semseg, H, W, keep

    def integrate(self, observations: list):
        rgb, pc, sem_gt = observations[0]
        if not self.use_gt_sem:
            sem_gt = None
        T_new_prev = np.asarray(self.pose_provider(pc), dtype=np.float64)
        self.T_prev_origin = np.matmul(self.T_prev_origin, T_new_prev)
        if self._sweep_dev is not None:    # the device ICP uploaded this very sweep: K1 reads that copy
            host, dev = self._sweep_dev
            self._sweep_dev = None
            if host is pc:
                pc = dev
        fast = self._obs_pointers(rgb, pc, sem_gt) if self._fast else None
        general = self._frame_tensors(rgb, pc, sem_gt) if fast is None else None
        if len(self._track) > 0:           # move everything stored so far into the new ego frame (K2, owed to the next reader)
            self.update_sem_pcs(T_new_prev)
        if fast is not None:
            # ONE library call: staging + upload of the host arrays, K1, and the pose bookkeeping of the frame (update_poses,
            # the new pose [0,0,0], the newest path segment, the horizon eviction: sem_pc_accum.py:156-228)
            obs, semseg, H, W, keep = fast
            if self._defer_k1 != getattr(self.store.ctx, 'k1_defer', False):
                self.store.set_defer_k1(self._defer_k1)
            idx, path_length = self.store.append_kitti_obs(obs, self.P_velo_frame, H, W, self.semseg_filters, self.sample_mode,
                                                           self._track, T_new_prev, self.horizon_dist, keep=keep)
            del keep
            if idx is None:                # a numpy pose track: its step stays here
                idx, path_length = self._track.step(T_new_prev, self.horizon_dist)
        else:
            frame, semseg, H, W = general
            self.store.append_kitti([frame], self.P_velo_frame, H, W, self.semseg_filters, sample_mode=self.sample_mode)
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
            if self._sweep_dev is not None:
                host, dev = self._sweep_dev
                self._sweep_dev = None
                if host is pc:
                    pc = dev
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
        if bev_num == 1 and gen_future and self._fast_bev_ok(present_idx):
            return [self._generate_bev_fast(present_idx)]
        pcs, trajs = self._window_inputs(present_idx, gen_future)
        return self._run_bev(pcs, trajs, bev_num)

    def generate_bev_device(self, present_idx: int, out=None):
        """Extension (no reference counterpart): generate_bev(present_idx, 1, gen_future=True)[0] with the 21 planes left in
        HBM -- {'planes_f16': cuda float16 [21,px,px] (`out` if given), 'trajs_present' / '_future' / '_full'} -- for callers
        that gather or post-process on the device (the sharded runner, the benchmark)."""
        if self._fast_bev_ok(present_idx):
            return self._generate_bev_fast(present_idx, out=out, to_host=False)
        pcs, trajs = self._window_inputs(present_idx, True)
        return self.sem_bev_generator.generate(pcs, trajs, device_only=True, out=out)

    def _fast_bev_ok(self, present_idx):
        gen = self.sem_bev_generator
        if not self._fast or type(gen) is not _mods()[4] or gen.do_aug or gen.do_warp or self._store is None \
                or getattr(self._track, '_h', None) is None or os.environ.get('PCA_SYNC_BEV'):
            return False
        if not isinstance(present_idx, (int, np.integer)):
            return False
        n = self._store.n_frames
        split = int(present_idx) if present_idx >= 0 else n + int(present_idx)
        return 0 < split < n and n == len(self._track)

    def _generate_bev_fast(self, present_idx, out=None, to_host=True):
        """One un-augmented sample through ONE library call (pca_kitti_generate_bev): ego polylines, raster with the owed
        re-transforms riding along, and the planes' way to the host.  Same numbers as the general path (_window_inputs ->
        generate -> rasterise -> to_host_async), which stays for everything else (augmentation, warp, bev_num > 1, ...)."""
        C, torch, LazyBev, _PendingCopy, _, hl, _ = _mods()
        st, gen, track = self.store, self.sem_bev_generator, self._track
        ctx = st.ctx
        st.poll_status()
        n = st.n_frames
        present_idx = int(present_idx)
        split = present_idx if present_idx >= 0 else n + present_idx
        lo = max(split - 2, 0)
        near = track.poses_window(lo, split + 1)                # the sample's pose and the two before it
        origin = near[split - lo]
        # heading from the last two PRESENT poses (bev_generator.py:87-93), evaluated with numpy as the general path does
        rot_mat = hl.rotation_matrix_3d(hl.heading_rot_ang(near[:split - lo] - origin))
        px = gen.pixel_size
        prm = gen._raster_params(origin, rot_mat, 0., 0., 1. * gen.view_size, st.intensity_div255)
        max_points = st.bev_workspace(px)
        n_pend, pend_T, pend_ends, write_back = st.bev_pending(0, n)
        if out is None:
            out = torch.empty((21, px, px), dtype=torch.float16, device=st.device)
        else:
            assert out.dtype == torch.float16 and out.is_contiguous() and tuple(out.shape) == (21, px, px)
        host = torch.empty((1, 21, px, px), dtype=torch.float16, pin_memory=True) if to_host else None
        tb = self._ga_traj                                      # landing buffers of the polylines, kept between calls
        if tb is None or tb[0].shape[0] < 2 * n or tb[1].shape[0] < n + 1:
            tb = self._ga_traj = (np.empty((4 * n + 64, 3)), np.zeros(2 * n + 64, dtype=np.int32))
            self._ga_traj_addr = (tb[0].ctypes.data, tb[1].ctypes.data)
        start = tb[1]
        cst = st.c_store()
        # the call's arguments live in a block kept between calls (pca_kitti_generate_bev_v): store, parameter block, owed-chain
        # buffers, scratch and track are written when one of them is replaced, the slots / outputs / view hint per call
        a = self._ga
        if a is None:
            from pca_amd._lib import PcaKittiGenerateBevArgs
            a = self._ga = PcaKittiGenerateBevArgs()
            self._ga_const = None
        pend_T, pend_ends = st._pend_T, st._pend_ends          # (the store's buffers, whether or not anything is owed)
        const = (id(cst), id(st.frame_off), id(prm), id(pend_T), id(pend_ends), id(st._ws), track._h)
        if const != self._ga_const:
            a.store, a.frame_off, a.prm = C.addressof(cst), st.frame_off.data_ptr(), C.addressof(prm)
            a.pending_Ts, a.pending_slot_ends = C.addressof(pend_T), C.addressof(pend_ends)
            a.workspace, a.workspace_bytes = st._ws.data_ptr(), st._ws.numel()
            a.track = getattr(track._h, 'value', track._h)
            self._ga_const = const
            self._ga_keep = (cst, st.frame_off, prm, pend_T, pend_ends, st._ws)      # (what the addresses point into)
        a.slot_begin, a.slot_split, a.slot_end, a.max_points = st.head, st.head + split, st.head + n, max_points
        a.n_pending, a.write_back = n_pend, write_back
        a.planes_f16, a.host_planes = out.data_ptr(), (None if host is None else host.data_ptr())
        a.traj_rows, a.traj_start = self._ga_traj_addr
        a.stream = ctx.stream_int() or None
        a.hint_F = st.view_hint_into(a, 0, n)                  # (the library skips it if this call writes owed transforms back)
        ticket = ctx.lib.pca_kitti_generate_bev_v(ctx.h, C.addressof(a))
        if ticket < 0:
            ctx.check(ticket)
        st.hints_taken += a.hinted
        st.bev_done(write_back)
        rows = tb[0][:a.n_rows].copy()
        empty = np.zeros((0, 3))
        ego_p = rows[:start[split - 1]].copy() if split >= 2 else empty      # edges 0 .. split-2 belong to the present polyline
        ego_f = rows[start[split]:].copy() if n - split >= 2 else empty
        if not to_host:
            return {'planes_f16': out, 'trajs_present': [ego_p], 'trajs_future': [ego_f], 'trajs_full': [rows]}
        return LazyBev(host, 0, _PendingCopy(ctx, ticket, (out, host)), ([ego_p], [ego_f], [rows]))
