"""Semantic point-cloud accumulator base class -- drop-in for the reference's
``sem_pc_accum.SemanticPointCloudAccumulator`` (same constructor, attributes and method names).

State that the reference keeps as Python lists of (M,10) f64 numpy arrays (``self.sem_pcs``) lives in a
device-resident structure-of-arrays store (pca_amd.device_store.DeviceStore); ``sem_pcs`` is still
readable (it materialises host copies on access).  Poses, segment distances, images stay host lists
exactly like the reference.  There is no CPU fallback: constructing an accumulator without a visible
MI355X raises.
"""
import gzip
import os
import pickle

import numpy as np

from bev_generator.bev_generator import DeviceWindow
from bev_generator.rgb_bev import RGBBEVGenerator  # noqa: F401  (import surface of the reference module)
from bev_generator.sem_bev import SemBEVGenerator
from pca_amd import host_logic as hl


class SemanticPointCloudAccumulator:

    def __init__(self, horizon_dist: float, icp_threshold: float, semseg_onnx_path: str, semseg_filters: list,
                 sem_idxs: dict, use_gt_sem: bool, bev_params: dict):
        self.semseg_model = None
        if use_gt_sem is False:
            self.semseg_model = SemSegONNX(semseg_onnx_path)
        self.semseg_filters = semseg_filters
        self.sem_idxs = sem_idxs
        self.use_gt_sem = use_gt_sem
        self.icp_threshold = icp_threshold
        self.icp_trans_init = np.eye(4)
        self.T_prev_origin = np.eye(4)
        self.pcd_prev = None
        self.horizon_dist = horizon_dist

        self._track = hl.PoseTrack()     # poses (N) and seg_dists (N-1)
        self.rgbs = []
        self.semsegs = []
        self._store = None               # created on first use (needs the GPU)
        self._store_args = {}
        # opt-in voxel de-duplication of the accumulation buffer (extension, off by default: the reference only
        # evicts whole frames).  Set the attribute, or PCA_VOXEL_DEDUP=<voxel size in m> [PCA_VOXEL_DEDUP_EVERY=<k>].
        env = os.environ.get('PCA_VOXEL_DEDUP')
        self.voxel_dedup = float(env) if env else None
        self.voxel_dedup_every = int(os.environ.get('PCA_VOXEL_DEDUP_EVERY', '1'))
        self._integrated = 0
        self._aug_worker0 = 0            # number of the first augmentation "worker" of a generate_bev call (see _run_bev)
        # opt-in sample mode of the projection kernels (extension, default = the reference's nearest pixel):
        # 'bilinear' mixes r, g, b of the four neighbours (PCA_SAMPLE_MODE=bilinear), the class stays the nearest pixel's
        self.sample_mode = os.environ.get('PCA_SAMPLE_MODE', 'nearest')

        self.sem_bev_generator = None
        if bev_params['type'] == 'sem':
            self.sem_bev_generator = SemBEVGenerator(
                self.sem_idxs, bev_params['view_size'], bev_params['pixel_size'], bev_params['max_trans_radius'],
                bev_params['zoom_thresh'], bev_params['do_warp'], bev_params['int_scaler'],
                bev_params['int_sep_scaler'], bev_params['int_mid_threshold'], bev_params['height_filter'])
        elif bev_params['type'] == 'rgb':
            raise NotImplementedError('Needs refactoring')

    # ---- state views -------------------------------------------------------------------------
    @property
    def store(self):
        if self._store is None:
            from pca_amd.device_store import DeviceStore
            self._store = DeviceStore(**self._store_args)
        return self._store

    @property
    def poses(self):
        """List of [x,y,z] of the live frames (a copy: mutate through the accumulator, not this list)."""
        return self._track.poses

    @poses.setter
    def poses(self, value):
        self._track.poses = value

    @property
    def seg_dists(self):
        return self._track.seg_dists

    @seg_dists.setter
    def seg_dists(self, value):
        self._track.seg_dists = value

    @property
    def sem_pcs(self):
        """Host copies of the stored frames: list of (M,10) f64 arrays [x,y,z,i,r,g,b,sem,inst,dyn]."""
        if self._store is None:
            return []
        return self._store.frame_rows()

    # ---- integrate (platform specific) ---------------------------------------------------------
    def integrate(self, observations: list):
        raise NotImplementedError()

    def obs2sem_vec_space(self, rgb, pc, sem_gt=None) -> tuple:
        raise NotImplementedError()

    def _after_integrate(self):
        self._integrated += 1
        # errors of kernels that have finished (an overflowing store, a timed-out compaction, ...) surface on the next call
        # at the latest; the read is a few words of mapped host memory, no stream operation
        self.store.poll_status()
        if self.voxel_dedup and self._integrated % max(self.voxel_dedup_every, 1) == 0:
            self.store.voxel_dedup(self.voxel_dedup)

    def update_poses(self, T_new_prev):
        self._track.apply_transform(T_new_prev)

    def update_sem_pcs(self, T_new_prev):
        """Every stored point p <- T_new_prev p, in place on the device (K2; deferred so that a BEV that
        follows immediately applies it in its first pass -- same roundings, one read of the store less)."""
        self.store.retransform(np.asarray(T_new_prev, dtype=np.float64), defer=True)

    def remove_observations(self):
        """Appends the newest path segment and evicts frames beyond the memory horizon."""
        path_length = self._track.push_segment()
        idx = self._track.evict_beyond(self.horizon_dist, path_length)
        if idx:
            self.store.evict(idx)
            self.rgbs = self.rgbs[idx:]
            self.semsegs = self.semsegs[idx:]
        return idx, path_length

    # ---- small host helpers --------------------------------------------------------------------
    @staticmethod
    def comp_incr_path_dist(seg_dists: list):
        return hl.incremental_path_dists(seg_dists)

    def get_segment_dists(self) -> list:
        return self._track.seg_array().tolist()

    def get_incremental_path_dists(self) -> np.array:
        return self._track.incr()

    def get_pose(self, idx: int = None) -> np.array:
        return self._track.as_array() if idx is None else self._track.pose(idx)

    def get_rgb(self, idx: int = None) -> list:
        return self.rgbs if idx is None else [self.rgbs[idx]]

    def get_semseg(self, idx: int = None) -> list:
        return self.semsegs if idx is None else [self.semsegs[idx]]

    @staticmethod
    def dist(pose_0: np.array, pose_1: np.array):
        return hl.pose_dist(pose_0, pose_1)

    @staticmethod
    def write_compressed_pickle(obj, filename, write_dir):
        """gzip(pickle(obj)) -> write_dir/filename.gz, the reference's container (sem_pc_accum.py:280-294).  A BEV sample
        whose planes are still in flight (LazyBev) goes to the process-wide background writer (pca_amd.writer: device wait,
        pickling and compression off the driver's thread; flushed at exit); PCA_ASYNC_WRITE=0 keeps it synchronous."""
        from bev_generator.sem_bev import LazyBev
        if isinstance(obj, LazyBev) and os.environ.get('PCA_ASYNC_WRITE', '1') != '0':
            from pca_amd.writer import shared_writer
            shared_writer().submit(obj, filename, write_dir)
            return
        path = os.path.join(write_dir, f"{filename}.gz")
        blob = pickle.dumps(obj)
        try:
            with gzip.open(path, "wb") as f:
                f.write(blob)
        except IOError as error:
            print(error)

    @staticmethod
    def read_compressed_pickle(path):
        from pca_amd import writer
        writer.flush_shared()                 # samples handed to the background writer are on disk before anything is read
        try:
            with gzip.open(path, "rb") as f:
                return pickle.loads(f.read())
        except IOError as error:
            print(error)

    @staticmethod
    def pc2pcd(pc):
        import open3d as o3d
        pcd = o3d.geometry.PointCloud()
        pcd.points = o3d.utility.Vector3dVector(pc[:, :3])
        pcd.estimate_normals()
        return pcd

    # ---- projection helpers of the reference's public surface (device-backed) ---------------------
    def _k1_rows(self, pc_velo, rgb, sem, P_velo_frame, filters):
        import torch
        from pca_amd.device_store import DeviceStore
        H, W = sem.shape[:2]
        tmp = DeviceStore(capacity=max(pc_velo.shape[0], 1), max_frames=2)
        dev = tmp.device
        frame = dict(pts=torch.from_numpy(np.ascontiguousarray(pc_velo, dtype=np.float32)).to(dev),
                     rgb=torch.from_numpy(np.ascontiguousarray(rgb, dtype=np.uint8)).to(dev),
                     sem=torch.from_numpy(np.ascontiguousarray(sem, dtype=np.uint8)).to(dev))
        tmp.append_kitti([frame], P_velo_frame, H, W, filters)
        return tmp.rows(0)

    def filter_semseg_pc(self, pc):
        keep = ~np.isin(pc[:, -1], list(self.semseg_filters))
        return pc[keep]

    def gen_semantic_pc(self, pc_velo, semantic_map, P_velo_frame):
        """(M, 4+K) rows [x,y,z,i, map channels] of the points that project inside the map."""
        semantic_map = np.asarray(semantic_map)
        K = semantic_map.shape[2]
        if K == 3:
            rows = self._k1_rows(pc_velo, semantic_map, np.zeros(semantic_map.shape[:2], np.uint8), P_velo_frame, [])
            return rows[:, :7]
        if K == 1:
            rgb0 = np.zeros(semantic_map.shape[:2] + (3, ), np.uint8)
            rows = self._k1_rows(pc_velo, rgb0, semantic_map[..., 0], P_velo_frame, [])
            return np.concatenate([rows[:, :4], rows[:, 7:8]], axis=1)
        raise NotImplementedError('semantic_map must have 1 or 3 layers')

    @staticmethod
    def velo2frame(pc_velo, P_velo_frame):
        """(N,3) velodyne -> (N,3) homogeneous image-frame coordinates (host helper, numpy)."""
        homo = np.concatenate((pc_velo, np.ones((pc_velo.shape[0], 1))), axis=1)
        return np.matmul(P_velo_frame, homo.T).T

    def velo2img(self, pc_velo, P_velo_frame, img_h, img_w, max_depth=np.inf):
        """(N,4) velodyne rows -> (M,6) rows [x, y, z, i, u, v] of the points that project inside the image (host
        helper for direct callers, the reference's expressions: sem_pc_accum.py:367-402; integrate() runs K1)."""
        frame = self.velo2frame(pc_velo[:, :3], P_velo_frame)
        depth = frame[:, 2]
        depth[depth == 0] = -1e-6
        u = np.round(frame[:, 0] / np.abs(depth)).astype(int)
        v = np.round(frame[:, 1] / np.abs(depth)).astype(int)
        inside = (u >= 0) & (u < img_w) & (v >= 0) & (v < img_h) & (depth > 0) & (depth < max_depth)
        return np.concatenate([pc_velo, u[:, None], v[:, None]], axis=1)[inside]

    def viz_sem_vec_space(self):
        """Open3D window with every stored point (host copy of the device store) and the ego path."""
        self.viz_sem_pc(np.concatenate(self.sem_pcs, axis=0), self.poses)

    @staticmethod
    def viz_sem_pc(sem_pc: np.array, poses: list = []):
        """sem_pc: (N,>=7) rows [x, y, z, intensity, r, g, b, ...]; poses: list of [x, y, z].  Needs open3d."""
        import open3d as o3d
        cloud = o3d.geometry.PointCloud()
        cloud.points = o3d.utility.Vector3dVector(sem_pc[:, :3])
        cloud.colors = o3d.utility.Vector3dVector(np.asarray(sem_pc[:, 4:7], dtype=np.float64) / 255.)
        frame = o3d.geometry.TriangleMesh.create_coordinate_frame(size=1, origin=poses[0] if len(poses) else [0, 0, 0])
        path = o3d.geometry.LineSet(points=o3d.utility.Vector3dVector(poses),
                                    lines=o3d.utility.Vector2iVector([[k, k + 1] for k in range(len(poses) - 1)]))
        path.colors = o3d.utility.Vector3dVector([[1, 0, 0]] * max(len(poses) - 1, 0))
        o3d.visualization.draw_geometries([frame, path, cloud])

    # ---- BEV -------------------------------------------------------------------------------------
    def viz_bev(self, bev, file_path, rgbs: list = [], semsegs: list = []):
        self.sem_bev_generator.viz_bev(bev, file_path, rgbs, semsegs)

    def generate_bev(self, present_idx: int = None, bev_num: int = 1, gen_future: bool = False):
        raise NotImplementedError()

    def _window_inputs(self, present_idx, gen_future, other_trajs=None, gt_lanes=None):
        """The reference's (pcs, trajs) dicts for one BEV sample, with device window handles in place of
        the concatenated host arrays (kitti360_sem_pc_accum.py:180-228 / nuscenes_oracle_...py:521-595)."""
        poses = self._track.as_array()
        origin = poses[-1].copy() if present_idx is None else poses[present_idx].copy()
        split = self.store.n_frames if present_idx is None else \
            (present_idx if present_idx >= 0 else self.store.n_frames + present_idx)
        if split <= 0:
            raise ValueError('need at least one array to concatenate')     # np.concatenate([]) in the reference
        all_sets = present_idx is None and gen_future        # the reference slices [:None] and [None:]: everything
        win = DeviceWindow(self.store, split, origin, future_is_present=all_sets)
        pcs = {'pc_present': win.part('present')}
        rel = poses - origin
        trajs = {'ego_traj_present': rel[:present_idx]}
        others = other_trajs if other_trajs is not None else ([], [], [])
        trajs['other_trajs_present'] = [np.concatenate([t]) - origin for t in others[0]]
        if gt_lanes is not None:
            trajs['gt_lanes'] = [lane - origin for lane in gt_lanes]
        if gen_future:
            if split >= self.store.n_frames and not all_sets:
                raise ValueError('need at least one array to concatenate')
            pcs['pc_future'] = win.part('future')
            pcs['pc_full'] = win.part('full')
            trajs['ego_traj_future'] = rel[present_idx:]
            trajs['ego_traj_full'] = rel
            if not all_sets:
                trajs['_ego_split'] = split
            trajs['other_trajs_future'] = [np.concatenate([t]) - origin for t in others[1]]
            trajs['other_trajs_full'] = [np.concatenate([t]) - origin for t in others[2]]
        else:
            for k in ('pc_future', 'pc_full'):
                pcs[k] = None
            for k in ('ego_traj_future', 'other_trajs_future', 'ego_traj_full', 'other_trajs_full'):
                trajs[k] = None
        return pcs, trajs

    def _run_bev(self, pcs, trajs, bev_num):
        """bev_num samples of one window.  The reference forks a multiprocessing.Pool for bev_num > 1 (pickling the window
        per worker, kitti360_sem_pc_accum.py:236-241); here the bev_num rasters are enqueued back to back on the device
        into one [bev_num,21,px,px] tensor and leave it in ONE asynchronous copy: the returned dicts (LazyBev) fill in on
        first access.  Each raster has its own random augmentation, drawn as generate_rand_aug draws it: sample k reseeds
        as the reference's k-th Pool worker would (pid + k in place of the worker's own pid), so the samples of one
        window differ although they are drawn in one process within the same second."""
        import torch
        from bev_generator.bev_generator import WindowPart
        gen = self.sem_bev_generator
        if self._store is not None:
            self._store.poll_status()
        if not isinstance(pcs['pc_present'], WindowPart) or os.environ.get('PCA_SYNC_BEV'):
            return [gen.generate_multiproc((pcs, self._copy_trajs(trajs)), worker=self._aug_worker0 + k)
                    for k in range(bev_num)]
        px = gen.pixel_size
        planes = torch.empty((bev_num, 21, px, px), dtype=torch.float16, device=self.store.device)
        with gen.raster_batch():                  # bev_num > 1: ONE launch of each kernel for all the samples
            results = [gen.generate_multiproc((pcs, self._copy_trajs(trajs)), device_only=True, out=planes[k],
                                              worker=self._aug_worker0 + k) for k in range(bev_num)]
        return gen.to_host_async(planes, results)

    def _window_inputs_for(self, present_idx, gen_future):
        """(pcs, trajs) of one sample as generate_bev builds them (overridden where other agents' trajectories exist)."""
        return self._window_inputs(present_idx, gen_future)

    def generate_bev_many(self, present_idxs, gen_future: bool = True):
        """Extension (no reference counterpart): the samples generate_bev(idx, 1, gen_future)[0] would return for every idx of
        `present_idxs`, rasterised in ONE launch of each kernel (the store does not change between them: the driver's sweep
        over present_idx of a finished scene, run_nuscenes_bev_gen.py:245-271) and copied to the host in one asynchronous
        transfer.  Returns the list of dicts (LazyBev)."""
        import torch
        gen = self.sem_bev_generator
        present_idxs = list(present_idxs)
        if not present_idxs:
            return []
        self.store.poll_status()
        px = gen.pixel_size
        planes = torch.empty((len(present_idxs), 21, px, px), dtype=torch.float16, device=self.store.device)
        with gen.raster_batch():
            results = []
            for k, idx in enumerate(present_idxs):
                pcs, trajs = self._window_inputs_for(idx, gen_future)
                results.append(gen.generate_multiproc((pcs, self._copy_trajs(trajs)), device_only=True, out=planes[k]))
        return gen.to_host_async(planes, results)

    @staticmethod
    def _copy_trajs(trajs):
        out = {}
        for k, v in trajs.items():
            if isinstance(v, list):
                out[k] = [np.array(t) for t in v]
            elif v is None or isinstance(v, int):
                out[k] = v
            else:
                out[k] = np.array(v)
        return out


def SemSegONNX(path):
    """Late import so that GT-semantics runs never need onnxruntime."""
    from utils.onnx_utils import SemSegONNX as _S
    return _S(path)
