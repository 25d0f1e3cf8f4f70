"""BEV generator base class -- drop-in for the reference's ``bev_generator.bev_generator.BEVGenerator``
(same constructor, attributes, method names and argument meaning), with the per-point work
(rotate / translate / crop / height filter / floor-to-grid / per-cell reductions) executed by the gfx950
rasteriser (pca_bev_generate, csrc/pca_bev.hip) instead of numpy.

Inputs of ``generate``: the reference's ``pcs`` / ``trajs`` dicts.  ``pcs`` values are either host
(N,10) f64 arrays (uploaded, slow path) or ``WindowPart`` handles onto a device-resident window handed
out by the accumulators (fast path, no copies).  Unlike the reference, host input arrays are NOT
mutated (bev_generator.py:226-231 of the reference rotates its inputs in place).
"""
import os
import time
from abc import ABC, abstractmethod

import ctypes

import numpy as np

from pca_amd import host_logic as hl

SETS = ('present', 'future', 'full')
PLANES = ('road', 'intensity', 'r', 'g', 'b', 'dynamic', 'elevation')


class DeviceWindow:
    """Live frames [first, last) of a DeviceStore; frames before `split` form the 'present' set.
    `origin` is the BEV frame origin (the reference subtracts it on the host while concatenating)."""

    def __init__(self, store, split, origin, first=0, last=None, future_is_present=False):
        self.store = store
        self.split = int(split)
        self.origin = np.asarray(origin, dtype=np.float64)
        self.first = int(first)
        self.last = store.n_frames if last is None else int(last)
        # generate_bev(present_idx=None, gen_future=True) of the reference: sem_pcs[:None] and sem_pcs[None:] are both
        # the whole window, so 'present', 'future' and 'full' are the same point set
        self.future_is_present = bool(future_is_present)

    def part(self, name):
        return WindowPart(self, name)


class WindowPart:
    """What an accumulator puts under pcs['pc_present' | 'pc_future' | 'pc_full']."""

    def __init__(self, window, name):
        self.window = window
        self.name = name

    def to_numpy(self):
        """Materialises the reference's host array (concatenated frames minus the origin)."""
        w = self.window
        rows = w.store.frame_rows()
        lo, hi = {'present': (w.first, w.split), 'future': (w.split, w.last), 'full': (w.first, w.last)}[self.name]
        if w.future_is_present:
            lo, hi = w.first, w.last
        out = np.concatenate(rows[lo:hi]) if hi > lo else np.zeros((0, 10))
        out[:, :3] = out[:, :3] - w.origin
        return out


class BEVGenerator(ABC):

    def __init__(self,
                 view_size: int,
                 pixel_size: int,
                 max_trans_radius: float = 0.,
                 zoom_thresh: float = 0.,
                 do_warp: bool = False,
                 int_scaler: float = 1.,
                 int_sep_scaler: float = 1.,
                 int_mid_threshold: float = 0.5,
                 height_filter=None):
        self.view_size = view_size          # [m]
        self.pixel_size = pixel_size        # [px]
        self.max_trans_radius = max_trans_radius
        self.zoom_thresh = zoom_thresh
        self.do_warp = do_warp
        self.do_aug = bool(self.max_trans_radius > 0. or self.zoom_thresh > 0.)
        self.int_scaler = int_scaler
        self.int_sep_scaler = int_sep_scaler
        self.int_mid_threshold = int_mid_threshold
        self.sem_idx = 7                    # column of the semantic class in a (N,10) row
        self.height_filter = height_filter
        if self.height_filter is not None:
            print("NOTE: Removes points above ego-vehicle height!")
        self._tmp = {}                      # scratch device stores of the host-array path
        self._frame = None
        self._device_only = False
        self._out16 = None
        self._defer = None                  # a list while rasters are being collected for ONE launch (raster_batch)

    def __getstate__(self):                 # device handles never travel through pickle
        d = dict(self.__dict__)
        d['_tmp'] = {}
        d['_defer'] = None
        return d

    class _RasterBatch:
        def __init__(self, gen):
            self.gen = gen

        def __enter__(self):
            self.gen._defer = []
            return self

        def __exit__(self, exc_type, exc, tb):
            jobs, self.gen._defer = self.gen._defer, None
            if exc_type is None and jobs:
                store = jobs[0][0]
                assert all(j[0] is store for j in jobs)
                outs = [j[2] for j in jobs]
                base = outs[0]._base if outs[0]._base is not None else outs[0]
                px = int(jobs[0][1][1].px)
                contiguous = all(o.data_ptr() == base.data_ptr() + k * 21 * px * px * 2 for k, o in enumerate(outs))
                if len(jobs) > 1 and contiguous and base.numel() >= len(jobs) * 21 * px * px:
                    store.bev_many([j[1] for j in jobs], base.view(-1)[:len(jobs) * 21 * px * px].view(len(jobs), 21, px, px))
                else:                           # a single raster (owed transforms ride along), or outputs scattered in memory
                    for _, (split, prm, first, last), out in jobs:
                        store.bev(split, prm, first_frame=first, last_frame=last, out16=out)
            return False

    def raster_batch(self):
        """Context manager: device_only generate() calls made inside it (with `out=` slices of one [k,21,px,px] tensor, no
        warp) only RECORD their raster; leaving the block launches all of them at once (DeviceStore.bev_many)."""
        return BEVGenerator._RasterBatch(self)

    # ------------------------------------------------------------------ raster (device) ----
    def _raster_params(self, origin, rot_mat, dx, dy, aug_view_size, intensity_div255):
        """pca_bev_params of one sample.  The constant part (grid, intensity transform, class sets) is built once per
        generator; origin / rotation / shift / view -- the first 15 doubles of the struct -- are copied in per call."""
        key = ('prm', bool(intensity_div255), self.pixel_size, self.height_filter, self.int_scaler,
               self.int_sep_scaler, self.int_mid_threshold, getattr(self, 'rgb_fill', 0))
        prm = self._tmp.get(key)
        if prm is None:
            from pca_amd.device_store import make_bev_params
            sem_idxs = getattr(self, 'sem_idxs', None) or {}
            dyn_cls = [sem_idxs[k] for k in ('car', 'truck', 'bus', 'motorcycle') if k in sem_idxs]
            prm = self._tmp[key] = make_bev_params(np.zeros(3), np.eye(3), 0., 0., 1., self.pixel_size,
                                                   self.height_filter, self.int_scaler, self.int_sep_scaler,
                                                   self.int_mid_threshold, sem_idxs.get('road', -1), dyn_cls,
                                                   intensity_div255, getattr(self, 'rgb_fill', 0))
        head = self._tmp.get(id(prm))                   # the struct's first 15 doubles as a numpy array over the same memory
        if head is None:
            head = self._tmp[id(prm)] = np.frombuffer((ctypes.c_double * 15).from_address(ctypes.addressof(prm)), dtype=np.float64)
        head[0:3] = origin
        head[3:12] = np.asarray(rot_mat, dtype=np.float64).reshape(9)
        head[12], head[13], head[14] = dx, dy, aug_view_size
        return prm

    def _tmp_store(self, key):
        from pca_amd.device_store import DeviceStore
        if key not in self._tmp:
            self._tmp[key] = DeviceStore(capacity=1 << 16, max_frames=4)
        return self._tmp[key]

    def rasterise(self, pc_present, pc_future, pc_full, rot_mat, dx, dy, aug_view_size, want_f64=False, out16=None):
        """Returns (planes_f16, planes_f64|None) as cuda tensors [21,px,px], set-major
        {present,future,full} x {road,intensity,r,g,b,dynamic,elevation}."""
        if isinstance(pc_present, WindowPart):
            w = pc_present.window
            prm = self._raster_params(w.origin, rot_mat, dx, dy, aug_view_size, w.store.intensity_div255)
            if self._defer is not None and out16 is not None and not want_f64 and not w.future_is_present:
                from pca_amd._lib import PcaBevParams
                self._defer.append((w.store, (w.split, PcaBevParams.from_buffer_copy(prm), w.first, w.last), out16))
                return out16, None
            return w.store.bev(w.split, prm, want_f64=want_f64, first_frame=w.first, last_frame=w.last, out16=out16)
        # host arrays: 'present' and 'future' share one launch; 'full' is an independent input in the
        # reference's interface, so it gets its own launch (as the present set of a second window)
        zero = np.zeros(3)
        prm = self._raster_params(zero, rot_mat, dx, dy, aug_view_size, False)
        st = self._tmp_store('pf')
        i64 = st.load_rows([pc_present, pc_future])
        a16, a64 = st.bev(1, prm, want_f64=want_f64, intensity64=i64)
        st2 = self._tmp_store('full')
        i64 = st2.load_rows([pc_full])
        b16, b64 = st2.bev(1, prm, want_f64=want_f64, intensity64=i64)
        a16[14:21] = b16[0:7]
        if want_f64:
            a64[14:21] = b64[0:7]
        return a16, a64

    # ------------------------------------------------------------------ generate ----------
    @abstractmethod
    def generate_bev(self, pc_present, pc_future, pc_full, trajs_present, trajs_future, trajs_full,
                     gt_lane_trajs=None):
        pass

    def generate(self,
                 pcs: dict,
                 trajs: dict,
                 rot_ang: float = 0.,
                 trans_dx: float = 0.,
                 trans_dy: float = 0.,
                 zoom_scalar: float = 1.,
                 do_warping: bool = False,
                 device_only: bool = False,
                 out=None):
        """device_only=True (extension): the 21 planes stay in HBM -- the returned dict holds one cuda
        float16 tensor 'planes_f16' [21,px,px] instead of 15 host arrays (used by the sharded runner and
        the benchmark, where BEVs are gathered with RCCL); `out` may name the cuda tensor to write into."""
        pc_present, pc_future, pc_full = self.extract_pc_dict(pcs)
        ego_present, ego_future, ego_full = self.extract_ego_traj_dict(trajs)
        oth_present, oth_future, oth_full = self.extract_other_traj_dicts(trajs)
        lanes = self.extract_gt_lane_dicts(trajs) if 'gt_lanes' in trajs.keys() else None

        aug_view_size = zoom_scalar * self.view_size
        if do_warping is False:
            rot_ang = hl.heading_rot_ang(ego_present)
        rot_mat = hl.rotation_matrix_3d(rot_ang)

        def to_grid(traj_list):
            # (the lists are this call's own copies: nobody reads the in-place rotation the reference leaves behind)
            return [hl.transform_traj(t, rot_mat, trans_dx, trans_dy, aug_view_size, self.pixel_size, mutate=False)
                    for t in traj_list]

        split = trajs.get('_ego_split')          # set by the accumulators: the ego polylines are slices of one array
        if split is not None and pc_future is not None:
            ego_p, ego_f, ego_a = hl.transform_ego_split(ego_full, split, rot_mat, trans_dx, trans_dy, aug_view_size,
                                                        self.pixel_size)
            trajs_present = [ego_p] + to_grid(oth_present)
            trajs_future = [ego_f] + to_grid(oth_future)
            trajs_full = [ego_a] + to_grid(oth_full)
        else:
            trajs_present = to_grid([ego_present] + oth_present)
        if lanes is not None:
            lanes = [lane for lane in to_grid(lanes) if lane.shape[0] > 0]
        if pc_future is None:
            # the reference only defines the future/full trajectories inside `if pc_future is not None`
            raise UnboundLocalError("local variable 'trajs_future' referenced before assignment")
        if split is None:
            trajs_future = to_grid([ego_future] + oth_future)
            trajs_full = to_grid([ego_full] + oth_full)

        self._frame = (rot_mat, trans_dx, trans_dy, aug_view_size)
        self._device_only = device_only
        self._out16 = out if device_only else None
        return self.generate_bev(pc_present, pc_future, pc_full, trajs_present, trajs_future, trajs_full, lanes)

    def preprocess_pc_and_trajs(self, pc, trajs, rot_ang, trans_dx, trans_dy, aug_view_size):
        """Host-array convenience with the reference's return values (grid coordinates in columns 0,1).
        Row selection and cell indices come from the device binning kernel's definition; this helper
        exists for API completeness and evaluates the same expressions with numpy."""
        rot_mat = hl.rotation_matrix_3d(rot_ang)
        pc = self.geometric_transform(pc, rot_ang, trans_dx, trans_dy, aug_view_size)
        out_trajs = [hl.transform_traj(t, rot_mat, trans_dx, trans_dy, aug_view_size, self.pixel_size)
                     for t in trajs]
        if self.height_filter is not None:
            pc = pc[pc[:, 2] < self.height_filter]
        return self.pos2grid(pc, aug_view_size), out_trajs

    def generate_rand_aug(self, pcs: dict, trajs: dict, do_warping: bool = True, device_only: bool = False, out=None,
                          worker: int = 0):
        # The reference seeds from pid * time (bev_generator.py:168) and draws each of the bev_num samples in its own
        # Pool worker, i.e. under its own pid.  Here the samples of one window are drawn in ONE process: `worker` (the
        # sample's number in the batch) stands in for the worker's pid offset, so that the samples differ.
        np.random.seed(((os.getpid() + int(worker)) * int(time.time())) % 123456789)
        rot_ang = 2 * np.pi * np.random.random()
        trans_r = self.max_trans_radius * np.random.random()
        trans_ang = 2 * np.pi * np.random.random()
        trans_dx = trans_r * np.cos(trans_ang)
        trans_dy = trans_r * np.sin(trans_ang)
        zoom_scalar = np.random.normal(0, 0.1)
        zoom_scalar = 1 + min(max(zoom_scalar, -self.zoom_thresh), self.zoom_thresh)
        return self.generate(pcs, trajs, rot_ang, trans_dx, trans_dy, zoom_scalar, do_warping, device_only, out)

    def generate_multiproc(self, bev_gen_inputs, device_only: bool = False, out=None, worker: int = 0):
        pcs, trajs = bev_gen_inputs
        if self.do_aug:
            return self.generate_rand_aug(pcs, trajs, device_only=device_only, out=out, worker=worker)
        return self.generate(pcs, trajs, device_only=device_only, out=out)

    def generate_rand_aug_multiproc(self, bev_gen_inputs):
        pcs, trajs = bev_gen_inputs
        return self.generate_rand_aug(pcs, trajs, do_warping=True)

    # ------------------------------------------------------------------ host helpers ------
    def geometric_transform(self, pc_mat, rot_ang, trans_dx, trans_dy, aug_view_size, is_traj=False):
        rot_mat = self.rotation_matrix_3d(rot_ang)
        pc_mat[:, :3] = np.matmul(rot_mat, pc_mat[:, :3].T).T
        pc_mat[:, 0] += trans_dx
        pc_mat[:, 1] += trans_dy
        if is_traj:
            return self.crop_trajectory(pc_mat, aug_view_size)
        return self.crop_view(pc_mat, aug_view_size)

    @staticmethod
    def crop_view(pc_mat, aug_view_size):
        h = 0.5 * aug_view_size
        pc_mat = pc_mat[np.logical_and(pc_mat[:, 0] > -h, pc_mat[:, 0] < h)]
        return pc_mat[np.logical_and(pc_mat[:, 1] > -h, pc_mat[:, 1] < h)]

    def crop_trajectory(self, traj, aug_view_size, thresh=1e-4):
        return hl.crop_trajectory(traj, aug_view_size, thresh)

    @staticmethod
    def point_in_box(pnt_x, pnt_y, box_x0, box_y0, box_x1, box_y1):
        return (box_x0 < pnt_x and pnt_x < box_x1) and (box_y0 < pnt_y and pnt_y < box_y1)

    def cal_intersec_pnt(self, x0, y0, x1, y1, bbox, thresh=1e-4):
        return hl.bisect_box_crossing(x0, y0, x1, y1, bbox, thresh)

    @staticmethod
    def partition_semantic_pc(pc_mat, sems, sem_idx):
        mask = np.isin(pc_mat[:, sem_idx], list(sems))
        return pc_mat[mask], pc_mat[~mask]

    @staticmethod
    def dirichlet_dist_expectation(gridmaps, obs_weight=1):
        g = np.stack(gridmaps) * obs_weight + 1.
        g /= np.sum(g, axis=0)
        return [g[k] for k in range(g.shape[0])]

    @staticmethod
    def warp_dense_probmaps(probmaps, a_1, a_2, b_1, b_2):
        return hl.warp_dense_probmaps(probmaps, a_1, a_2, b_1, b_2)

    def warp_sparse_points(self, pnts, a_1, a_2, b_1, b_2, i_mid, j_mid, i_warp, j_warp):
        return hl.warp_sparse_points(pnts, a_1, a_2, j_warp, j_mid, self.pixel_size)

    @staticmethod
    def warp_point(x, y, a_1, a_2, b_1, b_2, I, J):
        return hl.warp_point(x, y, a_1, a_2, b_1, b_2, I, J)

    def warp_points(self, pnt_list, a_1, a_2, b_1, b_2, I, J):
        return [hl.warp_point(p[0], p[1], a_1, a_2, b_1, b_2, I, J) for p in pnt_list]

    @staticmethod
    def get_random_warp_params(mean_ratio, max_ratio, I, J):
        return hl.get_random_warp_params(mean_ratio, max_ratio, I, J)

    @staticmethod
    def cal_warp_params(idx_0, idx_1, idx_max):
        return hl.cal_warp_params(idx_0, idx_1, idx_max)

    def warp_trajs(self, trajs, a_1, a_2, b_1, b_2, i_mid, j_mid, i_warp, j_warp):
        return [self.warp_sparse_points(t, a_1, a_2, b_1, b_2, i_mid, j_mid, i_warp, j_warp) for t in trajs]

    @staticmethod
    def extract_pc_dict(pcs: dict):
        return pcs['pc_present'], pcs['pc_future'], pcs['pc_full']

    @staticmethod
    def extract_ego_traj_dict(trajs: dict) -> tuple:
        return trajs['ego_traj_present'], trajs['ego_traj_future'], trajs['ego_traj_full']

    @staticmethod
    def extract_other_traj_dicts(trajs: dict) -> tuple:
        return trajs['other_trajs_present'], trajs['other_trajs_future'], trajs['other_trajs_full']

    @staticmethod
    def extract_gt_lane_dicts(trajs: dict) -> tuple:
        return trajs['gt_lanes']

    @staticmethod
    def extract_aug_dict(augs: dict):
        return augs['max_translation_radius'], augs['zoom_threshold']

    @staticmethod
    def rotation_matrix_3d(ang):
        return hl.rotation_matrix_3d(ang)

    def pos2grid(self, pc_mat, view_size):
        return hl.pos2grid_inplace(pc_mat, view_size, self.pixel_size)

    @abstractmethod
    def viz_bev(self):
        pass
