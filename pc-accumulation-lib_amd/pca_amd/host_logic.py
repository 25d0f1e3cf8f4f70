"""Host-side (CPU, tiny) parts of the hot path: the ego-pose track, path distances, horizon eviction,
BEV heading, trajectory clipping and the polynomial warp.

These are O(#frames) or O(#trajectory vertices) per call and stay on the host exactly as in the
reference; the per-point work is in the HIP library.  Every numpy expression that feeds a value the
device consumes (pose origin, rotation matrix) is evaluated with the same numpy calls as the
reference so that the f64 bit patterns agree.

Reference behaviour restated (file:line relative to the reference root):
  sem_pc_accum.py:156-165   update_poses           -> PoseTrack.apply_transform
  sem_pc_accum.py:185-209   remove_observations    -> PoseTrack.push_segment_and_evict
  sem_pc_accum.py:211-228   comp_incr_path_dist    -> incremental_path_dists
  sem_pc_accum.py:404-415   dist                   -> pose_dist
  bev_generator/bev_generator.py:87-93    heading  -> heading_rot_ang
  bev_generator/bev_generator.py:207-237  geometric_transform(is_traj=True) -> transform_traj
  bev_generator/bev_generator.py:257-371  crop_trajectory / cal_intersec_pnt
  bev_generator/bev_generator.py:482-698  warp family
"""
import math
import random
from collections.abc import Sequence

import numpy as np


# ----------------------------------------------------------------------------------------------
#  poses / path distances
# ----------------------------------------------------------------------------------------------
def pose_dist(p0, p1):
    """Euclidean distance over all stored components (xyz, despite the reference docstring)."""
    return np.sqrt(np.sum((p1 - p0)**2))


_TRI = {}


def incremental_path_dists(seg_dists):
    """Cumulative path length as lower-triangular-ones @ d (NOT np.cumsum: last bits differ)."""
    d = np.array(seg_dists)
    n = len(d)
    tri = _TRI.get(n)
    if tri is None:
        if len(_TRI) > 64:
            _TRI.clear()
        tri = _TRI[n] = np.tri(n)
    return np.matmul(tri, d)


class ArrayList(Sequence):
    """Read-only list view of an array snapshot: behaves like the reference's list (of lists) for len / index /
    iteration / comparison / np.array(), but costs nothing until an element is actually looked at (the drivers ask
    for len(acc.poses) once per frame)."""

    def __init__(self, arr):
        self._a = arr

    def __len__(self):
        return self._a.shape[0]

    def __getitem__(self, i):
        return self._a[i].tolist()

    def __iter__(self):
        return iter(self._a.tolist())

    def __array__(self, dtype=None, copy=None):
        return np.array(self._a, dtype=dtype)

    def tolist(self):
        return self._a.tolist()

    def __eq__(self, other):
        return self._a.tolist() == (other.tolist() if isinstance(other, (ArrayList, np.ndarray)) else other)

    def __repr__(self):
        return repr(self._a.tolist())


class NumpyPoseTrack:
    """Ego poses of the live frames and the segment distances between them -- the numpy form, expression by expression
    what the reference evaluates (the C form below is checked against it when the library is bound).

    Stored as one (F,3) f64 array; ``poses`` hands out the reference's list-of-lists form.  The pose
    update keeps the reference's arithmetic -- one (4,4)@(4,1) product PER POSE (numpy routes those to
    gemv, whose rounding differs from a (4,4)@(4,F) gemm) -- but issues them as one stacked matmul."""

    def __init__(self):
        self._H = np.zeros((0, 4, 1))    # homogeneous column vectors [x, y, z, 1]: the matmul operand, kept as is
        self._D = np.zeros(0)            # segment distances between consecutive poses

    @property
    def _P(self):
        return self._H[:, :3, 0]

    @property
    def seg_dists(self):
        return ArrayList(self._D.copy())

    @seg_dists.setter
    def seg_dists(self, value):
        self._D = np.array(value, dtype=np.float64).reshape(-1)

    def seg_array(self):
        return self._D

    @property
    def poses(self):
        return ArrayList(self._P.copy())

    @poses.setter
    def poses(self, value):
        P = np.array(value, dtype=np.float64).reshape(-1, 3)
        self._H = np.ones((P.shape[0], 4, 1))
        self._H[:, :3, 0] = P

    def __len__(self):
        return self._H.shape[0]

    def pose(self, idx):
        return self._P[idx].copy()

    def as_array(self):
        return self._P.copy()

    def apply_transform(self, T):
        """Every stored pose p <- (T @ [p,1])[:3]."""
        if self._H.shape[0] == 0:
            return
        H = np.matmul(T, self._H)        # (F,4,1): one gemv per pose, as the reference issues them
        H[:, 3, 0] = 1.0
        self._H = H

    def append(self, pose):
        h = np.ones((1, 4, 1))
        h[0, :3, 0] = np.asarray(pose, dtype=np.float64).reshape(3)
        self._H = np.concatenate([self._H, h])

    def push_segment(self):
        """Appends the distance between the two newest poses; returns the total path length."""
        self._D = np.append(self._D, pose_dist(self._H[-1, :3, 0], self._H[-2, :3, 0]))
        return np.sum(self._D)

    def evict_beyond(self, horizon_dist, path_length):
        """Number of oldest frames to drop so that the remaining path fits the horizon (0 if it fits)."""
        if not path_length > horizon_dist:
            return 0
        incr = incremental_path_dists(self._D)
        incr -= path_length - horizon_dist
        k = int((incr > 0.).argmax())
        self._H = self._H[k:]
        self._D = self._D[k:]
        return k

    def incr(self):
        return incremental_path_dists(self._D)

    def step(self, T_new_prev, horizon_dist):
        """The pose bookkeeping of one integrate(): returns (frames evicted, path length or None)."""
        if len(self) > 0:
            self.apply_transform(T_new_prev)
        self.append([0., 0., 0.])
        if len(self) < 2:
            return 0, None
        path_length = self.push_segment()
        return self.evict_beyond(horizon_dist, path_length), path_length

    def trigger(self, bev_horizon, previous_idx, min_step):
        """Sample trigger of run_kitti360_bev_gen.py:218-240 on the track as it stands: present index or None."""
        if len(self) < 2:
            return None
        d = self.incr()
        if d[-1] < bev_horizon:
            return None
        present_idx = int(((d - bev_horizon) > 0).argmax())
        if d[-1] - d[present_idx] < bev_horizon:
            return None
        if pose_dist(self.pose(previous_idx), self.pose(present_idx)) < min_step:
            return None
        return present_idx


def _numpy_gemv_entry():
    """Address of the cblas_dgemv (64-bit integer interface) of the OpenBLAS numpy has loaded, or None."""
    import ctypes as C
    try:
        from threadpoolctl import threadpool_info
        paths = [p['filepath'] for p in threadpool_info() if p.get('user_api') == 'blas' and p.get('filepath')]
    except Exception:
        return None
    for path in paths:
        try:
            blas = C.CDLL(path)
        except OSError:
            continue
        for sym in ('scipy_cblas_dgemv64_', 'cblas_dgemv64_'):
            fn = getattr(blas, sym, None)
            if fn is not None:
                return C.cast(fn, C.c_void_p).value, blas
    return None


class CPoseTrack:
    """The same track kept by the C library (pca_host_track_*, csrc/pca_host.hip): one call per integrate() instead of a
    dozen numpy expressions.  Same numbers bit for bit: the matrix products go through numpy's own BLAS entry point."""
    _lib = None
    _gemv = None
    _blas = None

    def __init__(self):
        import ctypes as C
        self._C = C
        h = C.c_void_p()
        if self._lib.pca_host_track_create(C.byref(h), self._gemv) != 0:
            raise RuntimeError('pca_host_track_create failed')
        self._h = h
        self._pl = C.c_double(0.0)
        self._win = None         # (address, poses, view) behind poses_window

    def __del__(self):
        try:
            self._lib.pca_host_track_destroy(self._h)
        except Exception:
            pass

    def __len__(self):
        return self._lib.pca_host_track_len(self._h)

    def _view(self, ptr, n, width):
        if n == 0:
            return np.zeros((0, width)) if width > 1 else np.zeros(0)
        a = np.ctypeslib.as_array((self._C.c_double * (n * width)).from_address(ptr))
        return a.reshape(n, width) if width > 1 else a

    def as_array(self):
        n = len(self)
        return self._view(self._lib.pca_host_track_poses(self._h), n, 4)[:, :3].copy()

    def poses_window(self, first, last):
        """Copies of poses [first, last) (rows of x, y, z) without a copy of the whole track: a view over the library's buffer is
        kept while the buffer stays where it is (the address is asked for on every call; a grown track gets a new view)."""
        n = len(self)
        ptr = self._lib.pca_host_track_poses(self._h)
        c = self._win
        if c is None or c[0] != ptr or c[1] < n:
            c = self._win = (ptr, n, self._view(ptr, n, 4))
        return c[2][first:last, :3].copy()

    def pose(self, idx):
        n = len(self)
        i = int(idx)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError('pose index out of range')
        return self._view(self._lib.pca_host_track_poses(self._h), n, 4)[i, :3].copy()

    def seg_array(self):
        nd = self._lib.pca_host_track_n_segments(self._h)
        return self._view(self._lib.pca_host_track_segments(self._h), nd, 1).copy()

    @property
    def poses(self):
        return ArrayList(self.as_array())

    @poses.setter
    def poses(self, value):
        P = np.ascontiguousarray(np.array(value, dtype=np.float64).reshape(-1, 3))
        D = self.seg_array()
        self._lib.pca_host_track_set(self._h, P.ctypes.data, P.shape[0], D.ctypes.data, D.shape[0])

    @property
    def seg_dists(self):
        return ArrayList(self.seg_array())

    @seg_dists.setter
    def seg_dists(self, value):
        D = np.ascontiguousarray(np.array(value, dtype=np.float64).reshape(-1))
        P = np.ascontiguousarray(self.as_array())
        self._lib.pca_host_track_set(self._h, P.ctypes.data, P.shape[0], D.ctypes.data, D.shape[0])

    def apply_transform(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64)
        self._lib.pca_host_track_transform(self._h, T.ctypes.data)

    def append(self, pose):
        p = np.ascontiguousarray(np.asarray(pose, dtype=np.float64).reshape(3))
        self._lib.pca_host_track_append(self._h, p.ctypes.data)

    def push_segment(self):
        if self._lib.pca_host_track_push_segment(self._h, self._C.byref(self._pl)) != 0:
            raise IndexError('push_segment needs two poses')
        return np.float64(self._pl.value)

    def evict_beyond(self, horizon_dist, path_length):
        return int(self._lib.pca_host_track_evict_beyond(self._h, float(horizon_dist), float(path_length)))

    def incr(self):
        out = np.empty(self._lib.pca_host_track_n_segments(self._h))
        self._lib.pca_host_track_incr(self._h, out.ctypes.data)
        return out

    def step(self, T_new_prev, horizon_dist):
        T = np.ascontiguousarray(T_new_prev, dtype=np.float64)
        k = self._lib.pca_host_track_step(self._h, T.ctypes.data, float(horizon_dist), self._C.byref(self._pl))
        pl = self._pl.value
        return int(k), (None if pl != pl else np.float64(pl))

    def trigger(self, bev_horizon, previous_idx, min_step):
        k = self._lib.pca_host_track_trigger(self._h, float(bev_horizon), int(previous_idx), float(min_step))
        if k == -2:
            raise IndexError('pose index out of range')
        return None if k < 0 else int(k)


def _bind_c_track():
    """CPoseTrack if the library is built, numpy's BLAS entry point is found AND a randomised run agrees with the numpy
    form bit for bit; else None (the numpy form stays)."""
    import os
    if os.environ.get('PCA_HOST_TRACK', 'c') == 'numpy':
        return None
    lib = _c_library()
    entry = _numpy_gemv_entry()
    if lib is None or entry is None or not hasattr(lib, 'pca_host_track_step'):
        return None
    CPoseTrack._lib, CPoseTrack._gemv, CPoseTrack._blas = lib, entry[0], entry[1]
    rng = np.random.default_rng(20261004)
    # the per-pose (4,4)@(4,1) product: a closed form instead of a BLAS call per pose, if one reproduces numpy's bits here
    n = 4096
    Tm = rng.normal(size=(n, 4, 4)) * rng.choice([1e-3, 1., 1e3], size=(n, 1, 1))
    Tm[::3, 3] = [0., 0., 0., 1.]                                     # rigid transforms' last row
    X = np.ones((n, 4, 1))
    X[:, :3, 0] = rng.normal(size=(n, 3)) * rng.choice([1e-2, 1., 1e2, 1e4], size=(n, 1))
    want = np.concatenate([np.matmul(Tm[i], X[i:i + 1]) for i in range(n)])[:, :, 0]
    got = np.empty((n, 4))
    lib.pca_host_gemv4_mode(0)
    for mode in (1, 2, 3, 4, 5):
        lib.pca_host_gemv4_probe(entry[0], mode, Tm.ctypes.data, np.ascontiguousarray(X[:, :, 0]).ctypes.data, n, got.ctypes.data)
        if np.array_equal(got, want):
            lib.pca_host_gemv4_mode(mode)
            break
    # tri(n) @ d in groups of 8 rows instead of the full product, if the groups come out as numpy's full product does
    lib.pca_host_incr_blocks(0)
    ok = True
    for n in (2, 7, 8, 9, 63, 64, 65, 127, 200, 201, 255, 333, 512, 1000):
        d = np.ascontiguousarray(rng.uniform(0.2, 1.7, n))
        want = np.matmul(np.tri(n), d)
        out = np.empty(8)
        for r0 in sorted({0, 8 * ((n - 1) // 8), 8 * (n // 16), 8 * (n // 24)}):
            r1 = min(r0 + 8, n)
            lib.pca_host_incr_probe(entry[0], d.ctypes.data, n, r0, r1, out.ctypes.data)
            ok = ok and np.array_equal(out[:r1 - r0], want[r0:r1])
    lib.pca_host_incr_blocks(1 if ok else 0)
    a, b = CPoseTrack(), NumpyPoseTrack()
    prev = 0
    for f in range(160):
        ang = rng.uniform(-0.05, 0.05)
        T = np.eye(4)
        T[:2, :2] = [[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]
        T[:3, 3] = rng.uniform(-1.5, 0.5, 3) * [1.0, 0.05, 0.01]
        ra, rb = a.step(T, 37.5), b.step(T, 37.5)
        prev -= ra[0]
        ta, tb = (a.trigger(11.0, prev, 1.0), b.trigger(11.0, prev, 1.0)) if -len(b) <= prev < len(b) else (None, None)
        if ra != rb or ta != tb or not np.array_equal(a.as_array(), b.as_array()) \
                or not np.array_equal(a.seg_array(), b.seg_array()) or not np.array_equal(a.incr(), b.incr()):
            return None
        if ta is not None:
            prev = ta
    return CPoseTrack


_TRACK_CLASS = []


def PoseTrack():
    """The accumulator's pose track: the C form when it is available and agrees with numpy on this machine (checked once
    per process), else the numpy form.  PCA_HOST_TRACK=numpy forces the latter."""
    if not _TRACK_CLASS:
        try:
            _TRACK_CLASS.append(_bind_c_track() or NumpyPoseTrack)
        except Exception:
            _TRACK_CLASS.append(NumpyPoseTrack)
    return _TRACK_CLASS[0]()


# ----------------------------------------------------------------------------------------------
#  BEV frame
# ----------------------------------------------------------------------------------------------
def rotation_matrix_3d(ang):
    c, s = np.cos(ang), np.sin(ang)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def heading_rot_ang(ego_traj_present):
    """Rotation that points the last present ego step 'up' in the BEV."""
    ang = 0.5 * np.pi
    if len(ego_traj_present) > 1:
        dx = ego_traj_present[-1][0] - ego_traj_present[-2][0]
        dy = ego_traj_present[-1][1] - ego_traj_present[-2][1]
        ang += np.arctan2(dy, dx)
    return np.pi - ang


def pos2grid_inplace(mat, view, px):
    mat[:, 0:2] = np.floor(mat[:, 0:2] / view * px + 0.5 * px)
    return mat


# ----------------------------------------------------------------------------------------------
#  trajectories
# ----------------------------------------------------------------------------------------------
def _inside(x, y, b0x, b0y, b1x, b1y):
    return (b0x < x and x < b1x) and (b0y < y and y < b1y)


def bisect_box_crossing(x0, y0, x1, y1, bbox, thresh=1e-4):
    """Midpoint refinement of the point where segment (p0,p1) crosses the box; exactly one end inside.
    Plain Python floats (IEEE doubles, same roundings as numpy scalars, several times faster)."""
    x0, y0, x1, y1 = float(x0), float(y0), float(x1), float(y1)
    b0x, b0y, b1x, b1y = (float(b) for b in bbox)
    gap = math.inf
    iters = 0
    xm = ym = 0.0
    while gap > thresh:
        xm = 0.5 * (x0 + x1)
        ym = 0.5 * (y0 + y1)
        p0_in = (b0x < x0 and x0 < b1x) and (b0y < y0 and y0 < b1y)
        mid_in = (b0x < xm and xm < b1x) and (b0y < ym and ym < b1y)
        # the midpoint replaces whichever end lies on its own side of the border
        if mid_in == p0_in:
            gap = math.sqrt((xm - x0)**2 + (ym - y0)**2)
            x0, y0 = xm, ym
        else:
            gap = math.sqrt((xm - x1)**2 + (ym - y1)**2)
            x1, y1 = xm, ym
        iters += 1
    return xm, ym, iters


def crop_trajectory(traj, view, thresh=1e-4):
    """Clips a polyline to the open view box.  Walks EDGES (a -> b): an inside `a` is emitted, an edge that
    crosses the border additionally emits the bisected crossing point.  Consequences kept from the
    reference: the final vertex is never emitted on its own and every emitted point carries
    the z of its edge's first vertex.  Vertex classification is vectorised; only crossing edges loop."""
    n = traj.shape[0]
    if n < 2:
        return np.zeros((0, 3))
    h = 0.5 * view
    lo = -h
    x, y = traj[:, 0], traj[:, 1]
    inside = (lo < x) & (x < h) & (lo < y) & (y < h)
    a_in, b_in = inside[:-1], inside[1:]
    crossing = a_in != b_in
    per_edge = a_in.astype(np.int64) + crossing.astype(np.int64)
    total = int(per_edge.sum())
    if total == 0:
        return np.zeros((0, 3))
    start = np.cumsum(per_edge) - per_edge
    out = np.empty((total, 3))
    ka = np.nonzero(a_in)[0]
    out[start[ka]] = traj[ka, :3]
    bbox = [lo, lo, h, h]
    for k in np.nonzero(crossing)[0]:
        ix, iy, _ = bisect_box_crossing(traj[k, 0], traj[k, 1], traj[k + 1, 0], traj[k + 1, 1], bbox, thresh)
        out[start[k] + (1 if a_in[k] else 0)] = (ix, iy, traj[k, 2])
    return out


def _crop_edges(traj, view, thresh=1e-4):
    """Per-edge emission of crop_trajectory: returns (rows (M,3), first output row of every edge (E+1,))."""
    n = traj.shape[0]
    if n < 2:
        return np.zeros((0, 3)), np.zeros(max(n, 1), dtype=np.int64)
    h = 0.5 * view
    lo = -h
    x, y = traj[:, 0], traj[:, 1]
    inside = (lo < x) & (x < h) & (lo < y) & (y < h)
    a_in, b_in = inside[:-1], inside[1:]
    crossing = a_in != b_in
    per_edge = a_in.astype(np.int64) + crossing.astype(np.int64)
    start = np.concatenate([[0], np.cumsum(per_edge)])
    out = np.empty((int(start[-1]), 3))
    ka = np.nonzero(a_in)[0]
    out[start[ka]] = traj[ka, :3]
    bbox = [lo, lo, h, h]
    for k in np.nonzero(crossing)[0]:
        ix, iy, _ = bisect_box_crossing(traj[k, 0], traj[k, 1], traj[k + 1, 0], traj[k + 1, 1], bbox, thresh)
        out[start[k] + (1 if a_in[k] else 0)] = (ix, iy, traj[k, 2])
    return out, start


_CLIB = []


def _c_library():
    """The C-ABI library if it is built (its host helpers need no GPU); None otherwise -> numpy forms."""
    if not _CLIB:
        try:
            from . import _lib
            _CLIB.append(_lib.load())
        except Exception:
            _CLIB.append(None)
    return _CLIB[0]


def transform_ego_split(full, split, rot_mat, dx, dy, view, px):
    """The three ego trajectories of one sample -- present = full[:split], future = full[split:], full -- in
    grid coordinates, from ONE rotate / translate / clip / floor pass.  Every step is row- or edge-local, so
    the slices equal what three separate transform_traj calls return (the present set simply lacks the edge
    split-1 -> split, the future set starts at edge split)."""
    t = np.array(full, dtype=np.float64)
    n = t.shape[0]
    lib = _c_library()
    if lib is not None and t.ndim == 2 and t.shape[1] == 3:
        # same arithmetic in one C call (pca_host_ego_to_grid): the numpy form below costs ~40 us of Python per sample
        R = np.ascontiguousarray(rot_mat, dtype=np.float64)
        rows = np.empty((max(2 * (n - 1), 1), 3))
        start = np.zeros(max(n, 1), dtype=np.int32)
        m = lib.pca_host_ego_to_grid(t.ctypes.data, n, R.ctypes.data, float(dx), float(dy), float(view), int(px),
                                     rows.ctypes.data, start.ctypes.data)
        rows = rows[:m]
    else:
        t[:, :3] = np.matmul(rot_mat, t[:, :3].T).T
        t[:, 0] += dx
        t[:, 1] += dy
        rows, start = _crop_edges(t, view)
        rows = pos2grid_inplace(rows, view, px)
    empty = np.zeros((0, 3))
    e_p = max(split - 1, 0)                                  # edges 0 .. split-2 belong to the present polyline
    present = rows[:start[e_p]].copy() if split >= 2 else empty
    future = rows[start[split]:].copy() if n - split >= 2 else empty
    return present, future, rows


def transform_traj(traj, rot_mat, dx, dy, view, px, mutate=True):
    """rotate -> translate -> clip -> grid coordinates, for one (k,3) trajectory.  Mutates `traj` like the
    reference (bev_generator.py:226-231) unless the caller does not need that (mutate=False: the same arithmetic in one C
    call, pca_host_ego_to_grid, as transform_ego_split uses it -- a NuScenes sample has three polylines per agent)."""
    if not mutate:
        lib = _c_library()
        if lib is not None and isinstance(traj, np.ndarray) and traj.ndim == 2 and traj.shape[1] == 3:
            t = np.ascontiguousarray(traj, dtype=np.float64)
            n = t.shape[0]
            if n < 2:
                return np.zeros((0, 3))
            R = np.ascontiguousarray(rot_mat, dtype=np.float64)
            rows = np.empty((2 * (n - 1), 3))
            start = np.empty(n, dtype=np.int32)
            m = lib.pca_host_ego_to_grid(t.ctypes.data, n, R.ctypes.data, float(dx), float(dy), float(view), int(px),
                                         rows.ctypes.data, start.ctypes.data)
            return rows[:m]
    traj[:, :3] = np.matmul(rot_mat, traj[:, :3].T).T
    traj[:, 0] += dx
    traj[:, 1] += dy
    return pos2grid_inplace(crop_trajectory(traj, view), view, px)


# ----------------------------------------------------------------------------------------------
#  polynomial warp (data augmentation, --bev_do_warp)
# ----------------------------------------------------------------------------------------------
def cal_warp_params(idx_0, idx_1, idx_max):
    a_1 = (idx_1 - idx_0**2 / idx_max) / (idx_0 * (1.0 - idx_0 / idx_max))
    a_2 = (1.0 - a_1) / idx_max
    return (a_1, a_2)


def get_random_warp_params(mean_ratio, max_ratio, I, J):
    max_val = max_ratio * (I / 2.0)
    mean_val = mean_ratio * max_val
    i_warp = np.random.normal(mean_val, max_val)
    j_warp = np.random.normal(mean_val, max_val)
    if abs(i_warp) > max_val:
        i_warp = max_val
    if abs(j_warp) > max_val:
        j_warp = max_val
    if random.random() < 0.5:
        i_warp = -i_warp
    if random.random() < 0.5:
        j_warp = -j_warp
    return (int(I / 2) + i_warp, int(J / 2) + j_warp)


def warp_source_index(coef_1, coef_2, n):
    """Source index for every warped index 0..n-1: clamp(rint(c1*k + c2*k^2))."""
    k = np.arange(n, dtype=np.float64)
    src = np.rint(coef_1 * k + coef_2 * (k * k)).astype(np.int64)
    return np.clip(src, 0, n - 1)


def warp_dense_probmaps(maps, a_1, a_2, b_1, b_2):
    """out[:, jw, iw] = maps[:, j(jw), i(iw)] -- vectorised gather form of the reference's 2-D loop
    (note the reference's transposed write, bev_generator.py:523)."""
    _, I, J = maps.shape
    i_src = warp_source_index(a_1, a_2, I)
    j_src = warp_source_index(b_1, b_2, J)
    # reference loops i_warp over range(I) (clamped to I) and j_warp over range(J) (clamped to J) and
    # writes B[:, j_warp, i_warp] = A[:, j, i]
    out = np.zeros(maps.shape)
    out[:, :J, :I] = maps[:, j_src[:, None], i_src[None, :]]
    return out


def warp_point(x, y, a_1, a_2, b_1, b_2, I, J):
    if math.isclose(a_2, 0.0, abs_tol=1e-6):
        xw = x
    else:
        xw = int(np.rint((-a_1 + np.sqrt(a_1**2 + 4.0 * a_2 * x)) / (2 * a_2)))
    if math.isclose(b_2, 0.0, abs_tol=1e-6):
        yw = y
    else:
        yw = int(np.rint((-b_1 + np.sqrt(b_1**2 + 4.0 * b_2 * y)) / (2 * b_2)))
    xw = 0 if xw < 0 else (I - 1 if xw >= I else xw)
    yw = 0 if yw < 0 else (J - 1 if yw >= J else yw)
    return (xw, yw)


def warp_sparse_points(pnts, a_1, a_2, j_warp, j_mid, px):
    """Inverse warp of trajectory vertices; the j axis uses the mirrored warp centre."""
    b_1r, b_2r = cal_warp_params(px - j_warp, j_mid, px - 1)
    for k in range(pnts.shape[0]):
        xw, yw = warp_point(pnts[k, 0], pnts[k, 1], a_1, a_2, b_1r, b_2r, px, px)
        pnts[k, 0] = xw
        pnts[k, 1] = yw
    return pnts
