"""ctypes front-end of the CPU oracle (oracle/pca_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package never does.  The oracle is pinned against golden vectors made by running the real
reference (tests/test_oracle_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'pca_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'liboracle.so'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_kitti_project_sample_filter.restype = C.c_int64
        _LIB.orc_nusc_sample_filter_transform.restype = C.c_int64
        _LIB.orc_f64_to_f16.restype = C.c_uint16
        _LIB.orc_f64_to_f16.argtypes = [C.c_double]
    return _LIB


class _CStore(C.Structure):
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('z', C.c_void_p), ('intensity', C.c_void_p),
                ('rgbs', C.c_void_p), ('inst', C.c_void_p), ('dyn', C.c_void_p)]


class BevParams(C.Structure):
    _fields_ = [('origin', C.c_double * 3), ('R', C.c_double * 9), ('dx', C.c_double), ('dy', C.c_double),
                ('view', C.c_double), ('height_filter', C.c_double), ('int_scaler', C.c_double),
                ('int_sep_scaler', C.c_double), ('int_mid_threshold', C.c_double), ('rgb_fill', C.c_double),
                ('px', C.c_int32), ('road_class', C.c_int32), ('dynobj_mask', C.c_uint64 * 4),
                ('intensity_div255', C.c_int32), ('pad', C.c_int32)]


def class_mask(classes):
    m = [0, 0, 0, 0]
    for c in classes:
        c = int(c)
        if 0 <= c < 256:
            m[c >> 6] |= 1 << (c & 63)
    return np.array(m, dtype=np.uint64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Store:
    """Host SoA point store with the device store's layout."""

    def __init__(self, capacity, intensity_div255=False):
        self.cap = int(capacity)
        self.n = 0
        self.x = np.zeros(self.cap)
        self.y = np.zeros(self.cap)
        self.z = np.zeros(self.cap)
        self.intensity = np.zeros(self.cap, np.float32)
        self.rgbs = np.zeros(self.cap, np.uint32)
        self.inst = np.zeros(self.cap, np.int32)
        self.dyn = np.zeros(self.cap, np.uint8)
        self.intensity_div255 = bool(intensity_div255)

    def c(self):
        return _CStore(*[a.ctypes.data for a in (self.x, self.y, self.z, self.intensity, self.rgbs, self.inst,
                                                  self.dyn)])

    def rows(self, lo=0, hi=None):
        """(M,10) f64 rows exactly as the reference stores them in sem_pcs."""
        hi = self.n if hi is None else hi
        s = slice(lo, hi)
        out = np.empty((hi - lo, 10))
        out[:, 0], out[:, 1], out[:, 2] = self.x[s], self.y[s], self.z[s]
        i = self.intensity[s].astype(np.float64)
        out[:, 3] = i / 255. if self.intensity_div255 else i
        r = self.rgbs[s]
        out[:, 4], out[:, 5], out[:, 6], out[:, 7] = r & 255, (r >> 8) & 255, (r >> 16) & 255, r >> 24
        out[:, 8] = self.inst[s]
        out[:, 9] = self.dyn[s]
        return out

    @staticmethod
    def from_rows(rows, intensity_div255=False):
        """Inverse of rows() for (M,10) arrays whose columns 3..9 are representable."""
        rows = np.asarray(rows, dtype=np.float64)
        st = Store(max(rows.shape[0], 1), intensity_div255)
        m = rows.shape[0]
        st.n = m
        st.x[:m], st.y[:m], st.z[:m] = rows[:, 0], rows[:, 1], rows[:, 2]
        raw = np.rint(rows[:, 3] * 255.) if intensity_div255 else rows[:, 3]
        st.intensity[:m] = raw.astype(np.float32)
        c = rows[:, 4:8].astype(np.uint32)
        st.rgbs[:m] = c[:, 0] | (c[:, 1] << 8) | (c[:, 2] << 16) | (c[:, 3] << 24)
        st.inst[:m] = rows[:, 8].astype(np.int32)
        st.dyn[:m] = rows[:, 9].astype(np.uint8)
        return st


def set_sample_mode(mode):
    """0 nearest (the reference), 1 bilinear rgb (the opt-in mode of K1 / K1n)."""
    lib().orc_set_sample_mode(int(mode))


def sample_bilinear(feat_map, uv):
    """pts_feat_from_img(uv, feat_map, 'bilinear') for a 2-D map."""
    feat_map = np.ascontiguousarray(feat_map, np.float64)
    uv = np.ascontiguousarray(uv, np.float64)
    out = np.empty(uv.shape[0])
    rc = lib().orc_sample_bilinear(_p(feat_map), int(feat_map.shape[0]), int(feat_map.shape[1]), _p(uv),
                                   C.c_int64(uv.shape[0]), _p(out))
    if rc:
        raise AssertionError('pts_uv must be all inside image')
    return out


def kitti_project_sample_filter(st, pts, P, rgb, sem, sem_gt, H, W, filters, want_uv=False):
    pts = np.ascontiguousarray(pts, np.float32)
    n = pts.shape[0]
    P = np.ascontiguousarray(P, np.float64)
    rgb = None if rgb is None else np.ascontiguousarray(rgb, np.uint8)
    sem = None if sem is None else np.ascontiguousarray(sem, np.uint8)
    sem_gt = None if sem_gt is None else np.ascontiguousarray(sem_gt, np.uint8).ravel()
    fm = class_mask(filters)
    mask = np.zeros(n, np.uint8) if want_uv else None
    u = np.zeros(n, np.int64) if want_uv else None
    v = np.zeros(n, np.int64) if want_uv else None
    assert st.n + n <= st.cap
    cs = st.c()
    m = lib().orc_kitti_project_sample_filter(_p(pts), C.c_int64(n), _p(P), _p(rgb), _p(sem), _p(sem_gt), int(H),
                                              int(W), _p(fm), C.byref(cs), C.c_int64(st.n), _p(mask), _p(u), _p(v))
    st.n += m
    return (m, mask.astype(bool), u, v) if want_uv else m


def retransform(st, T, lo=0, hi=None):
    hi = st.n if hi is None else hi
    T = np.ascontiguousarray(T, np.float64)
    x, y, z = st.x[lo:hi], st.y[lo:hi], st.z[lo:hi]
    lib().orc_retransform(_p(x), _p(y), _p(z), C.c_int64(hi - lo), _p(T))


def homo_transform(T, pts):
    T = np.ascontiguousarray(T, np.float64)
    pts = np.ascontiguousarray(pts, np.float64)
    out = np.empty_like(pts)
    lib().orc_homo_transform(_p(T), _p(pts), C.c_int64(pts.shape[0]), _p(out))
    return out


def nusc_sample_filter_transform(st, pc, cam_idx, imgs, sems, T, filters):
    pc = np.ascontiguousarray(pc, np.float64)
    cam_idx = np.ascontiguousarray(cam_idx, np.int64)
    imgs = np.ascontiguousarray(imgs, np.uint8)
    sems = np.ascontiguousarray(sems, np.uint8)
    ncam, H, W = sems.shape
    T = np.ascontiguousarray(T, np.float64)
    fm = class_mask(filters)
    assert st.n + pc.shape[0] <= st.cap
    cs = st.c()
    m = lib().orc_nusc_sample_filter_transform(_p(pc), _p(cam_idx), C.c_int64(pc.shape[0]), _p(imgs), _p(sems),
                                               int(ncam), int(H), int(W), _p(T), _p(fm), C.byref(cs),
                                               C.c_int64(st.n))
    if m < 0:
        raise AssertionError('pts_uv must be all inside image')
    st.n += m
    return m


def mark_dynamic(st, begin, end, inst_idx):
    lib().orc_mark_dynamic(_p(st.inst), _p(st.dyn), C.c_int64(begin), C.c_int64(end), C.c_int32(inst_idx))


def nusc_project_cams(pc_lidar, T_ego_from_lidar, T_glob_from_ego, T_cam_from_glob, K, wh):
    pc_lidar = np.ascontiguousarray(pc_lidar, np.float64)
    n = pc_lidar.shape[0]
    a = [np.ascontiguousarray(t, np.float64) for t in (T_ego_from_lidar, T_glob_from_ego, T_cam_from_glob, K, wh)]
    ncam = a[2].shape[0]
    ego = np.empty((n, 3))
    uv = np.empty((n, 2))
    cam = np.empty(n, np.int64)
    lib().orc_nusc_project_cams(_p(pc_lidar), C.c_int64(n), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]),
                                int(ncam), _p(ego), _p(uv), _p(cam))
    return ego, uv, cam


def f64_to_f16(a):
    a = np.ascontiguousarray(a, np.float64)
    f = lib().orc_f64_to_f16
    return np.array([f(float(v)) for v in a.ravel()], np.uint16).reshape(a.shape).view(np.float16)


PLANES = ('road', 'intensity', 'r', 'g', 'b', 'dynamic', 'elevation')
SETS = ('present', 'future', 'full')


def make_bev_params(origin, R, dx, dy, view, px, height_filter, int_scaler, int_sep_scaler, int_mid_threshold,
                    road_class, dynobj_classes, intensity_div255, rgb_fill=0.):
    prm = BevParams()
    prm.origin[:] = [float(v) for v in origin]
    prm.R[:] = [float(v) for v in np.asarray(R).ravel()]
    prm.dx, prm.dy, prm.view = float(dx), float(dy), float(view)
    prm.height_filter = float('nan') if height_filter is None else float(height_filter)
    prm.int_scaler, prm.int_sep_scaler, prm.int_mid_threshold = float(int_scaler), float(int_sep_scaler), float(
        int_mid_threshold)
    prm.rgb_fill = float(rgb_fill)
    prm.px = int(px)
    prm.road_class = int(road_class)
    prm.dynobj_mask[:] = [int(v) for v in class_mask(dynobj_classes)]
    prm.intensity_div255 = int(bool(intensity_div255))
    return prm


def bev(st, n_split, prm, n=None, intensity64=None, want_cells=False):
    """Returns dict: planes f64 [21,px,px], f16 [21,px,px], intraw [3,px,px] (, cells [n])."""
    n = st.n if n is None else n
    px = prm.px
    planes = np.zeros((21, px, px))
    f16 = np.zeros((21, px, px), np.uint16)
    intraw = np.zeros((3, px, px))
    cells = np.zeros(max(n, 1), np.int64) if want_cells else None
    cs = st.c()
    i64 = None if intensity64 is None else np.ascontiguousarray(intensity64, np.float64)
    rc = lib().orc_bev(C.byref(cs), _p(i64), C.c_int64(n), C.c_int64(n_split), C.byref(prm), _p(planes), _p(f16),
                       _p(intraw), _p(cells))
    assert rc == 0
    out = dict(planes=planes, f16=f16.view(np.float16), intraw=intraw)
    if want_cells:
        out['cells'] = cells[:n]
    return out
