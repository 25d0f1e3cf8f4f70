"""numpy-shaped CPU baseline.  TEST / BENCHMARK INFRASTRUCTURE ONLY (never imported by the product).

The reference is pure numpy; the reference itself never travels to the GPU box, so `bench.py`'s `cpu_baseline` times this
restatement, written in the reference's ALGORITHMIC SHAPE (whole-array numpy expressions, one temporary per step, the
projection evaluated twice per frame, one boolean-mask copy per filtered class, per-frame re-transform loop, histogram2d
count maps) -- SURVEY.md 8(d).  The two per-cell reductions the reference runs as Python loops (min z: sem_bev.py:535-554,
channel medians: :619-669) exist here in two forms: `vectorised` (np.lexsort segments -- what a numpy user would write)
and `loops` (the reference's own shape: a Python loop over points / over cells), timed separately by the bench.
Checked against the C oracle on small inputs by tests/test_oracle_golden.py::test_numpy_shape_matches_c_oracle.
"""
import numpy as np


# ---------------------------------------------------------------------------------------------- integrate (a1-a6)
def project(pc, P, H, W):
    """rows [x, y, z, i, u, v] of the points inside the image (sem_pc_accum.py:347-402)."""
    homo = np.concatenate((pc[:, :3], np.ones((pc.shape[0], 1))), axis=1)
    frame = np.matmul(P, homo.T).T
    depth = frame[:, 2]
    depth[depth == 0] = -1e-6
    u = np.round(frame[:, 0] / np.abs(depth)).astype(int)
    v = np.round(frame[:, 1] / np.abs(depth)).astype(int)
    inside = np.logical_and(np.logical_and(np.logical_and(u >= 0, u < W), v >= 0), v < H)
    inside = np.logical_and(np.logical_and(inside, depth > 0), depth < np.inf)
    return np.concatenate([pc, u[:, None], v[:, None]], axis=1)[inside]


def sample(pc, P, feat_map):
    rows = project(pc, P, feat_map.shape[0], feat_map.shape[1])
    feat = feat_map[rows[:, 5].astype(int), rows[:, 4].astype(int), :]
    return np.concatenate([rows[:, :4], feat], axis=1)


def integrate_frame(pc, P, img, sem, filters):
    """(M,10) rows of one KITTI frame (kitti360_sem_pc_accum.py:129-156): projection + gather twice, class filter."""
    a = sample(pc, P, img)
    b = sample(pc, P, sem[..., None])
    rows = np.concatenate((a, b[:, -1:]), axis=1)
    for c in filters:
        rows = rows[rows[:, -1] != c]
    return np.concatenate([rows, np.zeros((rows.shape[0], 2))], axis=1)


def retransform(frames, T):
    """sem_pc_accum.py:167-183: every stored frame, one homogeneous product each."""
    for rows in frames:
        homo = np.concatenate((rows[:, :3], np.ones((rows.shape[0], 1))), axis=1)
        rows[:, :3] = np.matmul(T, homo.T).T[:, :3]


# ---------------------------------------------------------------------------------------------- BEV (a16-a23)
def _prep(rows, R, view, px):
    rows = rows.copy()
    rows[:, :3] = np.matmul(R, rows[:, :3].T).T
    h = 0.5 * view
    rows = rows[np.logical_and(rows[:, 0] > -h, rows[:, 0] < h)]
    rows = rows[np.logical_and(rows[:, 1] > -h, rows[:, 1] < h)]
    rows[:, 0:2] = np.floor(rows[:, 0:2] / view * px + 0.5 * px)
    return rows


def _counts(rows, px, weights=None):
    g, _, _ = np.histogram2d(rows[:, 1], rows[:, 0], bins=px, range=[[0, px], [0, px]], weights=weights)
    return np.flip(g, axis=0)


def _min_z_vectorised(rows, px):
    out = np.zeros((px, px))
    if rows.shape[0]:
        cell = (px - 1 - rows[:, 1].astype(int)) * px + rows[:, 0].astype(int)
        order = np.lexsort((rows[:, 2], cell))
        first = np.concatenate([[True], cell[order][1:] != cell[order][:-1]])
        out.ravel()[cell[order][first]] = rows[order, 2][first]
    return out


def _median_vectorised(rows, px, col):
    out = np.zeros((px, px))
    if rows.shape[0]:
        cell = (px - 1 - rows[:, 1].astype(int)) * px + rows[:, 0].astype(int)
        order = np.lexsort((rows[:, col], cell))
        c, v = cell[order], rows[order, col]
        start = np.flatnonzero(np.concatenate([[True], c[1:] != c[:-1]]))
        n = np.diff(np.concatenate([start, [c.size]]))
        lo, hi = start + (n - 1) // 2, start + n // 2
        out.ravel()[c[start]] = 0.5 * (v[lo] + v[hi])
    return out


def _min_z_loops(rows, px):
    """The reference's shape: one Python iteration per point."""
    out = np.zeros((px, px))
    seen = np.zeros((px, px), dtype=bool)
    for k in range(rows.shape[0]):
        i, j, z = int(rows[k, 0]), int(rows[k, 1]), rows[k, 2]
        r = px - 1 - j
        if not seen[r, i] or z < out[r, i]:
            out[r, i] = z
            seen[r, i] = True
    return out


def _median_loops(rows, px, col):
    """The reference's shape: bucket the points per cell in Python, then one np.median per cell."""
    buckets = {}
    for k in range(rows.shape[0]):
        buckets.setdefault((px - 1 - int(rows[k, 1]), int(rows[k, 0])), []).append(rows[k, col])
    out = np.zeros((px, px))
    for r in range(px):
        for c in range(px):
            vals = buckets.get((r, c))
            if vals:
                out[r, c] = np.median(vals)
    return out


def bev_set(rows, R, view, px, road_class, dyn_classes, int_params, loops=False, median_cols=(4, 5, 6)):
    """Seven planes of one point set (sem_bev.py:57-118)."""
    g = _prep(rows, R, view, px)
    static = g[g[:, 9] != 1]
    road = static[static[:, 7] == road_class]
    notroad = static[static[:, 7] != road_class]
    n_road, n_not = _counts(road, px), _counts(notroad, px)
    p_road = (n_road + 1.) / (n_road + n_not + 2.)
    dynm = np.isin(static[:, 7], dyn_classes)
    n_dyn, n_nd = _counts(static[dynm], px), _counts(static[~dynm], px)
    p_dyn = (n_dyn + 1.) / (n_dyn + n_nd + 2.)
    inten = _counts(road, px, weights=road[:, 3]) / (n_road + 1.)
    s, k, m = int_params
    inten = np.minimum(1., s * (1. / (1. + np.exp(-k * (inten - m)))))
    zmin = (_min_z_loops if loops else _min_z_vectorised)(static, px)
    med = _median_loops if loops else _median_vectorised
    rgb = [med(static, px, c) / 255. if c in median_cols else np.zeros((px, px)) for c in (4, 5, 6)]
    return [p_road, inten, rgb[0], rgb[1], rgb[2], p_dyn, zmin]


def bev(frames, split, origin, R, view, px, road_class, dyn_classes, int_params, loops=False):
    """21 float16 planes of a window (list of (M,10) frames, 'present' = frames[:split])."""
    def cat(fs):
        out = np.concatenate(fs) if fs else np.zeros((0, 10))
        out[:, :3] = out[:, :3] - origin
        return out
    sets = [cat(frames[:split]), cat(frames[split:]), cat(frames)]
    return np.stack([p for s in sets for p in bev_set(s, R, view, px, road_class, dyn_classes, int_params, loops)
                     ]).astype(np.float16)
