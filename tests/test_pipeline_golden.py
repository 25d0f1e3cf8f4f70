"""Full integrate()/generate_bev() sequences of the reference (golden fixtures) reproduced with the
oracle kernels + the product's host logic (pose track, eviction, tracker, trajectories).  CPU only."""
import pytest
import numpy as np

from oracle import oracle as orc
from pca_amd import host_logic as hl
from pca_amd.tracker import InstanceTracker

KITTI_FILTERS = [10, 11, 12, 16, 18, 255]
NUSC_FILTERS = [10, 11, 12, 16, 18]
SEM_IDXS = {'road': 0, 'car': 13, 'truck': 14, 'bus': 15, 'motorcycle': 17}
DYNOBJ = [13, 14, 15, 17]


def fake_semseg(img):
    a = img.astype(np.int64)
    return ((a[..., 0] + 2 * a[..., 1] + 3 * a[..., 2]) % 19).astype(np.uint8)


def check_bev_dict(g, out, trajs, prefix='bev_'):
    for s, name in enumerate(orc.SETS):
        F = out['f16'][7 * s:7 * s + 7]
        for k, key in ((0, 'road'), (5, 'dynamic'), (6, 'elevation')):
            assert np.array_equal(F[k].view(np.uint16), g[f'{prefix}{key}_{name}'].view(np.uint16)), (key, name)
        assert np.array_equal(F[2:5].view(np.uint16), g[f'{prefix}rgb_{name}'].view(np.uint16))
        a = F[1].view(np.uint16).astype(np.int32)
        b = g[f'{prefix}intensity_{name}'].view(np.uint16).astype(np.int32)
        assert np.abs(a - b).max() <= 1
        n = int(g[f'{prefix}trajs_{name}_n'])
        assert n == len(trajs[name])
        for k in range(n):
            assert np.array_equal(trajs[name][k], g[f'{prefix}trajs_{name}_{k}']), (name, k)


def run_bev(st, lo, sizes, present_idx, track, view, px, height_filter, ints, div255, others=None):
    origin = np.array(track.poses[present_idx])
    n_split = int(np.sum(sizes[:present_idx]))
    poses = np.concatenate([track.poses])
    ego = {'present': poses[:present_idx] - origin, 'future': poses[present_idx:] - origin, 'full': poses - origin}
    rot = hl.heading_rot_ang(ego['present'])
    R = hl.rotation_matrix_3d(rot)
    prm = orc.make_bev_params(origin, R, 0., 0., view, px, height_filter, *ints, SEM_IDXS['road'], DYNOBJ, div255)
    sub = orc.Store(st.n - lo + 1, div255)
    for name in ('x', 'y', 'z', 'intensity', 'rgbs', 'inst', 'dyn'):
        getattr(sub, name)[:st.n - lo] = getattr(st, name)[lo:st.n]
    sub.n = st.n - lo
    out = orc.bev(sub, n_split, prm)
    trajs = {}
    for k, name in enumerate(orc.SETS):
        lst = [ego[name]] + ([] if others is None else [np.concatenate([t]) - origin for t in others[k]])
        trajs[name] = [hl.transform_traj(t.copy(), R, 0., 0., view, px) for t in lst]
    return out, trajs


def test_kitti_integrate_and_bev(golden):
    g = golden('kitti_accum')
    F, H, W = int(g['F']), int(g['H']), int(g['W'])
    st = orc.Store(F * 3000)
    lo = 0
    sizes = []
    track = hl.PoseTrack()
    removed = []
    for k in range(F):
        T = g['Ts'][k]
        img = g[f'img_{k}']
        if track.poses:
            track.apply_transform(T)
            orc.retransform(st, T, lo, st.n)
        m = orc.kitti_project_sample_filter(st, g[f'pc_{k}'], g['P'], img, fake_semseg(img), None, H, W,
                                            KITTI_FILTERS)
        sizes.append(m)
        track.append([0., 0., 0.])
        ev = 0
        if len(track.poses) > 1:
            ev = track.evict_beyond(float(g['horizon']), track.push_segment())
            lo += int(np.sum(sizes[:ev]))
            sizes = sizes[ev:]
        removed.append(ev)
        if f'step{k}_sizes' in g:
            assert np.array_equal(np.array(sizes), g[f'step{k}_sizes'])
            assert np.array_equal(st.rows(lo), g[f'step{k}_sem_pcs'])
            assert np.array_equal(np.array(track.poses), g[f'step{k}_poses'])
            assert np.array_equal(np.array(track.seg_dists), g[f'step{k}_seg_dists'])
    assert np.array_equal(np.array(removed), g['removed'])
    assert np.array_equal(hl.incremental_path_dists(track.seg_dists), g['incr_path_dists'])
    out, trajs = run_bev(st, lo, sizes, int(g['present_idx']), track, 40, 32, None, (20., 20., 0.5), False)
    check_bev_dict(g, out, trajs)


def test_kitti_use_gt_sem(golden):
    g = golden('kitti_gtsem')
    st = orc.Store(4 * 2000)
    track = hl.PoseTrack()
    sizes = []
    for k in range(4):
        T = g['Ts'][k]
        if track.poses:
            track.apply_transform(T)
            orc.retransform(st, T)
        sem_gt = g[f'sem_gt_{k}'][:, -1].astype(np.uint8)      # int16 trainIds 0..18 / 255
        sizes.append(orc.kitti_project_sample_filter(st, g[f'pc_{k}'], g['P'], None, None, sem_gt, 64, 96,
                                                     KITTI_FILTERS))
        track.append([0., 0., 0.])
        if len(track.poses) > 1:
            assert track.evict_beyond(50., track.push_segment()) == 0
    assert np.array_equal(np.array(sizes), g['sizes'])
    assert np.array_equal(st.rows(), g['sem_pcs'])
    assert np.array_equal(np.array(track.poses), g['poses'])
    out, trajs = run_bev(st, 0, sizes, 2, track, 40, 32, None, (20., 20., 0.5), False)
    check_bev_dict(g, out, trajs)


def test_nuscenes_oracle_integrate_tracker_and_bev(golden):
    g = golden('nusc_oracle')
    F = int(g['F'])
    st = orc.Store(F * 2500, intensity_div255=True)
    track = hl.PoseTrack()
    tracker = InstanceTracker()
    offs = [0]
    T_global_world = None
    for k in range(F):
        T_ego_global = g[f'T_{k}']
        if T_global_world is None:
            T_global_world = np.linalg.inv(T_ego_global)
        T_ego_world = T_global_world @ T_ego_global
        pose = T_ego_world[:3, -1].tolist()
        pose[2] += 1.
        imgs = g[f'imgs_{k}']
        sems = np.stack([fake_semseg(im) for im in imgs])
        m = orc.nusc_sample_filter_transform(st, g[f'pc_{k}'], g[f'cam_idx_{k}'], imgs, sems, T_ego_world,
                                             NUSC_FILTERS)
        offs.append(offs[-1] + m)
        track.append(pose)
        if k == 0:
            assert np.array_equal(st.rows(0, m), g['frame0_after_integrate'])
        tokens = str(g['inst_tokens'][k]).split(',')
        centers = [orc.homo_transform(T_global_world, c[None])[0] for c in g[f'inst_center_{k}']]
        for ts, inst_idx in tracker.observe(k, tokens, list(g[f'inst_cls_{k}']), centers):
            orc.mark_dynamic(st, offs[ts], offs[ts + 1], inst_idx)
        if len(track.poses) > 1:
            track.push_segment()
    sizes = np.diff(offs)
    assert np.array_equal(sizes, g['sizes'])
    assert np.array_equal(st.rows(), g['sem_pcs'])
    assert g['sem_pcs'][:, 9].sum() > 0
    assert np.array_equal(np.array(track.poses), g['poses'])
    assert np.array_equal(np.array(track.seg_dists), g['seg_dists'])
    assert list(g['dyn_instances']) == tracker.dynamic
    pi = int(g['present_idx'])
    out, trajs = run_bev(st, 0, list(sizes), pi, track, 30, 32, 3., (1., 30., 0.12), True,
                         others=tracker.split_trajectories(pi))
    check_bev_dict(g, out, trajs)
    assert int(g['bev_trajs_full_n']) > 1


def test_host_logic_utils(golden):
    g = golden('utils')
    assert np.array_equal(hl.incremental_path_dists(g['pd_seg']), g['pd_incr'])
    for k in range(int(g['ct_n'])):
        assert np.array_equal(hl.crop_trajectory(g[f'ct_in_{k}'].copy(), 20.), g[f'ct_out_{k}'])
    a1, a2 = hl.cal_warp_params(19.7, 16, 31)
    b1, b2 = hl.cal_warp_params(13.8, 16, 31)
    assert np.array_equal(np.array([a1, a2, b1, b2]), g['wp_params'])
    assert np.array_equal(hl.warp_dense_probmaps(g['wp_in'], a1, a2, b1, b2), g['wp_out'])
    assert np.array_equal(hl.warp_sparse_points(g['ws_in'].copy(), a1, a2, 13.8, 16, 32), g['ws_out'])


def test_ego_split_transform_equals_three_separate_transforms():
    """The accumulators transform the ego polyline once and slice it; must equal the reference's three calls."""
    rng = np.random.default_rng(12)
    for trial in range(200):
        n = int(rng.integers(2, 60))
        step = rng.uniform(0.3, 3.0)
        ang = np.cumsum(rng.normal(0, 0.15, n))
        pts = np.stack([np.cumsum(step * np.cos(ang)), np.cumsum(step * np.sin(ang)), rng.normal(0, 0.1, n)], 1)
        pts -= pts[int(rng.integers(0, n))]
        split = int(rng.integers(1, n))
        R = hl.rotation_matrix_3d(rng.uniform(0, 6.28))
        dx, dy, view, px = rng.uniform(-2, 2), rng.uniform(-2, 2), float(rng.choice([10., 20., 51.2])), 64
        p, f, a = hl.transform_ego_split(pts, split, R, dx, dy, view, px)
        want = [hl.transform_traj(t.copy(), R, dx, dy, view, px) for t in (pts[:split], pts[split:], pts)]
        for got, w in zip((p, f, a), want):
            assert got.shape == w.shape and np.array_equal(got, w), (trial, n, split)


def test_transform_traj_c_call_equals_numpy_form():
    """transform_traj(mutate=False) -- the one C call generate() uses for the other agents' polylines -- returns what the
    reference's in-place numpy form returns, on random polylines incl. ones that leave and re-enter the view."""
    rng = np.random.default_rng(21)
    for trial in range(300):
        n = int(rng.integers(0, 40))
        pts = np.cumsum(rng.normal(0, 2.5, (n, 3)), axis=0) + rng.normal(0, 8, 3) if n else np.zeros((0, 3))
        R = hl.rotation_matrix_3d(rng.uniform(0, 6.28))
        dx, dy, view, px = rng.uniform(-2, 2), rng.uniform(-2, 2), float(rng.choice([10., 20., 51.2])), int(rng.choice([32, 256]))
        keep = pts.copy()
        got = hl.transform_traj(pts, R, dx, dy, view, px, mutate=False)
        assert np.array_equal(pts, keep)                                   # left alone
        want = hl.transform_traj(keep.copy(), R, dx, dy, view, px)
        assert got.shape == want.shape and np.array_equal(got, want), (trial, n)


def test_pts_feat_from_img_nearest_and_optin_bilinear(golden):
    """datasets/nuscenes_utils.py:181-214: 'nearest' is what every caller uses (and what K1n fuses); 'bilinear' is
    the reference's unused branch (2-D maps only), kept as an opt-in and pinned on its own output: here the oracle's
    restatement, in tests/test_gpu_kernels.py the device kernel (pca_sample_bilinear) against the same fixture."""
    from datasets.nuscenes_utils import pts_feat_from_img
    from oracle import oracle as orc
    g = golden('utils')
    assert np.array_equal(pts_feat_from_img(g['pf_uv'], g['pf_img'], 'nearest'), g['pf_nearest'])
    assert np.array_equal(orc.sample_bilinear(g['pf_img'][..., 0], g['pf_uv_bil']), g['pf_bilinear'])
    with pytest.raises(AssertionError):
        orc.sample_bilinear(g['pf_img'][..., 0], np.array([[0.5, 5.0]]))
    with pytest.raises(AssertionError):
        pts_feat_from_img(np.array([[0.5, 5.0]]), g['pf_img'], 'nearest')
