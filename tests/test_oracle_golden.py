"""Pins the CPU oracle (oracle/pca_oracle.c) against golden vectors produced by the real reference.

Kernel-level functions only; the full integrate()/generate_bev() sequences are pinned in
test_pipeline_golden.py (oracle kernels + product host logic)."""
import numpy as np
import pytest

from oracle import oracle as orc

KITTI_FILTERS = [10, 11, 12, 16, 18, 255]
SEM_IDXS = {'road': 0, 'car': 13, 'truck': 14, 'bus': 15, 'motorcycle': 17}
DYNOBJ = [13, 14, 15, 17]


def test_f64_to_f16_matches_numpy():
    rng = np.random.default_rng(0)
    v = np.concatenate([
        rng.uniform(-2, 2, 20000), rng.uniform(-70000, 70000, 2000), 10.0**rng.uniform(-12, 6, 20000),
        -10.0**rng.uniform(-12, 6, 2000), [0.0, -0.0, np.inf, -np.inf, 65504., 65519.9, 65520., 6.1e-5, 5.96e-8,
                                           2.98e-8, 2.9802322387695312e-08, 2.99e-8, 1e-9],
        np.arange(0, 2049) / 2048., (np.arange(0, 4096) + 0.5) / 4096.
    ])
    with np.errstate(over='ignore'):
        want = v.astype(np.float16)
    got = orc.f64_to_f16(v)
    assert np.array_equal(want.view(np.uint16), got.view(np.uint16))


def _rows(st):
    return st.rows()


@pytest.mark.parametrize('suffix,Pkey', [('', 'P'), ('2', 'P2')])
def test_k1_velo2img_and_gather(golden, suffix, Pkey):
    g = golden('k1')
    pc = g['pc' + suffix]
    H, W = int(g['H']), int(g['W'])
    st = orc.Store(pc.shape[0])
    # no class filter -> rows == gen_semantic_pc output
    m, mask, u, v = orc.kitti_project_sample_filter(st, pc, g[Pkey], g['img'], g['sem'].astype(np.uint8), None, H,
                                                    W, [], want_uv=True)
    ref = g['velo2img' + suffix]                    # (M,6) [x,y,z,i,u,v] of in-frustum points
    assert m == ref.shape[0] == int(mask.sum())
    assert np.array_equal(pc[mask].astype(np.float64), ref[:, :4])
    assert np.array_equal(u[mask], ref[:, 4].astype(np.int64))
    assert np.array_equal(v[mask], ref[:, 5].astype(np.int64))
    rows = _rows(st)
    assert np.array_equal(rows[:, :7], g['sem_rgb' + suffix])
    assert np.array_equal(rows[:, 7], g['sem_sem' + suffix][:, -1])


def test_k1_filter(golden):
    g = golden('k1')
    st = orc.Store(g['pc'].shape[0])
    orc.kitti_project_sample_filter(st, g['pc'], g['P'], g['img'], g['sem'].astype(np.uint8), None, int(g['H']),
                                    int(g['W']), KITTI_FILTERS)
    assert np.array_equal(_rows(st)[:, :8], g['filtered'])


def test_homo_transform(golden):
    g = golden('utils')
    assert np.array_equal(orc.homo_transform(g['ht_T'], g['ht_pts']), g['ht_out'])


def test_project_cams_last_camera_wins(golden):
    g = golden('utils')
    Tc = np.stack([np.linalg.inv(T) for T in g['c6_glob_from_cam']])
    K = np.tile(g['pp_K'].ravel(), (6, 1))
    wh = np.tile(g['pp_wh'], (6, 1))
    ego, uv, cam = orc.nusc_project_cams(g['c6_pc'], g['c6_ego_from_lidar'], g['c6_glob_from_ego'], Tc, K, wh)
    assert np.array_equal(ego, g['c6_pc_in_ego'])
    assert np.array_equal(cam, g['c6_cam_idx'])
    assert np.array_equal(uv, g['c6_uv'])
    assert (cam >= 0).sum() > 100


def test_project_single_cam(golden):
    g = golden('utils')
    eye = np.eye(4)
    ego, uv, cam = orc.nusc_project_cams(g['pp_pc'], eye, eye, eye[None], g['pp_K'].ravel()[None], g['pp_wh'][None])
    m = g['pp_mask']
    assert np.array_equal(cam >= 0, m)
    assert np.array_equal(uv[m], g['pp_uv'][m])


def _bev_case(g, view, px, height_filter, ints, div255, args=None):
    present, future = g['pc_present'], g['pc_future']
    rows = np.concatenate([present, future])
    st = orc.Store.from_rows(rows, intensity_div255=div255)
    assert np.array_equal(st.rows(), rows)          # representable -> lossless SoA
    ego = g['in_ego_traj_present']
    if args is None:
        from pca_amd import host_logic as hl
        rot, dx, dy, zoom = hl.heading_rot_ang(ego), 0., 0., 1.
    else:
        rot, dx, dy, zoom = args
    c, s = np.cos(rot), np.sin(rot)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    prm = orc.make_bev_params([0., 0., 0.], R, dx, dy, zoom * view, px, height_filter, *ints, SEM_IDXS['road'],
                              DYNOBJ, div255)
    return orc.bev(st, present.shape[0], prm, want_cells=True)


def _check_bev(out, g, pre=True):
    px = out['planes'].shape[-1]
    for s, name in enumerate(orc.SETS):
        P = out['planes'][7 * s:7 * s + 7]
        F = out['f16'][7 * s:7 * s + 7]
        if pre:
            assert np.array_equal(P[0], g[f'pre_road_{name}'])
            assert np.array_equal(P[2:5], g[f'pre_rgb_{name}'])
            assert np.array_equal(P[5], g[f'pre_dynamic_{name}'])
            assert np.array_equal(P[6], g[f'pre_elevation_{name}'])
            assert np.array_equal(out['intraw'][s], g[f'pre_intraw_{name}'])      # bincount order kept
            np.testing.assert_allclose(P[1], g[f'pre_intensity_{name}'], rtol=0, atol=1e-14)
        for k, key in ((0, 'road'), (5, 'dynamic'), (6, 'elevation')):
            assert np.array_equal(F[k].view(np.uint16), g[f'bev_{key}_{name}'].view(np.uint16)), key
        assert np.array_equal(F[2:5].view(np.uint16), g[f'bev_rgb_{name}'].view(np.uint16))
        # exp() is libm here and numpy's SIMD exp in the reference: allow one f16 ulp on the intensity plane
        a = F[1].view(np.uint16).astype(np.int32)
        b = g[f'bev_intensity_{name}'].view(np.uint16).astype(np.int32)
        assert np.abs(a - b).max() <= 1
        assert (a != b).mean() < 1e-3
        assert F.shape == (7, px, px)


def test_bev_a(golden):
    g = golden('bev_a')
    out = _bev_case(g, 20, 32, None, (20., 20., 0.5), False)
    _check_bev(out, g)
    # cell ids against the reference's floor()-ed grid coordinates of the kept 'present' rows
    rows = g['pre_grid_rows_present']
    n_p = g['pc_present'].shape[0]
    cells = out['cells'][:n_p]
    dyn = g['pc_present'][:, 9] == 1
    kept = cells >= 0
    # reference keeps dynamic points through preprocess; oracle drops them at binning time
    ref_cells = (32 - 1 - rows[:, 1].astype(int)) * 32 + rows[:, 0].astype(int)
    assert np.array_equal(cells[kept], ref_cells[rows[:, 9] != 1])
    assert kept.sum() + (dyn & (ref_cells.size > 0)).sum() >= kept.sum()


def test_bev_b_nuscenes_params_height_filter_explicit_args(golden):
    g = golden('bev_b')
    out = _bev_case(g, 51.2, 64, 3., (1., 30., 0.12), True, args=tuple(g['args']))
    _check_bev(out, g)


def test_bev_c_256(golden):
    g = golden('bev_c')
    out = _bev_case(g, 80, 256, None, (20., 20., 0.5), False)
    _check_bev(out, g)


def test_bev_d_empty_future_single_pose(golden):
    g = golden('bev_d')
    out = _bev_case(g, 20, 16, None, (20., 20., 0.5), False)
    _check_bev(out, g, pre=False)


def test_rgb_bev_medians(golden):
    g = golden('utils')
    rows = g['rg_pc'].copy()            # columns 0,1 already hold grid coordinates
    px, view = 16, 20.
    # invert pos2grid: put every point at its cell centre
    rows[:, 0] = (rows[:, 0] + 0.5 - 0.5 * px) * view / px
    rows[:, 1] = (rows[:, 1] + 0.5 - 0.5 * px) * view / px
    rows[:, 9] = 0                      # RGBBEVGenerator has no dynamic partition
    st = orc.Store.from_rows(rows)
    prm = orc.make_bev_params([0, 0, 0], np.eye(3), 0, 0, view, px, None, 1, 1, 0.5, 0, DYNOBJ, False, rgb_fill=7.)
    out = orc.bev(st, st.n, prm)
    assert np.array_equal(out['planes'][2:5] * 255., g['rg_out'])


def test_numpy_shape_matches_c_oracle():
    """oracle/numpy_shape.py (the numpy-shaped CPU baseline bench.py times) against the C oracle: integrate rows equal,
    BEV planes equal in both reduction forms (vectorised / the reference's Python loops)."""
    from oracle import numpy_shape as ns
    from oracle import oracle as orc
    rng = np.random.default_rng(12)
    H, W, n = 48, 80, 4000
    cam_to_velo = np.array([[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418],
                            [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
                            [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824], [0, 0, 0, 1]])
    P = np.array([[40., 0, W / 2, 0], [0, 40., H / 2, 0], [0, 0, 1, 0]]) @ np.linalg.inv(cam_to_velo)
    filters = [10, 11, 12, 16, 18, 255]
    frames, st, sizes = [], orc.Store(4 * n), []
    T = np.eye(4)
    T[:3, :3] = [[np.cos(0.01), -np.sin(0.01), 0], [np.sin(0.01), np.cos(0.01), 0], [0, 0, 1]]
    T[0, 3] = -1.0
    for k in range(4):
        pc = np.stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n), rng.uniform(-2, 3, n), rng.uniform(0, 1, n)],
                      1).astype(np.float32)
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        sem = rng.integers(0, 19, (H, W)).astype(np.uint8)
        if k:
            ns.retransform(frames, T)
            orc.retransform(st, T)
        frames.append(ns.integrate_frame(pc.astype(np.float64), P, img, sem, filters))
        sizes.append(orc.kitti_project_sample_filter(st, pc, P, img, sem, None, H, W, filters))
    assert np.array_equal(np.concatenate(frames), st.rows())
    R = np.array([[np.cos(0.4), -np.sin(0.4), 0], [np.sin(0.4), np.cos(0.4), 0], [0, 0, 1.]])
    origin = np.array([0.5, -0.25, 0.1])
    ref = orc.bev(st, sizes[0] + sizes[1], orc.make_bev_params(origin, R, 0., 0., 30., 32, None, 20., 20., 0.5, 0,
                                                               [13, 14, 15, 17], False))
    for loops in (False, True):
        got = ns.bev([f.copy() for f in frames], 2, origin, R, 30., 32, 0, [13, 14, 15, 17], (20., 20., 0.5), loops)
        d = np.abs(got.view(np.uint16).astype(int) - ref['f16'].view(np.uint16).astype(int))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3, loops          # intensity plane: summation order, 1 fp16 ulp
        for s in range(3):
            for k in (0, 2, 3, 4, 5, 6):
                assert np.array_equal(got[7 * s + k].view(np.uint16), ref['f16'][7 * s + k].view(np.uint16))
