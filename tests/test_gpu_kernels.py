"""GPU parity, kernel level: every HIP kernel (through the C ABI) against the CPU oracle on the same seeded
inputs and against the committed golden vectors.  Integers / masks / coordinates: bit-exact.
Floating-point planes: 1e-12 absolute against the oracle (contract: 1e-5 against the reference)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KITTI_FILTERS = [10, 11, 12, 16, 18, 255]
NUSC_FILTERS = [10, 11, 12, 16, 18]
SEM_IDXS = {'road': 0, 'car': 13, 'truck': 14, 'bus': 15, 'motorcycle': 17}
DYNOBJ = [13, 14, 15, 17]

CAM_TO_VELO = np.array([[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418],
                        [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
                        [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824], [0, 0, 0, 1]])
P_RECT = np.array([[552.554261, 0, 682.049453, 0], [0, 552.554261, 238.769549, 0], [0, 0, 1, 0]])
P_KITTI = P_RECT @ np.linalg.inv(CAM_TO_VELO)


@pytest.fixture(scope='module')
def T():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    return torch


@pytest.fixture(scope='module')
def orc():
    from oracle import oracle
    return oracle


def dev_store(**kw):
    from pca_amd.device_store import DeviceStore
    return DeviceStore(**kw)


def cu(T, a, dtype=None):
    t = T.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def kitti_frame(rng, n, H, W, lim=60.0):
    pc = np.stack([rng.uniform(-lim, lim, n), rng.uniform(-lim, lim, n), rng.uniform(-2, 3, n),
                   rng.uniform(0, 1, n)], 1).astype(np.float32)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    sem = rng.integers(0, 19, (H, W)).astype(np.uint8)
    sem[rng.random((H, W)) < 0.01] = 255
    return pc, img, sem


# ------------------------------------------------------------------------------------------- K1
@pytest.mark.parametrize('suffix,Pkey', [('', 'P'), ('2', 'P2')])
@pytest.mark.parametrize('filters', [[], KITTI_FILTERS])
def test_k1_golden(T, orc, golden, suffix, Pkey, filters):
    g = golden('k1')
    pc, img, sem = g['pc' + suffix], g['img'], g['sem'].astype(np.uint8)
    H, W = int(g['H']), int(g['W'])
    st = dev_store(capacity=8192, max_frames=4)
    st.append_kitti([dict(pts=cu(T, pc), rgb=cu(T, img), sem=cu(T, sem))], g[Pkey], H, W, filters)
    rows = st.rows(0)
    ost = orc.Store(pc.shape[0])
    orc.kitti_project_sample_filter(ost, pc, g[Pkey], img, sem, None, H, W, filters)
    assert np.array_equal(rows, ost.rows())
    if not filters:
        assert np.array_equal(rows[:, :7], g['sem_rgb' + suffix])
        assert np.array_equal(rows[:, 7], g['sem_sem' + suffix][:, -1])
    elif suffix == '':
        assert np.array_equal(rows[:, :8], g['filtered'])
    st.check_status()


def test_k1_full_size_batch_and_ragged(T, orc):
    """KITTI-shape frames (120k points, 376x1408 images), several per launch incl. an empty and a
    one-point frame and sizes around the tile size; stable order, segment offsets, bit-exact rows."""
    rng = np.random.default_rng(7)
    H, W = 376, 1408
    sizes = [120000, 0, 1, 1023, 1024, 1025, 120000, 4097]
    frames, host = [], []
    for n in sizes:
        pc, img, sem = kitti_frame(rng, n, H, W)
        host.append((pc, img, sem))
        frames.append(dict(pts=cu(T, pc), rgb=cu(T, img), sem=cu(T, sem)))
    st = dev_store(capacity=400000, max_frames=16)
    st.append_kitti(frames[:3], P_KITTI, H, W, KITTI_FILTERS)
    st.append_kitti(frames[3:], P_KITTI, H, W, KITTI_FILTERS)
    got = st.frame_rows()
    st.check_status()
    assert len(got) == len(sizes)
    kept = 0
    for (pc, img, sem), rows in zip(host, got):
        ost = orc.Store(max(pc.shape[0], 1))
        orc.kitti_project_sample_filter(ost, pc, P_KITTI, img, sem, None, H, W, KITTI_FILTERS)
        assert np.array_equal(rows, ost.rows())
        kept += ost.n
    assert 0.15 < kept / sum(sizes) < 0.30          # ~0.22 of the input survives


def test_k1_use_gt_sem(T, orc, golden):
    g = golden('kitti_gtsem')
    pc = g['pc_0']
    sem_gt = g['sem_gt_0'][:, -1].astype(np.uint8)
    st = dev_store(capacity=4096, max_frames=4)
    st.append_kitti([dict(pts=cu(T, pc), sem_gt=cu(T, sem_gt))], g['P'], 1, 1, KITTI_FILTERS)
    ost = orc.Store(pc.shape[0])
    orc.kitti_project_sample_filter(ost, pc, g['P'], None, None, sem_gt, 1, 1, KITTI_FILTERS)
    assert np.array_equal(st.rows(0), ost.rows())
    assert st.rows(0).shape[0] == int(g['sizes'][0])


def test_k1_capacity_overflow_is_reported(T):
    rng = np.random.default_rng(3)
    pc, img, sem = kitti_frame(rng, 5000, 64, 96, 10.0)
    pc[:, 2] = 0.0
    st = dev_store(capacity=16, max_frames=4)
    st.ub_tail = -10**9           # defeat the host-side planner on purpose
    sem_gt = np.zeros(5000, np.uint8)
    st.append_kitti([dict(pts=cu(T, pc), sem_gt=cu(T, sem_gt))], P_KITTI, 1, 1, [])
    with pytest.raises(RuntimeError, match='overflow'):
        st.check_status()


# ------------------------------------------------------------------------------------------- K2
@pytest.mark.parametrize('n_T', [1, 3, 16, 21])
def test_k2_retransform_chain(T, orc, n_T):
    rng = np.random.default_rng(11)
    rows = np.zeros((100003, 10))
    rows[:, :3] = rng.uniform(-100, 100, (100003, 3))
    parts = [rows[:1], rows[1:2], rows[2:50001], rows[50001:]]
    st = dev_store(capacity=200000, max_frames=8)
    st.load_rows(parts)
    st.evict(1)                                     # odd start offset: scalar head path
    Ts = []
    for k in range(n_T):
        a = rng.uniform(-0.05, 0.05)
        c, s = np.cos(a), np.sin(a)
        Tm = np.eye(4)
        Tm[:3, :3] = np.array([[c, -s, 0.001], [s, c, -0.002], [0.0005, 0.001, 1.0]])
        Tm[:3, 3] = rng.uniform(-2, 2, 3)
        Ts.append(Tm)
    st.retransform(np.stack(Ts))
    ost = orc.Store.from_rows(rows[1:])
    for Tm in Ts:
        orc.retransform(ost, Tm)
    assert np.array_equal(st.rows()[:, :3], ost.rows()[:, :3])


# ------------------------------------------------------------------------------------------ K1n / K3
def fake_semseg(img):
    a = img.astype(np.int64)
    return ((a[..., 0] + 2 * a[..., 1] + 3 * a[..., 2]) % 19).astype(np.uint8)


def test_k1n_and_k3_golden(T, orc, golden):
    g = golden('nusc_oracle')
    st = dev_store(capacity=1 << 15, max_frames=16, intensity_div255=True)
    ost = orc.Store(1 << 15, intensity_div255=True)
    T0 = np.linalg.inv(g['T_0'])
    offs = [0]
    for k in range(int(g['F'])):
        Tw = T0 @ g[f'T_{k}']
        imgs = g[f'imgs_{k}']
        sems = np.stack([fake_semseg(im) for im in imgs])
        st.append_nusc(cu(T, g[f'pc_{k}']), cu(T, g[f'cam_idx_{k}']), cu(T, imgs), cu(T, sems), Tw, NUSC_FILTERS)
        offs.append(offs[-1] + orc.nusc_sample_filter_transform(ost, g[f'pc_{k}'], g[f'cam_idx_{k}'], imgs, sems, Tw,
                                                                NUSC_FILTERS))
    assert np.array_equal(st.offsets(), np.array(offs))
    assert np.array_equal(st.rows(), ost.rows())
    pairs = [(0, 0), (2, 0), (3, 3), (6, 2), (6, -1)]
    st.mark_dynamic(pairs)
    for f, i in pairs:
        orc.mark_dynamic(ost, offs[f], offs[f + 1], i)
    assert np.array_equal(st.rows(), ost.rows())
    assert st.rows()[:, 9].sum() > 0
    st.check_status()


def test_k1n_uv_outside_image_raises(T, golden):
    g = golden('nusc_oracle')
    pc = g['pc_0'].copy()
    pc[5, 4] = 0.5                                   # u <= 1 for a point assigned to a camera
    cam = g['cam_idx_0'].copy()
    cam[5] = 2
    imgs = g['imgs_0']
    sems = np.stack([fake_semseg(im) for im in imgs])
    st = dev_store(capacity=4096, max_frames=4, intensity_div255=True)
    st.append_nusc(cu(T, pc), cu(T, cam), cu(T, imgs), cu(T, sems), np.eye(4), NUSC_FILTERS)
    with pytest.raises(AssertionError):
        st.check_status()


def test_k0n_project_cams_golden(T, orc, golden):
    from datasets.nuscenes_utils import project_to_cameras
    g = golden('utils')
    K = [g['pp_K']] * 6
    wh = [g['pp_wh']] * 6
    ego, uv, cam = project_to_cameras(g['c6_pc'], g['c6_ego_from_lidar'], g['c6_glob_from_ego'],
                                      list(g['c6_glob_from_cam']), K, wh)
    assert np.array_equal(ego, g['c6_pc_in_ego'])
    assert np.array_equal(cam, g['c6_cam_idx'])
    assert np.array_equal(uv, g['c6_uv'])


# ------------------------------------------------------------------------------------------- BEV
def run_dev_bev(T, rows_p, rows_f, view, px, hf, ints, div255, rot, dx=0., dy=0., origin=(0., 0., 0.)):
    from pca_amd import host_logic as hl
    from pca_amd.device_store import make_bev_params
    st = dev_store(capacity=max(rows_p.shape[0] + rows_f.shape[0], 1), max_frames=4, intensity_div255=div255)
    i64 = st.load_rows([rows_p, rows_f])
    prm = make_bev_params(origin, hl.rotation_matrix_3d(rot), dx, dy, view, px, hf, *ints, 0, DYNOBJ, div255)
    p16, p64 = st.bev(1, prm, want_f64=True, intensity64=i64)
    st.check_status()
    return p16.cpu().numpy(), p64.cpu().numpy(), i64 is not None


def run_orc_bev(orc, rows_p, rows_f, view, px, hf, ints, div255, rot, dx=0., dy=0., origin=(0., 0., 0.)):
    from pca_amd import host_logic as hl
    rows = np.concatenate([rows_p, rows_f])
    ost = orc.Store.from_rows(rows, intensity_div255=div255)
    prm = orc.make_bev_params(origin, hl.rotation_matrix_3d(rot), dx, dy, view, px, hf, *ints, 0, DYNOBJ, div255)
    return orc.bev(ost, rows_p.shape[0], prm)


def assert_planes_match(p16, p64, ref, name=''):
    P, F = ref['planes'], ref['f16']
    for s in range(3):
        for k, key in enumerate(('road', 'intensity', 'r', 'g', 'b', 'dynamic', 'elevation')):
            a, b = p64[7 * s + k], P[7 * s + k]
            if key == 'intensity':
                np.testing.assert_allclose(a, b, rtol=0, atol=1e-12, err_msg=f'{name} {key} set {s}')
                d = np.abs(p16[7 * s + k].view(np.uint16).astype(int) - F[7 * s + k].view(np.uint16).astype(int))
                assert d.max() <= 1 and (d != 0).mean() < 1e-3
            else:
                assert np.array_equal(a, b), f'{name} {key} set {s}'
                assert np.array_equal(p16[7 * s + k].view(np.uint16), F[7 * s + k].view(np.uint16))


BEV_CASES = {
    'bev_a': (20, 32, None, (20., 20., 0.5), False),
    'bev_b': (51.2, 64, 3., (1., 30., 0.12), True),
    'bev_c': (80, 256, None, (20., 20., 0.5), False),
    'bev_d': (20, 16, None, (20., 20., 0.5), False),
}


@pytest.mark.parametrize('case', sorted(BEV_CASES))
def test_bev_golden_and_oracle(T, orc, golden, case):
    from pca_amd import host_logic as hl
    g = golden(case)
    view, px, hf, ints, div255 = BEV_CASES[case]
    if case == 'bev_b':
        rot, dx, dy, zoom = g['args']
    else:
        rot, dx, dy, zoom = hl.heading_rot_ang(g['in_ego_traj_present']), 0., 0., 1.
    p16, p64, used_i64 = run_dev_bev(T, g['pc_present'], g['pc_future'], zoom * view, px, hf, ints, div255, rot, dx, dy)
    assert not used_i64                              # fixture intensities are representable
    ref = run_orc_bev(orc, g['pc_present'], g['pc_future'], zoom * view, px, hf, ints, div255, rot, dx, dy)
    assert_planes_match(p16, p64, ref, case)
    # and straight against the reference's own fp16 outputs
    for s, name in enumerate(('present', 'future', 'full')):
        for k, key in ((0, 'road'), (5, 'dynamic'), (6, 'elevation')):
            assert np.array_equal(p16[7 * s + k].view(np.uint16), g[f'bev_{key}_{name}'].view(np.uint16))
        assert np.array_equal(p16[7 * s + 2:7 * s + 5].view(np.uint16), g[f'bev_rgb_{name}'].view(np.uint16))
        d = np.abs(p16[7 * s + 1].view(np.uint16).astype(int) - g[f'bev_intensity_{name}'].view(np.uint16).astype(int))
        assert d.max() <= 1
        if f'pre_road_{name}' in g:                  # pre-cast f64 planes: the 1e-5 contract
            for k, key in ((0, 'road'), (1, 'intensity'), (5, 'dynamic'), (6, 'elevation')):
                np.testing.assert_allclose(p64[7 * s + k], g[f'pre_{key}_{name}'], rtol=0, atol=1e-5)
            np.testing.assert_allclose(p64[7 * s + 2:7 * s + 5], g[f'pre_rgb_{name}'], rtol=0, atol=1e-5)


def test_bev_f64_intensity_override(T, orc):
    """Column 3 not f32-representable -> the f64 side channel is used and matches the oracle."""
    rng = np.random.default_rng(5)
    rows = np.zeros((5000, 10))
    rows[:, :2] = rng.uniform(-9, 9, (5000, 2))
    rows[:, 2] = rng.uniform(-1, 2, 5000)
    rows[:, 3] = rng.uniform(0, 1, 5000)            # genuine f64
    rows[:, 4:7] = rng.integers(0, 256, (5000, 3))
    p16, p64, used = run_dev_bev(T, rows[:3000], rows[3000:], 20, 32, None, (20., 20., 0.5), False, 0.3)
    assert used
    ost = orc.Store.from_rows(rows)
    from pca_amd import host_logic as hl
    prm = orc.make_bev_params([0, 0, 0], hl.rotation_matrix_3d(0.3), 0, 0, 20, 32, None, 20., 20., 0.5, 0, DYNOBJ, False)
    ref = orc.bev(ost, 3000, prm, intensity64=rows[:, 3])
    assert_planes_match(p16, p64, ref)


def test_bev_full_size_window(T, orc):
    """Config-2-sized window (5.3 M stored points, 256x256, view 80): oracle parity + conservation."""
    rng = np.random.default_rng(9)
    n = 5_300_000
    rows = np.zeros((n, 10))
    rows[:, 0] = rng.uniform(-160, 60, n)           # frames strung along a 200 m path
    rows[:, 1] = rng.uniform(-60, 60, n)
    rows[:, 2] = rng.uniform(-2, 3, n)
    rows[:, 3] = rng.uniform(0, 1, n).astype(np.float32)
    rows[:, 4:7] = rng.integers(0, 256, (n, 3))
    rows[:, 7] = rng.choice([0, 0, 1, 2, 8, 9, 13, 14, 15, 17], n)
    rows[:, 9] = rng.random(n) < 0.02
    # hot cells: 60k points inside one cell, 20k with identical colour
    rows[:60000, 0:2] = rng.uniform(0.01, 0.30, (60000, 2))
    rows[:20000, 4:7] = 77
    split = 2_600_000
    origin = (-50., 1.5, 0.25)
    p16, p64, _ = run_dev_bev(T, rows[:split], rows[split:], 80, 256, None, (20., 20., 0.5), False, 1.234, 0., 0., origin)
    ref = run_orc_bev(orc, rows[:split], rows[split:], 80, 256, None, (20., 20., 0.5), False, 1.234, 0., 0., origin)
    assert_planes_match(p16, p64, ref, 'full-size')
    assert (p64[6] != 0).sum() > 30000               # most of the grid is observed


def _stress_config4(T, F):
    """BASELINE config 4 (1 M points per frame, all kept, 512x512 BEV, view 160 m) with F frames: size-independent
    properties checked against plain torch reductions on the device, accumulated frame by frame."""
    from pca_amd.device_store import make_bev_params
    n, px, view = 1_000_000, 512, 160.0
    st = dev_store(capacity=F * n, max_frames=F + 1)
    g = T.Generator(device='cuda').manual_seed(4)
    classes = T.tensor([0, 1, 2, 8, 9, 13, 14], device='cuda', dtype=T.uint8)       # none is filtered
    Tm = np.eye(4)
    Tm[0, 3] = -0.5 if F <= 100 else -0.03125                 # exact in binary: retransforms stay exact
    for k in range(F):
        pts = T.empty((n, 4), device='cuda', dtype=T.float32)
        pts[:, :2] = (T.rand((n, 2), device='cuda', generator=g) * 160 - 80).float()
        pts[:, 2] = (T.rand(n, device='cuda', generator=g) * 5 - 2).float()
        pts[:, 3] = T.rand(n, device='cuda', generator=g).float()
        sem_gt = classes[T.randint(0, 7, (n, ), device='cuda', generator=g)]
        if k:
            st.retransform(Tm)                                # K2 over everything stored so far, every frame
        st.append_kitti([dict(pts=pts.contiguous(), sem_gt=sem_gt)], P_KITTI, 1, 1, KITTI_FILTERS)
    sizes = st.sizes()
    assert sizes.tolist() == [n] * F                          # every point kept, segment offsets exact
    split = F // 2
    prm = make_bev_params((0.25, -0.5, 0.0), np.eye(3), 0., 0., view, px, None, 20., 20., 0.5, 0, DYNOBJ, False)
    p16, p64 = st.bev(split, prm, want_f64=True)
    st.check_status()
    # reference reductions with torch on the device (R = identity: the fma chain is exact), one frame at a time
    ncell = px * px
    cnt = T.zeros((2, ncell), device='cuda', dtype=T.float64)
    road, dyn = T.zeros_like(cnt), T.zeros_like(cnt)
    zmin = T.full((2, ncell), float('inf'), device='cuda', dtype=T.float64)
    for k in range(F):
        s = 0 if k < split else 1
        lo, hi = k * n, (k + 1) * n
        x = st.x[lo:hi] - 0.25
        y = st.y[lo:hi] + 0.5
        m = (x > -80) & (x < 80) & (y > -80) & (y < 80)
        i = T.floor(x / view * px + 0.5 * px).long().clamp(0, px - 1)
        j = T.floor(y / view * px + 0.5 * px).long().clamp(0, px - 1)
        c = ((px - 1 - j) * px + i)[m]
        sem = ((st.rgbs[lo:hi] >> 24) & 255)[m]
        cnt[s] += T.bincount(c, minlength=ncell)
        road[s] += T.bincount(c[sem == 0], minlength=ncell)
        dyn[s] += T.bincount(c[(sem == 13) | (sem == 14)], minlength=ncell)
        zmin[s] = zmin[s].scatter_reduce(0, c, st.z[lo:hi][m], 'amin')
    assert float(cnt.sum()) > 0.8 * F * n
    for s in range(3):
        cn = cnt[s] if s < 2 else cnt[0] + cnt[1]
        rd = road[s] if s < 2 else road[0] + road[1]
        dy = dyn[s] if s < 2 else dyn[0] + dyn[1]
        zm = zmin[s] if s < 2 else T.minimum(zmin[0], zmin[1])
        assert T.equal(p64[7 * s + 0].flatten(), (rd + 1) / ((rd + 1) + ((cn - rd) + 1)))
        assert T.equal(p64[7 * s + 5].flatten(), (dy + 1) / ((dy + 1) + ((cn - dy) + 1)))
        assert T.equal(p64[7 * s + 6].flatten(), T.where(cn > 0, zm, T.zeros_like(zm)))
    # rgb == 0 everywhere (GT-semantics mode): medians are exactly the fill colour
    assert float(p64[2:5].abs().max()) == 0.0
    assert T.equal(p16.double()[0], p64[0].half().double())
    return st


def test_stress_config4_scaled_properties(T):
    _stress_config4(T, 48)


def test_stress_config4_full_size_one_billion_points(T):
    """The whole of BASELINE config 4: 1000 frames x 1 M points = 1e9 stored points (37 GB of the 288 GB), re-
    transformed on every integrate, one 512x512 BEV over the full window (every tile is a heavy tile: ~244 k records)."""
    free, _ = T.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip('needs ~120 GB of free HBM')
    st = _stress_config4(T, 1000)
    assert int(st.offsets()[-1]) == 1_000_000_000
    del st
    T.cuda.empty_cache()


@pytest.mark.parametrize('px,view', [(7, 10.0), (30, 33.0), (100, 51.2), (512, 160.0), (1024, 200.0)])
def test_bev_grid_sizes_and_partial_tiles(T, orc, px, view):
    """Grids that are not a multiple of the 8x8 tile, smaller than one tile, and the largest supported one."""
    rng = np.random.default_rng(px)
    n = 40000
    rows = np.zeros((n, 10))
    rows[:, :2] = rng.uniform(-0.55 * view, 0.55 * view, (n, 2))
    rows[:, 2] = rng.uniform(-2, 4, n)
    rows[:, 3] = rng.uniform(0, 1, n).astype(np.float32)
    rows[:, 4:7] = rng.integers(0, 256, (n, 3))
    rows[:, 7] = rng.choice([0, 1, 13, 17], n)
    rows[:, 9] = rng.random(n) < 0.05
    p16, p64, _ = run_dev_bev(T, rows[:15000], rows[15000:], view, px, 3.0, (20., 20., 0.5), False, 0.9, 0.3, -0.2)
    ref = run_orc_bev(orc, rows[:15000], rows[15000:], view, px, 3.0, (20., 20., 0.5), False, 0.9, 0.3, -0.2)
    assert_planes_match(p16, p64, ref, f'px={px}')


def test_bev_empty_and_out_of_view_windows(T, orc):
    empty = np.zeros((0, 10))
    far = np.zeros((100, 10))
    far[:, 0] = 1e6
    for rows_p, rows_f in ((empty, empty), (far, empty), (empty, far)):
        p16, p64, _ = run_dev_bev(T, rows_p, rows_f, 20, 32, None, (20., 20., 0.5), False, 0.0)
        ref = run_orc_bev(orc, rows_p, rows_f, 20, 32, None, (20., 20., 0.5), False, 0.0)
        assert_planes_match(p16, p64, ref)
        assert np.all(p64[0] == 0.5) and np.all(p64[6] == 0.0)


def test_bev_dense_tile_goes_through_batches(T, orc):
    """One 8x8 tile holding more colour records than fit its LDS buffer, spread over many cells (batch path),
    plus cells above 64 values (histogram path) next to small ones (radix-select path)."""
    rng = np.random.default_rng(21)
    n = 30000
    rows = np.zeros((n, 10))
    # view 32 m, 64 px -> cell 0.5 m, tile 4 m: put 20k points into the tile [0,4) x [0,4)
    rows[:20000, :2] = rng.uniform(0.01, 3.99, (20000, 2))
    rows[20000:, :2] = rng.uniform(-15.9, 15.9, (10000, 2))
    rows[:, 2] = rng.uniform(-1, 2, n)
    rows[:, 3] = rng.integers(0, 256, n) / 255.
    rows[:, 4:7] = rng.integers(0, 256, (n, 3))
    rows[:, 7] = rng.choice([0, 2, 13], n)
    perm = rng.permutation(n)
    rows = rows[perm]
    p16, p64, _ = run_dev_bev(T, rows[:12000], rows[12000:], 32, 64, None, (1., 30., 0.12), True, 0.0)
    ref = run_orc_bev(orc, rows[:12000], rows[12000:], 32, 64, None, (1., 30., 0.12), True, 0.0)
    assert_planes_match(p16, p64, ref, 'dense tile')


def _skewed_rows(rng, n, sigma, colour_spread):
    """Points clustered around a driven path (density falling off away from it), narrow colour distribution."""
    rows = np.zeros((n, 10))
    rows[:, 0] = rng.uniform(-15.9, 15.9, n)
    rows[:, 1] = rng.normal(0, sigma, n)
    rows[:, 2] = rng.uniform(-1, 2, n)
    rows[:, 3] = rng.integers(0, 256, n) / 255.
    base = rng.integers(0, 256, 3)
    rows[:, 4:7] = np.clip(base + rng.integers(-colour_spread, colour_spread + 1, (n, 3)), 0, 255)
    rows[:, 7] = rng.choice([0, 1, 2, 13], n, p=[0.6, 0.2, 0.15, 0.05])
    rows[:, 9] = rng.random(n) < 0.02
    return rows


@pytest.mark.parametrize('n,sigma,spread', [(60000, 1.5, 8), (400000, 0.8, 3), (150000, 4.0, 0)])
def test_bev_ring_like_skew_heavy_and_light_tiles(T, orc, n, sigma, spread):
    """Real accumulations are skewed: cells next to the path hold hundreds to thousands of points.  Covers the
    per-wave histogram path (cells > 64 values in tiles that fit LDS), the heavy-tile kernel (tiles beyond the LDS
    colour buffer), identical colours (spread 0) and the mix of both tile kinds in one raster."""
    rng = np.random.default_rng(n)
    rows = _skewed_rows(rng, n, sigma, spread)
    cut = int(0.4 * n)
    p16, p64, _ = run_dev_bev(T, rows[:cut], rows[cut:], 32, 64, None, (1., 30., 0.12), True, 0.2)
    ref = run_orc_bev(orc, rows[:cut], rows[cut:], 32, 64, None, (1., 30., 0.12), True, 0.2)
    assert_planes_match(p16, p64, ref, f'skew n={n}')


def test_bev_heavy_tiles_without_and_with_the_heavy_kernel(T, orc):
    """The heavy kernel is only launched while heavy tiles were seen in one of the last 64 calls.  After 70 sparse
    rasters a dense one finds no heavy launch behind it: the light kernel's last workgroup works the queue off (exact,
    slow) and tells the host; the call after that goes through the heavy kernel.  Both equal the oracle."""
    rng = np.random.default_rng(5)
    sparse = _skewed_rows(rng, 2000, 6.0, 8)
    for _ in range(70):
        run_dev_bev(T, sparse[:800], sparse[800:], 32, 64, None, (1., 30., 0.12), True, 0.2)
    rows = _skewed_rows(rng, 200000, 0.9, 4)
    cut = 80000
    ref = run_orc_bev(orc, rows[:cut], rows[cut:], 32, 64, None, (1., 30., 0.12), True, 0.2)
    for which in ('drained by the light kernel', 'heavy kernel'):
        p16, p64, _ = run_dev_bev(T, rows[:cut], rows[cut:], 32, 64, None, (1., 30., 0.12), True, 0.2)
        assert_planes_match(p16, p64, ref, which)


def test_bev_cell_beyond_16bit_counters(T, orc):
    """One (cell, set) with more than 65 535 values: the heavy kernel's packed 16-bit histograms overflow there and the
    32-bit whole-workgroup path must take over for that cell only."""
    rng = np.random.default_rng(77)
    n = 180000
    rows = _skewed_rows(rng, n, 3.0, 5)
    rows[:70000, 0] = rng.uniform(2.01, 2.49, 70000)          # cell [2.0, 2.5) x [1.0, 1.5): 70k present points
    rows[:70000, 1] = rng.uniform(1.01, 1.49, 70000)
    rows[70000:100000, 0] = rng.uniform(2.51, 2.99, 30000)    # neighbour cell: 30k
    rows[70000:100000, 1] = rng.uniform(1.01, 1.49, 30000)
    rows[100000:140000, 0] = rng.uniform(2.01, 2.49, 40000)   # and 40k future points in the first cell
    rows[100000:140000, 1] = rng.uniform(1.01, 1.49, 40000)
    p16, p64, used_i64 = run_dev_bev(T, rows[:100000], rows[100000:], 32, 64, None, (1., 30., 0.12), True, 0.0)
    assert not used_i64
    ref = run_orc_bev(orc, rows[:100000], rows[100000:], 32, 64, None, (1., 30., 0.12), True, 0.0)
    assert_planes_match(p16, p64, ref, '16-bit overflow')
    # the same raster through the f64-intensity variant of the kernels (column 3 not f32-representable)
    rows[:, 3] = rng.uniform(0, 1, n)
    p16, p64, used_i64 = run_dev_bev(T, rows[:100000], rows[100000:], 32, 64, None, (20., 20., 0.5), False, 0.0)
    assert used_i64
    from pca_amd import host_logic as hl
    prm = orc.make_bev_params([0, 0, 0], hl.rotation_matrix_3d(0.0), 0, 0, 32, 64, None, 20., 20., 0.5, 0, DYNOBJ, False)
    ref = orc.bev(orc.Store.from_rows(rows), 100000, prm, intensity64=rows[:, 3])
    assert_planes_match(p16, p64, ref, '16-bit overflow, f64 intensity')


def test_bev_extra_reducers_max_mean(T):
    """Opt-in extra reducers (no reference counterpart): max z / mean z over a cell's static points, mean raw
    intensity over its road points; checked against numpy on an axis-aligned raster (R = I, no shift), light and
    heavy tiles."""
    from pca_amd import _lib, host_logic as hl
    from pca_amd.device_store import make_bev_params
    rng = np.random.default_rng(31)
    n, view, px = 120000, 32.0, 64
    rows = _skewed_rows(rng, n, 2.0, 6)
    rows[:, 3] = rng.uniform(0, 1, n).astype(np.float32)
    cut = 50000
    st = dev_store(capacity=n, max_frames=4)
    assert st.load_rows([rows[:cut], rows[cut:]]) is None
    prm = make_bev_params((0., 0., 0.), hl.rotation_matrix_3d(0.0), 0., 0., view, px, None, 20., 20., 0.5, 0, DYNOBJ, False)
    extra = T.zeros((3, len(_lib.BEV_EXTRA_PLANES), px, px), dtype=T.float64, device='cuda')
    p16, p64 = st.bev(1, prm, want_f64=True, extra=extra)
    st.check_status()
    ex = extra.cpu().numpy()
    elev = p64.cpu().numpy()
    x, y, z, inten, sem, dyn = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 7], rows[:, 9]
    keep = (x > -view / 2) & (x < view / 2) & (y > -view / 2) & (y < view / 2) & (dyn != 1)
    i = np.floor(x / view * px + 0.5 * px).astype(int)
    j = np.floor(y / view * px + 0.5 * px).astype(int)
    cell = (px - 1 - j) * px + i
    is_future = np.arange(n) >= cut
    for s, sel in enumerate((~is_future, is_future, np.ones(n, bool))):
        m = keep & sel
        cnt = np.bincount(cell[m], minlength=px * px).astype(float)
        zmax = np.full(px * px, -np.inf)
        np.maximum.at(zmax, cell[m], z[m])
        zmax[cnt == 0] = 0.0
        zsum = np.bincount(cell[m], weights=z[m], minlength=px * px)
        zmean = np.where(cnt > 0, zsum / np.maximum(cnt, 1), 0.0)
        mr = m & (sem == 0)
        cr = np.bincount(cell[mr], minlength=px * px).astype(float)
        isum = np.bincount(cell[mr], weights=inten[mr], minlength=px * px)
        imean = np.where(cr > 0, isum / np.maximum(cr, 1), 0.0)
        assert np.array_equal(ex[s, 0].ravel(), zmax), f'max z set {s}'
        np.testing.assert_allclose(ex[s, 1].ravel(), zmean, rtol=0, atol=1e-11)
        np.testing.assert_allclose(ex[s, 2].ravel(), imean, rtol=0, atol=1e-11)
        # the ordinary planes are untouched by the option, and max >= mean >= min where observed
        obs = cnt > 0
        assert np.all(ex[s, 0].ravel()[obs] >= ex[s, 1].ravel()[obs] - 1e-12)
        assert np.all(ex[s, 1].ravel()[obs] >= elev[7 * s + 6].ravel()[obs] - 1e-12)
    p16b, p64b = st.bev(1, prm, want_f64=True)
    assert T.equal(p64, p64b) and T.equal(p16, p16b)


@pytest.mark.parametrize('n_frames,size', [(1, 0.5), (7, 0.2), (40, 1.0)])
def test_voxel_dedup_first_point_per_voxel_stays(T, n_frames, size):
    """Opt-in voxel de-duplication (no reference counterpart): numpy model = first occurrence per voxel in store
    order; frames stay contiguous and ordered; a second call changes nothing; BEV still runs on the result."""
    rng = np.random.default_rng(n_frames)
    frames = []
    for f in range(n_frames):
        m = int(rng.integers(0, 9000)) if f % 5 else 0           # includes empty frames
        rows = np.zeros((m, 10))
        rows[:, :3] = rng.normal(0, 3.0, (m, 3)) + [0.1 * f, 0, 0]
        rows[:, 3] = rng.integers(0, 256, m)
        rows[:, 4:7] = rng.integers(0, 256, (m, 3))
        rows[:, 7] = rng.integers(0, 19, m)
        rows[:, 8] = rng.integers(-1, 5, m)
        rows[:, 9] = rng.random(m) < 0.1
        frames.append(rows)
    if n_frames == 1:
        dense = np.zeros((5000, 10))
        dense[:, :3] = rng.normal(0, 1.0, (5000, 3))
        frames = [np.concatenate([frames[0], dense])]
    st = dev_store(capacity=max(sum(r.shape[0] for r in frames), 1), max_frames=n_frames + 1)
    assert st.load_rows(frames) is None
    st.voxel_dedup(size)
    st.check_status()
    allrows = np.concatenate(frames)
    vox = np.floor(allrows[:, :3] / size).astype(np.int64)
    _, first = np.unique(vox, axis=0, return_index=True)
    keep = np.zeros(len(allrows), bool)
    keep[first] = True
    bounds = np.concatenate([[0], np.cumsum([r.shape[0] for r in frames])])
    got = st.frame_rows()
    assert len(got) == n_frames
    for f in range(n_frames):
        want = allrows[bounds[f]:bounds[f + 1]][keep[bounds[f]:bounds[f + 1]]]
        assert np.array_equal(got[f], want), f'frame {f}'
    sizes_before = st.sizes().copy()
    st.voxel_dedup(size)                                      # idempotent
    assert np.array_equal(st.sizes(), sizes_before)
    assert all(np.array_equal(a, b) for a, b in zip(st.frame_rows(), got))
    assert 0 < keep.sum() < len(allrows) or len(allrows) == 0


@pytest.mark.parametrize('seed', range(24))
def test_bev_randomised_configs_match_oracle(T, orc, seed):
    """Seeded random sweep over the rasteriser's parameter space -- grid size (including non-multiples of the 8x8 tile
    and single-tile grids), view, rotation, augmentation shift, height filter, intensity encoding, class mix, dynamic
    flags, clustered vs uniform points (light and heavy tiles, cells above and below 64 values), empty present or
    future sets -- every plane against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    px = int(rng.choice([1, 5, 8, 13, 32, 57, 64, 96, 129, 200, 256]))
    view = float(rng.choice([8.0, 20.0, 51.2, 80.0, 123.4]))
    n = int(rng.choice([0, 1, 70, 3000, 40000, 120000]))
    rows = np.zeros((n, 10))
    mode = rng.integers(0, 3)
    if mode == 0:                                            # uniform
        rows[:, :2] = rng.uniform(-0.6 * view, 0.6 * view, (n, 2))
    elif mode == 1:                                          # a few dense clusters + background
        centres = rng.uniform(-0.4 * view, 0.4 * view, (5, 2))
        pick = rng.integers(0, 5, n)
        rows[:, :2] = centres[pick] + rng.normal(0, 0.01 * view, (n, 2))
        bg = rng.random(n) < 0.2
        rows[bg, :2] = rng.uniform(-0.6 * view, 0.6 * view, (int(bg.sum()), 2))
    else:                                                    # along a path
        rows[:, 0] = rng.uniform(-0.6 * view, 0.6 * view, n)
        rows[:, 1] = rng.normal(0, 0.03 * view, n)
    rows[:, 2] = rng.uniform(-2, 4, n)
    div255 = bool(rng.integers(0, 2))
    rows[:, 3] = rng.integers(0, 256, n) / 255. if div255 else rng.uniform(0, 1, n).astype(np.float32)
    spread = int(rng.choice([0, 3, 40, 255]))
    base = rng.integers(0, 256, 3)
    rows[:, 4:7] = np.clip(base + rng.integers(-spread, spread + 1, (n, 3)), 0, 255)
    rows[:, 7] = rng.choice([0, 1, 2, 8, 13, 14, 15, 17], n)
    rows[:, 9] = rng.random(n) < float(rng.choice([0.0, 0.05, 0.5]))
    cut = int(rng.choice([0, n // 3, n // 2, n]))
    hf = None if rng.integers(0, 2) else float(rng.uniform(0, 3))
    ints = (20., 20., 0.5) if not div255 else (1., 30., 0.12)
    rot = float(rng.uniform(-np.pi, np.pi))
    dx, dy = (0., 0.) if rng.integers(0, 2) else tuple(rng.uniform(-0.1 * view, 0.1 * view, 2))
    origin = tuple(rng.uniform(-1, 1, 3))
    p16, p64, _ = run_dev_bev(T, rows[:cut], rows[cut:], view, px, hf, ints, div255, rot, dx, dy, origin)
    ref = run_orc_bev(orc, rows[:cut], rows[cut:], view, px, hf, ints, div255, rot, dx, dy, origin)
    assert_planes_match(p16, p64, ref, f'seed {seed}: px={px} view={view} n={n} mode={mode}')


@pytest.mark.parametrize('seed', range(12))
def test_k1_k2_randomised_batches_match_oracle(T, orc, seed):
    """Seeded random sweep over K1 + K2: image sizes (down to 1x2), projection matrices (random rotations, focal
    lengths, points behind / on the camera plane), filter sets, batch compositions (empty, tiny, tile-boundary and
    large frames, several launches), interleaved re-transforms -- stored rows and segment sizes against the oracle."""
    rng = np.random.default_rng(5000 + seed)
    H, W = int(rng.choice([1, 2, 7, 64, 376])), int(rng.choice([2, 3, 96, 1408]))
    f = float(rng.uniform(0.3, 2.0)) * W
    K = np.array([[f, 0, W / 2 + rng.uniform(-3, 3), 0], [0, f, H / 2 + rng.uniform(-3, 3), 0], [0, 0, 1, 0]])
    ang = rng.uniform(-np.pi, np.pi, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
    Rz = np.array([[np.cos(ang[2]), -np.sin(ang[2]), 0], [np.sin(ang[2]), np.cos(ang[2]), 0], [0, 0, 1]])
    E = np.eye(4)
    E[:3, :3] = Rz @ Rx
    E[:3, 3] = rng.uniform(-2, 2, 3)
    P = K @ E
    filters = sorted(set(rng.integers(0, 20, int(rng.integers(0, 6))).tolist()) | ({255} if rng.integers(0, 2) else set()))
    pool = [0, 1, 2, 63, 64, 65, 2047, 2048, 2049, 5000, 30000, 121111]
    st = dev_store(capacity=700000, max_frames=64)
    ost = orc.Store(700000)
    sizes = []
    for launch in range(int(rng.integers(1, 4))):
        if launch and rng.integers(0, 2):
            Tm = np.eye(4)
            a = rng.uniform(-0.05, 0.05)
            Tm[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
            Tm[:3, 3] = rng.uniform(-1.5, 1.5, 3)
            st.retransform(Tm, defer=bool(rng.integers(0, 2)))
            orc.retransform(ost, Tm)
        frames = []
        for _ in range(int(rng.integers(1, 6))):
            n = int(rng.choice(pool))
            pc, img, sem = kitti_frame(rng, n, H, W, lim=float(rng.choice([3.0, 60.0])))
            if n > 4:
                pc[0, :3] = 0.0                              # exactly on the camera centre / plane
                pc[1, :3] = -pc[2, :3]
            frames.append(dict(pts=cu(T, pc), rgb=cu(T, img), sem=cu(T, sem)))
            sizes.append(orc.kitti_project_sample_filter(ost, pc, P, img, sem, None, H, W, filters))
        st.append_kitti(frames, P, H, W, filters)
    st.check_status()
    assert st.sizes().tolist() == sizes
    assert np.array_equal(st.rows(), ost.rows()), f'seed {seed}: H={H} W={W} filters={filters}'


@pytest.mark.parametrize('seed', range(10))
def test_nuscenes_kernels_randomised_match_oracle(T, orc, seed):
    """Seeded random sweep over the NuScenes kernels: K0n (1..6 cameras, random poses / intrinsics, points behind the
    cameras), K1n (random images, assignments incl. 'on no camera', class filters, poses), K3 (dynamic marking of
    instances over several frames)."""
    from datasets.nuscenes_utils import project_to_cameras
    rng = np.random.default_rng(9000 + seed)
    ncam = int(rng.integers(1, 7))
    H, W = int(rng.choice([8, 90, 225])), int(rng.choice([12, 160, 400]))

    def pose():
        a = rng.uniform(-np.pi, np.pi)
        M = np.eye(4)
        M[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        M[:3, 3] = rng.uniform(-30, 30, 3) * [1, 1, 0.05]
        return M
    # K0n
    n0 = int(rng.choice([1, 64, 3333, 20000]))
    lidar = np.c_[rng.uniform(-40, 40, (n0, 2)), rng.uniform(-2, 3, n0)]
    ego_from_lidar, glob_from_ego = pose(), pose()
    cam_R = np.array([[0, -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1.]])      # camera looks along ego x
    cams = [glob_from_ego @ pose() @ np.linalg.inv(cam_R) for _ in range(ncam)]
    Ks = [np.array([[rng.uniform(0.5, 1.5) * W, 0, W / 2], [0, rng.uniform(0.5, 1.5) * W, H / 2], [0, 0, 1.]])
          for _ in range(ncam)]
    whs = [np.array([W, H], dtype=float)] * ncam
    ego, uv, cam = project_to_cameras(lidar, ego_from_lidar, glob_from_ego, cams, Ks, whs)
    o_ego, o_uv, o_cam = orc.nusc_project_cams(lidar, ego_from_lidar, glob_from_ego,
                                               np.stack([np.linalg.inv(c) for c in cams]), np.stack(Ks), np.stack(whs))
    assert np.array_equal(ego, o_ego) and np.array_equal(cam, o_cam) and np.array_equal(uv, o_uv)
    # K1n + K3 over a few frames
    filters = sorted(set(rng.integers(0, 19, int(rng.integers(0, 5))).tolist()))
    st = dev_store(capacity=200000, max_frames=16, intensity_div255=True)
    ost = orc.Store(200000, intensity_div255=True)
    offs = [0]
    for k in range(int(rng.integers(1, 5))):
        n = int(rng.choice([0, 1, 255, 256, 257, 9000, 34720]))
        pc = np.zeros((n, 7))
        pc[:, :2] = rng.uniform(-50, 50, (n, 2))
        pc[:, 2] = rng.uniform(-2, 4, n)
        pc[:, 3] = rng.integers(0, 256, n)
        pc[:, 4] = rng.uniform(1.01, W - 1.01, n)
        pc[:, 5] = rng.uniform(1.01, H - 1.01, n)
        pc[:, 6] = rng.integers(-1, 5, n)
        cidx = rng.integers(-1, ncam, n)
        imgs = rng.integers(0, 256, (ncam, H, W, 3), dtype=np.uint8)
        sems = rng.integers(0, 19, (ncam, H, W)).astype(np.uint8)
        Tw = pose()
        st.append_nusc(cu(T, pc), cu(T, cidx), cu(T, imgs), cu(T, sems), Tw, filters)
        offs.append(offs[-1] + orc.nusc_sample_filter_transform(ost, pc, cidx, imgs, sems, Tw, filters))
    st.check_status()
    assert np.array_equal(st.offsets(), np.array(offs))
    assert np.array_equal(st.rows(), ost.rows())
    nf = len(offs) - 1
    pairs = [(int(rng.integers(0, nf)), int(rng.integers(-1, 5))) for _ in range(int(rng.integers(0, 7)))]
    st.mark_dynamic(pairs)
    for f, i in pairs:
        orc.mark_dynamic(ost, offs[f], offs[f + 1], i)
    assert np.array_equal(st.rows(), ost.rows())


# ------------------------------------------------------------------------------------------- opt-in bilinear sampling
def test_bilinear_sampler_matches_reference_fixture(T, golden):
    """pts_feat_from_img(..., 'bilinear') runs on the device (pca_sample_bilinear) and reproduces the reference's own
    output bit for bit; errors as the reference: AssertionError outside the image, ValueError for multi-channel maps."""
    from datasets.nuscenes_utils import pts_feat_from_img
    g = golden('utils')
    got = pts_feat_from_img(g['pf_uv_bil'], g['pf_img'][..., 0], 'bilinear')
    assert got.dtype == np.float64 and np.array_equal(got, g['pf_bilinear'])
    with pytest.raises(AssertionError):
        pts_feat_from_img(np.array([[0.5, 5.0]]), g['pf_img'][..., 0], 'bilinear')
    with pytest.raises(ValueError):
        pts_feat_from_img(g['pf_uv_bil'], g['pf_img'], 'bilinear')


@pytest.mark.parametrize('seed', [0, 1])
def test_k1_bilinear_mode_matches_oracle(T, orc, seed):
    """sample_mode='bilinear' of K1 (fused and split launch forms) against the oracle's restatement of the mode; the
    default mode is untouched (the other K1 tests)."""
    rng = np.random.default_rng(100 + seed)
    H, W = (94, 352) if seed == 0 else (376, 1408)
    frames, host = [], []
    for n in ([30000, 5] if seed == 0 else [120000] * 3):
        pc, img, sem = kitti_frame(rng, n, H, W)
        host.append((pc, img, sem))
        frames.append(dict(pts=cu(T, pc), rgb=cu(T, img), sem=cu(T, sem)))
    Pm = P_KITTI if seed else np.array([[138., 0, 176, 0], [0, 138., 47, 0], [0, 0, 1, 0]]) @ np.linalg.inv(
        np.vstack([np.array([[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418],
                             [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
                             [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824]]), [0, 0, 0, 1]]))
    st = dev_store(capacity=400000, max_frames=8)
    st.append_kitti(frames, Pm, H, W, KITTI_FILTERS, sample_mode='bilinear')
    got = st.frame_rows()
    st.check_status()
    orc.set_sample_mode(1)
    try:
        differs = 0
        for (pc, img, sem), rows in zip(host, got):
            ost = orc.Store(max(pc.shape[0], 1))
            orc.kitti_project_sample_filter(ost, pc, Pm, img, sem, None, H, W, KITTI_FILTERS)
            assert np.array_equal(rows, ost.rows())
            orc.set_sample_mode(0)
            near = orc.Store(max(pc.shape[0], 1))
            orc.kitti_project_sample_filter(near, pc, Pm, img, sem, None, H, W, KITTI_FILTERS)
            orc.set_sample_mode(1)
            assert np.array_equal(near.rows()[:, [0, 1, 2, 3, 7]], rows[:, [0, 1, 2, 3, 7]])     # same points, same class
            differs += int((near.rows()[:, 4:7] != rows[:, 4:7]).any(axis=1).sum())
        assert differs > 100                                                                      # the colours do differ
    finally:
        orc.set_sample_mode(0)


def test_k1n_bilinear_mode_matches_oracle(T, orc):
    rng = np.random.default_rng(77)
    n, ncam, H, W = 9000, 3, 60, 100
    pc = np.stack([rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), rng.uniform(-2, 4, n),
                   rng.integers(0, 256, n).astype(float), rng.uniform(1.01, W - 1.01, n), rng.uniform(1.01, H - 1.01, n),
                   rng.integers(-1, 4, n).astype(float)], 1)
    pc[:50, 4] = np.floor(pc[:50, 4])                       # integer coordinates: the upper neighbour has weight 0
    pc[:50, 4] = np.clip(pc[:50, 4], 2, W - 2)
    cam = rng.integers(-1, ncam, n)
    imgs = rng.integers(0, 256, (ncam, H, W, 3), dtype=np.uint8)
    sems = rng.integers(0, 19, (ncam, H, W)).astype(np.uint8)
    Tm = np.eye(4)
    Tm[:3, 3] = [5., -2., 0.5]
    filters = [10, 11, 12, 16, 18]
    st = dev_store(capacity=n, max_frames=4, intensity_div255=True)
    st.append_nusc(cu(T, pc), cu(T, cam), cu(T, imgs), cu(T, sems), Tm, filters, sample_mode='bilinear')
    st.check_status()
    orc.set_sample_mode(1)
    try:
        ost = orc.Store(n, intensity_div255=True)
        orc.nusc_sample_filter_transform(ost, pc, cam, imgs, sems, Tm, filters)
    finally:
        orc.set_sample_mode(0)
    assert np.array_equal(st.rows(0), ost.rows()) and ost.n > 1000


def test_bev_chain_of_owed_transforms_equals_eager_retransform(T):
    """Up to four re-transforms stay owed: the rasteriser applies them to what it reads and writes back every fourth
    time (pca_bev_generate_chain).  Planes and stored coordinates equal those of a store that re-transforms eagerly --
    with frames appended and evicted in between, a BEV missing now and then (the chain fills up and is flushed by K2)
    and the rows read back mid-chain (flush of a partial chain)."""
    from pca_amd.device_store import DeviceStore, make_bev_params
    rng = np.random.default_rng(123)

    def frame(n):
        rows = np.zeros((n, 10))
        rows[:, 0:2] = rng.uniform(-40, 40, (n, 2))
        rows[:, 2] = rng.uniform(-2, 2, n)
        rows[:, 3] = rng.integers(0, 256, n)
        rows[:, 4:7] = rng.integers(0, 256, (n, 3))
        rows[:, 7] = rng.integers(0, 19, n)
        rows[:, 9] = rng.integers(0, 2, n)
        return rows
    lazy, eager = DeviceStore(capacity=1 << 18, max_frames=32), DeviceStore(capacity=1 << 18, max_frames=32)
    lazy.CHAIN_K, eager.CHAIN_K = 4, 1
    assert lazy.load_rows([frame(6000) for _ in range(3)]) is None
    assert eager.load_rows(lazy.frame_rows()) is None
    prm = make_bev_params(np.zeros(3), np.eye(3), 0., 0., 80., 128, None, 20., 20., 0.5, 0, [13, 14, 15, 17], False, 0)
    for step in range(14):
        a = 0.01 * (step + 1)
        Tm = np.eye(4)
        Tm[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        Tm[0, 3], Tm[1, 3] = 0.9, -0.05 * step
        n_new = 5000 + 100 * step                          # a new frame through K1's ground-truth-semantics branch
        host = rng.uniform(-40, 40, (n_new, 4)).astype(np.float32)
        host[:, 3] = rng.integers(0, 256, n_new)
        pts = T.from_numpy(host).cuda()
        sem = T.from_numpy(rng.integers(0, 19, n_new).astype(np.uint8)).cuda()
        for st in (lazy, eager):
            st.retransform(Tm, defer=True)
            st.append_kitti([dict(pts=pts, sem_gt=sem)], np.eye(3, 4), 1, 1, [])
        if step in (5, 9):
            for st in (lazy, eager):
                st.evict(1)
        if step % 5 != 3:                                  # no BEV on steps 3, 8, 13: the chain grows past the write-back point
            pl, pe = lazy.bev(lazy.n_frames // 2, prm)[0], eager.bev(eager.n_frames // 2, prm)[0]
            assert T.equal(pl.view(T.int16), pe.view(T.int16)), step
        if step == 6:                                      # mid-chain read-back
            assert np.array_equal(lazy.rows(), eager.rows())
    assert np.array_equal(lazy.rows(), eager.rows())
    lazy.check_status()
    eager.check_status()


def _tilting_transform(step):
    """A re-transform that changes z: yaw + a small pitch + a translation with a z component."""
    a, b = 0.01 * (step + 1), 0.004 * ((step % 3) - 1) + 0.002
    Rz = np.array([[np.cos(a), -np.sin(a), 0.], [np.sin(a), np.cos(a), 0.], [0., 0., 1.]])
    Ry = np.array([[np.cos(b), 0., np.sin(b)], [0., 1., 0.], [-np.sin(b), 0., np.cos(b)]])
    Tm = np.eye(4)
    Tm[:3, :3] = Rz @ Ry
    Tm[:3, 3] = [0.9, -0.05 * step, 0.07 * (1 + step % 2)]
    return Tm


@pytest.mark.parametrize('variant', ['registers', 'memory_path', 'intensity64'])
def test_bev_chain_with_z_changing_transforms_matches_eager_and_oracle(T, orc, monkeypatch, variant):
    """The owed transforms of the chain change z (pitch + z translation).  Every record's z has to be the transformed
    one on all three routes of bev_tile_bin: points kept in registers, the memory path of pass B (forced with one
    level-1 workgroup: PCA_BEV_G=1 puts 18 000+ points in a chunk, beyond the 12 288 of the registers) and the f64
    intensity route (no register path at all).  Lazy (write-back every fourth raster) == eager == the oracle."""
    from pca_amd.device_store import DeviceStore, make_bev_params
    rng = np.random.default_rng(321)
    if variant == 'memory_path':
        monkeypatch.setenv('PCA_BEV_G', '1')

    def frame(n):
        rows = np.zeros((n, 10))
        rows[:, 0:2] = rng.uniform(-40, 40, (n, 2))
        rows[:, 2] = rng.uniform(-2, 2, n)
        rows[:, 3] = rng.uniform(0, 1, n) if variant == 'intensity64' else rng.integers(0, 256, n)
        rows[:, 4:7] = rng.integers(0, 256, (n, 3))
        rows[:, 7] = rng.integers(0, 19, n)
        rows[:, 9] = rng.integers(0, 2, n)
        return rows
    host = [frame(6000) for _ in range(3)]
    lazy, eager = DeviceStore(capacity=1 << 18, max_frames=32), DeviceStore(capacity=1 << 18, max_frames=32)
    lazy.CHAIN_K, eager.CHAIN_K = 4, 1
    i64_l, i64_e = lazy.load_rows(host), eager.load_rows(host)
    assert (i64_l is not None) == (variant == 'intensity64')
    allrows = np.concatenate(host)
    ost = orc.Store.from_rows(allrows)
    o_i64 = allrows[:, 3].copy() if variant == 'intensity64' else None
    args = [np.array([0.3, -0.2, 0.1]), np.eye(3), 0., 0., 80., 128, 2.5, 20., 20., 0.5, 0, [13, 14, 15, 17], False, 0]
    prm = make_bev_params(*args)
    sizes = [r.shape[0] for r in host]
    for step in range(9):
        Tm = _tilting_transform(step)
        for st in (lazy, eager):
            st.retransform(Tm, defer=True)
        orc.retransform(ost, Tm)
        if step == 4:                                       # no raster here: two transforms are owed at the next one
            continue
        pl, pl64 = lazy.bev(2, prm, want_f64=True, intensity64=i64_l)
        pe, _ = eager.bev(2, prm, want_f64=True, intensity64=i64_e)
        assert T.equal(pl.view(T.int16), pe.view(T.int16)), step
        ref = orc.bev(ost, sizes[0] + sizes[1], orc.make_bev_params(*args), intensity64=o_i64)['planes']
        got = pl64.cpu().numpy()
        for s in range(3):
            for k in (0, 2, 3, 4, 5, 6):                    # 6 = elevation: min z of the TRANSFORMED points
                assert np.array_equal(got[7 * s + k], ref[7 * s + k]), (step, s, k)
            assert np.abs(got[7 * s + 1] - ref[7 * s + 1]).max() < 1e-12
    assert np.array_equal(lazy.rows(), eager.rows())
    assert np.array_equal(lazy.rows(), ost.rows())
    lazy.check_status()
    eager.check_status()


def test_k1_one_pixel_image(T, orc):
    """A 1x1 image (3 bytes): the colour gather is one 4-byte load, so the library hands the kernel a padded copy."""
    rng = np.random.default_rng(5)
    n = 3000
    pc = np.stack([rng.uniform(1, 30, n), rng.uniform(-3, 3, n), rng.uniform(-1, 1, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
    P = np.array([[0.4, 0., 0.5, 0.], [0., 0.4, 0.5, 0.], [0., 0., 1., 0.]]) @ np.array(
        [[0., -1., 0., 0.], [0., 0., -1., 0.], [1., 0., 0., 0.], [0., 0., 0., 1.]])
    imgs = [np.array([[[7, 200, 31]]], np.uint8), np.array([[[255, 0, 128]]], np.uint8)]
    sems = [np.array([[3]], np.uint8), np.array([[13]], np.uint8)]
    st = dev_store(capacity=4 * n, max_frames=4)
    st.append_kitti([dict(pts=cu(T, pc), rgb=cu(T, im), sem=cu(T, se)) for im, se in zip(imgs, sems)], P, 1, 1, KITTI_FILTERS)
    st.check_status()
    got = st.frame_rows()
    for im, se, rows in zip(imgs, sems, got):
        ost = orc.Store(n)
        orc.kitti_project_sample_filter(ost, pc, P, im, se, None, 1, 1, KITTI_FILTERS)
        assert ost.n > 10 and np.array_equal(rows, ost.rows())


def test_tile_kernel_start_offsets_change_nothing_in_a_subprocess():
    """PCA_BEV_STAGGER (read once per process; an experiment's knob, off by default): the workgroups of bev_tile_cells that share
    a CU start a few microseconds apart.  Timing only -- the golden BEV and the randomised configurations come out bit for bit."""
    import subprocess
    import sys
    env = dict(os.environ, PCA_BEV_STAGGER='3')
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(here, 'test_gpu_kernels.py'), '-x', '-q', '-m', 'gpu', '-k',
                        'bev_golden_and_oracle or randomised_configs'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'no tests ran' not in r.stdout


def test_light_tile_kernel_thread_contiguous_mode_in_a_subprocess():
    """PCA_BEV_HEAVY_MIN is read once per process.  Raised to the LDS colour capacity (4096), tiles of 2561..4096 records
    stay with the light tile kernel and take its thread-contiguous mode (per-thread runs); the skewed and the dense-tile
    parity cases are repeated that way in a child process."""
    import subprocess
    import sys
    env = dict(os.environ, PCA_BEV_HEAVY_MIN='4096')
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(here, 'test_gpu_kernels.py'), '-x', '-q', '-m', 'gpu', '-k',
                        'ring_like_skew or dense_tile_goes or randomised_configs'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def _ab_switches_child():
    """(run in a child process by the test below) hashes of a skewed raster's planes and of a batched K1's stored rows."""
    import hashlib
    import torch as T_
    rng = np.random.default_rng(123)
    rows = _skewed_rows(rng, 150000, 0.9, 4)
    p16, p64, _ = run_dev_bev(T_, rows[:60000], rows[60000:], 32, 64, None, (1., 30., 0.12), True, 0.2)
    p16b, _, _ = run_dev_bev(T_, rows[:60000], rows[60000:], 32, 64, None, (1., 30., 0.12), True, 0.2)      # (second call: heavy kernel)
    h_bev = hashlib.sha256(np.asarray(p16).tobytes() + np.asarray(p16b).tobytes() + np.asarray(p64).tobytes()).hexdigest()
    from pca_amd.device_store import DeviceStore
    H, W, n = 94, 352, 30000
    frames = []
    for k in range(6):
        pc = np.stack([rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-2, 3, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
        frames.append(dict(pts=T_.from_numpy(pc).cuda(), rgb=T_.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda(),
                           sem=T_.from_numpy(rng.integers(0, 19, (H, W)).astype(np.uint8)).cuda()))
    cam_to_velo = np.array([[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418], [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
                            [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824], [0, 0, 0, 1]])
    P = np.array([[138., 0, 176, 0], [0, 138., 47, 0], [0, 0, 1, 0]]) @ np.linalg.inv(cam_to_velo)
    st = DeviceStore(capacity=6 * n, max_frames=8)
    os.environ['PCA_K1_MODE'] = 'split'
    st.append_kitti(frames, P, H, W, [10, 11, 12, 16, 18, 255])
    st.check_status()
    h_k1 = hashlib.sha256(st.rows().tobytes()).hexdigest()
    print('ABHASH', h_bev, h_k1, int(st.offsets()[-1]))


def test_ab_switches_of_round5_change_no_result():
    """PCA_BEV_SPLIT (pieces ordered by half of the tile: off / register path / always) and PCA_K1_APPEND (tiles per
    k1_append workgroup, non-temporal accesses) are speed switches read once per process: every setting gives the bits of the
    default (a raster with heavy and light tiles, f16 and f64 planes; the stored rows of a six-frame split-form K1 batch)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ('import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_kernels as t; '
            't._ab_switches_child()' % (here, os.path.join(os.path.dirname(here), 'pc-accumulation-lib_amd'), os.path.dirname(here)))
    got = {}
    for name, env in (('default', {}), ('split0', {'PCA_BEV_SPLIT': '0'}), ('split2', {'PCA_BEV_SPLIT': '2'}),
                      ('append4nt', {'PCA_K1_APPEND': '4,nt'}), ('staging_plain', {'PCA_STAGING_NT': '0'})):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stdout[-1500:] + r.stderr[-1500:])
        got[name] = [ln for ln in r.stdout.splitlines() if ln.startswith('ABHASH ')][-1].split()[1:]
    assert int(got['default'][2]) > 20000
    for name, v in got.items():
        assert v == got['default'], name


def test_k1_defer_semantics_at_the_c_abi(T, orc):
    """pca_k1_defer at the store level: a noted K1 rides only in a raster whose window ends with its slot (same store, 5-plane
    f32-intensity layout); any other call -- a raster of another window, an f64-intensity raster, a re-transform, a second
    observation, offsets() -- runs it first, on its own.  Store and planes equal a store that never defers, call by call,
    and the oracle at the end."""
    import ctypes as C

    from pca_amd import _lib
    from pca_amd.device_store import make_bev_params
    rng = np.random.default_rng(5)
    H, W = 64, 96
    P = np.array([[40., 0, 48, 0], [0, 40., 32, 0], [0, 0, 1, 0]]) @ np.array(
        [[0., -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]])

    def frame(n):
        pc = np.stack([rng.uniform(0.5, 20, n), rng.uniform(-9, 9, n), rng.uniform(-1, 2, n), rng.uniform(0, 1, n)],
                      1).astype(np.float32)
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        sem = rng.integers(0, 19, (H, W)).astype(np.uint8)
        return pc, img, sem
    frames = [frame(int(n)) for n in (3000, 1, 9000, 200, 5000, 7000, 4097, 4096)]
    dev = [tuple(T.from_numpy(a).cuda() for a in f) for f in frames]
    stores = [dev_store(capacity=1 << 17, max_frames=32) for _ in range(2)]
    ctx = stores[0].ctx
    prm = make_bev_params((3.0, 0.5, 0.0), np.eye(3), 0., 0., 40., 64, None, 20., 20., 0.5, 0, DYNOBJ, False)
    Tm = np.eye(4)
    Tm[:3, 3] = [-0.5, 0.01, 0.0]

    def obs_of(k, host):
        pc, img, sem = frames[k] if host else dev[k]
        o = _lib.PcaKittiObs()
        ptr = (lambda a: a.ctypes.data) if host else (lambda a: a.data_ptr())
        o.pts, o.rgb, o.sem, o.sem_gt, o.n, o.host_mask = ptr(pc), ptr(img), ptr(sem), None, len(pc), (7 if host else 0)
        return o, (pc, img, sem)

    def planes(st, **kw):
        p16, p64 = st.bev(max(st.n_frames - 2, 1), prm, **kw)
        return p16.cpu().numpy().view(np.uint16), None if p64 is None else p64.cpu().numpy()

    def step(st, k):
        if st.n_frames:
            st.retransform(Tm, defer=True)
        o, alive = obs_of(k, host=(k % 3 == 2))
        st.append_kitti_obs(o, P, H, W, KITTI_FILTERS, keep=alive)
        if k == 1:
            return None                                   # no raster: the next observation runs this K1
        if k == 3:
            return planes(st, last_frame=st.n_frames - 1)  # a window that ends BEFORE the noted frame
        if k == 4:
            return planes(st, want_f64=True)              # rides (the f64 planes are the tile kernel's business)
        if k == 5:
            st.flush_pending()                            # the owed re-transform, eagerly: K1 first
            return planes(st)
        if k == 6:
            return planes(st, first_frame=2)              # a shorter window (the owed re-transforms are applied first: K1 too)
        return planes(st)
    results, rows, launches = [], [], []
    for w, st in enumerate(stores):                       # the whole sequence with deferral, then without
        st.set_defer_k1(w == 0)
        per_step, res = [], []
        for k in range(len(frames)):
            ctx.profile(True)
            res.append(step(st, k))
            per_step.append(ctx.profile_read()['kitti_project_sample_filter'][1])
            ctx.profile(False)
        results.append(res)
        rows.append(st.rows())
        launches.append(per_step)
    stores[0].set_defer_k1(False)
    for a, b in zip(*results):
        assert (a is None) == (b is None)
        if a is not None:
            assert np.array_equal(a[0], b[0])
            assert (a[1] is None) == (b[1] is None) and (a[1] is None or np.array_equal(a[1], b[1], equal_nan=True))
    # K1 launches of its own, step by step: without deferral one per step; with it none where the K1 rode (k = 0, 4, 7),
    # the K1 of the step before at k = 2 (k = 1 had no raster), this step's at k = 3 (another window), k = 5 (re-transform)
    # and k = 6 (a window that starts later: DeviceStore applies the owed re-transforms eagerly first)
    assert all(v >= 1 for v in launches[1]), launches
    assert [launches[0][k] for k in (0, 1, 4, 7)] == [0] * 4 and all(launches[0][k] >= 1 for k in (2, 3, 5, 6)), launches
    assert np.array_equal(rows[0], rows[1]) and rows[0].shape[0] > 5000
    assert np.array_equal(stores[0].offsets(), stores[1].offsets())
    # ... and the oracle on the final store
    ost = orc.Store(1 << 17)
    for k, (pc, img, sem) in enumerate(frames):
        if k:
            orc.retransform(ost, Tm, 0, ost.n)
        orc.kitti_project_sample_filter(ost, pc, P, img, sem, None, H, W, KITTI_FILTERS)
    assert np.array_equal(rows[0], ost.rows(0))
    for st in stores:
        st.check_status()
