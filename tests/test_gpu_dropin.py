"""GPU parity, API level: the drop-in classes (same names / constructors / methods as the reference) run
the golden sequences end to end on the device and must reproduce the reference's own outputs."""
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KITTI_FILTERS = [10, 11, 12, 16, 18, 255]
NUSC_FILTERS = [10, 11, 12, 16, 18]
SEM_IDXS = {'road': 0, 'car': 13, 'truck': 14, 'bus': 15, 'motorcycle': 17}
BEV_KITTI = dict(type='sem', view_size=40, pixel_size=32, max_trans_radius=0., zoom_thresh=0., do_warp=False,
                 int_scaler=20., int_sep_scaler=20., int_mid_threshold=0.5, height_filter=None)
BEV_NUSC = dict(type='sem', view_size=30, pixel_size=32, max_trans_radius=0., zoom_thresh=0., do_warp=False,
                int_scaler=1., int_sep_scaler=30., int_mid_threshold=0.12, height_filter=3.)


class FakeSemSeg:
    """Same stand-in for the ONNX model that generated the fixtures (tools/make_golden.py)."""

    def pred(self, rgb):
        a = np.asarray(rgb).astype(np.int64)
        return ((a[..., 0] + 2 * a[..., 1] + 3 * a[..., 2]) % 19)[None, None]


@pytest.fixture(autouse=True)
def fake_model(monkeypatch):
    import torch
    assert torch.cuda.is_available()
    import sem_pc_accum
    monkeypatch.setattr(sem_pc_accum, 'SemSegONNX', lambda path: FakeSemSeg())


def check_bev(bev, g, prefix='bev_'):
    keys = [k[len(prefix):] for k in g.files if k.startswith(prefix)]
    assert keys
    for s in ('present', 'future', 'full'):
        for key in ('road', 'dynamic', 'elevation', 'rgb'):
            got, want = bev[f'{key}_{s}'], g[f'{prefix}{key}_{s}']
            assert got.dtype == np.float16 and got.shape == want.shape
            assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), (key, s)
        d = np.abs(bev[f'intensity_{s}'].view(np.uint16).astype(int) - g[f'{prefix}intensity_{s}'].view(np.uint16).astype(int))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3
        n = int(g[f'{prefix}trajs_{s}_n'])
        assert len(bev[f'trajs_{s}']) == n
        for k in range(n):
            assert np.array_equal(bev[f'trajs_{s}'][k], g[f'{prefix}trajs_{s}_{k}']), (s, k)
    assert set(bev.keys()) == {f'{a}_{s}' for a in ('road', 'trajs', 'intensity', 'rgb', 'dynamic', 'elevation')
                               for s in ('present', 'future', 'full')} | ({'gt_lanes'} & set(bev.keys()))


@pytest.mark.parametrize('track', ['default', 'numpy'])
def test_kitti_accumulator_sequence(golden, capsys, track):
    """track='numpy': the fallback pose track (the reference's numpy expressions themselves, what the accumulator uses when a
    probe of the C track fails on a machine) through the same golden sequence, on the GPU box."""
    from PIL import Image

    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    from pca_amd import host_logic as hl
    g = golden('kitti_accum')
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': g['P']}
    acc = Kitti360SemanticPointCloudAccumulator(float(g['horizon']), calib, 1e3, 'fake.onnx', KITTI_FILTERS, SEM_IDXS,
                                                False, dict(BEV_KITTI))
    if track == 'numpy':
        acc._track = hl.NumpyPoseTrack()
    queue = list(g['Ts'])
    acc.pose_provider = lambda pc: queue.pop(0)
    removed = []
    for k in range(int(g['F'])):
        removed.append(acc.integrate([(Image.fromarray(g[f'img_{k}']), g[f'pc_{k}'], None)]))
        if f'step{k}_sizes' in g.files:
            sem_pcs = acc.sem_pcs
            assert np.array_equal(np.array([a.shape[0] for a in sem_pcs]), g[f'step{k}_sizes'])
            assert np.array_equal(np.concatenate(sem_pcs), g[f'step{k}_sem_pcs'])
            assert np.array_equal(np.array(acc.poses), g[f'step{k}_poses'])
            assert np.array_equal(np.array(acc.seg_dists), g[f'step{k}_seg_dists'])
    assert np.array_equal(np.array(removed), g['removed'])
    assert np.array_equal(acc.get_incremental_path_dists(), g['incr_path_dists'])
    assert len(acc.get_rgb(2)) == 1 and acc.get_semseg(2)[0].shape == (64, 96)
    bevs = acc.generate_bev(int(g['present_idx']), 1, gen_future=True)
    assert len(bevs) == 1
    check_bev(bevs[0], g)
    acc.store.check_status()
    assert '#pc' in capsys.readouterr().out


def test_kitti_accumulator_gt_sem_and_io(golden, tmp_path):
    from PIL import Image

    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    g = golden('kitti_gtsem')
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': g['P']}
    acc = Kitti360SemanticPointCloudAccumulator(50., calib, 1e3, None, KITTI_FILTERS, SEM_IDXS, True, dict(BEV_KITTI))
    queue = list(g['Ts'])
    acc.pose_provider = lambda pc: queue.pop(0)
    dummy = Image.fromarray(np.zeros((64, 96, 3), np.uint8))
    for k in range(4):
        assert acc.integrate([(dummy, g[f'pc_{k}'], g[f'sem_gt_{k}'])]) == 0
    assert np.array_equal(np.concatenate(acc.sem_pcs), g['sem_pcs'])
    assert np.array_equal(np.array(acc.poses), g['poses'])
    bev = acc.generate_bev(2, 1, gen_future=True)[0]
    check_bev(bev, g)
    # output writer: same container format as the reference (gzip + pickle of the dict)
    acc.write_compressed_pickle(bev, 'bev_000.pkl', str(tmp_path))
    back = acc.read_compressed_pickle(os.path.join(str(tmp_path), 'bev_000.pkl.gz'))
    assert set(back.keys()) == set(bev.keys())
    assert np.array_equal(back['rgb_full'].view(np.uint16), bev['rgb_full'].view(np.uint16))
    acc.viz_bev(bev, os.path.join(str(tmp_path), 'viz.png'), acc.get_rgb(2), acc.get_semseg(2))
    assert os.path.getsize(os.path.join(str(tmp_path), 'viz.png')) > 1000
    # two augmentation-free copies are identical
    b2 = acc.generate_bev(2, 2, gen_future=True)
    assert len(b2) == 2 and np.array_equal(b2[0]['road_full'].view(np.uint16), b2[1]['road_full'].view(np.uint16))


def test_nuscenes_oracle_accumulator_sequence(golden):
    from PIL import Image

    from nuscenes_oracle_sem_pc_accum import NuScenesOracleSemanticPointCloudAccumulator
    g = golden('nusc_oracle')
    acc = NuScenesOracleSemanticPointCloudAccumulator('fake.onnx', NUSC_FILTERS, SEM_IDXS, False, dict(BEV_NUSC),
                                                      'boston', False, None)
    for k in range(int(g['F'])):
        T = g[f'T_{k}']
        obs = dict(images=[Image.fromarray(im) for im in g[f'imgs_{k}']], pc=g[f'pc_{k}'],
                   pc_cam_idx=g[f'cam_idx_{k}'], ego_at_lidar_ts=T, ego_global_x=T[0, 3], ego_global_y=T[1, 3],
                   inst_tokens=str(g['inst_tokens'][k]).split(','), inst_cls=list(g[f'inst_cls_{k}']),
                   inst_center=list(g[f'inst_center_{k}']))
        assert acc.integrate([obs]) is None
        if k == 0:
            assert np.array_equal(acc.sem_pcs[0], g['frame0_after_integrate'])
    assert np.array_equal(np.array([a.shape[0] for a in acc.sem_pcs]), g['sizes'])
    assert np.array_equal(np.concatenate(acc.sem_pcs), g['sem_pcs'])          # incl. retroactive dyn flags
    assert np.array_equal(np.array(acc.poses), g['poses'])
    assert np.array_equal(np.array(acc.seg_dists), g['seg_dists'])
    assert list(g['dyn_instances']) == acc.dyn_instances
    assert np.array_equal(acc.get_incremental_path_dists(), g['incr_path_dists'])
    assert acc.map == 'boston' and len(acc.ego_global_xs) == int(g['F'])
    bev = acc.generate_bev(int(g['present_idx']), 1, gen_future=True)[0]
    check_bev(bev, g)


@pytest.mark.parametrize('case,ctor,args', [
    ('bev_a', (20, 32, 0., 0., False, 20., 20., 0.5, None), None),
    ('bev_b', (51.2, 64, 0., 0., False, 1., 30., 0.12, 3.), 'file'),
    ('bev_c', (80, 256, 0., 0., False, 20., 20., 0.5, None), None),
    ('bev_d', (20, 16, 0., 0., False, 20., 20., 0.5, None), None),
    ('bev_e', (20, 32, 0., 0., True, 20., 20., 0.5, None), None),
])
def test_sem_bev_generator_host_arrays(golden, case, ctor, args):
    """SemBEVGenerator.generate on host (N,10) arrays -- the reference's own calling convention."""
    from bev_generator.sem_bev import SemBEVGenerator
    g = golden(case)
    gen = SemBEVGenerator(SEM_IDXS, *ctor)
    if case == 'bev_e':
        w = tuple(g['warp'])
        gen.get_random_warp_params = lambda *a: w
    present, future = g['pc_present'], g['pc_future']
    pcs = dict(pc_present=present.copy(), pc_future=future.copy(), pc_full=np.concatenate([present, future]))
    trajs = {}
    for k in ('ego_traj_present', 'ego_traj_future', 'ego_traj_full'):
        trajs[k] = g['in_' + k].copy()
    for k in ('other_trajs_present', 'other_trajs_future', 'other_trajs_full'):
        trajs[k] = [g[f'in_{k}_{i}'].copy() for i in range(int(g[f'in_{k}_n']))]
    keep = present.copy()
    if args == 'file':
        a = g['args']
        bev = gen.generate(pcs, trajs, a[0], a[1], a[2], a[3], True)
    else:
        bev = gen.generate(pcs, trajs)
    check_bev(bev, g)
    assert np.array_equal(pcs['pc_present'], keep)          # inputs are not mutated (documented difference)


def test_generate_without_future_raises_like_reference(golden):
    from bev_generator.sem_bev import SemBEVGenerator
    g = golden('bev_d')
    gen = SemBEVGenerator(SEM_IDXS, 20, 16)
    pcs = dict(pc_present=g['pc_present'], pc_future=None, pc_full=None)
    trajs = dict(ego_traj_present=g['in_ego_traj_present'], ego_traj_future=None, ego_traj_full=None,
                 other_trajs_present=[], other_trajs_future=None, other_trajs_full=None)
    with pytest.raises(UnboundLocalError):
        gen.generate(pcs, trajs)


def test_rgb_bev_generator_medians(golden):
    from bev_generator.rgb_bev import RGBBEVGenerator
    from bev_generator.sem_bev import SemBEVGenerator
    g = golden('utils')
    rgen = RGBBEVGenerator(20, 16, 7)
    r, gr, b = rgen.get_rgb_maps(g['rg_pc'])
    assert np.array_equal(np.stack([r, gr, b]), g['rg_out'])
    sgen = SemBEVGenerator(SEM_IDXS, 20, 16, rgb_fill=7)
    r2, g2, b2 = sgen.get_rgb_maps(g['rg_pc'])
    assert np.array_equal(np.stack([r2, g2, b2]), g['rg_out'])
    elev, mask = sgen.get_elevation_map(g['rg_pc'])
    ij = g['rg_pc'][:, :2].astype(int)
    want = np.zeros((16, 16))
    seen = np.zeros((16, 16), bool)
    for (i, j), z in zip(ij, g['rg_pc'][:, 2]):
        if not seen[15 - j, i] or z < want[15 - j, i]:
            want[15 - j, i] = z
            seen[15 - j, i] = True
    assert np.array_equal(elev, want) and np.array_equal(mask, seen)


def test_helper_methods_match_reference(golden):
    """velo2img-level helpers of the accumulator's public surface."""
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    g = golden('k1')
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': g['P']}
    acc = Kitti360SemanticPointCloudAccumulator(8., calib, 1e3, None, KITTI_FILTERS, SEM_IDXS, True, dict(BEV_KITTI))
    rgb_rows = acc.gen_semantic_pc(g['pc'], g['img'], g['P'])
    sem_rows = acc.gen_semantic_pc(g['pc'], g['sem'][..., None], g['P'])
    assert np.array_equal(rgb_rows, g['sem_rgb'])
    assert np.array_equal(sem_rows, g['sem_sem'])
    both = np.concatenate((rgb_rows, sem_rows[:, -1:]), axis=1)
    assert np.array_equal(acc.filter_semseg_pc(both), g['filtered'])


def test_driver_shaped_run_on_fake_kitti_tree(tmp_path, monkeypatch):
    """The call sequence of run_kitti360_bev_gen.py (dataloader -> calibration -> accumulator -> trigger ->
    generate_bev -> pickle + png) on a synthetic KITTI-360 tree, GT semantics, poses from PCA_KITTI_T_FILE;
    the last BEV and the stored points are checked against the oracle pipeline."""
    from fake_kitti import SEQ, write_tree
    from oracle import oracle as orc

    from datasets.kitti360_utils import get_camera_intrinsics, get_transf_matrices
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    from obs_dataloaders.kitti360_obs_dataloader import Kitti360Dataloader
    from pca_amd import host_logic as hl
    root = str(tmp_path / 'KITTI-360')
    n_frames = 40
    frames, Ts, _ = write_tree(root, first_idx=130, n_frames=n_frames)
    monkeypatch.setenv('PCA_KITTI_T_FILE', os.path.join(root, 'T_new_prev.npy'))
    filters = [10, 11, 12, 16, 18, 255]
    h_cam_velo, h_velo_cam = get_transf_matrices(root)
    p_cam_frame = get_camera_intrinsics(root)
    calib = {'h_velo_cam': h_velo_cam, 'p_cam_frame': p_cam_frame, 'p_velo_frame': np.matmul(p_cam_frame, h_velo_cam)}
    bev_params = dict(BEV_KITTI, view_size=30, pixel_size=64)
    acc = Kitti360SemanticPointCloudAccumulator(20., calib, 1e3, 'none.onnx', filters, SEM_IDXS, True, bev_params)
    loader = Kitti360Dataloader(root, 1, [SEQ], [130], [130 + n_frames])
    bev_horizon, previous_idx, n_bev, last = 6, 0, 0, None
    for observations in loader:
        previous_idx -= acc.integrate(observations)
        if len(acc.poses) < 2:
            continue
        d = acc.get_incremental_path_dists()
        if d[-1] < bev_horizon:
            continue
        present_idx = ((d - bev_horizon) > 0).argmax()
        if d[-1] - d[present_idx] < bev_horizon:
            continue
        if acc.dist(acc.get_pose(previous_idx), acc.get_pose(present_idx)) < 1:
            continue
        previous_idx = present_idx
        bevs = acc.generate_bev(present_idx, 1, gen_future=True)
        out_dir = str(tmp_path / 'bevs' / 'subdir000')
        os.makedirs(out_dir, exist_ok=True)
        acc.write_compressed_pickle(bevs[0], f'bev_{n_bev:03d}.pkl', out_dir)
        acc.viz_bev(bevs[0], os.path.join(out_dir, f'viz_{n_bev:03d}.png'), acc.get_rgb(present_idx),
                    acc.get_semseg(present_idx))
        n_bev += 1
        last = (bevs[0], int(present_idx))
    from pca_amd import writer
    writer.flush_shared()          # samples go to the background writer; a driver process gets this flush at exit
    assert n_bev >= 5 and len(os.listdir(str(tmp_path / 'bevs' / 'subdir000'))) == 2 * n_bev

    # oracle replay of the same sequence
    idx2idx = loader.idx2idx
    st = orc.Store(n_frames * 3000)
    track = hl.PoseTrack()
    sizes, lo = [], 0
    for k, (pc, img, lab) in enumerate(frames):
        sem = lab.astype(np.int16).copy()[:, None]
        from datasets.kitti360_utils import conv_semantic_ids
        sem = conv_semantic_ids(sem, idx2idx)[:, 0].astype(np.uint8)
        if len(track):
            track.apply_transform(Ts[k])
            orc.retransform(st, Ts[k], lo, st.n)
        sizes.append(orc.kitti_project_sample_filter(st, pc, calib['p_velo_frame'], None, None, sem, 1, 1, filters))
        track.append([0., 0., 0.])
        if len(track) > 1:
            ev = track.evict_beyond(20., track.push_segment())
            lo += int(np.sum(sizes[:ev]))
            sizes = sizes[ev:]
    assert np.array_equal(np.concatenate(acc.sem_pcs), st.rows(lo))
    assert np.array_equal(np.array(acc.poses), np.array(track.poses))
    bev, pidx = last
    origin = np.array(track.poses[pidx])
    ego = np.array(track.poses[:pidx]) - origin
    R = hl.rotation_matrix_3d(hl.heading_rot_ang(ego))
    prm = orc.make_bev_params(origin, R, 0., 0., 30, 64, None, 20., 20., 0.5, 0, [13, 14, 15, 17], False)
    sub = orc.Store(st.n - lo + 1)
    for name in ('x', 'y', 'z', 'intensity', 'rgbs', 'inst', 'dyn'):
        getattr(sub, name)[:st.n - lo] = getattr(st, name)[lo:st.n]
    sub.n = st.n - lo
    ref = orc.bev(sub, int(np.sum(sizes[:pidx])), prm)
    F = ref['f16']
    for s, name in enumerate(('present', 'future', 'full')):
        assert np.array_equal(bev[f'road_{name}'].view(np.uint16), F[7 * s].view(np.uint16))
        assert np.array_equal(bev[f'elevation_{name}'].view(np.uint16), F[7 * s + 6].view(np.uint16))
        assert np.array_equal(bev[f'dynamic_{name}'].view(np.uint16), F[7 * s + 5].view(np.uint16))
        assert np.abs(bev[f'intensity_{name}'].view(np.uint16).astype(int) - F[7 * s + 1].view(np.uint16).astype(int)).max() <= 1


def test_long_stream_with_window_slides_and_growth():
    """1200 frames through a deliberately small store: the live window has to slide to the front many times and
    the slot table has to grow; state must stay identical to the oracle's."""
    import torch
    from oracle import oracle as orc
    from pca_amd import host_logic as hl
    from pca_amd.device_store import DeviceStore
    rng = np.random.default_rng(8)
    st = DeviceStore(capacity=6000, max_frames=16)
    ost = orc.Store(1200 * 300)
    track = hl.PoseTrack()
    P = np.array([[40., 0, 48, 0], [0, 40., 32, 0], [0, 0, 1, 0]])
    sizes, lo = [], 0
    T = np.eye(4)
    T[:3, :3] = hl.rotation_matrix_3d(0.01)
    T[:3, 3] = [-1.0, 0.02, 0.0]
    for k in range(1200):
        n = int(rng.integers(50, 300))
        pc = np.stack([rng.uniform(-9, 9, n), rng.uniform(-9, 9, n), rng.uniform(-1, 2, n), rng.uniform(0, 1, n)],
                      1).astype(np.float32)
        sem = rng.integers(0, 19, n).astype(np.uint8)
        if len(track):
            track.apply_transform(T)
            st.retransform(T, defer=(k % 3 == 0))
            orc.retransform(ost, T, lo, ost.n)
        st.append_kitti([dict(pts=torch.from_numpy(pc).cuda(), sem_gt=torch.from_numpy(sem).cuda())], P, 1, 1,
                        [10, 11, 12, 16, 18, 255])
        sizes.append(orc.kitti_project_sample_filter(ost, pc, P, None, None, sem, 1, 1, [10, 11, 12, 16, 18, 255]))
        track.append([0., 0., 0.])
        if len(track) > 1:
            ev = track.evict_beyond(25., track.push_segment())
            st.evict(ev)
            lo += int(np.sum(sizes[:ev]))
            sizes = sizes[ev:]
        if k in (0, 17, 400, 1199):
            assert np.array_equal(st.sizes(), np.array(sizes))
            assert np.array_equal(st.rows(), ost.rows(lo))
    st.check_status()
    assert st.max_frames > 16 and st.capacity >= 6000


def test_prefetching_loader_and_async_writer(tmp_path, monkeypatch):
    """Ingest + writer (the 'next' rows either side of the path): same observations, same stored points, same
    files as the synchronous loader / writer."""
    from fake_kitti import SEQ, write_tree

    from datasets.kitti360_utils import get_camera_intrinsics, get_transf_matrices
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    from obs_dataloaders.kitti360_obs_dataloader import Kitti360Dataloader
    from pca_amd.ingest import PrefetchingLoader, compose_label_lut
    from pca_amd.writer import AsyncBevWriter
    root = str(tmp_path / 'KITTI-360')
    n_frames = 24
    write_tree(root, first_idx=0, n_frames=n_frames)
    _, h_velo_cam = get_transf_matrices(root)
    p_cam = get_camera_intrinsics(root)
    calib = {'h_velo_cam': h_velo_cam, 'p_cam_frame': p_cam, 'p_velo_frame': np.matmul(p_cam, h_velo_cam)}
    Ts = np.load(os.path.join(root, 'T_new_prev.npy'))

    def run(loader, writer):
        acc = Kitti360SemanticPointCloudAccumulator(50., calib, 1e3, 'none', KITTI_FILTERS, SEM_IDXS, True,
                                                    dict(BEV_KITTI, view_size=30, pixel_size=32))
        it = iter(Ts)
        acc.pose_provider = lambda pc: next(it)
        for observations in loader:
            acc.integrate(observations)
        bev = acc.generate_bev(12, 1, gen_future=True)[0]
        bev_dev = acc.sem_bev_generator.generate(*acc._window_inputs(12, True), device_only=True)
        if writer is None:
            acc.write_compressed_pickle(bev, 'a.pkl', str(tmp_path / 'sync'))
        else:
            writer.submit(bev, 'a.pkl', str(tmp_path / 'async'))
            writer.submit(bev_dev, 'b.pkl', str(tmp_path / 'async'))
        return np.concatenate(acc.sem_pcs), bev, acc

    os.makedirs(str(tmp_path / 'sync'))
    base = Kitti360Dataloader(root, 1, [SEQ], [0], [n_frames])
    rows_a, bev_a, acc = run(base, None)
    w = AsyncBevWriter(n_threads=2)
    rows_b, bev_b, _ = run(PrefetchingLoader(Kitti360Dataloader(root, 1, [SEQ], [0], [n_frames]), depth=3), w)
    w.close()
    assert np.array_equal(rows_a, rows_b)
    for k in bev_a:
        if not k.startswith('trajs'):
            assert np.array_equal(bev_a[k].view(np.uint16), bev_b[k].view(np.uint16)), k
    rd = acc.read_compressed_pickle
    sync, a, b = rd(str(tmp_path / 'sync' / 'a.pkl.gz')), rd(str(tmp_path / 'async' / 'a.pkl.gz')), rd(str(tmp_path / 'async' / 'b.pkl.gz'))
    assert set(sync) == set(a) == set(b)
    for k in sync:
        if not k.startswith('trajs'):
            assert np.array_equal(sync[k], a[k]) and np.array_equal(sync[k], b[k]), k
    # composed label table == the reference's sequential remap
    from datasets.kitti360_utils import conv_semantic_ids
    table, lo = compose_label_lut(base.idx2idx)
    ids = np.arange(-1, 46, dtype=np.int16)[:, None]
    assert np.array_equal(table[:47], conv_semantic_ids(ids.copy(), base.idx2idx)[:, 0])


def test_prefetching_loader_order_and_reader_errors(tmp_path, monkeypatch):
    """Eight reader threads deliver the frames in order; with PCA_INGEST_IMAGES=0 the image is decoded on first use only;
    a reader's exception (a missing file) surfaces in the consumer at that frame."""
    import torch
    from fake_kitti import SEQ, write_tree

    from obs_dataloaders.kitti360_obs_dataloader import Kitti360Dataloader
    from pca_amd.ingest import DeviceImage, PrefetchingLoader
    root = str(tmp_path / 'KITTI-360')
    n_frames = 20
    frames, _, _ = write_tree(root, first_idx=0, n_frames=n_frames, n_pts=500, H=16, W=24)
    monkeypatch.setenv('PCA_INGEST_THREADS', '8')
    for images in ('1', '0'):
        monkeypatch.setenv('PCA_INGEST_IMAGES', images)
        k = 0
        for obs in PrefetchingLoader(Kitti360Dataloader(root, 1, [SEQ], [0], [n_frames]), depth=6):
            img, pc, sem = obs[0]                 # device buffers belong to a ring: look at a batch while it is current
            assert isinstance(img, DeviceImage) and isinstance(pc, torch.Tensor) and pc.is_cuda
            assert np.array_equal(pc.cpu().numpy(), frames[k][0])
            assert np.array_equal(np.asarray(img), frames[k][1]) and np.array_equal(img.dev.cpu().numpy(), frames[k][1])
            k += 1
        assert k == n_frames
    os.remove(os.path.join(root, 'data_3d_raw', SEQ, 'velodyne_points', 'data', f'{7:010d}.bin'))
    seen = 0
    with pytest.raises((FileNotFoundError, OSError)):
        for obs in PrefetchingLoader(Kitti360Dataloader(root, 1, [SEQ], [0], [n_frames]), depth=6):
            seen += 1
    assert seen == 7


def test_kitti_accumulator_voxel_dedup_option(golden):
    """Opt-in extension: acc.voxel_dedup = size de-duplicates the buffer after every integrate().  Model: the same
    sequence WITHOUT the option gives every point's coordinates (transforms are per point, so survivors are bitwise
    identical); a numpy survivor mask (first point per voxel in store order) is carried from step to step."""
    from PIL import Image

    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    g = golden('kitti_accum')
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': g['P']}
    size = 0.4

    def make():
        a = Kitti360SemanticPointCloudAccumulator(1e9, calib, 1e3, 'fake.onnx', KITTI_FILTERS, SEM_IDXS, False,
                                                  dict(BEV_KITTI))
        q = list(g['Ts'])
        a.pose_provider = lambda pc: q.pop(0)
        return a
    plain, dd = make(), make()
    assert dd.voxel_dedup is None                       # off by default
    dd.voxel_dedup = size
    alive = []                                           # per frame: survivor mask over the plain frame's rows
    n_steps = min(int(g['F']), 8)
    for k in range(n_steps):
        obs = [(Image.fromarray(g[f'img_{k}']), g[f'pc_{k}'], None)]
        plain.integrate(obs)
        dd.integrate(obs)
        frames = plain.sem_pcs
        alive.append(np.ones(frames[-1].shape[0], bool))
        cand = np.concatenate([f[m] for f, m in zip(frames, alive)])
        vox = np.floor(cand[:, :3] / size).astype(np.int64)
        _, first = np.unique(vox, axis=0, return_index=True)
        keep = np.zeros(len(cand), bool)
        keep[first] = True
        pos = 0
        for j, m in enumerate(alive):
            cnt = int(m.sum())
            idx = np.flatnonzero(m)
            m[idx[~keep[pos:pos + cnt]]] = False
            pos += cnt
        got = dd.sem_pcs
        assert len(got) == len(frames)
        for j, (f, m) in enumerate(zip(frames, alive)):
            assert np.array_equal(got[j], f[m]), (k, j)
    assert sum(int(m.sum()) for m in alive) < sum(len(m) for m in alive)
    bevs = dd.generate_bev(n_steps // 2, 1, gen_future=True)     # the rasteriser runs on the thinned buffer
    assert bevs[0]['road_full'].shape == (32, 32)
    dd.store.check_status()


def test_nuscenes_dataloader_projection_chain_on_device(golden):
    """obs_dataloaders/nuscenes_obs_dataloader.py:162-202 through the drop-in loader: fake dataset tables and sweep
    provider (dataset walking is out of scope), the reference's transform chain + 6-camera projection + last-camera-
    wins from the golden vectors."""
    from types import SimpleNamespace

    from obs_dataloaders.nuscenes_obs_dataloader import NuScenesDataloader
    g = golden('utils')
    n = g['c6_pc'].shape[0]
    rng = np.random.default_rng(4)
    sweep = np.zeros((n, 8))
    sweep[:, :3] = g['c6_pc']
    sweep[:, 3] = rng.integers(0, 256, n)
    sweep[:, 6] = rng.integers(-1, 3, n)

    class FakeNusc:
        scene = [{'first_sample_token': 's0'}]
        tables = {('sample', 's0'): {'next': 's1', 'scene_token': 'sc', 'data': {'LIDAR_TOP': 'l0'}},
                  ('sample', 's1'): {'next': '', 'scene_token': 'sc', 'data': {'LIDAR_TOP': 'l1'}},
                  ('sample_data', 'l0'): {'ego_pose_token': 'e0'}, ('sample_data', 'l1'): {'ego_pose_token': 'e0'},
                  ('ego_pose', 'e0'): {'translation': [411.5, 1180.25, 0.0]}}

        def get(self, table, token):
            return self.tables[(table, token)]

    class Loader(NuScenesDataloader):
        def _lidar(self, sample):
            return SimpleNamespace(ego_from_self=g['c6_ego_from_lidar'], glob_from_ego=g['c6_glob_from_ego'])

        def _cameras(self, sample):
            return [SimpleNamespace(img=f'img{j}', glob_from_self=g['c6_glob_from_cam'][j], cam_K=g['pp_K'],
                                    img_wh=g['pp_wh']) for j in range(6)]

    seen = {}

    def provider(nusc, token, **cfg):
        seen.update(cfg, token=token)
        return {'points': sweep, 'instances_token': ['a', 'b'], 'instances_name': [np.int64(0), np.int64(3)],
                'instances_center': [np.zeros(3), np.ones(3)]}

    loader = Loader(FakeNusc(), scene_ids=[0], batch_size=1, num_sweeps=5)
    loader.sweep_provider = provider
    assert len(loader) == 2 and loader.sample_tokens == ['s0', 's1']
    obs = next(iter(loader))[0]
    assert seen['token'] == 's0' and seen['n_sweeps'] == 5 and seen['map_point_feat2idx']['inst_idx'] == 6
    assert np.array_equal(obs['pc'][:, :3], g['c6_pc_in_ego'])
    assert np.array_equal(obs['pc'][:, 4:6], g['c6_uv'])
    assert np.array_equal(obs['pc_cam_idx'], g['c6_cam_idx'])
    assert np.array_equal(obs['pc'][:, 3], sweep[:, 3]) and np.array_equal(obs['pc'][:, 6], sweep[:, 6])
    assert obs['pc'].shape == (n, 7) and obs['images'] == [f'img{j}' for j in range(6)]
    assert obs['inst_cls'] == [0, 3] and obs['inst_tokens'] == ['a', 'b']
    assert obs['ego_global_x'] == 411.5 and obs['ego_global_y'] == 1180.25
    assert np.array_equal(obs['ego_at_lidar_ts'], g['c6_glob_from_ego'])
    assert set(obs) == {'meta', 'ego_at_lidar_ts', 'images', 'pc_cam_idx', 'pc', 'inst_tokens', 'inst_cls', 'inst_center',
                        'ego_global_x', 'ego_global_y'}
    with pytest.raises(TypeError):
        NuScenesDataloader(FakeNusc())                       # scene_ids=None: range(list), as in the reference


# ---------------------------------------------------------------------------------------------------------------
#  round-2 additions: device warp, all-sets window, rotation check, pose source, velo2img
# ---------------------------------------------------------------------------------------------------------------
def _bev_e_inputs(g):
    present, future = g['pc_present'], g['pc_future']
    pcs = dict(pc_present=present.copy(), pc_future=future.copy(), pc_full=np.concatenate([present, future]))
    trajs = {}
    for k in ('ego_traj_present', 'ego_traj_future', 'ego_traj_full'):
        trajs[k] = g['in_' + k].copy()
    for k in ('other_trajs_present', 'other_trajs_future', 'other_trajs_full'):
        trajs[k] = [g[f'in_{k}_{i}'].copy() for i in range(int(g[f'in_{k}_n']))]
    return pcs, trajs


def test_device_only_with_warp_is_warped_on_the_device(golden):
    """device_only=True + do_warp=True: the planes that stay in HBM are the reference's warped planes (pca_bev_warp),
    also when the caller names the output tensor; trajectories are warped too."""
    import torch

    from bev_generator.sem_bev import SemBEVGenerator
    g = golden('bev_e')
    for use_out in (False, True):
        gen = SemBEVGenerator(SEM_IDXS, 20, 32, 0., 0., True, 20., 20., 0.5, None)
        w = tuple(g['warp'])
        gen.get_random_warp_params = lambda *a: w
        pcs, trajs = _bev_e_inputs(g)
        out = torch.zeros((21, 32, 32), dtype=torch.float16, device='cuda') if use_out else None
        res = gen.generate(pcs, trajs, device_only=True, out=out)
        p16 = res['planes_f16']
        if use_out:
            assert p16.data_ptr() == out.data_ptr()
        bev = SemBEVGenerator.pack_bev(p16.cpu().numpy(), res['trajs_present'], res['trajs_future'], res['trajs_full'])
        check_bev(bev, g)


def test_warp_kernel_matches_numpy_gather(golden):
    """pca_bev_warp on random planes and coefficient pairs against the vectorised numpy form (itself pinned on the
    reference's loop by utils.npz: wp_in / wp_out)."""
    import torch

    from bev_generator.sem_bev import SemBEVGenerator
    from pca_amd import host_logic as hl
    rng = np.random.default_rng(4)
    for px, iw, jw in ((32, 19.7, 13.8), (64, 25.0, 40.5), (256, 140.2, 111.9), (31, 12.2, 18.8)):
        a1, a2 = hl.cal_warp_params(iw, int(px / 2), px - 1)
        b1, b2 = hl.cal_warp_params(jw, int(px / 2), px - 1)
        planes = rng.random((5, px, px))
        want = hl.warp_dense_probmaps(planes, a1, a2, b1, b2).astype(np.float16)
        got = SemBEVGenerator.warp_planes_device(torch.from_numpy(planes.astype(np.float16)).cuda(), a1, a2, b1, b2)
        assert np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16)), px


def test_bev_rejects_a_rotation_that_is_not_about_z():
    import torch

    from pca_amd import host_logic as hl
    from pca_amd.device_store import DeviceStore, make_bev_params
    st = DeviceStore(capacity=1024, max_frames=4)
    rows = np.zeros((10, 10))
    rows[:, :3] = np.random.default_rng(0).uniform(-5, 5, (10, 3))
    st.load_rows([rows])
    c, s = np.cos(0.3), np.sin(0.3)
    tilt = np.array([[1, 0, 0], [0, c, -s], [0, s, c]]) @ hl.rotation_matrix_3d(0.2)
    args = ([0., 0., 0.], tilt, 0., 0., 20., 16, None, 20., 20., 0.5, 0, [13], False)
    with pytest.raises(RuntimeError, match='rotation about the z axis'):
        st.bev(1, make_bev_params(*args))
    args = ([0., 0., 0.], hl.rotation_matrix_3d(0.2), 0., 0., 20., 16, None, 20., 20., 0.5, 0, [13], False)
    p16, _ = st.bev(1, make_bev_params(*args))
    assert tuple(p16.shape) == (21, 16, 16) and torch.isfinite(p16.float()).all()


def test_generate_bev_without_present_idx_uses_the_whole_window_for_every_set(golden):
    """generate_bev(present_idx=None, gen_future=True): the reference slices [:None] and [None:], so present, future and
    full are all the whole window (and all poses)."""
    from PIL import Image

    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    g = golden('kitti_gtsem')
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': g['P']}
    acc = Kitti360SemanticPointCloudAccumulator(50., calib, 1e3, None, KITTI_FILTERS, SEM_IDXS, True, dict(BEV_KITTI))
    queue = list(g['Ts'])
    acc.pose_provider = lambda pc: queue.pop(0)
    dummy = Image.fromarray(np.zeros((64, 96, 3), np.uint8))
    for k in range(4):
        acc.integrate([(dummy, g[f'pc_{k}'], g[f'sem_gt_{k}'])])
    bev_all = acc.generate_bev(None, 1, gen_future=True)[0]
    for key in ('road', 'intensity', 'rgb', 'dynamic', 'elevation'):
        for s in ('present', 'future'):
            assert np.array_equal(bev_all[f'{key}_{s}'].view(np.uint16), bev_all[f'{key}_full'].view(np.uint16)), (key, s)
    assert (bev_all['road_full'] != np.float16(0.5)).sum() > 10          # the window is in view
    n = len(acc.poses)
    assert len(bev_all['trajs_present']) == len(bev_all['trajs_future']) == len(bev_all['trajs_full']) == 1
    # 'full' of an ordinary call with the same origin and heading holds the same points: present_idx = -1 is not the
    # same call (its present set ends one frame early), so the equality is checked against the host-array path
    from bev_generator.sem_bev import SemBEVGenerator
    rows = np.concatenate(acc.sem_pcs)
    rows[:, :3] -= np.array(acc.poses[-1])
    rel = np.array(acc.poses) - np.array(acc.poses[-1])
    gen = SemBEVGenerator(SEM_IDXS, BEV_KITTI['view_size'], BEV_KITTI['pixel_size'], 0., 0., False,
                          BEV_KITTI['int_scaler'], BEV_KITTI['int_sep_scaler'], BEV_KITTI['int_mid_threshold'],
                          BEV_KITTI['height_filter'])
    host = gen.generate(dict(pc_present=rows, pc_future=rows, pc_full=rows),
                        dict(ego_traj_present=rel, ego_traj_future=rel, ego_traj_full=rel, other_trajs_present=[],
                             other_trajs_future=[], other_trajs_full=[]))
    for key in ('road', 'rgb', 'dynamic', 'elevation'):
        assert np.array_equal(bev_all[f'{key}_full'].view(np.uint16), host[f'{key}_full'].view(np.uint16)), key
    assert n == 4


def test_default_pose_source_is_open3d_and_raises_without_it(monkeypatch):
    import builtins

    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    monkeypatch.delenv('PCA_POSE_PROVIDER', raising=False)
    monkeypatch.delenv('PCA_KITTI_T_FILE', raising=False)
    real_import = builtins.__import__

    def no_open3d(name, *a, **k):
        if name == 'open3d':
            raise ImportError('no open3d')
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, '__import__', no_open3d)
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': np.eye(4)[:3]}
    acc = Kitti360SemanticPointCloudAccumulator(8., calib, 1e3, None, KITTI_FILTERS, SEM_IDXS, True, dict(BEV_KITTI))
    with pytest.raises(ImportError, match='PCA_POSE_PROVIDER=gpu_icp'):
        acc.pose_provider(np.zeros((10, 4), np.float32))


def test_velo2img_matches_reference(golden):
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    g = golden('k1')
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': g['P']}
    acc = Kitti360SemanticPointCloudAccumulator(8., calib, 1e3, None, KITTI_FILTERS, SEM_IDXS, True, dict(BEV_KITTI))
    assert np.array_equal(acc.velo2img(g['pc'].copy(), g['P'], int(g['H']), int(g['W'])), g['velo2img'])
    assert np.array_equal(acc.velo2img(g['pc2'].copy(), g['P2'], int(g['H']), int(g['W'])), g['velo2img2'])


def test_nuscenes_full_size_scene_matches_oracle_pipeline():
    """BASELINE configs[2] at full size: 40 frames x 34 720 points, 6 x 900x1600 images, two GT
    instances (one moving 0.5 m / frame: retroactive dynamic marking), 256^2 BEV at view 51.2 m with height filter 3 m
    and the NuScenes intensity transform -- the drop-in accumulator against the CPU oracle pipeline."""
    from PIL import Image

    from nuscenes_oracle_sem_pc_accum import NuScenesOracleSemanticPointCloudAccumulator
    from oracle import oracle as orc
    from pca_amd import host_logic as hl
    from pca_amd.tracker import InstanceTracker
    F, n, ncam, H, W = 40, 34_720, 6, 900, 1600
    rng = np.random.default_rng(2024)
    fake = FakeSemSeg()
    img_sets = [rng.integers(0, 256, (ncam, H, W, 3), dtype=np.uint8) for _ in range(2)]
    sem_sets = [np.stack([fake.pred(im)[0, 0] for im in s]).astype(np.uint8) for s in img_sets]
    pil_sets = [[Image.fromarray(im) for im in s] for s in img_sets]
    bev_params = dict(type='sem', view_size=51.2, pixel_size=256, max_trans_radius=0., zoom_thresh=0., do_warp=False,
                      int_scaler=1., int_sep_scaler=30., int_mid_threshold=0.12, height_filter=3.)
    acc = NuScenesOracleSemanticPointCloudAccumulator('fake.onnx', NUSC_FILTERS, SEM_IDXS, False, bev_params, 'boston',
                                                      False, None)
    st = orc.Store(F * n, intensity_div255=True)
    track, tracker, offs, T_global_world = hl.PoseTrack(), InstanceTracker(), [0], None
    for k in range(F):
        a = 0.002 * k
        T = np.eye(4)
        T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        T[:3, 3] = [1000. + 1.0 * k, 500., 0.]
        pc = np.stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-2, 4, n),
                       rng.integers(0, 256, n).astype(float), rng.uniform(1.01, W - 1.01, n),
                       rng.uniform(1.01, H - 1.01, n), rng.integers(-1, 5, n).astype(float)], 1)
        cam = rng.integers(-1, ncam, n)
        tokens, cls = ['parked', 'moving'], [0, 0]
        centers = [np.array([1010., 505., 0.5]), np.array([1005. + 0.5 * k, 495., 0.5])]
        obs = dict(images=pil_sets[k % 2], pc=pc, pc_cam_idx=cam, ego_at_lidar_ts=T, ego_global_x=T[0, 3],
                   ego_global_y=T[1, 3], inst_tokens=tokens, inst_cls=cls, inst_center=centers)
        assert acc.integrate([obs]) is None
        # ---- the oracle pipeline, as tests/test_pipeline_golden.py::test_nuscenes_oracle_integrate_tracker_and_bev ----
        if T_global_world is None:
            T_global_world = np.linalg.inv(T)
        T_ego_world = T_global_world @ T
        pose = T_ego_world[:3, -1].tolist()
        pose[2] += 1.
        m = orc.nusc_sample_filter_transform(st, pc, cam, img_sets[k % 2], sem_sets[k % 2], T_ego_world, NUSC_FILTERS)
        offs.append(offs[-1] + m)
        track.append(pose)
        for ts, inst_idx in tracker.observe(k, tokens, cls, [orc.homo_transform(T_global_world, c[None])[0] for c in centers]):
            orc.mark_dynamic(st, offs[ts], offs[ts + 1], inst_idx)
        if len(track.poses) > 1:
            track.push_segment()
    acc.store.check_status()
    sizes = np.diff(offs)
    assert np.array_equal(acc.store.sizes(), sizes) and 20_000 < sizes.mean() < 27_000
    rows = acc.store.rows()
    assert np.array_equal(rows, st.rows())
    assert rows[:, 9].sum() > 1000 and tracker.dynamic == acc.dyn_instances == ['moving']     # earlier frames marked too
    assert np.array_equal(np.array(acc.poses), np.array(track.poses))
    pi = 20
    bev = acc.generate_bev(pi, 1, gen_future=True)[0]
    origin = np.array(track.poses[pi])
    poses = np.array(track.poses)
    R = hl.rotation_matrix_3d(hl.heading_rot_ang(poses[:pi] - origin))
    prm = orc.make_bev_params(origin, R, 0., 0., 51.2, 256, 3., 1., 30., 0.12, 0, [13, 14, 15, 17], True)
    ref = orc.bev(st, int(sizes[:pi].sum()), prm)['f16']
    for s, name in enumerate(('present', 'future', 'full')):
        for k, key in ((0, 'road'), (5, 'dynamic'), (6, 'elevation')):
            assert np.array_equal(bev[f'{key}_{name}'].view(np.uint16), ref[7 * s + k].view(np.uint16)), (key, name)
        assert np.array_equal(bev[f'rgb_{name}'].view(np.uint16), ref[7 * s + 2:7 * s + 5].view(np.uint16)), name
        d = np.abs(bev[f'intensity_{name}'].view(np.uint16).astype(int) - ref[7 * s + 1].view(np.uint16).astype(int))
        assert d.max() <= 1
    assert (bev['road_full'] != np.float16(0.5)).mean() > 0.9                          # the view is covered
    others = tracker.split_trajectories(pi)
    assert len(bev['trajs_full']) == 1 + len(others[2])


def test_semseg_wrapper_keeps_the_class_map_on_the_device(monkeypatch):
    """utils/onnx_utils.SemSegONNX with a GPU execution provider (onnxruntime faked: tests/fake_ort.py): normalisation on
    the device == the reference's ToTensor + Normalize arithmetic, IOBinding on device pointers, a DeviceMap out whose
    cuda tensor feeds K1 without a host round trip; without a GPU provider it is the reference's numpy path."""
    import sys

    import torch

    import fake_ort
    from PIL import Image
    monkeypatch.setitem(sys.modules, 'onnxruntime', fake_ort)
    from utils import onnx_utils
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    # host path (CPU provider only): numpy (1,1,H,W), as the reference
    monkeypatch.setitem(fake_ort.STATE, 'providers', ['CPUExecutionProvider'])
    host_model = onnx_utils.SemSegONNX('model.onnx')
    assert host_model.keep_on_device is False and not host_model.accepts_device
    want = host_model.pred(Image.fromarray(img))
    assert isinstance(want, np.ndarray) and want.shape == (1, 1, 64, 96) and want.dtype == np.int64
    # the normalisation is the reference's: ToTensor (x / 255) then Normalize ((x - mean) / std), all f32
    x = onnx_utils.SemSegONNX.input_preproc(img)
    ref = ((img.astype(np.float32) / np.float32(255.)) - np.float32(onnx_utils.MEAN)) / np.float32(onnx_utils.STD)
    assert np.array_equal(x, np.transpose(ref, (2, 0, 1))) and x.dtype == np.float32
    xd = onnx_utils.SemSegONNX.input_preproc_device(torch.from_numpy(img).cuda())
    assert np.array_equal(xd.cpu().numpy()[0], x)
    # device path
    monkeypatch.setitem(fake_ort.STATE, 'providers', ['ROCMExecutionProvider', 'CPUExecutionProvider'])
    model = onnx_utils.SemSegONNX('model.onnx')
    model.ort_session_semseg.owner = model
    assert model.keep_on_device and model.accepts_device
    before = fake_ort.STATE['bound_runs']
    got = model.pred(torch.from_numpy(img).cuda())
    assert fake_ort.STATE['bound_runs'] == before + 1
    assert isinstance(got, onnx_utils.DeviceMap) and got.shape == (1, 1, 64, 96) and got.dev.is_cuda
    assert np.array_equal(np.asarray(got), want) and np.array_equal(np.asarray(got[0, 0]), want[0, 0])
    # through the accumulator: the map reaches K1 as the cuda tensor it already is
    import sem_pc_accum
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    monkeypatch.setattr(sem_pc_accum, 'SemSegONNX', lambda path: model)
    P = np.array([[40., 0, 48, 0], [0, 40., 32, 0], [0, 0, 1, 0]]) @ np.linalg.inv(np.array(
        [[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418], [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
         [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824], [0, 0, 0, 1]]))
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': P}
    acc = Kitti360SemanticPointCloudAccumulator(50., calib, 1e3, 'model.onnx', KITTI_FILTERS, SEM_IDXS, False, dict(BEV_KITTI))
    acc.pose_provider = lambda pc: np.eye(4)
    pc = np.stack([rng.uniform(-20, 20, 3000), rng.uniform(-20, 20, 3000), rng.uniform(-2, 2, 3000), rng.uniform(0, 1, 3000)],
                  1).astype(np.float32)
    acc.integrate([(Image.fromarray(img), pc, None)])
    rows = acc.sem_pcs[0]
    from oracle import oracle as orc
    ost = orc.Store(3000)
    orc.kitti_project_sample_filter(ost, pc, P, img, want[0, 0].astype(np.uint8), None, 64, 96, KITTI_FILTERS)
    assert np.array_equal(rows, ost.rows()) and rows.shape[0] > 100
    assert isinstance(acc.get_semseg(0)[0], onnx_utils.DeviceMap)
    # the ONE-CALL path (a device image in: pca_kitti_integrate) takes the DeviceMap's cuda tensor as it is -- no copy of the
    # class map to the host and back up (counted: DeviceMap.__array__ is the only way down) -- and a DeviceImage whose upload
    # has not happened hands over its host array without its lazy `dev` property starting a pageable upload
    from pca_amd.ingest import DeviceImage
    downs, lazy_ups = [], []
    real_array = onnx_utils.DeviceMap.__array__
    monkeypatch.setattr(onnx_utils.DeviceMap, '__array__', lambda self, *a, **k: (downs.append(1), real_array(self, *a, **k))[1])
    real_dev = DeviceImage.dev.fget
    monkeypatch.setattr(DeviceImage, 'dev', property(lambda self: (lazy_ups.append(self._dev is None), real_dev(self))[1]))
    assert acc._fast
    acc.integrate([(torch.from_numpy(img).cuda(), pc, None)])                       # cuda tensor in
    acc.integrate([(DeviceImage(img, torch.from_numpy(img).cuda()), pc, None)])     # DeviceImage, copy already there
    assert downs == [] and lazy_ups == []
    rows2 = acc.sem_pcs
    assert np.array_equal(rows2[1], ost.rows()) and np.array_equal(rows2[2], ost.rows())
    host_only = DeviceImage(img, None)                                              # DeviceImage without a device copy
    acc.semseg_model = type('HostModel', (), {'pred': lambda self, rgb: want})()    # (a host model: the image stays a host array)
    acc.integrate([(host_only, pc, None)])
    assert lazy_ups == [] and host_only._dev is None
    assert np.array_equal(acc.sem_pcs[3], ost.rows())


def test_bev_num_batch_equals_single_calls_and_lazy_dicts(golden, monkeypatch, tmp_path):
    """generate_bev(bev_num=4) with random augmentation (rotation / shift / zoom / warp): the four rasters run back to
    back on the device and leave in one asynchronous copy (LazyBev); with the random draws pinned they equal four single
    calls.  A LazyBev pickles as the plain dict; write_compressed_pickle hands it to the background writer."""
    import gzip
    import pickle
    import random

    from PIL import Image

    from bev_generator.sem_bev import LazyBev
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    from pca_amd import writer
    g = golden('kitti_gtsem')
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': g['P']}
    bev_params = dict(BEV_KITTI, max_trans_radius=3., zoom_thresh=0.2, do_warp=True)

    def build():
        acc = Kitti360SemanticPointCloudAccumulator(50., calib, 1e3, None, KITTI_FILTERS, SEM_IDXS, True, dict(bev_params))
        queue = list(g['Ts'])
        acc.pose_provider = lambda pc: queue.pop(0)
        dummy = Image.fromarray(np.zeros((64, 96, 3), np.uint8))
        for k in range(4):
            acc.integrate([(dummy, g[f'pc_{k}'], g[f'sem_gt_{k}'])])
        return acc
    # generate_rand_aug reseeds from (pid + worker) * int(time): freeze the clock so that the draws can be repeated
    import bev_generator.bev_generator as bg
    monkeypatch.setattr(bg.time, 'time', lambda: 1700000000.25)
    acc = build()
    random.seed(5)
    batch = acc.generate_bev(2, 4, gen_future=True)
    assert len(batch) == 4 and all(isinstance(b, LazyBev) for b in batch)
    random.seed(5)
    monkeypatch.setenv('PCA_SYNC_BEV', '1')                          # the one-after-the-other form
    singles = []
    for k in range(4):
        acc._aug_worker0 = k                                         # sample k of the batch = worker k
        singles.append(acc.generate_bev(2, 1, gen_future=True)[0])
    acc._aug_worker0 = 0
    monkeypatch.delenv('PCA_SYNC_BEV')
    assert not any(isinstance(b, LazyBev) for b in singles)
    for b, s in zip(batch, singles):
        assert set(b.keys()) == set(s.keys())
        for key in s:
            if key.startswith('trajs'):
                assert len(b[key]) == len(s[key]) and all(np.array_equal(x, y) for x, y in zip(b[key], s[key]))
            else:
                assert np.array_equal(np.asarray(b[key]).view(np.uint16), np.asarray(s[key]).view(np.uint16)), key
    # the augmentations of one window differ from sample to sample (nothing about the seeding is patched but the clock)
    for i in range(4):
        for j in range(i + 1, 4):
            assert not np.array_equal(np.asarray(batch[i]['road_full']), np.asarray(batch[j]['road_full'])), (i, j)
    # container: a LazyBev pickles as the reference's plain dict; the writer thread produces the same file
    fresh = acc.generate_bev(2, 1, gen_future=True)[0]
    assert isinstance(fresh, LazyBev)
    acc.write_compressed_pickle(fresh, 'bev_000.pkl', str(tmp_path))
    writer.shared_writer().flush()
    with gzip.open(str(tmp_path / 'bev_000.pkl.gz'), 'rb') as f:
        back = pickle.loads(f.read())
    assert type(back) is dict and back['rgb_full'].dtype == np.float16 and back['rgb_full'].shape == (3, 32, 32)
    assert np.array_equal(back['road_present'], fresh['road_present'])


def test_config2_shaped_stream_every_step_against_the_oracle_pipeline():
    """BASELINE configs[1] shape at full frame size: 120 000-point frames with 376x1408 images through the drop-in
    accumulator -- a 60-frame fill (integrate_many), then 14 steps of integrate() + one 256x256 BEV each, so that owed
    chains of 1..4 transforms, the write-back and an eviction all occur with real K1 output in the store.  EVERY step's
    planes and the final stored rows are compared with the oracle pipeline (not with another HIP path)."""
    import torch

    import sem_pc_accum
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    from oracle import oracle as orc
    from pca_amd import host_logic as hl
    H, W, N = 376, 1408, 120_000
    cam_to_velo = np.array([[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418],
                            [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
                            [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824], [0, 0, 0, 1]])
    P = np.array([[552.554261, 0, 682.049453, 0], [0, 552.554261, 238.769549, 0], [0, 0, 1, 0]]) @ np.linalg.inv(cam_to_velo)

    def frame(k):
        rng = np.random.default_rng(4000 + k)
        pc = np.stack([rng.uniform(-60, 60, N), rng.uniform(-60, 60, N), rng.uniform(-2, 3, N), rng.uniform(0, 1, N)],
                      1).astype(np.float32)
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        sem = rng.integers(0, 19, (H, W)).astype(np.uint8)
        sem[rng.random((H, W)) < 0.01] = 255
        return pc, img, sem
    pool = [frame(k) for k in range(3)]
    dev_pool = [(torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda(), torch.from_numpy(s).cuda()) for p, i, s in pool]

    class Resident:
        def pred(self, rgb):
            return by_ptr[rgb.data_ptr()][None, None]
    by_ptr = {d[0].data_ptr(): d[2] for d in dev_pool}
    sem_pc_accum.SemSegONNX = lambda path: Resident()
    a = -0.004
    T = np.array([[np.cos(a), -np.sin(a), 0, 0], [np.sin(a), np.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.]]) @ \
        np.array([[1, 0, 0, -1.0], [0, 1, 0, 0.01], [0, 0, 1, 0.002], [0, 0, 0, 1.]])
    horizon, bev_h, view, px = 70.0, 30.0, 80, 256
    bev_params = dict(BEV_KITTI, view_size=view, pixel_size=px)
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': P}
    acc = Kitti360SemanticPointCloudAccumulator(horizon, calib, 1e3, 'resident', KITTI_FILTERS, SEM_IDXS, False, bev_params)
    acc._store_args = dict(capacity=1 << 23, max_frames=256)
    acc.pose_provider = lambda pc: T
    ost = orc.Store(80 * 40000)
    track = hl.PoseTrack()
    sizes, lo, removed_dev, removed_orc = [], 0, [], []

    def oracle_step(k):
        nonlocal sizes, lo
        pc, img, sem = pool[k % 3]
        if len(track):
            track.apply_transform(T)
            orc.retransform(ost, T, lo, ost.n)
        sizes.append(orc.kitti_project_sample_filter(ost, pc, P, img, sem, None, H, W, KITTI_FILTERS))
        track.append([0., 0., 0.])
        ev = 0
        if len(track) > 1:
            ev = track.evict_beyond(horizon, track.push_segment())
            lo += int(np.sum(sizes[:ev]))
            sizes = sizes[ev:]
        removed_orc.append(ev)
    fill = 60
    removed_dev += acc.integrate_many([[(dev_pool[k % 3][0], dev_pool[k % 3][1], None)] for k in range(fill)])
    for k in range(fill):
        oracle_step(k)
    chains = set()
    for k in range(fill, fill + 14):
        removed_dev.append(acc.integrate([(dev_pool[k % 3][0], dev_pool[k % 3][1], None)]))
        oracle_step(k)
        if k == fill + 6:                                   # a step without a raster: the chain is one longer next time
            continue
        d = acc.get_incremental_path_dists()
        pidx = int(((d - bev_h) > 0).argmax())
        chains.add(len(acc.store._pending))
        bev = acc.generate_bev(pidx, 1, gen_future=True)[0]
        origin = np.array(track.poses[pidx])
        R = hl.rotation_matrix_3d(hl.heading_rot_ang(np.array(track.poses[:pidx]) - origin))
        prm = orc.make_bev_params(origin, R, 0., 0., view, px, None, 20., 20., 0.5, 0, [13, 14, 15, 17], False)
        sub = orc.Store(1)
        for name in ('x', 'y', 'z', 'intensity', 'rgbs', 'inst', 'dyn'):
            setattr(sub, name, getattr(ost, name)[lo:ost.n])
        sub.n = sub.cap = ost.n - lo
        F = orc.bev(sub, int(np.sum(sizes[:pidx])), prm)['f16']
        for s, name in enumerate(('present', 'future', 'full')):
            for key, pl in (('road', 0), ('dynamic', 5), ('elevation', 6)):
                assert np.array_equal(bev[f'{key}_{name}'].view(np.uint16), F[7 * s + pl].view(np.uint16)), (k, key, name)
            assert np.array_equal(bev[f'rgb_{name}'].view(np.uint16), F[7 * s + 2:7 * s + 5].view(np.uint16)), (k, name)
            di = np.abs(bev[f'intensity_{name}'].view(np.uint16).astype(int) - F[7 * s + 1].view(np.uint16).astype(int))
            assert di.max() <= 1 and (di != 0).mean() < 1e-3, (k, name)
    assert chains >= {1, 2, 3, 4}, chains                   # every chain length was rasterised at least once
    assert removed_dev == removed_orc and sum(removed_dev) >= 1
    assert np.array_equal(np.concatenate(acc.sem_pcs), ost.rows(lo))
    assert np.array_equal(np.array(acc.poses), np.array(track.poses))
    acc.store.check_status()


def test_nuscenes_integrate_many_equals_integrate_call_by_call(golden):
    """integrate_many of the NuScenes class (one batched K1n front + append launch, one K3 launch for all the tracker's
    marks) leaves the state call-by-call integrate() leaves: the golden scene with retroactive dynamic marking, once as a
    whole and once split 3 + 4, rows incl. the dynamic flags, poses, tracker, and the BEV."""
    from PIL import Image

    from nuscenes_oracle_sem_pc_accum import NuScenesOracleSemanticPointCloudAccumulator
    g = golden('nusc_oracle')
    F = int(g['F'])

    def obs_of(k):
        T = g[f'T_{k}']
        return [dict(images=[Image.fromarray(im) for im in g[f'imgs_{k}']], pc=g[f'pc_{k}'], pc_cam_idx=g[f'cam_idx_{k}'],
                     ego_at_lidar_ts=T, ego_global_x=T[0, 3], ego_global_y=T[1, 3],
                     inst_tokens=str(g['inst_tokens'][k]).split(','), inst_cls=list(g[f'inst_cls_{k}']),
                     inst_center=list(g[f'inst_center_{k}']))]
    for cuts in ([F], [3, F - 3], [1, 1, F - 2]):
        acc = NuScenesOracleSemanticPointCloudAccumulator('fake.onnx', NUSC_FILTERS, SEM_IDXS, False, dict(BEV_NUSC),
                                                          'boston', False, None)
        k = 0
        for n in cuts:
            assert acc.integrate_many([obs_of(j) for j in range(k, k + n)]) is None
            k += n
        assert np.array_equal(np.array([a.shape[0] for a in acc.sem_pcs]), g['sizes'])
        assert np.array_equal(np.concatenate(acc.sem_pcs), g['sem_pcs'])          # incl. retroactive dyn flags
        assert np.array_equal(np.array(acc.poses), g['poses'])
        assert np.array_equal(np.array(acc.seg_dists), g['seg_dists'])
        assert list(g['dyn_instances']) == acc.dyn_instances
        assert len(acc.ego_global_xs) == F and acc.ts == F
        bev = acc.generate_bev(int(g['present_idx']), 1, gen_future=True)[0]
        check_bev(bev, g)
        acc.store.check_status()


def test_nuscenes_batched_k1n_full_size_ragged_frames_match_oracle():
    """pca_nusc_sample_filter_transform_batch at BASELINE configs[2] frame size (34 720 points, 6 x 900x1600 images) with
    ragged frames -- empty, one point, around the 512-point tile -- each with its own image stack and T_ego_world: every
    frame's rows equal the oracle's."""
    import torch
    from oracle import oracle as orc
    from pca_amd.device_store import DeviceStore
    rng = np.random.default_rng(31)
    ncam, H, W = 6, 900, 1600
    sizes = [34720, 0, 1, 511, 512, 513, 34720, 2049]
    stacks = [(rng.integers(0, 256, (ncam, H, W, 3), dtype=np.uint8), rng.integers(0, 19, (ncam, H, W)).astype(np.uint8))
              for _ in range(2)]
    frames, host = [], []
    for k, n in enumerate(sizes):
        pc = np.stack([rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), rng.uniform(-2, 4, n),
                       rng.integers(0, 256, n).astype(float), rng.uniform(1.01, W - 1.01, n), rng.uniform(1.01, H - 1.01, n),
                       rng.integers(-1, 5, n).astype(float)], 1).reshape(n, 7)
        cam = rng.integers(-1, ncam, n)
        a = 0.01 * k
        T = np.eye(4)
        T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        T[:3, 3] = [1.0 * k, -0.3 * k, 0.02 * k]
        imgs, sems = stacks[k % 2]
        host.append((pc, cam, imgs, sems, T))
    dev_stacks = [(torch.from_numpy(i).cuda(), torch.from_numpy(s).cuda()) for i, s in stacks]
    for k, (pc, cam, imgs, sems, T) in enumerate(host):
        frames.append(dict(pc=torch.from_numpy(pc).cuda(), cam_idx=torch.from_numpy(cam).cuda(), imgs=dev_stacks[k % 2][0],
                           sems=dev_stacks[k % 2][1], T=T))
    st = DeviceStore(capacity=sum(sizes) + 16, max_frames=16, intensity_div255=True)
    st.append_nusc_many(frames[:3], NUSC_FILTERS)
    st.append_nusc_many(frames[3:], NUSC_FILTERS)
    st.check_status()
    got = st.frame_rows()
    assert len(got) == len(sizes)
    for (pc, cam, imgs, sems, T), rows in zip(host, got):
        ost = orc.Store(max(pc.shape[0], 1), intensity_div255=True)
        orc.nusc_sample_filter_transform(ost, pc, cam, imgs, sems, T, NUSC_FILTERS)
        assert np.array_equal(rows, ost.rows())


def test_generate_bev_many_equals_single_calls(golden):
    """generate_bev_many(idxs): all samples of a sweep over present_idx in ONE launch of each raster kernel and one
    asynchronous copy -- every dict equal to generate_bev(idx, 1, gen_future=True)[0] (planes bit for bit, trajectories),
    for the NuScenes class (other agents' trajectories, dynamic flags) and for the KITTI class; bev_num copies go through
    the same batched launch."""
    from PIL import Image

    from bev_generator.sem_bev import LazyBev
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    from nuscenes_oracle_sem_pc_accum import NuScenesOracleSemanticPointCloudAccumulator
    g = golden('nusc_oracle')
    F = int(g['F'])
    acc = NuScenesOracleSemanticPointCloudAccumulator('fake.onnx', NUSC_FILTERS, SEM_IDXS, False, dict(BEV_NUSC), 'boston',
                                                      False, None)
    batch = []
    for k in range(F):
        T = g[f'T_{k}']
        batch.append([dict(images=[Image.fromarray(im) for im in g[f'imgs_{k}']], pc=g[f'pc_{k}'], pc_cam_idx=g[f'cam_idx_{k}'],
                           ego_at_lidar_ts=T, ego_global_x=T[0, 3], ego_global_y=T[1, 3],
                           inst_tokens=str(g['inst_tokens'][k]).split(','), inst_cls=list(g[f'inst_cls_{k}']),
                           inst_center=list(g[f'inst_center_{k}']))])
    acc.integrate_many(batch)
    idxs = list(range(1, F))
    many = acc.generate_bev_many(idxs, gen_future=True)
    assert len(many) == len(idxs) and all(isinstance(b, LazyBev) for b in many)

    def same(b, s):
        assert set(b.keys()) == set(s.keys())
        for key in s:
            if key.startswith('trajs') or key == 'gt_lanes':
                assert len(b[key]) == len(s[key]) and all(np.array_equal(x, y) for x, y in zip(b[key], s[key])), key
            else:
                assert np.array_equal(np.asarray(b[key]).view(np.uint16), np.asarray(s[key]).view(np.uint16)), key
    for idx, b in zip(idxs, many):
        same(b, acc.generate_bev(idx, 1, gen_future=True)[0])
    check_bev(many[idxs.index(int(g['present_idx']))], g)
    # KITTI class + bev_num copies
    gk = golden('kitti_gtsem')
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': gk['P']}
    kacc = Kitti360SemanticPointCloudAccumulator(50., calib, 1e3, None, KITTI_FILTERS, SEM_IDXS, True, dict(BEV_KITTI))
    queue = list(gk['Ts'])
    kacc.pose_provider = lambda pc: queue.pop(0)
    dummy = Image.fromarray(np.zeros((64, 96, 3), np.uint8))
    for k in range(4):
        kacc.integrate([(dummy, gk[f'pc_{k}'], gk[f'sem_gt_{k}'])])
    km = kacc.generate_bev_many([1, 2, 3], gen_future=True)
    for idx, b in zip([1, 2, 3], km):
        same(b, kacc.generate_bev(idx, 1, gen_future=True)[0])
    check_bev(km[1], gk)
    copies = kacc.generate_bev(2, 3, gen_future=True)
    for c in copies:
        same(c, km[1])
    # more samples than one launch group holds (48 argument blocks in constant memory): the second group's blocks are
    # fetched from an offset inside the mapped host block
    idxs = [1, 2, 3] * 17
    km = kacc.generate_bev_many(idxs, gen_future=True)
    ref = {idx: kacc.generate_bev(idx, 1, gen_future=True)[0] for idx in (1, 2, 3)}
    assert len(km) == 51
    for idx, b in zip(idxs, km):
        same(b, ref[idx])


def test_nuscenes_prefetching_loader_equals_plain_loader_and_feeds_the_accumulator(golden, monkeypatch):
    """PCA_PREFETCH=1 on the NuScenes loader: read_host on reader threads, the six images through ONE pinned block, K0n and
    the (N,7) rows on the device.  Every observation equals the plain loader's (rows, camera indices, images, metadata), and
    an accumulator fed from it -- frame by frame, and through integrate_many with more batches collected than the loader's
    ring holds -- ends in the state the plain loader produces."""
    from types import SimpleNamespace

    import torch

    from nuscenes_oracle_sem_pc_accum import NuScenesOracleSemanticPointCloudAccumulator
    from obs_dataloaders.nuscenes_obs_dataloader import NuScenesDataloader
    from pca_amd.ingest import DeviceImages
    g = golden('utils')
    n = g['c6_pc'].shape[0]
    W, H = int(g['pp_wh'][0]), int(g['pp_wh'][1])
    n_samples = 7
    rng = np.random.default_rng(14)
    sweeps, images = [], []
    for k in range(n_samples):
        sw = np.zeros((n, 8))
        sw[:, :3] = g['c6_pc'] + rng.normal(0, 0.05, (n, 3))
        sw[:, 3] = rng.integers(0, 256, n)
        sw[:, 6] = rng.integers(-1, 3, n)
        sweeps.append(sw)
        images.append([rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(6)])
    toks = [f's{k}' for k in range(n_samples)]

    class FakeNusc:
        scene = [{'first_sample_token': 's0'}]

        def get(self, table, token):
            if table == 'sample':
                k = toks.index(token)
                return {'next': toks[k + 1] if k + 1 < n_samples else '', 'scene_token': 'sc', 'data': {'LIDAR_TOP': 'l' + token[1:]}}
            if table == 'sample_data':
                return {'ego_pose_token': 'e' + token[1:]}
            k = int(token[1:])
            return {'translation': [411.5 + 0.8 * k, 1180.25 - 0.1 * k, 0.0]}

    def glob_from_ego(k):
        T = np.array(g['c6_glob_from_ego'])
        T[0, 3] += 0.8 * k
        T[1, 3] -= 0.1 * k
        return T

    class Loader(NuScenesDataloader):
        def _lidar(self, sample):
            k = int(sample['data']['LIDAR_TOP'][1:])
            return SimpleNamespace(ego_from_self=g['c6_ego_from_lidar'], glob_from_ego=glob_from_ego(k))

        def _cameras(self, sample):
            k = int(sample['data']['LIDAR_TOP'][1:])
            return [SimpleNamespace(img=images[k][j], glob_from_self=g['c6_glob_from_cam'][j], cam_K=g['pp_K'], img_wh=g['pp_wh'])
                    for j in range(6)]

    def provider(nusc, token, **cfg):
        k = toks.index(token)
        return {'points': sweeps[k], 'instances_token': ['a', 'b'], 'instances_name': [np.int64(0), np.int64(3)],
                'instances_center': [np.array([5.0 + 0.6 * k, 1.0, 0.5]), np.ones(3)]}

    def make_loader():
        ld = Loader(FakeNusc(), scene_ids=[0], batch_size=1, num_sweeps=5)
        ld.sweep_provider = provider
        return ld
    plain = [b[0] for b in make_loader()]
    monkeypatch.setenv('PCA_PREFETCH', '1')
    fast = [b[0] for b in make_loader()]
    monkeypatch.delenv('PCA_PREFETCH')
    assert len(plain) == len(fast) == n_samples
    for k in range(n_samples):                              # (the ring holds 64 batches: all seven are still intact)
        p, f = plain[k], fast[k]
        assert isinstance(f['images'], DeviceImages) and isinstance(f['pc'], torch.Tensor)
        assert np.array_equal(f['pc'].cpu().numpy(), p['pc']) and np.array_equal(f['pc_cam_idx'].cpu().numpy(), p['pc_cam_idx'])
        assert np.array_equal(f['images'].dev.cpu().numpy(), np.stack(p['images']))
        assert all(np.array_equal(a, b) for a, b in zip(f['images'], p['images']))
        for key in ('meta', 'inst_tokens', 'inst_cls', 'ego_global_x', 'ego_global_y'):
            assert f[key] == p[key]
        assert np.array_equal(f['ego_at_lidar_ts'], p['ego_at_lidar_ts'])

    def run(loader_env, many):
        if loader_env:
            monkeypatch.setenv('PCA_PREFETCH', '1')
        acc = NuScenesOracleSemanticPointCloudAccumulator('fake.onnx', NUSC_FILTERS, SEM_IDXS, False, dict(BEV_NUSC), 'boston',
                                                          False, None)
        ld = make_loader()
        if many:
            acc.integrate_many(list(ld))                    # all seven batches collected first: more than the ring holds
        else:
            for observations in ld:
                acc.integrate(observations)
        if loader_env:
            monkeypatch.delenv('PCA_PREFETCH')
        acc.store.check_status()
        return np.concatenate(acc.sem_pcs), np.array(acc.poses), acc.dyn_instances
    want = run(False, False)
    assert want[0].shape[0] > 1000
    for loader_env, many in ((True, False), (True, True), (False, True)):
        got = run(loader_env, many)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    # a ring smaller than what is collected is refused, not read after it was overwritten
    monkeypatch.setenv('PCA_PREFETCH_RING', '4')
    with pytest.raises(ValueError, match='reuses its device buffers'):
        run(True, True)


def test_staged_upload_ragged_sizes_and_threads():
    """pca_host_stage_h2d behind PinnedUploader: arrays of 0 bytes, 1 byte, a non-multiple of the 128 KB slice and several MB,
    of different dtypes, arrive intact; the pinned blocks are reused over more calls than the ring is deep; several calling
    threads take turns."""
    import threading

    import torch
    from pca_amd.ingest import PinnedUploader
    up = PinnedUploader(torch.device('cuda', 0))
    rng = np.random.default_rng(5)
    shapes = [((0, 4), np.float32), ((1, ), np.uint8), ((131073, ), np.uint8), ((376, 1408, 3), np.uint8), ((120000, 4), np.float32),
              ((33, 7), np.float64), ((5, ), np.int64)]
    for rep in range(7):
        items = []
        for k, (shape, dt) in enumerate(shapes):
            a = (rng.random(shape) * 200).astype(dt)
            items.append((f'k{k}', a if rep % 2 else np.asfortranarray(a) if a.ndim > 1 else a))
        out = up.upload_many(items)
        torch.cuda.synchronize()
        for (_, a), d in zip(items, out):
            assert tuple(d.shape) == a.shape and np.array_equal(d.cpu().numpy(), a)
    stack = [rng.integers(0, 255, (90, 160, 3), dtype=np.uint8) for _ in range(6)]
    assert np.array_equal(up.upload_stack('imgs', stack).cpu().numpy(), np.stack(stack))

    errors = []

    def worker(seed):
        try:
            mine = PinnedUploader(torch.device('cuda', 0))
            r = np.random.default_rng(seed)
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(20):
                    a = r.random((50000 + seed, 3)).astype(np.float32)
                    d = mine('pts', a)
                    torch.cuda.current_stream().synchronize()
                    if not np.array_equal(d.cpu().numpy(), a):
                        errors.append(seed)
        except Exception as e:                           # noqa: BLE001
            errors.append(repr(e))
    threads = [threading.Thread(target=worker, args=(s, )) for s in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def _kitti_gt_accumulator(capacity=None):
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    P = np.array([[40., 0, 48, 0], [0, 40., 32, 0], [0, 0, 1, 0]])
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': P}
    acc = Kitti360SemanticPointCloudAccumulator(50., calib, 1e3, None, KITTI_FILTERS, SEM_IDXS, True, dict(BEV_KITTI))
    acc.pose_provider = lambda pc: np.eye(4)
    if capacity is not None:
        acc._store_args = dict(capacity=capacity, max_frames=8)
    return acc


def _gt_obs(rng, n, intensity_sign=1.0):
    from PIL import Image
    pc = np.stack([rng.uniform(-9, 9, n), rng.uniform(-9, 9, n), rng.uniform(-1, 2, n),
                   intensity_sign * rng.uniform(0.1, 1, n)], 1).astype(np.float32)
    sem = np.zeros((n, 1), np.int64)                       # road: the intensity plane takes every point
    return [(Image.fromarray(np.zeros((64, 96, 3), np.uint8)), pc, sem)]


def test_device_errors_surface_through_the_kitti_dropin_without_check_status(tmp_path):
    """The reference fails synchronously (IndexError in sem_bev.py:543-551, AssertionError in nuscenes_utils.py:191-195).
    Here a kernel raises a status bit; the drop-in must turn it into an exception on its own -- when a sample is looked at,
    when it is handed to the writer, on the next integrate() -- without anybody calling check_status()."""
    import torch
    from pca_amd import writer
    rng = np.random.default_rng(5)
    # (1) an overflowing store: points were dropped -> no sample made from the rest may be handed out.  The raise is seen as
    # soon as the kernel has stored it: by the integrate() that caused it if the kernel is quick, else by the next call or the
    # first look at the sample -- whichever comes first, without anybody asking
    acc = _kitti_gt_accumulator(capacity=64)
    acc.store.ub_tail = -10**9                             # defeat the host-side planner on purpose (as the K1 test does)
    with pytest.raises(RuntimeError, match='overflow'):
        acc.integrate(_gt_obs(rng, 5000))
        bevs = acc.generate_bev(None, 1, gen_future=True)
        bevs[0]['road_present']                            # first access waits for the copy and looks at the status mirror
    assert acc.store.ctx.peek_status() == 0                # raised once, cleared
    # (2) ... and on the next integrate() at the latest, for a driver that never looks at its samples
    acc = _kitti_gt_accumulator(capacity=64)
    acc.store.ub_tail = -10**9
    with pytest.raises(RuntimeError, match='overflow'):
        acc.integrate(_gt_obs(rng, 5000))
        torch.cuda.synchronize()
        acc.store.ub_tail = 0
        acc.integrate(_gt_obs(rng, 10))
    # (3) an error raised by the RASTER (a negative lidar intensity on the f32 path: the exact integer sums assume >= 0):
    # integrate() and generate_bev() only enqueue, the sample's first access waits for its copy and raises
    acc = _kitti_gt_accumulator()
    acc.integrate(_gt_obs(rng, 2000, intensity_sign=-1.0))
    bev = acc.generate_bev(None, 1, gen_future=True)[0]
    with pytest.raises(ValueError, match='negative lidar intensity'):
        bev['intensity_full']
    # (4) the same through the background writer: the error comes back on the thread that talks to the driver
    acc = _kitti_gt_accumulator()
    acc.integrate(_gt_obs(rng, 2000, intensity_sign=-1.0))
    acc.write_compressed_pickle(acc.generate_bev(None, 1, gen_future=True)[0], 'bev_bad.pkl', str(tmp_path))
    with pytest.raises(ValueError, match='negative lidar intensity'):
        writer.flush_shared()
    assert not os.path.exists(os.path.join(str(tmp_path), 'bev_bad.pkl.gz'))
    # (5) a clean run stays silent and costs no synchronisation: the mirror reads zero
    acc = _kitti_gt_accumulator()
    acc.integrate(_gt_obs(rng, 2000))
    assert acc.generate_bev(None, 1, gen_future=True)[0]['road_full'].shape == (32, 32)
    assert acc.store.ctx.peek_status() == 0


def test_one_call_paths_equal_the_general_paths():
    """integrate() / generate_bev() go through ONE library call each (pca_kitti_integrate, pca_kitti_generate_bev) where that
    applies; `_fast = False` takes the general Python path.  Same stored rows, poses, evictions, planes and polylines -- with
    host arrays, device tensors and a mix of both as inputs, through eviction and every state of the owed chain."""
    import torch
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    from pca_amd import host_logic as hl
    rng = np.random.default_rng(21)
    P = np.array([[40., 0, 48, 0], [0, 40., 32, 0], [0, 0, 1, 0]]) @ np.array(
        [[0., -1, 0, 0], [0, 0, -1, 0], [1, 0, 0, 0], [0, 0, 0, 1]])
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': P}
    T = np.eye(4)
    T[:3, :3] = hl.rotation_matrix_3d(-0.01)
    T[:3, 3] = [-1.0, 0.02, 0.0]
    for use_gt in (False, True):
        accs = []
        for fast in (True, False):
            acc = Kitti360SemanticPointCloudAccumulator(9., calib, 1e3, 'fake.onnx', KITTI_FILTERS, SEM_IDXS, use_gt,
                                                        dict(BEV_KITTI))
            acc.pose_provider = lambda pc: T
            acc._fast = fast
            accs.append(acc)
        frames = []
        for k in range(16):
            n = int(rng.integers(1, 3000))
            pc = np.stack([rng.uniform(0.5, 20, n), rng.uniform(-9, 9, n), rng.uniform(-1, 2, n), rng.uniform(0, 1, n)],
                          1).astype(np.float32)
            img = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
            sem_gt = rng.integers(0, 19, (n, 1)).astype(np.int64)
            frames.append((img, pc, sem_gt))
        for k, (img, pc, sem_gt) in enumerate(frames):
            # host arrays, device tensors, and one of each
            if k % 3 == 0:
                obs = (img, pc, sem_gt)
            elif k % 3 == 1:
                obs = (torch.from_numpy(img).cuda(), torch.from_numpy(pc).cuda(), torch.from_numpy(sem_gt[:, -1].astype(np.uint8)).cuda())
            else:
                obs = (torch.from_numpy(img).cuda(), pc, sem_gt)
            if not use_gt and isinstance(obs[0], torch.Tensor):
                obs = (img, ) + obs[1:]                  # (the fake model of these tests reads host images)
            removed = [acc.integrate([obs]) for acc in accs]
            assert removed[0] == removed[1]
            if k >= 3:
                idx = len(accs[0].poses) - 2 - (k % 2)
                a, b = (acc.generate_bev(idx, 1, gen_future=True)[0] for acc in accs)
                assert set(a.keys()) == set(b.keys())
                for key in a.keys():
                    if key.startswith('trajs_'):
                        assert len(a[key]) == len(b[key]) and all(np.array_equal(x, y) for x, y in zip(a[key], b[key])), key
                    else:
                        assert np.array_equal(a[key].view(np.uint16), b[key].view(np.uint16)), (k, key)
                d = accs[0].generate_bev_device(idx)
                assert np.array_equal(d['planes_f16'].cpu().numpy()[0].view(np.uint16), a['road_present'].view(np.uint16))
                assert np.array_equal(d['trajs_full'][0], a['trajs_full'][0])
        assert np.array_equal(np.concatenate(accs[0].sem_pcs), np.concatenate(accs[1].sem_pcs))
        assert np.array_equal(np.array(accs[0].poses), np.array(accs[1].poses))
        assert np.array_equal(np.array(accs[0].seg_dists), np.array(accs[1].seg_dists))
        assert sum(1 for _ in accs[0].rgbs) == len(accs[0].poses)
        accs[0].store.check_status()


def test_one_call_integrate_edge_cases():
    """pca_kitti_integrate with the inputs the reference's loader can produce at the edges: an empty sweep, a single point, a
    one-pixel image (smaller than the 4-byte colour gather), non-contiguous / wrongly typed host arrays (converted like the
    general path converts them), and shapes that do not fit together (refused before anything changes)."""
    import torch
    rng = np.random.default_rng(2)
    accs = []
    for fast in (True, False):
        acc = _kitti_gt_accumulator()
        acc.use_gt_sem = False
        acc.semseg_model = FakeSemSeg()
        acc._fast = fast
        accs.append(acc)
    big = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    cases = [
        (big, np.zeros((0, 4), np.float32)),                                            # empty sweep
        (big, np.array([[5.0, 0.1, 0.2, 0.5]], np.float32)),                            # one point
        (np.full((1, 1, 3), 7, np.uint8), np.array([[5.0, 0.0, 0.0, 0.5]] * 3, np.float32)),   # one-pixel image
        (big[:, ::-1], np.asfortranarray(rng.uniform(1, 9, (500, 4)).astype(np.float32))),      # strided / F-ordered inputs
        (big.astype(np.int64), rng.uniform(1, 9, (300, 4))),                            # int64 image, float64 points
    ]
    for img, pc in cases:
        out = [acc.integrate([(img, pc, None)]) for acc in accs]
        assert out[0] == out[1]
    a, b = (np.concatenate(acc.sem_pcs) for acc in accs)
    assert a.shape[0] > 100 and np.array_equal(a, b)
    assert np.array_equal(np.array(accs[0].poses), np.array(accs[1].poses))
    n_before = accs[0].store.n_frames
    with pytest.raises(ValueError):
        accs[0].integrate([(big, rng.uniform(1, 9, (10, 3)).astype(np.float32), None)])          # (N,3) points
    assert accs[0].store.n_frames == n_before and len(accs[0].poses) == n_before
    accs[0].store.check_status()


def test_k1_riding_in_the_raster_equals_k1_on_its_own():
    """integrate() notes its K1 and the generate_bev() that follows runs it as the first workgroups of the raster's first kernel
    (pca_k1_defer; `_defer_k1 = False`: K1 is a launch of its own inside integrate()).  Same planes, stored rows, offsets and
    polylines, step for step, on full-size frames (30 K1 tiles riding along) and at the edges: a frame of which nothing is
    kept, a one-point frame, two integrate() in a row (the first K1 runs on its own), reading the store between integrate()
    and generate_bev() (the noted K1 runs first), host and device inputs, the image path and the per-point-label path."""
    import torch

    import sem_pc_accum
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    from pca_amd import _lib
    H, W, N = 376, 1408, 120_000
    cam_to_velo = np.array([[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418],
                            [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
                            [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824], [0, 0, 0, 1]])
    P = np.array([[552.554261, 0, 682.049453, 0], [0, 552.554261, 238.769549, 0], [0, 0, 1, 0]]) @ np.linalg.inv(cam_to_velo)
    rng = np.random.default_rng(77)

    def frame(n, spread=60.):
        pc = np.stack([rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n), rng.uniform(-2, 3, n),
                       rng.uniform(0, 1, n)], 1).astype(np.float32)
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        sem = rng.integers(0, 19, (H, W)).astype(np.uint8)
        sem[rng.random((H, W)) < 0.01] = 255
        return pc, img, sem
    frames = [frame(N) for _ in range(3)]
    frames.append(frame(1))                                               # one point
    behind = frame(5000)
    behind[0][:, 0] = -np.abs(behind[0][:, 0]) - 1.0                      # everything behind the camera: nothing kept
    frames.append(behind)
    frames.append(frame(70_000, 20.))                                     # dense near the sensor: most of it in view
    dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda(), torch.from_numpy(s).cuda()) for p, i, s in frames]
    by_ptr = {d[0].data_ptr(): d[2] for d in dev}
    by_id = {}

    class Resident:
        def pred(self, rgb):
            if isinstance(rgb, torch.Tensor):
                return by_ptr[rgb.data_ptr()][None, None]
            return by_id[id(rgb)][None, None]
    for (p, i, s) in frames:
        by_id[id(i)] = s
    sem_pc_accum.SemSegONNX = lambda path: Resident()
    a = -0.004
    T = np.array([[np.cos(a), -np.sin(a), 0, 0], [np.sin(a), np.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.]]) @ \
        np.array([[1, 0, 0, -1.0], [0, 1, 0, 0.01], [0, 0, 1, 0.002], [0, 0, 0, 1.]])
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': P}
    ctx = _lib.Context.get()
    # (a store of an EARLIER test that dies between an integrate() and its generate_bev() runs the context's noted K1 from its
    # __del__ -- harmless, but it would be counted below as a launch of its own: collect such stores now)
    import gc
    gc.collect()
    for use_gt in (False, True):
        accs = []
        for defer in (True, False):
            acc = Kitti360SemanticPointCloudAccumulator(40., calib, 1e3, 'resident', KITTI_FILTERS, SEM_IDXS, use_gt,
                                                        dict(BEV_KITTI, view_size=80, pixel_size=256))
            acc._store_args = dict(capacity=1 << 22, max_frames=64)
            acc.pose_provider = lambda pc: T
            assert acc._defer_k1                                             # the default
            acc._defer_k1 = defer
            accs.append(acc)
        order = [0, 1, 2, 0, 3, 1, 4, 2, 5, 0, 1, 5, 2, 4, 4, 0, 1, 2, 0, 1, 2, 5, 0, 1]
        launches = [{}, {}]
        for k, f in enumerate(order):
            pc, img, sem = frames[f]
            if use_gt:
                sem_gt = by_ptr[dev[f][0].data_ptr()].cpu().numpy()          # any labels will do: one per point
                sem_gt = np.resize(sem_gt.ravel(), (len(pc), 1)).astype(np.int64)
                obs = (img, pc if k % 2 else dev[f][1], sem_gt)
            else:
                obs = (img, pc, None) if k % 4 == 3 else (dev[f][0], dev[f][1], None)    # host arrays every fourth step
            for w, acc in enumerate(accs):
                ctx.profile(True)
                removed = acc.integrate([obs])
                if k in (6, 13):                                             # two integrate() in a row
                    launches[w][k] = ctx.profile_read()['kitti_project_sample_filter'][1]
                    ctx.profile(False)
                    continue
                if k == 9:                                                   # a reader between integrate() and the raster
                    acc.store.offsets()
                n = len(acc.poses)
                out = acc.generate_bev(n - 2 - (k % 2), 1, gen_future=True)[0] if n >= 3 else None
                launches[w][k] = ctx.profile_read()['kitti_project_sample_filter'][1]
                ctx.profile(False)
                if w == 0:
                    first = (removed, out)
                else:
                    assert removed == first[0]
                    if out is not None:
                        for key in out.keys():
                            if key.startswith('trajs_'):
                                assert all(np.array_equal(x, y) for x, y in zip(out[key], first[1][key])), key
                            else:
                                assert np.array_equal(out[key].view(np.uint16), first[1][key].view(np.uint16)), (use_gt, k, key)
        ra, rb = (np.concatenate(acc.sem_pcs) for acc in accs)
        assert ra.shape[0] > 100_000 and np.array_equal(ra, rb)
        assert np.array_equal(accs[0].store.offsets(), accs[1].store.offsets())
        # K1 launches of its own: in every step without deferral; with it, none in a step whose K1 the raster of the same step
        # took (every step but the first three, the two pairs and the steps after them, and the reader at k = 9)
        assert all(v >= 1 for v in launches[1].values()), launches
        clean = [k for k in range(3, len(order)) if k not in (6, 7, 9, 13, 14)]
        assert [launches[0][k] for k in clean] == [0] * len(clean), launches
        assert launches[0][9] >= 1, launches          # (the K1s of the pairs are run by the OTHER accumulator's next call: same context)
        for acc in accs:
            acc.store.check_status()
        accs[0].store.set_defer_k1(False)


def test_frames_that_cannot_reach_the_view_are_left_out_and_nothing_changes(monkeypatch):
    """The raster is told which slots can reach its view (DeviceStore.view_hint -> pca_bev_view_hint: the frames' boxes read back
    from K1, the camera cone for the frames whose box has not been seen yet, the product of the transforms applied since).  With
    the hint and without (`store.cull = False`): the same planes, sample for sample, on a winding path with tilting transforms
    -- a 24 m view inside a 60 m horizon, so frames at both ends of the window are left out --, through window slides, with the
    camera test and with per-point labels (no cone: boxes only), host and device inputs."""
    import torch

    import sem_pc_accum
    from kitti360_sem_pc_accum import Kitti360SemanticPointCloudAccumulator
    from pca_amd import host_logic as hl
    from pca_amd.device_store import DeviceStore
    H, W, N = 96, 320, 6000
    cam_to_velo = np.array([[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418],
                            [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
                            [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824], [0, 0, 0, 1]])
    P = np.array([[130.0, 0, 160.0, 0], [0, 130.0, 48.0, 0], [0, 0, 1, 0]]) @ np.linalg.inv(cam_to_velo)
    rng = np.random.default_rng(12)
    monkeypatch.setattr(DeviceStore, 'BOX_EVERY', 2)          # boxes come back quickly: both ends of the window get cut

    def frame():
        pc = np.stack([rng.uniform(-25, 25, N), rng.uniform(-25, 25, N), rng.uniform(-2, 3, N), rng.uniform(0, 1, N)],
                      1).astype(np.float32)
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        sem = rng.integers(0, 19, (H, W)).astype(np.uint8)
        return pc, img, sem
    frames = [frame() for _ in range(5)]
    by_id = {id(f[1]): f[2] for f in frames}

    class Resident:
        def pred(self, rgb):
            return by_id[id(rgb)][None, None]
    sem_pc_accum.SemSegONNX = lambda path: Resident()

    def T_of(k):                                              # ~1 m steps on a curve that turns left, then right, with a little tilt
        yaw = 0.03 * np.sin(k / 9.0) + 0.01
        pitch = 0.004 * np.cos(k / 5.0)
        R = hl.rotation_matrix_3d(yaw)
        Rp = np.array([[np.cos(pitch), 0, np.sin(pitch)], [0, 1, 0], [-np.sin(pitch), 0, np.cos(pitch)]])
        T = np.eye(4)
        T[:3, :3] = Rp @ R
        T[:3, 3] = [-(0.8 + 0.4 * ((k * 7) % 5) / 5.0), 0.02 * np.sin(k), 0.003]
        return T
    calib = {'h_velo_cam': None, 'p_cam_frame': None, 'p_velo_frame': P}
    for use_gt in (False, True):
        accs = []
        for cull in (True, False):
            acc = Kitti360SemanticPointCloudAccumulator(60., calib, 1e3, 'resident', KITTI_FILTERS, SEM_IDXS, use_gt,
                                                        dict(BEV_KITTI, view_size=24, pixel_size=64))
            acc._store_args = dict(capacity=1 << 19, max_frames=48)       # slides while the stream runs
            accs.append(acc)
        for w, acc in enumerate(accs):
            Ts = iter([T_of(k) for k in range(400)])
            acc.pose_provider = lambda pc, Ts=Ts: next(Ts)
            acc.store.cull = w == 0
        n_samples = 0
        for k in range(150):
            pc, img, sem = frames[k % 5]
            if use_gt:
                obs = (img, pc if k % 2 else torch.from_numpy(pc).cuda(), (np.arange(len(pc))[:, None] * 7 + k) % 19)
            else:
                obs = (img, pc, None)
            outs = []
            for acc in accs:
                acc.integrate([obs])
                d = acc.get_incremental_path_dists()
                ok = len(d) > 3 and d[-1] > 22.0
                if ok and k % 3 != 2:                                    # (a step without a raster now and then: longer owed chains)
                    pidx = int(((d - (d[-1] - 20.0)) > 0).argmax())          # the pose ~20 m of path behind the newest
                    pidx = min(max(pidx, 1), len(d) - 2)
                    outs.append(acc.generate_bev(pidx, 1, gen_future=True)[0])
            if len(outs) == 2:
                n_samples += 1
                for key in outs[0].keys():
                    if key.startswith('trajs_'):
                        assert all(np.array_equal(x, y) for x, y in zip(outs[0][key], outs[1][key])), key
                    else:
                        assert np.array_equal(outs[0][key].view(np.uint16), outs[1][key].view(np.uint16)), (use_gt, k, key)
        assert n_samples > 60
        hints = [acc.store.hints_taken for acc in accs]
        assert hints[0] >= 8 and hints[1] == 0, (hints, n_samples)    # frames really were left out (never in a call that writes back)
        assert np.array_equal(np.concatenate(accs[0].sem_pcs), np.concatenate(accs[1].sem_pcs))
        for acc in accs:
            acc.store.check_status()
