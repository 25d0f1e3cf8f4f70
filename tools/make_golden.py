#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REAL reference.

Test infrastructure only.  Runs in the build container (needs the read-only
reference checkout at /root/reference); never runs on the GPU box.  Nothing of
the reference is copied: the reference modules are imported from where they
lie, executed on small seeded synthetic inputs, and only the numeric inputs and
outputs are written to ``tests/golden/*.npz``.

Third-party modules the reference imports at module level but that are absent
from this image (open3d, onnxruntime, torchvision, nuscenes-devkit,
pyquaternion) are replaced by empty ``types.ModuleType`` stubs; the only
behaviour injected is
  * a fake semseg model  ``pred(rgb) -> (1,1,H,W)``  (see ``fake_semseg``),
  * a fake ICP result carrying the pose we inject,
  * ``view_points`` implementing the documented nuscenes-devkit formula
    (viewpad @ [p;1], normalise by row 2) -- parity at that boundary is
    "unpinned" (SURVEY.md 8c), the stub pins our own restatement only.

Usage:  python tools/make_golden.py [--out tests/golden]
"""
import argparse
import os
import random
import sys
import types

import numpy as np

REF = '/root/reference'


# --------------------------------------------------------------------------
#  reference import with stubs
# --------------------------------------------------------------------------
class _FakeICPResult:
    def __init__(self, T):
        self.transformation = T


class _FakePointCloud:
    def __init__(self):
        self.points = None

    def estimate_normals(self):
        pass


_icp_queue = []


def _fake_registration_icp(target, source, thr, init, est):
    return _FakeICPResult(_icp_queue.pop(0))


def view_points_stub(points, view, normalize):
    """Documented behaviour of nuscenes.utils.geometry_utils.view_points."""
    viewpad = np.eye(4)
    viewpad[:view.shape[0], :view.shape[1]] = view
    nbr_points = points.shape[1]
    points = np.concatenate((points, np.ones((1, nbr_points))))
    points = np.dot(viewpad, points)
    points = points[:3, :]
    if normalize:
        points = points / points[2:3, :].repeat(3, 0).reshape(3, nbr_points)
    return points


def import_reference():
    assert os.path.isdir(REF), 'reference checkout not present'
    sys.dont_write_bytecode = True
    names = [
        'open3d', 'onnxruntime', 'torchvision', 'torchvision.transforms',
        'nuscenes', 'nuscenes.nuscenes', 'nuscenes.utils',
        'nuscenes.utils.data_classes', 'nuscenes.utils.geometry_utils',
        'nuscenes.map_expansion', 'nuscenes.map_expansion.map_api',
        'pyquaternion'
    ]
    for name in names:
        sys.modules[name] = types.ModuleType(name)
    sys.modules['nuscenes.nuscenes'].NuScenes = object
    sys.modules['nuscenes.utils.data_classes'].LidarPointCloud = object
    sys.modules['nuscenes.utils.geometry_utils'].transform_matrix = None
    sys.modules['nuscenes.utils.geometry_utils'].view_points = view_points_stub
    sys.modules['nuscenes.map_expansion.map_api'].NuScenesMap = object
    sys.modules['pyquaternion'].Quaternion = object
    o3d = sys.modules['open3d']
    o3d.geometry = types.SimpleNamespace(PointCloud=_FakePointCloud)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: a)
    o3d.pipelines = types.SimpleNamespace(registration=types.SimpleNamespace(
        registration_icp=_fake_registration_icp,
        TransformationEstimationPointToPlane=lambda: None))
    ds = types.ModuleType('datasets')
    ds.__path__ = [os.path.join(REF, 'datasets')]
    sys.modules['datasets'] = ds
    sys.path.insert(0, REF)
    import sem_pc_accum  # noqa
    import kitti360_sem_pc_accum  # noqa
    import nuscenes_oracle_sem_pc_accum  # noqa
    import datasets.nuscenes_utils as nu  # noqa
    from bev_generator.sem_bev import SemBEVGenerator  # noqa
    from bev_generator.rgb_bev import RGBBEVGenerator  # noqa
    return types.SimpleNamespace(
        sem_pc_accum=sem_pc_accum,
        kitti=kitti360_sem_pc_accum,
        oracle=nuscenes_oracle_sem_pc_accum,
        nu=nu,
        SemBEVGenerator=SemBEVGenerator,
        RGBBEVGenerator=RGBBEVGenerator)


# --------------------------------------------------------------------------
#  synthetic inputs (shared with the tests through the stored arrays)
# --------------------------------------------------------------------------
CAM_TO_VELO = np.array(
    [[0.04307104361, -0.08829286498, 0.995162929, 0.8043914418],
     [-0.999004371, 0.007784614041, 0.04392796942, 0.2993489574],
     [-0.01162548558, -0.9960641394, -0.08786966659, -0.1770225824],
     [0, 0, 0, 1]])


def small_calib(H, W, f):
    p_cam = np.array([[f, 0, W / 2 + 0.049453, 0], [0, f, H / 2 - 0.230451, 0],
                      [0, 0, 1, 0]], dtype=float)
    h_velo_cam = np.linalg.inv(CAM_TO_VELO)
    return {
        'h_velo_cam': h_velo_cam,
        'p_cam_frame': p_cam,
        'p_velo_frame': np.matmul(p_cam, h_velo_cam)
    }


def fake_semseg(rgb):
    """Deterministic stand-in for SemSegONNX.pred: (1,1,H,W) int64, 0..18."""
    a = np.asarray(rgb).astype(np.int64)
    sem = (a[..., 0] + 2 * a[..., 1] + 3 * a[..., 2]) % 19
    return sem[None, None]


class FakeSemSeg:
    def pred(self, rgb):
        return fake_semseg(rgb)


def rigid(rx, ry, rz, tx, ty, tz):
    cx, sx = np.cos(rx), np.sin(rx)
    cy, sy = np.cos(ry), np.sin(ry)
    cz, sz = np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [tx, ty, tz]
    return T


def kitti_frame(rng, N, H, W, lim=30.0):
    from PIL import Image
    pc = np.stack([
        rng.uniform(-lim, lim, N),
        rng.uniform(-lim, lim, N),
        rng.uniform(-2, 3, N),
        rng.uniform(0, 1, N)
    ], 1).astype(np.float32)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    return pc, Image.fromarray(img), img


BEV_PARAMS_KITTI = dict(type='sem', view_size=40, pixel_size=32,
                        max_trans_radius=0., zoom_thresh=0., do_warp=False,
                        int_scaler=20., int_sep_scaler=20.,
                        int_mid_threshold=0.5, height_filter=None)
BEV_PARAMS_NUSC = dict(type='sem', view_size=30, pixel_size=32,
                       max_trans_radius=0., zoom_thresh=0., do_warp=False,
                       int_scaler=1., int_sep_scaler=30.,
                       int_mid_threshold=0.12, height_filter=3.)
SEM_IDXS = {'road': 0, 'car': 13, 'truck': 14, 'bus': 15, 'motorcycle': 17}
KITTI_FILTERS = [10, 11, 12, 16, 18, 255]
NUSC_FILTERS = [10, 11, 12, 16, 18]


def flat_bev(prefix, bev, out):
    """Flatten a reference BEV dict into npz entries."""
    for k, v in bev.items():
        if k.startswith('trajs_') or k == 'gt_lanes':
            out[f'{prefix}{k}_n'] = np.array(len(v))
            for i, t in enumerate(v):
                out[f'{prefix}{k}_{i}'] = np.asarray(t, dtype=float)
        else:
            out[f'{prefix}{k}'] = np.asarray(v)


# --------------------------------------------------------------------------
#  case generators
# --------------------------------------------------------------------------
def case_k1(ref, out_dir):
    """velo2img / gen_semantic_pc / filter on one frame incl. edge cases."""
    rng = np.random.default_rng(101)
    H, W = 64, 96
    calib = small_calib(H, W, 40.0)
    acc = ref.kitti.Kitti360SemanticPointCloudAccumulator(
        8., calib, 1e3, None, KITTI_FILTERS, SEM_IDXS, True, BEV_PARAMS_KITTI)
    pc, _, img = kitti_frame(rng, 4096, H, W)
    sem = rng.integers(0, 19, (H, W)).astype(np.int64)
    sem[rng.random((H, W)) < 0.02] = 255
    P = calib['p_velo_frame']
    out = dict(pc=pc, img=img, sem=sem, P=P, H=np.array(H), W=np.array(W))
    out['velo2img'] = acc.velo2img(pc.copy(), P, H, W)
    out['sem_rgb'] = acc.gen_semantic_pc(pc.copy(), img, P)
    out['sem_sem'] = acc.gen_semantic_pc(pc.copy(), sem[..., None], P)
    rgbsem = np.concatenate((out['sem_rgb'], out['sem_sem'][:, -1:]), axis=1)
    out['filtered'] = acc.filter_semseg_pc(rgbsem)

    # axis-aligned camera: exact .5 rounding, depth == 0, <0, tiny, inf, nan
    P2 = np.array([[8., 0, 48, 0], [0, 8, 32, 0], [0, 0, 1, 0]])
    e = []
    for x in (0.25, 0.75, -24.25, -24.75, 23.25, 23.75, 23.5, -24.0):
        for y in (0.25, 0.75, -16.25, -16.75, 15.25, 15.75, 15.5, -16.0):
            e.append([x, y, 4.0, 0.5])
    e += [[1, 1, 0, .1], [0, 0, 0, .2], [1, 1, -4, .3], [1, 1, 1e-30, .4],
          [1e30, 1, 1, .5], [np.inf, 1, 1, .6], [1, np.nan, 1, .7],
          [1, 1, np.inf, .8], [np.nan, np.nan, np.nan, .9], [0, 0, 1e-38, 1.],
          [3, -2, 2, 0.]]
    pc2 = np.array(e, dtype=np.float32)
    pc2 = np.concatenate([pc2, kitti_frame(rng, 512, H, W, 8.0)[0]])
    pc2[-512:, 2] = np.abs(pc2[-512:, 2]) + 0.5
    out['pc2'] = pc2
    out['P2'] = P2
    with np.errstate(all='ignore'):
        out['velo2img2'] = acc.velo2img(pc2.copy(), P2, H, W)
        out['sem_rgb2'] = acc.gen_semantic_pc(pc2.copy(), img, P2)
        out['sem_sem2'] = acc.gen_semantic_pc(pc2.copy(), sem[..., None], P2)
    np.savez_compressed(os.path.join(out_dir, 'k1.npz'), **out)
    print('k1: kept', out['velo2img'].shape[0], 'of', pc.shape[0],
          '| filtered', out['filtered'].shape[0], '| edge kept',
          out['velo2img2'].shape[0], 'of', pc2.shape[0])


def case_kitti_accum(ref, out_dir):
    """Kitti360 accumulator: integrate() x14 with injected poses, eviction,
    then generate_bev (semseg from images) + a use_gt_sem variant."""
    rng = np.random.default_rng(202)
    H, W = 64, 96
    calib = small_calib(H, W, 40.0)
    ref.sem_pc_accum.SemSegONNX = lambda path: FakeSemSeg()
    acc = ref.kitti.Kitti360SemanticPointCloudAccumulator(
        8., calib, 1e3, 'fake.onnx', KITTI_FILTERS, SEM_IDXS, False,
        BEV_PARAMS_KITTI)
    F, N = 14, 3000
    out = dict(P=calib['p_velo_frame'], H=np.array(H), W=np.array(W),
               F=np.array(F), horizon=np.array(8.))
    Ts = []
    for k in range(F):
        T = rigid(0.001 * (k % 3), -0.002, -0.02 - 0.001 * k, -1.0 - 0.01 * k,
                  0.05, 0.01)
        Ts.append(T)
    out['Ts'] = np.stack(Ts)
    removed = []
    for k in range(F):
        pc, pil, img = kitti_frame(rng, N - 100 * (k % 4), H, W)
        out[f'pc_{k}'] = pc
        out[f'img_{k}'] = img
        _icp_queue.append(Ts[k])
        removed.append(acc.integrate([(pil, pc, None)]))
        if k in (0, 4, F - 1):
            out[f'step{k}_sizes'] = np.array([a.shape[0] for a in acc.sem_pcs])
            out[f'step{k}_sem_pcs'] = np.concatenate(acc.sem_pcs)
            out[f'step{k}_poses'] = np.array(acc.poses)
            out[f'step{k}_seg_dists'] = np.array(acc.seg_dists)
    out['removed'] = np.array(removed)
    out['incr_path_dists'] = acc.get_incremental_path_dists()
    present_idx = len(acc.poses) // 2
    out['present_idx'] = np.array(present_idx)
    bev = acc.generate_bev(present_idx, 1, gen_future=True)[0]
    flat_bev('bev_', bev, out)
    np.savez_compressed(os.path.join(out_dir, 'kitti_accum.npz'), **out)
    print('kitti_accum: live frames', len(acc.poses), 'removed', removed)

    # use_gt_sem variant (no projection at all, rgb = 0)
    acc = ref.kitti.Kitti360SemanticPointCloudAccumulator(
        50., calib, 1e3, None, KITTI_FILTERS, SEM_IDXS, True,
        BEV_PARAMS_KITTI)
    out = dict(P=calib['p_velo_frame'])
    Ts = [rigid(0, 0.001, 0.03, -1.5, 0.1, 0.0) for _ in range(4)]
    out['Ts'] = np.stack(Ts)
    for k in range(4):
        pc, pil, img = kitti_frame(rng, 2000, H, W, 15.0)
        sem_gt = rng.integers(0, 19, (2000, 1)).astype(np.int16)
        sem_gt[rng.random(2000) < 0.03] = 255
        out[f'pc_{k}'] = pc
        out[f'sem_gt_{k}'] = sem_gt
        _icp_queue.append(Ts[k])
        acc.integrate([(pil, pc, sem_gt)])
    out['sizes'] = np.array([a.shape[0] for a in acc.sem_pcs])
    out['sem_pcs'] = np.concatenate(acc.sem_pcs)
    out['poses'] = np.array(acc.poses)
    bev = acc.generate_bev(2, 1, gen_future=True)[0]
    flat_bev('bev_', bev, out)
    np.savez_compressed(os.path.join(out_dir, 'kitti_gtsem.npz'), **out)
    print('kitti_gtsem: sizes', out['sizes'])


def random_sem_pc(rng, n, lim, int255=False, dyn_frac=0.1):
    pc = np.zeros((n, 10))
    pc[:, 0] = rng.uniform(-lim, lim, n)
    pc[:, 1] = rng.uniform(-lim, lim, n)
    pc[:, 2] = rng.uniform(-2, 4, n)
    if int255:
        pc[:, 3] = rng.integers(0, 256, n) / 255.
    else:
        pc[:, 3] = rng.uniform(0, 1, n).astype(np.float32)
    pc[:, 4:7] = rng.integers(0, 256, (n, 3))
    pc[:, 7] = rng.choice([0, 0, 0, 1, 2, 8, 9, 13, 14, 15, 17, 5], n)
    pc[:, 8] = rng.integers(-1, 4, n)
    pc[:, 9] = (rng.random(n) < dyn_frac).astype(float)
    return pc


def bev_inputs(rng, n_p, n_f, lim, int255=False):
    pc_present = random_sem_pc(rng, n_p, lim, int255)
    pc_future = random_sem_pc(rng, n_f, lim, int255)
    pc_full = np.concatenate([pc_present, pc_future])
    k = 9
    s = np.linspace(-0.9 * lim, 0.9 * lim, 2 * k)
    traj = np.stack([s, 0.3 * s + 0.02 * s**2 / lim, 0 * s + 1.0], 1)
    other = [
        np.stack([s[:k] * 1.3 + 2.0, -0.5 * s[:k] - 1.0, 0 * s[:k]], 1),
        np.stack([0 * s[:4] + 1e3, s[:4], 0 * s[:4]], 1),   # fully outside
    ]
    pcs = dict(pc_present=pc_present, pc_future=pc_future, pc_full=pc_full)
    trajs = dict(ego_traj_present=traj[:k].copy(),
                 ego_traj_future=traj[k:].copy(), ego_traj_full=traj.copy(),
                 other_trajs_present=[o.copy() for o in other],
                 other_trajs_future=[other[0][::-1].copy()],
                 other_trajs_full=[])
    return pcs, trajs


def copy_inputs(pcs, trajs):
    p = {k: (None if v is None else v.copy()) for k, v in pcs.items()}
    t = {}
    for k, v in trajs.items():
        t[k] = [a.copy() for a in v] if isinstance(v, list) else v.copy()
    return p, t


def store_inputs(out, pcs, trajs):
    out['pc_present'] = pcs['pc_present']
    out['pc_future'] = pcs['pc_future']
    for k, v in trajs.items():
        if isinstance(v, list):
            out[f'in_{k}_n'] = np.array(len(v))
            for i, a in enumerate(v):
                out[f'in_{k}_{i}'] = a
        else:
            out[f'in_{k}'] = v


def ref_planes(gen, pcs, trajs, out, rot=None, dx=0., dy=0., zoom=1.):
    """Pre-fp16-cast f64 planes for the three point sets, obtained by calling
    the reference's own sub-functions in the order generate()/generate_bev()
    call them.  Stored as pre_<plane>_<set>."""
    p, t = copy_inputs(pcs, trajs)
    if rot is None:
        ego = t['ego_traj_present']
        rot = 0.5 * np.pi
        if len(ego) > 1:
            rot += np.arctan2(ego[-1][1] - ego[-2][1], ego[-1][0] - ego[-2][0])
        rot = np.pi - rot
    out['rot_ang'] = np.array(rot)
    aug = zoom * gen.view_size
    for name in ('present', 'future', 'full'):
        pc_g, _ = gen.preprocess_pc_and_trajs(p[f'pc_{name}'], [], rot, dx, dy,
                                              aug)
        out[f'pre_n_grid_{name}'] = np.array(pc_g.shape[0])
        _, pc_s = gen.partition_semantic_pc(pc_g, [1], 9)
        r, g, b = gen.get_rgb_maps(pc_s)
        out[f'pre_rgb_{name}'] = np.stack([r, g, b]) / 255.
        out[f'pre_elevation_{name}'] = gen.get_elevation_map(pc_s)[0]
        out[f'pre_road_{name}'] = gen.gen_sem_probmap(pc_s, ['road'])
        im = gen.gen_intensity_map(pc_s, 'road')
        out[f'pre_intraw_{name}'] = im
        out[f'pre_intensity_{name}'] = gen.road_marking_transform(
            im, gen.int_scaler, gen.int_sep_scaler, gen.int_mid_threshold)
        out[f'pre_dynamic_{name}'] = gen.gen_sem_probmap(
            pc_s, ['car', 'truck', 'bus', 'motorcycle'])
        if name == 'present':
            out['pre_grid_rows_present'] = pc_g


def case_bev(ref, out_dir):
    rng = np.random.default_rng(303)
    # ---- A: KITTI params, heading from trajectory, intermediates stored ----
    gen = ref.SemBEVGenerator(SEM_IDXS, 20, 32, 0., 0., False, 20., 20., 0.5,
                              None)
    pcs, trajs = bev_inputs(rng, 8000, 12000, 14.0)
    out = {}
    store_inputs(out, pcs, trajs)
    p, t = copy_inputs(pcs, trajs)
    flat_bev('bev_', gen.generate(p, t), out)
    ref_planes(gen, pcs, trajs, out)
    np.savez_compressed(os.path.join(out_dir, 'bev_a.npz'), **out)

    # ---- B: NuScenes params, height filter, explicit rot/trans/zoom ----
    gen = ref.SemBEVGenerator(SEM_IDXS, 51.2, 64, 0., 0., False, 1., 30., 0.12,
                              3.)
    pcs, trajs = bev_inputs(rng, 15000, 9000, 33.0, int255=True)
    out = {}
    store_inputs(out, pcs, trajs)
    args = (0.7, 1.5, -2.25, 1.1, True)
    out['args'] = np.array(args[:4])
    p, t = copy_inputs(pcs, trajs)
    flat_bev('bev_', gen.generate(p, t, *args), out)
    ref_planes(gen, pcs, trajs, out, *args[:4])
    np.savez_compressed(os.path.join(out_dir, 'bev_b.npz'), **out)

    # ---- C: 256 x 256, view 80 ----
    gen = ref.SemBEVGenerator(SEM_IDXS, 80, 256, 0., 0., False, 20., 20., 0.5,
                              None)
    pcs, trajs = bev_inputs(rng, 14000, 14000, 55.0)
    # pile-ups: many points in few cells (contention + long medians)
    pcs['pc_present'][:3000, :2] = rng.uniform(-0.6, 0.6, (3000, 2))
    pcs['pc_future'][:2000, :2] = rng.uniform(-0.6, 0.6, (2000, 2))
    pcs['pc_full'] = np.concatenate([pcs['pc_present'], pcs['pc_future']])
    out = {}
    store_inputs(out, pcs, trajs)
    p, t = copy_inputs(pcs, trajs)
    flat_bev('bev_', gen.generate(p, t), out)
    ref_planes(gen, pcs, trajs, out)
    np.savez_compressed(os.path.join(out_dir, 'bev_c.npz'), **out)

    # ---- D: empty future set, single-pose trajectory (no heading) ----
    gen = ref.SemBEVGenerator(SEM_IDXS, 20, 16, 0., 0., False, 20., 20., 0.5,
                              None)
    pcs, trajs = bev_inputs(rng, 500, 0, 14.0)
    trajs['ego_traj_present'] = trajs['ego_traj_present'][:1]
    out = {}
    store_inputs(out, pcs, trajs)
    p, t = copy_inputs(pcs, trajs)
    flat_bev('bev_', gen.generate(p, t), out)
    np.savez_compressed(os.path.join(out_dir, 'bev_d.npz'), **out)

    # ---- E: do_warp=True with pinned warp parameters ----
    gen = ref.SemBEVGenerator(SEM_IDXS, 20, 32, 0., 0., True, 20., 20., 0.5,
                              None)
    gen.get_random_warp_params = lambda *a: (16 + 3.7, 16 - 2.2)
    pcs, trajs = bev_inputs(rng, 6000, 6000, 14.0)
    out = dict(warp=np.array([16 + 3.7, 16 - 2.2]))
    store_inputs(out, pcs, trajs)
    p, t = copy_inputs(pcs, trajs)
    flat_bev('bev_', gen.generate(p, t), out)
    np.savez_compressed(os.path.join(out_dir, 'bev_e.npz'), **out)
    print('bev: a,b,c,d,e written')


def case_nusc(ref, out_dir):
    from PIL import Image
    rng = np.random.default_rng(404)
    H, W = 45, 80
    ref.sem_pc_accum.SemSegONNX = lambda path: FakeSemSeg()
    acc = ref.oracle.NuScenesOracleSemanticPointCloudAccumulator(
        'fake.onnx', NUSC_FILTERS, SEM_IDXS, False, BEV_PARAMS_NUSC, 'boston',
        False, None)
    F, N = 7, 2500
    out = dict(F=np.array(F), H=np.array(H), W=np.array(W))
    tokens_all = []
    for k in range(F):
        pc = np.zeros((N, 7))
        pc[:, 0] = rng.uniform(-20, 20, N)
        pc[:, 1] = rng.uniform(-20, 20, N)
        pc[:, 2] = rng.uniform(-2, 4, N)
        pc[:, 3] = rng.integers(0, 256, N)
        pc[:, 4] = rng.uniform(1.01, W - 1.01, N)
        pc[:, 5] = rng.uniform(1.01, H - 1.01, N)
        pc[:, 6] = rng.integers(-1, 4, N)
        # exact .5 pixel coordinates (round-half-even) on a few points
        pc[:8, 4] = [1.5, 2.5, 3.5, 4.5, 77.5, 78.5, 10.5, 11.5]
        pc[:8, 5] = [1.5, 2.5, 43.5, 42.5, 3.5, 4.5, 20.5, 21.5]
        cam_idx = rng.integers(-1, 6, N)
        imgs = rng.integers(0, 256, (6, H, W, 3), dtype=np.uint8)
        T = rigid(0.002 * k, -0.001 * k, 0.02 * k, 1000 + 1.0 * k,
                  500 + 0.1 * k, 0.3)
        tokens = ['a', 'b', 'c']
        clss = [0, 7, 1]
        centers = [
            np.array([1010 + 0.6 * k, 505., 0.5]),
            np.array([1005., 495. + 2.0 * k, 0.2]),
            np.array([990., 500., 0.4])
        ]
        if k >= 3:
            tokens.append('d')
            clss.append(5)
            centers.append(np.array([1000. - 0.8 * k, 510., 0.1]))
        if k == 5:      # 'a' unobserved at ts 5 -> split trajectory
            tokens, clss, centers = tokens[1:], clss[1:], centers[1:]
        obs = dict(images=[Image.fromarray(im) for im in imgs], pc=pc,
                   pc_cam_idx=cam_idx, ego_at_lidar_ts=T,
                   ego_global_x=T[0, 3], ego_global_y=T[1, 3],
                   inst_tokens=tokens, inst_cls=clss, inst_center=centers)
        out[f'pc_{k}'] = pc
        out[f'cam_idx_{k}'] = cam_idx
        out[f'imgs_{k}'] = imgs
        out[f'T_{k}'] = T
        out[f'inst_cls_{k}'] = np.array(clss)
        out[f'inst_center_{k}'] = np.stack(centers)
        tokens_all.append(','.join(tokens))
        acc.integrate([obs])
        if k == 0:
            out['frame0_after_integrate'] = acc.sem_pcs[0].copy()
    out['inst_tokens'] = np.array(tokens_all)
    out['sizes'] = np.array([a.shape[0] for a in acc.sem_pcs])
    out['sem_pcs'] = np.concatenate(acc.sem_pcs)
    out['poses'] = np.array(acc.poses)
    out['seg_dists'] = np.array(acc.seg_dists)
    out['dyn_instances'] = np.array(acc.dyn_instances)
    out['incr_path_dists'] = acc.get_incremental_path_dists()
    present_idx = 3
    out['present_idx'] = np.array(present_idx)
    bev = acc.generate_bev(present_idx, 1, gen_future=True)[0]
    flat_bev('bev_', bev, out)
    np.savez_compressed(os.path.join(out_dir, 'nusc_oracle.npz'), **out)
    print('nusc_oracle: sizes', out['sizes'], 'dyn', acc.dyn_instances)


def case_utils(ref, out_dir):
    rng = np.random.default_rng(505)
    nu = ref.nu
    out = {}
    # homo_transform
    T = rigid(0.1, -0.2, 0.3, 1000.5, -500.25, 3.125)
    pts = rng.uniform(-50, 50, (777, 3))
    out['ht_T'] = T
    out['ht_pts'] = pts
    out['ht_out'] = nu.homo_transform(T, pts)
    # pts_feat_from_img
    H, W = 45, 80
    img = rng.integers(0, 256, (H, W, 4)).astype(np.int64)
    uv = np.stack([rng.uniform(1.01, W - 1.01, 600),
                   rng.uniform(1.01, H - 1.01, 600)], 1)
    uv[:6] = [[1.5, 1.5], [2.5, 2.5], [3.5, 43.5], [78.5, 42.5], [10.5, 20.],
              [11., 21.5]]
    out['pf_img'] = img
    out['pf_uv'] = uv
    out['pf_nearest'] = nu.pts_feat_from_img(uv, img, 'nearest')
    # bilinear divides by zero for integer coordinates in the reference
    uvb = uv[uv[:, 0] != np.floor(uv[:, 0])]
    uvb = uvb[uvb[:, 1] != np.floor(uvb[:, 1])]
    out['pf_uv_bil'] = uvb
    # NOTE: the reference's bilinear branch only broadcasts for 2-D (H,W)
    # feature maps ((n,) weights times (n,C) features raises ValueError).
    out['pf_bilinear'] = nu.pts_feat_from_img(uvb, img[..., 0], 'bilinear')
    # project_pts3d (view_points stubbed with the documented formula)
    cam = nu.NuScenesCamera.__new__(nu.NuScenesCamera)
    cam.img_wh = np.array([1600, 900], dtype=float)
    cam.cam_K = np.array([[1266.417203046554, 0.0, 816.2670197447984],
                          [0.0, 1266.417203046554, 491.50706579294757],
                          [0.0, 0.0, 1.0]])
    pc = np.stack([rng.uniform(-30, 30, 900), rng.uniform(-10, 10, 900),
                   rng.uniform(-5, 60, 900)], 1)
    pc[:4, 2] = [0, 1e-3, 1.0000001e-3, -1]
    out['pp_K'] = cam.cam_K
    out['pp_wh'] = cam.img_wh
    out['pp_pc'] = pc
    out['pp_uv'], out['pp_mask'] = cam.project_pts3d(pc)
    # 6-camera projection loop shape (nuscenes_obs_dataloader.py:162-202)
    N = 1200
    pc_l = np.stack([rng.uniform(-40, 40, N), rng.uniform(-40, 40, N),
                     rng.uniform(-3, 5, N)], 1).astype(np.float32).astype(
                         float)
    ego_from_lidar = rigid(0.003, 0.01, -1.57, 0.94, 0.0, 1.84)
    glob_from_ego = rigid(0.01, -0.005, 0.6, 1010.2, 612.7, 0.1)
    cams_glob_from_self = []
    for j in range(6):
        ego_from_cam = rigid(-1.57, 0.0, -1.57 + j * 1.047, 1.5, 0.1 * j, 1.5)
        # a slightly different ego pose per camera timestamp
        g_e = rigid(0.01, -0.005, 0.6 + 0.001 * j, 1010.2 + 0.05 * j, 612.7,
                    0.1)
        cams_glob_from_self.append(g_e @ ego_from_cam)
    pc_in_ego = nu.homo_transform(ego_from_lidar, pc_l)
    pc_in_glob = nu.homo_transform(glob_from_ego, pc_in_ego)
    pc_uv = np.zeros((N, 2))
    pc_cam_idx = -np.ones(N, dtype=int)
    for j in range(6):
        pc_in_cam = nu.homo_transform(np.linalg.inv(cams_glob_from_self[j]),
                                      pc_in_glob)
        uvj, m = cam.project_pts3d(pc_in_cam)
        pc_uv[m] = uvj[m]
        pc_cam_idx[m] = j
    out['c6_pc'] = pc_l
    out['c6_ego_from_lidar'] = ego_from_lidar
    out['c6_glob_from_ego'] = glob_from_ego
    out['c6_glob_from_cam'] = np.stack(cams_glob_from_self)
    out['c6_pc_in_ego'] = pc_in_ego
    out['c6_uv'] = pc_uv
    out['c6_cam_idx'] = pc_cam_idx
    # path distances
    sd = rng.uniform(0.5, 1.5, 37)
    out['pd_seg'] = sd
    out['pd_incr'] = ref.sem_pc_accum.SemanticPointCloudAccumulator.\
        comp_incr_path_dist(sd)
    # trajectories: crop + intersection
    gen = ref.SemBEVGenerator(SEM_IDXS, 20, 32)
    trajs = [
        np.array([[-15., 0, 0], [-5, 1, 0], [0, 2, 1], [5, 14, 2],
                  [7, 3, 3], [8, 2, 4]]),
        np.array([[0., 0, 0], [1, 1, 1], [2, 2, 2]]),
        np.array([[-20., -20, 0], [20, 20, 1]]),
        np.array([[3., 3, 3]]),
        np.array([[9.9999, 0, 0], [10.0001, 0, 0], [9.5, 0.5, 0]]),
    ]
    out['ct_n'] = np.array(len(trajs))
    for i, tr in enumerate(trajs):
        out[f'ct_in_{i}'] = tr
        out[f'ct_out_{i}'] = gen.crop_trajectory(tr.copy(), 20.)
    # warps
    a1, a2 = gen.cal_warp_params(19.7, 16, 31)
    b1, b2 = gen.cal_warp_params(13.8, 16, 31)
    out['wp_params'] = np.array([a1, a2, b1, b2])
    maps = rng.uniform(0, 1, (3, 32, 32))
    out['wp_in'] = maps
    out['wp_out'] = gen.warp_dense_probmaps(maps, a1, a2, b1, b2)
    pts = np.stack([rng.integers(0, 32, 40).astype(float),
                    rng.integers(0, 32, 40).astype(float),
                    np.zeros(40)], 1)
    out['ws_in'] = pts
    out['ws_out'] = gen.warp_sparse_points(pts.copy(), a1, a2, b1, b2, 16, 16,
                                           19.7, 13.8)
    # RGB BEV generator medians
    rgen = ref.RGBBEVGenerator(20, 16, 7)
    pc = random_sem_pc(rng, 1500, 9.9)
    pc[:, :2] = np.floor(pc[:, :2] / 20 * 16 + 8)
    out['rg_pc'] = pc
    r, g, b = rgen.get_rgb_maps(pc)
    out['rg_out'] = np.stack([r, g, b])
    np.savez_compressed(os.path.join(out_dir, 'utils.npz'), **out)
    print('utils written')


def case_sweeps(ref, out_dir):
    """The reference's inst_centric_get_sweeps + load_data_to_tensor on a fake dataset object (tests/fake_nuscenes.py):
    inputs (the tables) and outputs (points, tokens, centres, last boxes, class indices)."""
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
    import fake_nuscenes as fk
    ref.nu.Quaternion = fk.FakeQuaternion          # pyquaternion stand-in (textbook formula; third-party, unpinned)
    tables = fk.synth_tables()
    with tempfile.TemporaryDirectory() as tmp:
        nusc = fk.FakeNuScenes(tables, tmp)
        res = ref.nu.inst_centric_get_sweeps(nusc, 'sample0', **fk.SWEEP_CFG)
        ref.nu.load_data_to_tensor(res)
    out = {'in_' + k: np.asarray(v) for k, v in tables.items()}
    out['points'] = res['points'].numpy()
    out['instances_token'] = np.array(res['instances_token'])
    out['instances_center'] = np.stack(res['instances_center'])
    out['instances_last_box'] = res['instances_last_box'].numpy()
    out['instances_name'] = res['instances_name'].numpy()
    assert (out['points'][:, 6] >= 0).sum() > 50 and len(set(res['instances_token'])) >= 4
    np.savez_compressed(os.path.join(out_dir, 'nusc_sweeps.npz'), **out)
    print('nusc_sweeps written:', out['points'].shape, len(res['instances_token']), 'labelled boxes')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(
        os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden'))
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    random.seed(0)
    np.random.seed(0)
    ref = import_reference()
    cases = dict(k1=case_k1, kitti=case_kitti_accum, bev=case_bev,
                 nusc=case_nusc, utils=case_utils, sweeps=case_sweeps)
    for name, fn in cases.items():
        if args.only and name not in args.only.split(','):
            continue
        fn(ref, args.out)


if __name__ == '__main__':
    main()
