"""Host time of the statements of the headline step's two Python hot paths (a replica of integrate() / _generate_bev_fast()
with a clock between the statements; bursts of 6 steps after a synchronize, so nothing waits for the GPU)."""
import builtins, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import numpy as np
import torch  # noqa: E402
import ctypes as C
from pca_amd import host_logic as hl
rp, builtins.print = builtins.print, (lambda *a, **k: None)
acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
st = bench.Stepper(acc, pool)
st.fill()
out = torch.empty((21, bench.PX, bench.PX), dtype=torch.float16, device='cuda')
for _ in range(30):
    st.step(out)
pc = time.perf_counter
T = {}
def tick(name, t0):
    t1 = pc(); T[name] = T.get(name, 0.0) + t1 - t0; return t1

def integrate(self, observations):
    t = pc()
    rgb, pcl, sem_gt = observations[0]
    if not self.use_gt_sem:
        sem_gt = None
    T_new_prev = np.asarray(self.pose_provider(pcl), dtype=np.float64)
    self.T_prev_origin = np.matmul(self.T_prev_origin, T_new_prev)
    t = tick('i1 pose provider, asarray, matmul', t)
    fast = self._obs_pointers(rgb, pcl, sem_gt)
    t = tick('i2 _obs_pointers (incl. model.pred)', t)
    if len(self._track) > 0:
        self.update_sem_pcs(T_new_prev)
    t = tick('i3 update_sem_pcs', t)
    obs, semseg, H, W, keep = fast
    if self._defer_k1 != getattr(self.store.ctx, 'k1_defer', False):
        self.store.set_defer_k1(self._defer_k1)
    idx, path_length = self.store.append_kitti_obs(obs, self.P_velo_frame, H, W, self.semseg_filters, self.sample_mode,
                                                   self._track, T_new_prev, self.horizon_dist, keep=keep)
    t = tick('i4 append_kitti_obs', t)
    self.rgbs.append(rgb)
    self.semsegs.append(semseg)
    if idx:
        self.store.evict(idx)
        self.rgbs = self.rgbs[idx:]
        self.semsegs = self.semsegs[idx:]
    if path_length is not None:
        print(f'    #pc {self.store.n_frames} |', f'path length {path_length:.2f}')
    self._after_integrate()
    t = tick('i5 lists, evict, print, _after_integrate', t)
    return idx

def gen(self, present_idx, out):
    from kitti360_sem_pc_accum import _mods
    t = pc()
    Cc, torch_, LazyBev, _PendingCopy, _, hl_, _ = _mods()
    st_, g, track = self.store, self.sem_bev_generator, self._track
    ctx = st_.ctx
    st_.poll_status()
    n = st_.n_frames
    split = int(present_idx)
    t = tick('g1 mods, poll_status', t)
    poses = track.as_array()
    origin = poses[split].copy()
    t = tick('g2 as_array, origin', t)
    rot_mat = hl_.rotation_matrix_3d(hl_.heading_rot_ang(poses[max(split - 2, 0):split] - origin))
    t = tick('g3 heading, rotation', t)
    px = g.pixel_size
    prm = g._raster_params(origin, rot_mat, 0., 0., 1. * g.view_size, st_.intensity_div255)
    t = tick('g4 _raster_params', t)
    max_points = st_.bev_workspace(px)
    n_pend, pend_T, pend_ends, write_back = st_.bev_pending(0, n)
    t = tick('g5 bev_workspace, bev_pending', t)
    assert out.dtype == torch.float16 and out.is_contiguous() and tuple(out.shape) == (21, px, px)
    rows = np.empty((max(2 * (n - 1), 1), 3))
    start = np.zeros(max(n, 1), dtype=np.int32)
    cst = st_.c_store()
    t = tick('g6 out checks, rows, start, c_store', t)
    a = self._ga
    pend_T, pend_ends = st_._pend_T, st_._pend_ends
    const = (id(cst), id(st_.frame_off), id(prm), id(pend_T), id(pend_ends), id(st_._ws), track._h)
    assert const == self._ga_const
    a.slot_begin, a.slot_split, a.slot_end, a.max_points = st_.head, st_.head + split, st_.head + n, max_points
    a.n_pending, a.write_back = n_pend, write_back
    a.planes_f16, a.host_planes = out.data_ptr(), None
    a.traj_rows, a.traj_start = rows.ctypes.data, start.ctypes.data
    a.stream = ctx.stream_int() or None
    a.hint_F = st_.view_hint_into(a, 0, n)
    t = tick('g7 argument block', t)
    ticket = ctx.lib.pca_kitti_generate_bev_v(ctx.h, Cc.addressof(a))
    t = tick('g8 C pca_kitti_generate_bev_v', t)
    st_.hints_taken += a.hinted
    st_.bev_done(write_back)
    rows = rows[:a.n_rows]
    empty = np.zeros((0, 3))
    ego_p = rows[:start[split - 1]].copy() if split >= 2 else empty
    ego_f = rows[start[split]:].copy() if n - split >= 2 else empty
    r = {'planes_f16': out, 'trajs_present': [ego_p], 'trajs_future': [ego_f], 'trajs_full': [rows]}
    t = tick('g9 polylines, dict', t)
    return r

n = 0
k = st.n
for rep in range(40):
    torch.cuda.synchronize()
    for _ in range(6):
        rgb, pcl, _ = pool[k % len(pool)]; k += 1
        integrate(acc, [(rgb, pcl, None)])
        t0 = pc(); idx = bench.present_index(acc); tick('t  trigger (the driver)', t0)
        gen(acc, idx, out)
        n += 1
builtins.print = rp
print('host us per step: %.1f' % (1e6 * sum(T.values()) / n))
for name in sorted(T):
    print('  %-46s %6.2f us' % (name, 1e6 * T[name] / n))
