"""Device-resident point store: the HBM replacement of the reference's ``self.sem_pcs`` list.

Layout (include/pca.h): structure-of-arrays, frames are contiguous segments, the segment boundaries
``frame_off`` live on the device so that integrate() never reads a count back.  The host only tracks
UPPER bounds of the append position (every input point kept) for capacity planning; exact counts are
fetched lazily (``sizes()``) when somebody asks for rows.

Slots: frame k of the live window occupies slot ``head + k``; eviction advances ``head``; when slots or
capacity run out the live window is moved to the front (rare, amortised).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import PcaBevParams, PcaKittiFrame, PcaStore

# PCA_BEV_CULL=0: no raster is told which frames cannot reach its view (A/B -- results are identical)
CULL = os.environ.get('PCA_BEV_CULL', '1') != '0'


class DeviceStore:
    """The accumulated points in HBM (SoA) with the host-side bookkeeping of slots and owed re-transforms.
    NOTE: the coordinate tensors x / y / z lag behind by the re-transforms in `_pending` (up to CHAIN_MAX of them);
    every method of this class that reads coordinates applies them first.  Code that reads the tensors directly calls
    `flush_pending()` before."""
    CHAIN_MAX = 4            # owed re-transforms at most (PCA_BEV_MAX_CHAIN of the C ABI)
    CHAIN_K = int(os.environ.get('PCA_BEV_CHAIN', '4'))   # the rasteriser writes them back when this many are owed

    def __init__(self, capacity=1 << 24, max_frames=4096, device=None, intensity_div255=False):
        self.ctx = _lib.Context.get(device)
        self.device = torch.device('cuda', self.ctx.device_index)
        self.intensity_div255 = bool(intensity_div255)
        self.max_frames = int(max_frames)
        self._alloc(int(capacity))
        self.frame_off = torch.zeros(self.max_frames + 1, dtype=torch.int64, device=self.device)
        self._moved = np.eye(4)  # product of every transform applied to all live frames so far (newest on the left)
        self.cull = CULL         # tell the rasters which slots can reach their view (view_hint)
        self.hints_taken = 0     # rasters of this store that were told to leave frames out
        self._alloc_frame_tables(self.max_frames)
        self.head = 0            # first live slot
        self.tail = 0            # one past the last live slot
        self.ub_tail = 0         # upper bound of frame_off[tail]
        self.lb_head = 0         # lower bound of frame_off[head] (exact after a sync)
        self._ub = []            # per live frame: upper bound of its size (its input point count)
        self._ub_sum = 0         # sum(self._ub), kept incrementally
        self._pending = []       # [(T 16 doubles, end slot)], oldest first: re-transforms owed to slots [head, end slot)
        self._k1_cache = None
        self._k1_src = None      # (P object, filters object, their values) of the last append_kitti_obs
        self._ia = None          # argument block of pca_kitti_integrate_v, kept between calls
        self._ws = None
        self._ws_many = None
        self._ws_points, self._ws_px = 0, 0
        self._dedup_ws = None
        self._obs_out = None
        self._k1_noted = None    # what the K1 noted by the last append_kitti_obs still reads (device tensors), or None
        self._pend_T = (C.c_double * (16 * self.CHAIN_MAX))()      # the owed chain as the C calls take it
        self._pend_T_np = np.frombuffer(self._pend_T, dtype=np.float64)
        self._pend_ends = (C.c_int * self.CHAIN_MAX)()

    # ---- memory ----------------------------------------------------------------------------
    def _alloc(self, cap):
        d = self.device
        self.capacity = cap
        self._cstore = None
        self.x = torch.empty(cap, dtype=torch.float64, device=d)
        self.y = torch.empty(cap, dtype=torch.float64, device=d)
        self.z = torch.empty(cap, dtype=torch.float64, device=d)
        self.intensity = torch.empty(cap, dtype=torch.float32, device=d)
        self.rgbs = torch.empty(cap, dtype=torch.int32, device=d)     # bit pattern of the u32
        self.inst = torch.empty(cap, dtype=torch.int32, device=d)
        self.dyn = torch.empty(cap, dtype=torch.uint8, device=d)

    def _arrays(self):
        return (self.x, self.y, self.z, self.intensity, self.rgbs, self.inst, self.dyn)

    # ---- where the frames are (for pca_host_view_hull: include/pca.h) --------------------------
    BOX_EVERY = 16           # the frames' boxes are read back every so many append_kitti_obs (asynchronously; the frames that
                             # need their box are the OLD ones behind the view -- the new ones ahead of it are out by their cone)

    def _alloc_frame_tables(self, max_frames, keep=None):
        """Per slot: the box K1 leaves on the device (frame_box), what the host has seen of it (_box: lo > hi = not seen), the
        store's `moved` matrix when the frame was created (_then; NaN = unknown: the frame always counts as visible) and
        whether K1's camera test applied to it (_has_cone).  keep = (first, n): those rows move to the front."""
        n1 = max_frames + 1
        box = torch.zeros((n1, 6), dtype=torch.int32, device=self.device)
        then = np.full((n1, 12), np.nan)
        seen = np.tile(np.array([1, -1] * 3, dtype=np.float32), (n1, 1))
        cone = np.zeros(n1, dtype=bool)
        if keep is not None:
            a, n = keep
            box[:n] = self.frame_box[a:a + n]
            then[:n], seen[:n], cone[:n] = self._then[a:a + n], self._box[a:a + n], self._has_cone[a:a + n]
        self.frame_box, self._then, self._box, self._has_cone = box, then, seen, cone
        self._then_addr, self._box_addr = then.ctypes.data, seen.ctypes.data
        self._box_pin = None     # pinned landing block of the read-backs
        self._box_copy = None    # (event, first slot, slots) of the read-back in flight
        self._since_box = 0
        self._cone = None        # (key, 15 doubles) of the camera the live frames were taken with
        self._n_nocone = 0 if keep is None else int(np.count_nonzero(~cone[:keep[1]] & np.isfinite(then[:keep[1], 0])))   # noted frames without the camera test
        self._cstore = None

    def _note_moved(self, Ts):
        """Ts (k,16): transforms applied -- now or owed -- to EVERY live frame, in order."""
        for T in Ts:
            M = T.reshape(4, 4) @ self._moved
            M[3] = (0., 0., 0., 1.)                 # (the kernels apply rows 0..2 only: whatever a T's last row says, it is not applied)
            self._moved = M

    def _note_frame(self, slot, cone_key):
        """A frame K1 is about to put into `slot`, created under the current `moved`; cone_key: (P bytes, H, W) if K1's camera
        test applies to it, else None."""
        self._then[slot] = self._moved[:3].reshape(12)      # (its row of _box says "not seen": every slot is used once between slides)
        if cone_key is not None:
            if self._cone is None or self._cone[0] != cone_key:
                if self._cone is not None:                  # another camera: the frames taken with the old one lose their cone
                    self._has_cone[self.head:self.tail] = False
                    self._n_nocone = self.tail - self.head
                out = np.empty(15)
                Pm = np.frombuffer(cone_key[0], dtype=np.float64)
                self.ctx.lib.pca_host_camera_cone(Pm.ctypes.data, int(cone_key[1]), int(cone_key[2]), out.ctypes.data)
                self._cone = (cone_key, out, out.ctypes.data)
            self._has_cone[slot] = True
        else:
            self._n_nocone += 1
        self._since_box += 1
        self._poll_boxes()
        if self._since_box >= self.BOX_EVERY and self._box_copy is None:
            n = self.tail - self.head
            if n > 0:
                if self._box_pin is None or self._box_pin.shape[0] < self.frame_box.shape[0]:
                    self._box_pin = torch.empty(tuple(self.frame_box.shape), dtype=torch.int32, pin_memory=True)
                self._box_pin[:n].copy_(self.frame_box[self.head:self.tail], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._box_copy = (ev, self.head, n)
                self._since_box = 0

    def _poll_boxes(self):
        """Takes in a finished read-back: the rows K1 had filled by then (a row of zeros stays unknown)."""
        if self._box_copy is None or not self._box_copy[0].query():
            return
        _, first, n = self._box_copy
        self._box_copy = None
        rows = self._box_pin[:n].numpy()
        dec = np.empty((n, 6), dtype=np.float32)
        self.ctx.lib.pca_f32_box_decode(rows.ctypes.data, n, dec.ctypes.data)
        known = dec[:, 0] <= dec[:, 1]
        tgt = self._box[first:first + n]
        tgt[known] = dec[known]

    def view_hint_into(self, args, first_frame, last_frame):
        """The arguments of pca_bev_view_hint for live frames [first_frame, last_frame) written into the hint_* fields of an
        argument block (pca_kitti_generate_bev_v); returns hint_F (0: no hint)."""
        n = last_frame - first_frame
        if not self.cull or n < 4:
            return 0
        a = self.head + first_frame
        args.hint_slot0 = a
        args.hint_then, args.hint_box = self._then_addr + 96 * a, self._box_addr + 24 * a
        args.hint_cone = self._cone[2] if (self._cone is not None and self._n_nocone == 0) else None
        args.hint_now = self._moved.ctypes.data
        return n

    def view_hint(self, prm, first_frame, last_frame, write_back):
        """Tells the context which slots of live frames [first_frame, last_frame) can reach the view of `prm` (pca_bev_view_hint)
        before a raster that writes nothing back.  Returns 1 if frames will be left out."""
        n = last_frame - first_frame
        if not self.cull or write_back or n < 4:
            return 0
        a = self.head + first_frame
        cone = self._cone[2] if (self._cone is not None and self._n_nocone == 0) else None
        return self._count_hint(self.ctx.lib.pca_bev_view_hint(self.ctx.h, a, n, self._then_addr + 96 * a, self._box_addr + 24 * a, cone,
                                              self._moved.ctypes.data, C.addressof(prm)))

    def _count_hint(self, r):
        self.hints_taken += 1 if r > 0 else 0
        return r

    def c_store(self):
        if self._cstore is None:
            self._cstore = PcaStore(self.x.data_ptr(), self.y.data_ptr(), self.z.data_ptr(), self.intensity.data_ptr(),
                                    self.rgbs.data_ptr(), self.inst.data_ptr(), self.dyn.data_ptr(), self.capacity,
                                    self.frame_box.data_ptr())
        return self._cstore

    @property
    def n_frames(self):
        return self.tail - self.head

    def set_defer_k1(self, on):
        """pca_k1_defer for this store's context: the K1 of append_kitti_obs is left for the raster that follows it (it rides in
        that raster's first kernel; see include/pca.h).  Every library call that touches a store runs a noted K1 first; the
        methods of this class that read the arrays with torch (offsets, rows, reserve, ...) call flush_k1() themselves.  Code
        that reads self.x ... self.frame_off directly must call flush_k1() first."""
        self.ctx.check(self.ctx.lib.pca_k1_defer(self.ctx.h, 1 if on else 0))
        self.ctx.k1_defer = bool(on)              # (the switch belongs to the context: every store on it sees it)
        if not on:
            self._k1_noted = None

    def flush_k1(self):
        if self._k1_noted is not None:
            self.ctx.check(self.ctx.lib.pca_k1_flush(self.ctx.h))
            self._k1_noted = None

    def __del__(self):
        # a K1 noted for THIS store must not outlive its arrays (the context would run it later, into memory that torch has
        # handed to someone else by then)
        try:
            if self._k1_noted is not None:
                self.flush_k1()
        except Exception:
            pass

    def offsets(self):
        """Exact segment boundaries of the live frames (host numpy int64, length n_frames+1).  Synchronises."""
        self.flush_k1()
        off = self.frame_off[self.head:self.tail + 1].cpu().numpy()
        self.lb_head = int(off[0])
        self.ub_tail = int(off[-1])
        self._ub = [int(v) for v in np.diff(off)]
        self._ub_sum = sum(self._ub)
        return off

    def sizes(self):
        return np.diff(self.offsets())

    def poll_status(self):
        """check_status() without the wait (the host-visible mirror of the status word); for every call."""
        self.ctx.poll_status()

    def check_status(self):
        self.ctx.check_status()

    def reserve(self, n_new, n_slots=1):
        """Make room for n_slots more frames holding at most n_new points in total."""
        if self.tail + n_slots <= self.max_frames and self.ub_tail + n_new <= self.capacity:
            return
        self.flush_pending()
        off = self.offsets()                       # synchronises: exact numbers
        a, b = int(off[0]), int(off[-1])
        live, nf = b - a, self.n_frames
        new_cap = self.capacity
        while live + n_new > new_cap:
            new_cap *= 2
        new_maxf = self.max_frames
        while nf + n_slots > new_maxf:
            new_maxf *= 2
        old = self._arrays()
        if new_cap != self.capacity:
            self._alloc(new_cap)
            for dst, src in zip(self._arrays(), old):
                dst[:live] = src[a:b]
        elif a > 0:                                # slide the live window to the front
            for t in old:
                t[:live] = t[a:b].clone()
        self.max_frames = new_maxf
        new_off = torch.zeros(new_maxf + 1, dtype=torch.int64, device=self.device)
        new_off[:nf + 1] = torch.from_numpy(off - a).to(self.device)
        self.frame_off = new_off
        cone = self._cone
        self._alloc_frame_tables(new_maxf, keep=(self.head, nf))   # the live frames keep their rows, every other slot starts empty
        self._cone = cone
        self.head, self.tail = 0, nf
        self.lb_head, self.ub_tail = 0, live

    def evict(self, k):
        k = int(k)
        if self._n_nocone and k:                   # (frames K1 noted without its camera test: the others have no `then` at all)
            sl = slice(self.head, self.head + k)
            self._n_nocone -= int(np.count_nonzero(~self._has_cone[sl] & np.isfinite(self._then[sl, 0])))
        self.head += k
        self._ub_sum -= sum(self._ub[:int(k)])
        del self._ub[:int(k)]

    def max_window_points(self):
        return max(self._ub_sum, 1)

    def clear(self):
        self.flush_k1()
        self.head = self.tail = 0
        self.ub_tail = self.lb_head = 0
        self._ub = []
        self._ub_sum = 0
        self._pending = []
        self.frame_off.zero_()
        self._alloc_frame_tables(self.max_frames)
        self._moved = np.eye(4)

    # ---- K1: KITTI ------------------------------------------------------------------------
    @staticmethod
    def kitti_descs(frames):
        """The C descriptors of a list of frame dicts (see append_kitti)."""
        descs = (PcaKittiFrame * len(frames))()
        for k, f in enumerate(frames):
            n = int(f['pts'].shape[0])
            assert f['pts'].dtype == torch.float32 and f['pts'].is_contiguous()
            descs[k].pts = f['pts'].data_ptr() if n else None
            descs[k].rgb = f['rgb'].data_ptr() if f.get('rgb') is not None else None
            descs[k].sem = f['sem'].data_ptr() if f.get('sem') is not None else None
            descs[k].sem_gt = f['sem_gt'].data_ptr() if f.get('sem_gt') is not None else None
            descs[k].n = n
        return descs

    def append_kitti(self, frames, P, H, W, filters, descs=None, sample_mode='nearest'):
        """frames: list of dicts {pts (n,4) f32 cuda, rgb (H,W,3) u8 cuda | None, sem (H,W) u8 cuda | None,
        sem_gt (n,) u8 cuda | None}.  One call appends all of them (stable order) as new slots.
        descs: kitti_descs(frames) built earlier (a caller that replays the same batch).
        sample_mode: 'nearest' (the reference) or 'bilinear' (opt-in: bilinear rgb, class of the nearest pixel)."""
        lib, ctx = self.ctx.lib, self.ctx
        n_in = sum(int(f['pts'].shape[0]) for f in frames)
        self.reserve(n_in, len(frames))
        if descs is None:
            descs = self.kitti_descs(frames)
        st = self.c_store()
        # keyed on VALUES (12 doubles + a short list): a calibration or filter list mutated in place is seen
        key = (np.asarray(P, dtype=np.float64).tobytes(), tuple(int(c) for c in (filters or ())))
        if self._k1_cache is None or self._k1_cache[0] != key:
            self._k1_cache = (key, _lib.f64_array(P, 12), _lib.class_mask(filters))
        Pc, fmask = self._k1_cache[1], self._k1_cache[2]
        ctx.check(lib.pca_kitti_project_sample_filter_ex(ctx.h, descs, len(frames), Pc, int(H), int(W),
                                                         fmask, C.byref(st), self.frame_off.data_ptr(), self.tail,
                                                         _lib.SAMPLE_MODES[sample_mode], ctx.stream()))
        self.tail += len(frames)
        self.ub_tail += n_in
        self._ub += [int(f['pts'].shape[0]) for f in frames]
        self._ub_sum += n_in

    def append_kitti_obs(self, obs, P, H, W, filters, sample_mode='nearest', track=None, T_new_prev=None, horizon=0., keep=None):
        """One observation through pca_kitti_integrate: `obs` is a _lib.PcaKittiObs whose pointers are device pointers or --
        where its host_mask says so -- host arrays (staged by the library: one pinned block, one H2D copy).  With `track` (a
        host_logic.CPoseTrack) the pose bookkeeping of the frame happens in the same call: returns (evicted frames, path
        length | None); without, (None, None) and the caller steps its own track.  The caller evicts.
        keep: the objects behind the struct's device pointers; with set_defer_k1 they are held until the noted K1 has run."""
        lib, ctx = self.ctx.lib, self.ctx
        n = int(obs.n)
        self.reserve(n, 1)
        # keyed on VALUES (a calibration or filter list mutated in place is seen); the same objects as last time take the short way
        src = self._k1_src
        if src is not None and src[0] is P and src[1] is filters and P.tobytes() == src[2] and tuple(filters or ()) == src[3]:
            key = self._k1_cache[0]
        else:
            key = (np.asarray(P, dtype=np.float64).tobytes(), tuple(int(c) for c in (filters or ())))
            ok = isinstance(P, np.ndarray) and P.dtype == np.float64 and all(type(c) is int for c in (filters or ()))
            self._k1_src = (P, filters, key[0], key[1]) if ok else None
        if self._k1_cache is None or self._k1_cache[0] != key:
            self._k1_cache = (key, _lib.f64_array(P, 12), _lib.class_mask(filters))
        Pc, fmask = self._k1_cache[1], self._k1_cache[2]
        st = self.c_store()
        th = getattr(track, '_h', None)
        # the call's arguments live in a block that is kept between calls (pca_kitti_integrate_v): what does not change from
        # frame to frame -- calibration, filter, store, track -- is written when it changes, the rest per call
        a = self._ia
        if a is None:
            a = self._ia = _lib.PcaKittiIntegrateArgs()
            self._ia_T = np.zeros(16)
            self._ia_T_addr = self._ia_T.ctypes.data
            self._ia_const = None
        const = (id(obs), id(Pc), int(H), int(W), id(st), id(self.frame_off), sample_mode, th, float(horizon))
        if const != self._ia_const:
            a.obs, a.P, a.H, a.W, a.filter_mask = C.addressof(obs), C.addressof(Pc), int(H), int(W), C.addressof(fmask)
            a.store, a.frame_off, a.sample_mode = C.addressof(st), self.frame_off.data_ptr(), _lib.SAMPLE_MODES[sample_mode]
            a.track, a.horizon = getattr(th, 'value', th), float(horizon)
            self._ia_const = const
            self._ia_keep = (obs, Pc, fmask, st, self.frame_off)     # (what the addresses point into)
        a.slot = self.tail
        if th is not None:
            self._ia_T[:] = np.asarray(T_new_prev, dtype=np.float64).reshape(16)
            a.T_new_prev = self._ia_T_addr
        else:
            a.T_new_prev = None
        a.stream = ctx.stream_int() or None
        ctx.check(lib.pca_kitti_integrate_v(ctx.h, C.addressof(a)))
        self._k1_noted = (keep, ) if getattr(ctx, 'k1_defer', False) else None      # (the K1 noted by the call before has run by now)
        self._note_frame(self.tail, (key[0], int(H), int(W)) if obs.sem_gt is None else None)
        self.tail += 1
        self.ub_tail += n
        self._ub.append(n)
        self._ub_sum += n
        if th is None:
            return None, None
        v = a.path_length
        return int(a.evicted), (None if v != v else np.float64(v))

    # ---- K1n: NuScenes --------------------------------------------------------------------
    def append_nusc(self, pc, cam_idx, imgs, sems, T, filters, sample_mode='nearest'):
        lib, ctx = self.ctx.lib, self.ctx
        n = int(pc.shape[0])
        self.reserve(n)
        ncam, H, W = sems.shape
        st = self.c_store()
        ctx.check(lib.pca_nusc_sample_filter_transform_ex(ctx.h, pc.data_ptr(), cam_idx.data_ptr(), n, imgs.data_ptr(),
                                                          sems.data_ptr(), int(ncam), int(H), int(W),
                                                          _lib.f64_array(T, 16), _lib.class_mask(filters),
                                                          C.byref(st), self.frame_off.data_ptr(), self.tail,
                                                          _lib.SAMPLE_MODES[sample_mode], ctx.stream()))
        self.tail += 1
        self.ub_tail += n
        self._ub.append(n)
        self._ub_sum += n

    def append_nusc_many(self, frames, filters, sample_mode='nearest'):
        """frames: list of dicts {pc (n,7) f64 cuda, cam_idx (n,) i64 cuda, imgs (ncam,H,W,3) u8 cuda, sems (ncam,H,W) u8
        cuda, T (4,4) T_ego_world}.  ONE front + one append launch for all of them (pca_nusc_sample_filter_transform_batch):
        the stored rows are those of len(frames) append_nusc calls in order."""
        lib, ctx = self.ctx.lib, self.ctx
        if not frames:
            return
        ncam, H, W = (int(v) for v in frames[0]['sems'].shape)
        max_tiles = 16384
        b0 = 0
        while b0 < len(frames):                               # the C call takes at most 16384 tiles of 512 points
            tiles, b1 = 0, b0
            while b1 < len(frames):
                t = max((int(frames[b1]['pc'].shape[0]) + 511) // 512, 1)
                if b1 > b0 and tiles + t > max_tiles:
                    break
                tiles += t
                b1 += 1
            part = frames[b0:b1]
            n_in = sum(int(f['pc'].shape[0]) for f in part)
            self.reserve(n_in, len(part))
            descs = (_lib.PcaNuscFrame * len(part))()
            keep = []
            for k, f in enumerate(part):
                assert f['pc'].dtype == torch.float64 and f['pc'].is_contiguous() and f['cam_idx'].dtype == torch.int64
                assert tuple(f['sems'].shape) == (ncam, H, W) and f['imgs'].is_contiguous() and f['sems'].is_contiguous()
                n = int(f['pc'].shape[0])
                Tc = _lib.f64_array(f['T'], 16)
                keep.append(Tc)
                descs[k].pc = f['pc'].data_ptr() if n else None
                descs[k].cam_idx = f['cam_idx'].data_ptr() if n else None
                descs[k].imgs, descs[k].sems = f['imgs'].data_ptr(), f['sems'].data_ptr()
                descs[k].n = n
                descs[k].T = C.cast(Tc, C.POINTER(C.c_double))
            st = self.c_store()
            ctx.check(lib.pca_nusc_sample_filter_transform_batch(ctx.h, descs, len(part), ncam, H, W, _lib.class_mask(filters),
                                                                 C.byref(st), self.frame_off.data_ptr(), self.tail,
                                                                 _lib.SAMPLE_MODES[sample_mode], ctx.stream()))
            self.tail += len(part)
            self.ub_tail += n_in
            self._ub += [int(f['pc'].shape[0]) for f in part]
            self._ub_sum += n_in
            b0 = b1

    # ---- K2 / K3 --------------------------------------------------------------------------
    def retransform(self, Ts, defer=False):
        """Applies the 4x4 transform(s) to every live point, in order (Ts: (4,4) or (k,4,4)).
        defer=True (single transform): the transform is only recorded; it is applied by the next consumer
        of the coordinates -- fused into the BEV rasteriser's first pass if that comes next (it reads every
        coordinate anyway), otherwise by a K2 launch (`flush_pending`).  Up to CHAIN_MAX transforms may be owed at a
        time: the rasteriser applies all of them to what it reads and writes the result back only every CHAIN_K-th time
        (the write-back is 24 B per stored point).  Results are bit-identical either way: one fma chain per transform."""
        if self.n_frames == 0:
            return
        Ts = np.ascontiguousarray(Ts, dtype=np.float64).reshape(-1, 16)
        self._note_moved(Ts)
        if defer and Ts.shape[0] == 1:
            if len(self._pending) >= self.CHAIN_MAX:
                self.flush_pending()
            self._pending.append((Ts[0].copy(), self.tail))
            return
        self.flush_pending()
        self._launch_retransform(Ts, self.head, self.tail)

    def _launch_retransform(self, Ts, begin_slot, end_slot):
        st = self.c_store()
        ctx = self.ctx
        ctx.check(ctx.lib.pca_retransform(ctx.h, C.byref(st), self.frame_off.data_ptr(), begin_slot, end_slot,
                                          _lib.f64_array(Ts, Ts.size), Ts.shape[0], ctx.stream()))

    def retransform_batch(self, Ts, n_new):
        """After append_kitti of `n_new` frames in one call: applies the per-frame transforms Ts (n_new,4,4) exactly as
        n_new successive integrate() steps would have -- frames stored before the batch get all of them, frame i of the
        batch gets Ts[i+1:] (nothing is ever transformed by its own frame's T)."""
        Ts = np.ascontiguousarray(Ts, dtype=np.float64).reshape(-1, 16)
        assert Ts.shape[0] == n_new
        self.flush_pending()
        self._note_moved(Ts)                      # (the batch's own frames have no `then`: they always count as visible)
        first_new = self.tail - n_new
        if first_new > self.head:
            st, ctx = self.c_store(), self.ctx
            ctx.check(ctx.lib.pca_retransform(ctx.h, C.byref(st), self.frame_off.data_ptr(), self.head, first_new,
                                              _lib.f64_array(Ts, Ts.size), Ts.shape[0], ctx.stream()))
        st, ctx = self.c_store(), self.ctx
        ctx.check(ctx.lib.pca_retransform_batch_tail(ctx.h, C.byref(st), self.frame_off.data_ptr(), first_new, n_new,
                                                     _lib.f64_array(Ts, Ts.size), ctx.stream()))

    def flush_pending(self):
        """Applies the owed transforms with K2: the slots between two consecutive end slots owe the same transforms (all
        the later ones), so every stored point is touched once."""
        pend, self._pending = self._pending, []
        begin = self.head
        for k, (_, end_slot) in enumerate(pend):
            if end_slot > begin:
                self._launch_retransform(np.stack([T for T, _ in pend[k:]]), begin, end_slot)
                begin = end_slot

    def mark_dynamic(self, pairs):
        """pairs: iterable of (frame index in the live window, instance index)."""
        pairs = list(pairs)
        if not pairs:
            return
        slots = (C.c_int32 * len(pairs))(*[self.head + int(f) for f, _ in pairs])
        insts = (C.c_int32 * len(pairs))(*[int(i) for _, i in pairs])
        st = self.c_store()
        ctx = self.ctx
        ctx.check(ctx.lib.pca_mark_dynamic(ctx.h, C.byref(st), self.frame_off.data_ptr(), slots, insts, len(pairs),
                                           ctx.stream()))

    def voxel_dedup(self, voxel_size):
        """Opt-in (no reference counterpart): of all live points in one voxel floor(xyz / voxel_size) only the first
        in store order (the oldest observation) stays; frames are compacted in place, order-preserving."""
        if self.n_frames == 0:
            return
        self.flush_pending()
        ctx, lib = self.ctx, self.ctx.lib
        max_points = self.max_window_points()
        need = lib.pca_voxel_dedup_workspace_bytes(max_points, self.n_frames)
        if self._dedup_ws is None or self._dedup_ws.numel() < need:
            self._dedup_ws = torch.empty(int(need) + 256, dtype=torch.uint8, device=self.device)
        st = self.c_store()
        ctx.check(lib.pca_voxel_dedup(ctx.h, C.byref(st), self.frame_off.data_ptr(), self.head, self.tail,
                                      float(voxel_size), max_points, self._dedup_ws.data_ptr(),
                                      self._dedup_ws.numel(), ctx.stream()))

    # ---- BEV ------------------------------------------------------------------------------
    def bev(self, split_frame, prm, want_f64=False, intensity64=None, first_frame=0, last_frame=None, out16=None,
            extra=None):
        """Rasterises live frames [first_frame, last_frame) with 'present' = frames before split_frame.
        Returns (planes_f16 [21,px,px] cuda float16, planes_f64 or None).
        extra: optional cuda f64 [3, len(_lib.BEV_EXTRA_PLANES), px, px] receiving the opt-in extra reducers."""
        ctx, lib = self.ctx, self.ctx.lib
        last_frame = self.n_frames if last_frame is None else last_frame
        px = int(prm.px)
        max_points = self.bev_workspace(px)
        if out16 is not None:
            assert out16.dtype == torch.float16 and out16.is_contiguous() and tuple(out16.shape) == (21, px, px)
        p16 = out16 if out16 is not None else torch.empty((21, px, px), dtype=torch.float16, device=self.device)
        p64 = torch.empty((21, px, px), dtype=torch.float64, device=self.device) if want_f64 else None
        st = self.c_store()
        n_pend, pend_T, pend_ends, write_back = self.bev_pending(first_frame, last_frame)
        if extra is not None:
            assert extra.dtype == torch.float64 and extra.is_contiguous() \
                and tuple(extra.shape) == (3, len(_lib.BEV_EXTRA_PLANES), px, px)
        self.view_hint(prm, first_frame, last_frame, n_pend > 0 and write_back)     # (right in front of the call it is meant for)
        ctx.check(lib.pca_bev_generate_chain(ctx.h, C.byref(st), None if intensity64 is None else intensity64.data_ptr(),
                                             self.frame_off.data_ptr(), self.head + first_frame, self.head + split_frame,
                                             self.head + last_frame, max_points, C.byref(prm), pend_T, pend_ends, n_pend,
                                             write_back, self._ws.data_ptr(), self._ws.numel(),
                                             None if p64 is None else p64.data_ptr(), p16.data_ptr(),
                                             None if extra is None else extra.data_ptr(), ctx.stream()))
        self.bev_done(write_back)                  # only now: a failed call above leaves the owed re-transforms owed
        return p16, p64

    def bev_workspace(self, px):
        """Makes sure the raster's scratch fits the live window at `px`; returns the window's point bound (max_points)."""
        max_points = self.max_window_points()
        if self._ws is None or max_points > self._ws_points or px != self._ws_px:
            # sized with headroom so that a window growing frame by frame does not reallocate every call
            self._ws_points, self._ws_px = int(max_points * 1.25) + 1, px
            need = self.ctx.lib.pca_bev_workspace_bytes(self._ws_points, px)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(int(need) + 256, dtype=torch.uint8, device=self.device)
        return max_points

    def bev_pending(self, first_frame, last_frame):
        """The owed re-transforms as a raster over live frames [first_frame, last_frame) takes them (oldest first; they are
        written back only when CHAIN_K of them are owed): (n_pend, Ts, slot ends, write_back) for pca_bev_generate_chain.
        A raster that does not cover what they are owed to gets them flushed first (K2)."""
        if not self._pending:
            return 0, None, None, 1
        if first_frame != 0 or self._pending[-1][1] > self.head + last_frame:
            self.flush_pending()
            return 0, None, None, 1
        n_pend = 0
        for T, e in self._pending:
            if e > self.head:
                self._pend_T_np[16 * n_pend:16 * n_pend + 16] = T
                self._pend_ends[n_pend] = int(e)
                n_pend += 1
        if n_pend == 0:
            return 0, None, None, 0
        return n_pend, self._pend_T, self._pend_ends, (1 if n_pend >= self.CHAIN_K else 0)

    def bev_done(self, write_back):
        """After a raster call of this store has returned."""
        if self._pending and write_back:
            self._pending = []
        self._k1_noted = None                      # (a raster takes a noted K1 along or runs it first: nothing is noted any more)

    def bev_many(self, jobs, out16):
        """jobs: [(split_frame, prm, first_frame, last_frame | None)]; out16: cuda float16 [len(jobs),21,px,px].  All rasters
        in ONE launch of each kernel (pca_bev_generate_many): equal to len(jobs) bev() calls.  Owed transforms are applied
        first (K2)."""
        ctx, lib = self.ctx, self.ctx.lib
        n = len(jobs)
        if n == 0:
            return out16
        self.flush_pending()
        px = int(jobs[0][1].px)
        assert out16.dtype == torch.float16 and out16.is_contiguous() and tuple(out16.shape) == (n, 21, px, px)
        max_points = self.max_window_points()
        per = (int(lib.pca_bev_workspace_bytes(max_points, px)) + 255) & ~255
        need = per * n + 512
        if self._ws_many is None or self._ws_many.numel() < need:
            self._ws_many = torch.empty(int(need * 1.25), dtype=torch.uint8, device=self.device)
        cj = (_lib.PcaBevJob * n)()
        plane_bytes = 21 * px * px * 2
        for k, (split, prm, first, last) in enumerate(jobs):
            last = self.n_frames if last is None else last
            cj[k].slot_begin, cj[k].slot_split, cj[k].slot_end = self.head + first, self.head + split, self.head + last
            C.memmove(C.addressof(cj[k].prm), C.addressof(prm), C.sizeof(PcaBevParams))
            cj[k].planes = None
            cj[k].planes_f16 = out16.data_ptr() + k * plane_bytes
        st = self.c_store()
        ctx.check(lib.pca_bev_generate_many(ctx.h, C.byref(st), None, self.frame_off.data_ptr(), cj, n, max_points,
                                            self._ws_many.data_ptr(), self._ws_many.numel(), ctx.stream()))
        return out16

    # ---- host views (synchronise) -----------------------------------------------------------
    def rows(self, frame=None):
        """(M,10) f64 rows exactly as the reference keeps them in sem_pcs (one frame, or all live points)."""
        self.flush_pending()
        off = self.offsets()
        lo, hi = (int(off[0]), int(off[-1])) if frame is None else (int(off[frame]), int(off[frame + 1]))
        out = np.empty((hi - lo, 10))
        out[:, 0] = self.x[lo:hi].cpu().numpy()
        out[:, 1] = self.y[lo:hi].cpu().numpy()
        out[:, 2] = self.z[lo:hi].cpu().numpy()
        i = self.intensity[lo:hi].cpu().numpy().astype(np.float64)
        out[:, 3] = i / 255. if self.intensity_div255 else i
        r = self.rgbs[lo:hi].cpu().numpy().view(np.uint32)
        out[:, 4], out[:, 5], out[:, 6], out[:, 7] = r & 255, (r >> 8) & 255, (r >> 16) & 255, r >> 24
        out[:, 8] = self.inst[lo:hi].cpu().numpy()
        out[:, 9] = self.dyn[lo:hi].cpu().numpy()
        return out

    def frame_rows(self):
        off = self.offsets()
        allrows = self.rows()
        base = int(off[0])
        return [allrows[int(off[k]) - base:int(off[k + 1]) - base] for k in range(self.n_frames)]

    def load_rows(self, rows_list, intensity64_out=None):
        """Replaces the content by host (M,10) f64 arrays, one frame each.  Returns a cuda f64 intensity
        tensor if column 3 is not representable in the store's f32(+/255) encoding, else None."""
        self.clear()
        total = int(sum(r.shape[0] for r in rows_list))
        if total > self.capacity or len(rows_list) > self.max_frames:
            self.max_frames = max(self.max_frames, len(rows_list))
            self._alloc(max(total, 1))
            self.frame_off = torch.zeros(self.max_frames + 1, dtype=torch.int64, device=self.device)
            self._alloc_frame_tables(self.max_frames)
        rows = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1, 10) for r in rows_list]) if rows_list \
            else np.zeros((0, 10))
        c = rows[:, 4:8]
        if rows.size and not (np.all(c == np.floor(c)) and c.min() >= 0 and c.max() <= 255):
            raise NotImplementedError('rgb / semantic columns must hold integers 0..255')
        if rows.size and not np.all(rows[:, 8] == np.floor(rows[:, 8])):
            raise NotImplementedError('instance column must hold integers')
        d = self.device
        self.x[:total] = torch.from_numpy(np.ascontiguousarray(rows[:, 0])).to(d)
        self.y[:total] = torch.from_numpy(np.ascontiguousarray(rows[:, 1])).to(d)
        self.z[:total] = torch.from_numpy(np.ascontiguousarray(rows[:, 2])).to(d)
        inten = rows[:, 3]
        raw = np.rint(inten * 255.) if self.intensity_div255 else inten
        raw32 = raw.astype(np.float32)
        back = raw32.astype(np.float64) / 255. if self.intensity_div255 else raw32.astype(np.float64)
        i64 = None
        if not np.array_equal(back, inten) or (inten.size and (inten.min() < 0 or np.signbit(inten).any())):
            i64 = torch.from_numpy(np.ascontiguousarray(inten)).to(d)
        self.intensity[:total] = torch.from_numpy(raw32).to(d)
        cu = c.astype(np.uint32)
        packed = (cu[:, 0] | (cu[:, 1] << 8) | (cu[:, 2] << 16) | (cu[:, 3] << 24)).astype(np.uint32)
        self.rgbs[:total] = torch.from_numpy(packed.view(np.int32)).to(d)
        self.inst[:total] = torch.from_numpy(rows[:, 8].astype(np.int32)).to(d)
        self.dyn[:total] = torch.from_numpy((rows[:, 9] == 1).astype(np.uint8)).to(d)
        off = np.concatenate([[0], np.cumsum([r.shape[0] for r in rows_list])]).astype(np.int64)
        self.frame_off[:off.size] = torch.from_numpy(off).to(d)
        self.head, self.tail = 0, len(rows_list)
        self.lb_head, self.ub_tail = 0, total
        self._ub = [int(r.shape[0]) for r in rows_list]
        self._ub_sum = sum(self._ub)
        return i64


def make_bev_params(origin, R, dx, dy, view, px, height_filter, int_scaler, int_sep_scaler, int_mid_threshold,
                    road_class, dynobj_classes, intensity_div255, rgb_fill=0.):
    prm = PcaBevParams()
    prm.origin[:] = [float(v) for v in origin]
    prm.R[:] = [float(v) for v in np.asarray(R, dtype=np.float64).ravel()]
    prm.dx, prm.dy, prm.view = float(dx), float(dy), float(view)
    prm.height_filter = float('nan') if height_filter is None else float(height_filter)
    prm.int_scaler = float(int_scaler)
    prm.int_sep_scaler = float(int_sep_scaler)
    prm.int_mid_threshold = float(int_mid_threshold)
    prm.rgb_fill = float(rgb_fill)
    prm.px = int(px)
    prm.road_class = int(road_class)
    prm.dynobj_mask[:] = list(_lib.class_mask(dynobj_classes))
    prm.intensity_div255 = int(bool(intensity_div255))
    return prm
