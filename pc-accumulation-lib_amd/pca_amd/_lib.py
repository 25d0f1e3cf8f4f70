"""ctypes binding of the gfx950 C-ABI library (include/pca.h).

There is no CPU fallback: if ``libpca_hip.so`` is missing or no GPU is visible the product path raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libpca_hip.so')

STATUS_STORE_OVERFLOW = 1
STATUS_UV_OUT_OF_IMAGE = 2
STATUS_NEGATIVE_INTENSITY = 4
STATUS_LOOKBACK_TIMEOUT = 8
SAMPLE_MODES = {'nearest': 0, 'bilinear': 1}

EXPORTS = ('pca_version', 'pca_ctx_create', 'pca_ctx_destroy', 'pca_last_error', 'pca_status', 'pca_status_peek', 'pca_status_mirror',
           'pca_kitti_project_sample_filter', 'pca_kitti_project_sample_filter_ex',
           'pca_nusc_sample_filter_transform', 'pca_nusc_sample_filter_transform_ex', 'pca_nusc_sample_filter_transform_batch', 'pca_sample_bilinear', 'pca_nusc_project_cams', 'pca_retransform', 'pca_retransform_batch_tail',
           'pca_mark_dynamic',
           'pca_bev_workspace_bytes', 'pca_bev_generate', 'pca_bev_generate_ex', 'pca_bev_generate_chain', 'pca_bev_generate_many', 'pca_bev_warp', 'pca_image_to_nchw_f32', 'pca_voxel_dedup_workspace_bytes', 'pca_voxel_dedup', 'pca_icp_workspace_bytes', 'pca_icp_register', 'pca_host_ego_to_grid',
           'pca_host_gemv4_probe', 'pca_host_gemv4_mode', 'pca_host_incr_probe', 'pca_host_incr_blocks', 'pca_host_track_create', 'pca_host_track_destroy', 'pca_host_track_len', 'pca_host_track_n_segments',
           'pca_host_track_poses', 'pca_host_track_segments', 'pca_host_track_set', 'pca_host_track_transform',
           'pca_host_track_append', 'pca_host_track_push_segment', 'pca_host_track_incr', 'pca_host_track_evict_beyond',
           'pca_host_track_step', 'pca_host_track_trigger', 'pca_host_stage_h2d', 'pca_host_d2h_async', 'pca_host_d2h_wait',
           'pca_kitti_integrate', 'pca_kitti_generate_bev', 'pca_k1_defer', 'pca_k1_flush', 'pca_host_camera_cone', 'pca_host_view_hull', 'pca_bev_bin_range', 'pca_bev_view_hint', 'pca_kitti_integrate_v', 'pca_kitti_generate_bev_v', 'pca_f32_box_decode',
           'pca_profile_enable', 'pca_profile_read')

KERNEL_IDS = ('kitti_project_sample_filter', 'nusc_sample_filter_transform', 'nusc_project_cams', 'retransform',
              'mark_dynamic', 'bev_bin', 'bev_scan', 'bev_scatter', 'bev_cells', 'bev_cells_heavy', 'voxel_dedup', 'bev_unit', 'icp')
BEV_EXTRA_PLANES = ('elevation_max', 'elevation_mean', 'intensity_mean')


class PcaStore(C.Structure):
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('z', C.c_void_p), ('intensity', C.c_void_p),
                ('rgbs', C.c_void_p), ('inst', C.c_void_p), ('dyn', C.c_void_p), ('capacity', C.c_int64),
                ('frame_box', C.c_void_p)]


class PcaKittiIntegrateArgs(C.Structure):
    _fields_ = [('obs', C.c_void_p), ('P', C.c_void_p), ('H', C.c_int32), ('W', C.c_int32), ('filter_mask', C.c_void_p),
                ('store', C.c_void_p), ('frame_off', C.c_void_p), ('slot', C.c_int32), ('sample_mode', C.c_int32),
                ('track', C.c_void_p), ('T_new_prev', C.c_void_p), ('horizon', C.c_double), ('evicted', C.c_int64),
                ('path_length', C.c_double), ('stream', C.c_void_p)]


class PcaKittiGenerateBevArgs(C.Structure):
    _fields_ = [('store', C.c_void_p), ('frame_off', C.c_void_p), ('slot_begin', C.c_int32), ('slot_split', C.c_int32),
                ('slot_end', C.c_int32), ('pad0', C.c_int32), ('max_points', C.c_int64), ('prm', C.c_void_p),
                ('pending_Ts', C.c_void_p), ('pending_slot_ends', C.c_void_p), ('n_pending', C.c_int32), ('write_back', C.c_int32),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_int64), ('planes_f16', C.c_void_p), ('host_planes', C.c_void_p),
                ('track', C.c_void_p), ('traj_rows', C.c_void_p), ('traj_start', C.c_void_p), ('n_rows', C.c_int32),
                ('pad1', C.c_int32), ('stream', C.c_void_p), ('hint_slot0', C.c_int32), ('hint_F', C.c_int32),
                ('hint_then', C.c_void_p), ('hint_box', C.c_void_p), ('hint_cone', C.c_void_p), ('hint_now', C.c_void_p),
                ('hinted', C.c_int32), ('pad2', C.c_int32)]


class PcaKittiObs(C.Structure):
    _fields_ = [('pts', C.c_void_p), ('rgb', C.c_void_p), ('sem', C.c_void_p), ('sem_gt', C.c_void_p),
                ('n', C.c_int32), ('host_mask', C.c_uint32)]


class PcaKittiFrame(C.Structure):
    _fields_ = [('pts', C.c_void_p), ('rgb', C.c_void_p), ('sem', C.c_void_p), ('sem_gt', C.c_void_p),
                ('n', C.c_int32), ('reserved', C.c_int32)]


class PcaNuscFrame(C.Structure):
    _fields_ = [('pc', C.c_void_p), ('cam_idx', C.c_void_p), ('imgs', C.c_void_p), ('sems', C.c_void_p),
                ('n', C.c_int32), ('reserved', C.c_int32), ('T', C.POINTER(C.c_double))]


class PcaBevParams(C.Structure):
    _fields_ = [('origin', C.c_double * 3), ('R', C.c_double * 9), ('dx', C.c_double), ('dy', C.c_double),
                ('view', C.c_double), ('height_filter', C.c_double), ('int_scaler', C.c_double),
                ('int_sep_scaler', C.c_double), ('int_mid_threshold', C.c_double), ('rgb_fill', C.c_double),
                ('px', C.c_int32), ('road_class', C.c_int32), ('dynobj_mask', C.c_uint64 * 4),
                ('intensity_div255', C.c_int32), ('pad', C.c_int32)]


class PcaBevJob(C.Structure):
    _fields_ = [('slot_begin', C.c_int32), ('slot_split', C.c_int32), ('slot_end', C.c_int32), ('reserved', C.c_int32),
                ('prm', PcaBevParams), ('planes', C.c_void_p), ('planes_f16', C.c_void_p)]


def class_mask(classes):
    """256-bit class set as 4 x uint64 (classes outside 0..255 can never match a u8 label)."""
    m = [0, 0, 0, 0]
    for c in classes or []:
        c = int(c)
        if 0 <= c < 256:
            m[c >> 6] |= 1 << (c & 63)
    return (C.c_uint64 * 4)(*m)


def f64_array(a, n):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    assert a.size == n, (a.size, n)
    return (C.c_double * n)(*a.tolist())


_lib = None


def load():
    """Loads the shared library (once) and declares the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; '
                           f'g.build()"` or `make -C pc-accumulation-lib_amd/csrc` (no CPU fallback exists)')
    # torch FIRST: it brings its own bundled HIP runtime (SONAME libamdhip64.so.7, the one this library needs too).  Loaded
    # before this library, the library binds to it and the process has ONE runtime; loaded after (a module that binds the host
    # helpers at import time, before anything imported torch), the process would hold the system's runtime AND torch's, and
    # the second of them to open the device finds "no ROCm-capable device" on the GPU boxes.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.pca_version.restype = C.c_int
    lib.pca_ctx_create.argtypes = [i32, C.POINTER(vp)]
    lib.pca_ctx_destroy.argtypes = [vp]
    lib.pca_ctx_destroy.restype = None
    lib.pca_last_error.argtypes = [vp]
    lib.pca_last_error.restype = C.c_char_p
    lib.pca_status.argtypes = [vp, vp, C.POINTER(C.c_uint32)]
    lib.pca_status_peek.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.pca_status_mirror.argtypes = [vp]
    lib.pca_status_mirror.restype = vp
    lib.pca_kitti_project_sample_filter.argtypes = [
        vp, C.POINTER(PcaKittiFrame), i32, C.POINTER(C.c_double), i32, i32, C.POINTER(C.c_uint64),
        C.POINTER(PcaStore), vp, i32, vp
    ]
    lib.pca_kitti_project_sample_filter_ex.argtypes = [
        vp, C.POINTER(PcaKittiFrame), i32, C.POINTER(C.c_double), i32, i32, C.POINTER(C.c_uint64),
        C.POINTER(PcaStore), vp, i32, i32, vp
    ]
    lib.pca_nusc_sample_filter_transform_ex.argtypes = [
        vp, vp, vp, C.c_int32, vp, vp, i32, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
        C.POINTER(PcaStore), vp, i32, i32, vp
    ]
    lib.pca_nusc_sample_filter_transform_batch.argtypes = [
        vp, C.POINTER(PcaNuscFrame), i32, i32, i32, i32, C.POINTER(C.c_uint64), C.POINTER(PcaStore), vp, i32, i32, vp
    ]
    lib.pca_sample_bilinear.argtypes = [vp, vp, i32, i32, vp, C.c_int32, vp, vp]
    lib.pca_nusc_sample_filter_transform.argtypes = [
        vp, vp, vp, C.c_int32, vp, vp, i32, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
        C.POINTER(PcaStore), vp, i32, vp
    ]
    lib.pca_nusc_project_cams.argtypes = [
        vp, vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
        C.POINTER(C.c_double), C.POINTER(C.c_double), i32, vp, vp, vp, vp
    ]
    lib.pca_retransform.argtypes = [vp, C.POINTER(PcaStore), vp, i32, i32, C.POINTER(C.c_double), i32, vp]
    lib.pca_retransform_batch_tail.argtypes = [vp, C.POINTER(PcaStore), vp, i32, i32, C.POINTER(C.c_double), vp]
    lib.pca_mark_dynamic.argtypes = [vp, C.POINTER(PcaStore), vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), i32, vp]
    lib.pca_bev_workspace_bytes.argtypes = [i64, i32]
    lib.pca_bev_workspace_bytes.restype = i64
    lib.pca_bev_generate.argtypes = [
        vp, C.POINTER(PcaStore), vp, vp, i32, i32, i32, i64, C.POINTER(PcaBevParams), C.POINTER(C.c_double), i32,
        vp, i64, vp, vp, vp
    ]
    lib.pca_bev_generate_ex.argtypes = [
        vp, C.POINTER(PcaStore), vp, vp, i32, i32, i32, i64, C.POINTER(PcaBevParams), C.POINTER(C.c_double), i32,
        vp, i64, vp, vp, vp, vp
    ]
    lib.pca_image_to_nchw_f32.argtypes = [vp, vp, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, vp]
    a = list(lib.pca_bev_generate_ex.argtypes)
    lib.pca_bev_generate_chain.argtypes = a[:9] + [vp, vp, i32, i32] + a[11:]
    lib.pca_bev_generate_many.argtypes = [vp, C.POINTER(PcaStore), vp, vp, C.POINTER(PcaBevJob), i32, i64, vp, i64, vp]
    lib.pca_bev_warp.argtypes = [vp, vp, vp, i32, i32, C.c_double, C.c_double, C.c_double, C.c_double, vp]
    lib.pca_voxel_dedup_workspace_bytes.restype = C.c_int64
    lib.pca_voxel_dedup_workspace_bytes.argtypes = [i64, i32]
    lib.pca_voxel_dedup.argtypes = [vp, C.POINTER(PcaStore), vp, i32, i32, C.c_double, i64, vp, i64, vp]
    lib.pca_icp_workspace_bytes.restype = C.c_int64
    lib.pca_icp_workspace_bytes.argtypes = [i32]
    lib.pca_icp_register.argtypes = [vp, vp, i32, vp, i32, C.c_double, C.POINTER(C.c_double), i32, C.c_double,
                                     C.c_double, vp, i64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_int), vp]
    lib.pca_host_ego_to_grid.argtypes = [vp, i32, vp, C.c_double, C.c_double, C.c_double, i32, vp, vp]
    dp = C.POINTER(C.c_double)
    lib.pca_host_gemv4_probe.argtypes = [vp, i32, vp, vp, i64, vp]
    lib.pca_host_gemv4_mode.argtypes = [i32]
    lib.pca_host_incr_probe.argtypes = [vp, vp, i64, i64, i64, vp]
    lib.pca_host_incr_blocks.argtypes = [i32]
    lib.pca_host_track_create.argtypes = [C.POINTER(vp), vp]
    lib.pca_host_track_destroy.argtypes = [vp]
    lib.pca_host_track_destroy.restype = None
    for name in ('pca_host_track_len', 'pca_host_track_n_segments'):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = i64
    for name in ('pca_host_track_poses', 'pca_host_track_segments'):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = vp
    lib.pca_host_track_set.argtypes = [vp, vp, i64, vp, i64]
    lib.pca_host_track_transform.argtypes = [vp, vp]
    lib.pca_host_track_append.argtypes = [vp, vp]
    lib.pca_host_track_push_segment.argtypes = [vp, dp]
    lib.pca_host_track_incr.argtypes = [vp, vp]
    lib.pca_host_track_evict_beyond.argtypes = [vp, C.c_double, C.c_double]
    lib.pca_host_track_evict_beyond.restype = i64
    lib.pca_host_track_step.argtypes = [vp, vp, C.c_double, dp]
    lib.pca_host_track_step.restype = i64
    lib.pca_host_track_trigger.argtypes = [vp, C.c_double, i64, C.c_double]
    lib.pca_host_track_trigger.restype = i64
    lib.pca_host_stage_h2d.argtypes = [i32, vp, vp, vp, vp, vp]
    lib.pca_host_d2h_async.argtypes = [vp, vp, vp, i64, vp]
    lib.pca_host_d2h_wait.argtypes = [vp, i32]
    lib.pca_kitti_integrate.argtypes = [vp, C.POINTER(PcaKittiObs), C.POINTER(C.c_double), i32, i32, C.POINTER(C.c_uint64),
                                        C.POINTER(PcaStore), vp, i32, i32, vp, vp, C.c_double, C.POINTER(C.c_int64),
                                        C.POINTER(C.c_double), vp]
    lib.pca_kitti_generate_bev.argtypes = [vp, C.POINTER(PcaStore), vp, i32, i32, i32, i64, C.POINTER(PcaBevParams), vp, vp,
                                           i32, i32, vp, i64, vp, vp, vp, vp, vp, C.POINTER(C.c_int32), vp]
    lib.pca_host_camera_cone.argtypes = [vp, i32, i32, vp]
    lib.pca_host_camera_cone.restype = None
    lib.pca_host_view_hull.argtypes = [i32, vp, vp, vp, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.pca_bev_bin_range.argtypes = [vp, i32, i32]
    lib.pca_bev_view_hint.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.pca_f32_box_decode.argtypes = [vp, i32, vp]
    lib.pca_f32_box_decode.restype = None
    lib.pca_kitti_integrate_v.argtypes = [vp, vp]
    lib.pca_kitti_generate_bev_v.argtypes = [vp, vp]
    lib.pca_k1_defer.argtypes = [vp, i32]
    lib.pca_k1_flush.argtypes = [vp]
    lib.pca_profile_enable.argtypes = [vp, i32]
    lib.pca_profile_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    _lib = lib
    return lib


class Context:
    """One pca_ctx per (process, GPU) -- plus one per LANE (see `Lane`): a second independent sequence on the same GPU gets
    its own context (look-back state, K1 / K1n staging, copy streams, status words are per context and assume one call
    sequence at a time).  All calls are enqueued on torch's current stream."""

    _by_device = {}
    _raw_stream = None
    _tls = threading.local()              # .lane: the Context this THREAD works on while inside `with Lane(...)`

    def __init__(self, device_index):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError('pca_amd needs an AMD GPU (torch.cuda.is_available() is False); '
                               'there is no CPU fallback')
        self.lib = load()
        self.device_index = int(device_index)
        h = C.c_void_p()
        if self.lib.pca_ctx_create(self.device_index, C.byref(h)) != 0:
            raise RuntimeError('pca_ctx_create failed')
        self.h = h
        # the device status bits as the host sees them without a stream operation (pca_status_peek): mapped once
        self._mirror = (C.c_uint32 * 8).from_address(self.lib.pca_status_mirror(h))
        self._mirror_q = (C.c_uint64 * 4).from_address(C.addressof(self._mirror))     # the same eight words, four reads

    @classmethod
    def get(cls, device=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError('pca_amd needs an AMD GPU (torch.cuda.is_available() is False); '
                               'there is no CPU fallback')
        idx = torch.cuda.current_device() if device is None else torch.device(device).index or 0
        lane = getattr(cls._tls, 'lane', None)
        if lane is not None and lane.device_index == idx:
            return lane
        if idx not in cls._by_device:
            cls._by_device[idx] = Context(idx)
        return cls._by_device[idx]

    def destroy(self):
        """Frees the context's device / pinned blocks (lanes; the per-device context lives as long as the process)."""
        if self.h is not None:
            self.lib.pca_ctx_destroy(self.h)
            self.h = None

    def stream(self):
        """Raw handle of torch's current stream on this device (every call targets it)."""
        return C.c_void_p(self.stream_int())

    def stream_int(self):
        """The same as a plain integer (0 = the default stream), e.g. for a field of an argument block."""
        raw = self._raw_stream
        if raw is None:
            import torch
            raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)      # ~10x cheaper than current_stream()
            if raw is None:
                return torch.cuda.current_stream(self.device_index).cuda_stream
            self._raw_stream = raw
        return raw(self.device_index)

    def check(self, rc):
        if rc != 0:
            raise RuntimeError('pca: ' + self.lib.pca_last_error(self.h).decode())

    def profile(self, on):
        """0/False off, 1/True per kernel launch, 2 whole units only (pca.h: pca_profile_enable)."""
        self.check(self.lib.pca_profile_enable(self.h, int(on)))

    def profile_read(self):
        """{kernel name: (total_ms, launches)} since profiling was enabled (synchronises)."""
        out = {}
        for k, name in enumerate(KERNEL_IDS):
            ms, n = C.c_double(0), C.c_int64(0)
            self.check(self.lib.pca_profile_read(self.h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def peek_status(self):
        """The status bits raised by kernels that have FINISHED, read from mapped host memory: no stream operation, no
        wait (a fraction of a microsecond).  Does not clear; `status()` does."""
        q = self._mirror_q
        if not (q[0] | q[1] | q[2] | q[3]):
            return 0
        m = self._mirror
        return sum(1 << b for b in range(8) if m[b])

    def check_status(self):
        """Synchronises, reads and clears the device status word and raises what the reference would have raised on the
        spot (AssertionError of datasets/nuscenes_utils.py:191-195) or a RuntimeError naming what went wrong."""
        st = self.status()
        if not st:
            return
        # every raised bit goes into ONE message (the word has been cleared: what is not said now is lost); the exception's
        # type is that of the first bit in the reference's own order of failure
        said = [(STATUS_UV_OUT_OF_IMAGE, AssertionError, 'pts_uv must be all inside image'),
                (STATUS_STORE_OVERFLOW, RuntimeError, 'pca: device point store overflow (points were dropped)'),
                (STATUS_LOOKBACK_TIMEOUT, RuntimeError,
                 'pca: a compaction workgroup timed out waiting for its predecessor (output invalid)'),
                (STATUS_NEGATIVE_INTENSITY, ValueError, 'pca: negative lidar intensity on the f32 path (pass intensity64 to bev())')]
        hit = [(exc, msg) for bit, exc, msg in said if st & bit]
        rest = st & ~(STATUS_UV_OUT_OF_IMAGE | STATUS_STORE_OVERFLOW | STATUS_LOOKBACK_TIMEOUT | STATUS_NEGATIVE_INTENSITY)
        if rest:
            hit.append((RuntimeError, 'pca: unknown device status bits 0x%x' % rest))
        raise hit[0][0]('; '.join(msg for _, msg in hit))

    def poll_status(self):
        """check_status() without the wait: only when the host-visible mirror shows a raised bit (a kernel that has
        finished raised it) does it synchronise and raise.  Cheap enough for every call."""
        if self.peek_status():
            self.check_status()

    def status(self):
        """Synchronises the stream and returns (and clears) the device status bits."""
        st = C.c_uint32(0)
        self.check(self.lib.pca_status(self.h, self.stream(), C.byref(st)))
        return st.value


class Lane:
    """An independent call sequence on one GPU: its own pca_ctx and its own torch stream.  Inside `with lane:` the calling
    THREAD's Context.get() is the lane's context and torch's current stream is the lane's stream, so that accumulators
    created and driven there neither share workspaces with, nor queue behind, another lane's.  What this is for: two
    independent sequences (scene shards: run_kitti360_bev_gen.py:161-173 of the reference loops over them) on one GPU, each
    driven from its own host thread -- the kernels of one fill the CUs the other's tails leave idle.  ctypes releases the GIL
    for the duration of a library call; the host-side staging pool and the constant-memory argument array of
    pca_bev_generate_many are shared by all lanes and take turns (mutex)."""

    def __init__(self, device=None, priority=0):
        import torch
        idx = torch.cuda.current_device() if device is None else torch.device(device).index or 0
        self.ctx = Context(idx)
        self.stream = torch.cuda.Stream(device=idx, priority=priority)
        self._guard = None
        self._outer = None

    def __enter__(self):
        import torch
        self._outer = getattr(Context._tls, 'lane', None)
        Context._tls.lane = self.ctx
        self._guard = torch.cuda.stream(self.stream)
        self._guard.__enter__()
        return self

    def __exit__(self, *exc):
        self._guard.__exit__(*exc)
        Context._tls.lane = self._outer
        self._guard = None
        return False

    def synchronize(self):
        self.stream.synchronize()

    def close(self):
        self.stream.synchronize()
        self.ctx.destroy()
