"""Host-side helpers that have no golden fixture of their own (CPU, no GPU needed)."""
import numpy as np
import pytest


def test_conv_semantic_ids_lookup_equals_sequential_masking():
    """datasets.kitti360_utils.conv_semantic_ids composes the sequential (old -> new) passes into one table for (N,1)
    integer arrays; the result must be that of the reference's in-order masked assignments (kitti360_utils.py:18-22 of the
    reference), chains and ids outside the table included."""
    from datasets.kitti360_utils import conv_semantic_ids
    from obs_dataloaders.kitti360_obs_dataloader import Kitti360Dataloader

    def sequential(a, d):
        for o, n in d.items():
            a[a[:, 0] == o] = n
        return a
    rng = np.random.default_rng(0)
    m = Kitti360Dataloader.gen_idx_mapping()
    for dt in (np.int16, np.int32, np.int64, np.uint8, np.float64):
        a = rng.integers(-1 if dt != np.uint8 else 0, 60, (20000, 1)).astype(dt)
        assert np.array_equal(conv_semantic_ids(a.copy(), m), sequential(a.copy(), m)), dt
    chain = {3: 7, 7: 100, 100: -5, -5: 3, 9: 9}
    a = rng.integers(-300, 300, (20000, 1)).astype(np.int16)
    assert np.array_equal(conv_semantic_ids(a.copy(), chain), sequential(a.copy(), chain))
    wide = rng.integers(0, 50, (1000, 3)).astype(np.int16)                 # not (N,1): whole rows are assigned
    assert np.array_equal(conv_semantic_ids(wide.copy(), m), sequential(wide.copy(), m))


def _random_T(rng):
    a = rng.uniform(-0.06, 0.06)
    T = np.eye(4)
    T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    T[:3, 3] = rng.uniform(-1.6, 0.4, 3) * [1.0, 0.05, 0.01]
    return T


def test_c_pose_track_equals_the_numpy_expressions_bit_for_bit():
    """pca_host_track_* (one C call per integrate()) against the numpy form that restates the reference's expressions:
    poses, segment distances, path length, evictions, incremental path distances and the driver's sample trigger --
    every value bit-equal over long random drives, incl. a track set from Python lists and negative previous_idx."""
    from pca_amd import host_logic as hl
    cls = hl._bind_c_track()
    assert cls is hl.CPoseTrack, 'the C track did not bind (library not built, or numpy\'s BLAS entry point not found)'
    rng = np.random.default_rng(77)
    for horizon, bev_h in ((37.5, 11.0), (200.0, 80.0), (5.0, 1.5)):
        c, n = hl.CPoseTrack(), hl.NumpyPoseTrack()
        prev = 0
        for f in range(420):
            T = _random_T(rng)
            rc, rn = c.step(T, horizon), n.step(T, horizon)
            assert rc == rn
            prev -= rc[0]
            if -len(n) <= prev < len(n):
                tc, tn = c.trigger(bev_h, prev, 1.0), n.trigger(bev_h, prev, 1.0)
                assert tc == tn
                if tn is not None:
                    prev = tn
            assert len(c) == len(n)
            assert np.array_equal(c.as_array(), n.as_array())
            assert np.array_equal(c.seg_array(), n.seg_array())
            if f % 7 == 0:
                assert np.array_equal(c.incr(), n.incr())
                assert c.poses == n.poses and c.seg_dists == n.seg_dists
    # the piecewise API the reference's methods map to, and assignment from lists
    c, n = hl.CPoseTrack(), hl.NumpyPoseTrack()
    for tr in (c, n):
        tr.poses = [[1., 2., 3.], [2., 2.5, 3.], [3.5, 2., 3.1]]
        tr.seg_dists = [1.1, 1.6]
        tr.apply_transform(_random_T(np.random.default_rng(5)))
        tr.append([0., 0., 0.])
        pl = tr.push_segment()
        tr.evict_beyond(2.0, pl)
    assert np.array_equal(c.as_array(), n.as_array()) and np.array_equal(c.seg_array(), n.seg_array())
    assert np.array_equal(c.pose(-1), n.pose(-1)) and len(c) == len(n)
    with pytest.raises(IndexError):
        c.pose(17)


def test_c_pose_track_full_product_fallbacks_agree():
    """With the closed form of the per-pose product and the row groups switched off (what a machine whose BLAS kernels
    differ gets), the C track still equals numpy: those paths call numpy's own dgemv."""
    from pca_amd import host_logic as hl
    assert hl._bind_c_track() is hl.CPoseTrack
    lib = hl.CPoseTrack._lib
    mode, blocks = lib.pca_host_gemv4_mode(0), lib.pca_host_incr_blocks(0)
    try:
        rng = np.random.default_rng(3)
        c, n = hl.CPoseTrack(), hl.NumpyPoseTrack()
        for f in range(150):
            T = _random_T(rng)
            assert c.step(T, 30.0) == n.step(T, 30.0)
            assert c.trigger(9.0, 0, 0.5) == n.trigger(9.0, 0, 0.5)
        assert np.array_equal(c.as_array(), n.as_array()) and np.array_equal(c.incr(), n.incr())
    finally:
        lib.pca_host_gemv4_mode(mode)
        lib.pca_host_incr_blocks(blocks)


def test_staging_pool_copies_without_a_gpu():
    """pca_host_stage_h2d's host side: the pool of threads copies ragged arrays (0 bytes .. several slices) into the staging
    blocks, also with several threads calling at once; without a GPU the H2D enqueue that follows fails (-2), which is the
    point where this test stops."""
    import ctypes as C
    import threading
    from pca_amd import _lib
    lib = _lib.load()
    errors = []

    def one(seed, reps):
        r = np.random.default_rng(seed)
        for _ in range(reps):
            n = int(r.integers(1, 5))
            srcs = [r.integers(0, 255, int(r.integers(0, 600000)), dtype=np.uint8) for _ in range(n)]
            pins = [np.zeros(max(len(s), 1), dtype=np.uint8) for s in srcs]
            vp = C.c_void_p * n
            rc = lib.pca_host_stage_h2d(n, vp(*[s.ctypes.data for s in srcs]), vp(*[p.ctypes.data for p in pins]),
                                        vp(*[p.ctypes.data for p in pins]), (C.c_int64 * n)(*[len(s) for s in srcs]), None)
            if any(len(s) for s in srcs) and rc == 0:
                import torch
                if not torch.cuda.is_available():
                    errors.append('H2D succeeded without a GPU')
            for s, p in zip(srcs, pins):
                if not np.array_equal(p[:len(s)], s):
                    errors.append(('copy', seed))
    one(1, 30)
    threads = [threading.Thread(target=one, args=(k, 25)) for k in range(2, 6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    assert lib.pca_host_stage_h2d(0, None, None, None, None, None) == 0
    assert lib.pca_host_stage_h2d(-1, None, None, None, None, None) == -1
