"""pca_host_view_hull / pca_host_camera_cone (host only, no GPU): a frame reported as unable to reach the view has no point of
its region inside the view square -- checked by sampling the cone and the box; frames whose numbers say nothing stay visible."""
import ctypes as C

import numpy as np
import pytest

from pca_amd import _lib


def _lib_or_skip():
    try:
        return _lib.load()
    except Exception as e:                                        # noqa: BLE001
        pytest.skip(f'library not built: {e}')


def _affine(rng, scale=1.0):
    a = rng.uniform(-np.pi, np.pi)
    b = rng.uniform(-0.05, 0.05)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    T = np.eye(4)
    T[:3, :3] = Ry @ Rz
    T[:3, 3] = rng.uniform(-60, 60, 3) * scale * [1, 1, 0.02]
    return T


def test_cone_contains_what_k1_keeps():
    lib = _lib_or_skip()
    rng = np.random.default_rng(0)
    cam_to_velo = np.array([[0.043, -0.088, 0.995, 0.80], [-0.999, 0.0078, 0.0439, 0.30], [-0.0116, -0.996, -0.0879, -0.18],
                            [0, 0, 0, 1.]])
    H, W = 376, 1408
    P = np.array([[552.55, 0, 682.05, 0], [0, 552.55, 238.77, 0], [0, 0, 1, 0]]) @ np.linalg.inv(cam_to_velo)
    P = np.ascontiguousarray(P)
    cone = np.empty(15)
    lib.pca_host_camera_cone(P.ctypes.data, H, W, cone.ctypes.data)
    apex, rays = cone[:3], cone[3:].reshape(4, 3)
    pts = rng.uniform(-80, 80, (200_000, 3))
    uvw = pts @ P[:, :3].T + P[:, 3]
    d = uvw[:, 2]
    u, v = np.rint(uvw[:, 0] / np.abs(d)), np.rint(uvw[:, 1] / np.abs(d))
    kept = pts[(u >= 0) & (u < W) & (v >= 0) & (v < H) & (d > 0)]
    assert len(kept) > 5000
    # a kept point minus the apex is a non-negative combination of the four edge rays: it lies on the inner side of the four
    # planes through the apex that adjacent rays span (rays ordered (u0,v0), (u1,v0), (u0,v1), (u1,v1))
    rel = kept - apex
    for i, j in ((0, 1), (1, 3), (3, 2), (2, 0)):
        nrm = np.cross(rays[i], rays[j])
        inside = np.sign(nrm @ rays[[k for k in range(4) if k not in (i, j)][0]])
        assert np.all((rel @ nrm) * inside >= -1e-9)


@pytest.mark.parametrize('with_cone', [True, False])
def test_frames_outside_the_hull_have_nothing_in_the_view(with_cone):
    lib = _lib_or_skip()
    rng = np.random.default_rng(3 if with_cone else 4)
    cam_to_velo = np.array([[0.043, -0.088, 0.995, 0.80], [-0.999, 0.0078, 0.0439, 0.30], [-0.0116, -0.996, -0.0879, -0.18],
                            [0, 0, 0, 1.]])
    P = np.ascontiguousarray(np.array([[552.55, 0, 682.05, 0], [0, 552.55, 238.77, 0], [0, 0, 1, 0]]) @ np.linalg.inv(cam_to_velo))
    cone = np.empty(15)
    lib.pca_host_camera_cone(P.ctypes.data, 376, 1408, cone.ctypes.data)
    apex, rays = cone[:3], cone[3:].reshape(4, 3)
    culled_total = 0
    for trial in range(60):
        F = int(rng.integers(4, 40))
        then = np.stack([_affine(rng)[:3].reshape(12) for _ in range(F)])
        now = _affine(rng)
        box = np.empty((F, 6), dtype=np.float32)
        for f in range(F):
            lo = rng.uniform(-40, 30, 3)
            box[f, 0::2] = lo
            box[f, 1::2] = lo + rng.uniform(0.1, 60, 3)
        unknown = rng.random(F) < 0.3
        box[unknown] = (1, -1, 1, -1, 1, -1)
        if not with_cone:
            then[rng.random(F) < 0.1] = np.nan                        # frames nobody knows anything about
        prm = _lib.PcaBevParams()
        ang = rng.uniform(-np.pi, np.pi)
        prm.origin[:] = list(rng.uniform(-40, 40, 3))
        prm.R[:] = [np.cos(ang), -np.sin(ang), 0, np.sin(ang), np.cos(ang), 0, 0, 0, 1]
        prm.dx, prm.dy, prm.view = float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)), float(rng.uniform(10, 90))
        first, last = C.c_int(0), C.c_int(0)
        nowm = np.ascontiguousarray(now[:3].reshape(12))
        rc = lib.pca_host_view_hull(F, then.ctypes.data, box.ctypes.data, cone.ctypes.data if with_cone else None,
                                    nowm.ctypes.data, C.addressof(prm), C.byref(first), C.byref(last))
        assert rc == 0
        lo_f, hi_f = (first.value, last.value) if first.value >= 0 else (F, F - 1)
        half = 0.5 * prm.view
        Rv = np.array(prm.R[:]).reshape(3, 3)
        for f in list(range(0, lo_f)) + list(range(hi_f + 1, F)):
            culled_total += 1
            assert np.all(np.isfinite(then[f]))                       # an unknown frame is never left out
            A = np.vstack([then[f].reshape(3, 4), [0, 0, 0, 1]])
            M = now @ np.linalg.inv(A)
            # sample the region the frame's points can be in: the box if it is known (and the cone, if given)
            if not unknown[f]:
                c = rng.uniform(box[f, 0::2], box[f, 1::2], (4000, 3))
            else:
                assert with_cone
                lam = rng.exponential(20.0, (4000, 4)) * (rng.random((4000, 4)) < 0.6)
                c = apex + lam @ rays
            if with_cone and not unknown[f]:
                rel = c - apex                                             # keep the samples that are inside the cone as well
                ok = np.ones(len(c), bool)
                for i, j in ((0, 1), (1, 3), (3, 2), (2, 0)):
                    nrm = np.cross(rays[i], rays[j])
                    inside = np.sign(nrm @ rays[[k for k in range(4) if k not in (i, j)][0]])
                    ok &= (rel @ nrm) * inside >= 0
                # (the frame may have been left out because of the cone OR the box: points of box AND cone are what exists)
                c = c[ok]
            p = c @ M[:3, :3].T + M[:3, 3]
            x, y = p[:, 0] - prm.origin[0], p[:, 1] - prm.origin[1]
            ax = Rv[0, 0] * x + Rv[0, 1] * y + prm.dx
            ay = Rv[1, 0] * x + Rv[1, 1] * y + prm.dy
            assert not np.any((np.abs(ax) < half) & (np.abs(ay) < half)), (trial, f)
    assert culled_total > 100
