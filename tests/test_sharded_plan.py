"""Planning half of the sharded runner on the CPU: the host replay of the driver's sample trigger and the chunk plan."""
import numpy as np

import sharded_common as sc
from pca_amd import shard
from pca_amd import sharded_run as sr


def _positions(Ts):
    acc = np.eye(4)
    out = np.zeros((Ts.shape[0], 3))
    for f in range(1, Ts.shape[0]):
        acc = Ts[f] @ acc
        out[f] = np.linalg.inv(acc)[:3, 3]
    return out


def test_replay_agrees_with_the_position_based_trigger():
    Ts = sc.transforms(400)
    oldest, samples = sr.replay(Ts, sc.ACCUM_H, sc.BEV_H, sc.SPACING)
    ref = shard.sample_frames(_positions(Ts), sc.ACCUM_H, sc.BEV_H, sc.SPACING)
    assert len(samples) > 100
    assert [f for f, _ in samples] == [f for f, _ in ref]
    assert [p - oldest[f] for f, p in samples] == [p for _, p in ref]       # frame number -> window index
    assert oldest == sorted(oldest) and oldest[-1] > 250


def test_plan_tiles_the_sequences_and_warms_up_one_horizon():
    seqs = [sc.transforms(n, seed=n) for n in (500, 900, 120, 700)]
    jobs, loads, samples = sr.plan(seqs, 4, sc.ACCUM_H, sc.BEV_H, sc.SPACING)
    flat = [j for r in jobs for j in r]
    for s, T in enumerate(seqs):
        edges = sorted((j.start, j.end) for j in flat if j.seq == s)
        assert edges[0][0] == 0 and edges[-1][1] == T.shape[0]
        assert all(a[1] == b[0] for a, b in zip(edges[:-1], edges[1:]))
        got = sorted(x for j in flat if j.seq == s for x in j.samples)
        assert got == samples[s]                                            # every sample job lands in exactly one chunk
    for j in flat:
        assert 0 <= j.warm_start <= j.start
        if j.start:
            pos = _positions(seqs[j.seq])
            path = np.linalg.norm(np.diff(pos[j.warm_start:j.start + 1], axis=0), axis=1).sum()
            assert sc.ACCUM_H <= path <= sc.ACCUM_H + 4.0                   # one memory horizon, not more
    assert max(loads) / (sum(loads) / 4) < 1.4                             # four short sequences: coarse pieces
