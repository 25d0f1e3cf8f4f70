"""Multi-GPU path on CPU: shard planner, pose-only sample trigger, and the rank-0 gather with the gloo
backend at world_size 2 (the same code runs over RCCL with backend 'nccl')."""
import os
import socket

import numpy as np
import pytest

from pca_amd import host_logic as hl
from pca_amd import shard

KITTI360_LENGTHS = [11270, 14384, 730, 11440, 6610, 9578, 2960, 13855, 3540]   # end - start per sequence


def test_lpt_assign_balances_scenes():
    rng = np.random.default_rng(0)
    costs = list(rng.integers(38, 42, 850))          # NuScenes scenes, ~40 frames each
    items, loads = shard.lpt_assign(costs, 8)
    assert sorted(i for it in items for i in it) == list(range(850))
    assert max(loads) - min(loads) <= 42


def test_kitti_chunks_reach_six_fold_scaling():
    total = sum(KITTI360_LENGTHS)
    # sequence granularity alone cannot: LPT over nine sequences
    _, loads = shard.lpt_assign(KITTI360_LENGTHS, 8)
    assert total / max(loads) < 6.0
    per_rank, loads = shard.plan_chunks(KITTI360_LENGTHS, 8, warmup_frames=200)
    chunks = [c for r in per_rank for c in r]
    for s, n in enumerate(KITTI360_LENGTHS):          # chunks tile every sequence exactly once
        edges = sorted((c.start, c.end) for c in chunks if c.seq == s)
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(edges[:-1], edges[1:]))
    assert all(c.warm_start == max(0, c.start - 200) for c in chunks)
    assert total / max(loads) >= 6.0                  # >= 6x scene-shard scaling incl. warm-up overhead


def test_sample_trigger_is_pose_only_and_chunkable():
    # 600 frames, 1 m per frame on a gentle curve
    ang = np.cumsum(np.full(600, 0.002))
    pos = np.stack([np.cumsum(np.cos(ang)), np.cumsum(np.sin(ang)), np.zeros(600)], 1)
    jobs = shard.sample_frames(pos, accum_horizon=200, bev_horizon=80, min_spacing=1)
    frames = [f for f, _ in jobs]
    assert len(jobs) > 200 and frames == sorted(frames)
    # present pose lies ~80 m of path behind the newest pose
    f, pidx = jobs[0]
    assert 155 < f < 170 and pidx > 0                 # first sample once 80 m lie behind AND ahead of 'present'
    # a chunk that starts later, after a warm-up of one accumulation horizon + one BEV horizon, emits the
    # same samples for its own frames
    start, warm = 400, 400 - 290
    sub = shard.sample_frames(pos[warm:], 200, 80, 1)
    sub = [(f + warm, p) for f, p in sub if f + warm >= start]
    ref = [(f, p) for f, p in jobs if f >= start]
    assert [f for f, _ in sub] == [f for f, _ in ref]
    assert [p for _, p in sub] == [p for _, p in ref]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n_local = 3 if rank == 0 else 5                   # ragged: ranks finish different numbers of BEVs
    local = torch.full((n_local, 21, 8, 8), float(rank + 1), dtype=torch.float16)
    local[:, 0, 0, 0] = torch.arange(n_local, dtype=torch.float16)
    got = shard.gather_to_rank0(local)
    if rank == 0:
        ok = len(got) == world and got[0].shape[0] == 3 and got[1].shape[0] == 5
        ok = ok and bool((got[1][:, 1] == 2).all()) and got[1][:, 0, 0, 0].tolist() == [0, 1, 2, 3, 4]
        # empty shard on some rank
    else:
        ok = got is None
    empty = torch.zeros((0 if rank == 1 else 2, 21, 8, 8), dtype=torch.float16)
    got2 = shard.gather_to_rank0(empty)
    if rank == 0:
        ok = ok and got2[0].shape[0] == 2 and got2[1].shape[0] == 0
    # asynchronous batches with known sizes: two gathers in flight, results in order
    b0 = torch.full((4, 21, 8, 8), float(10 + rank), dtype=torch.float16)
    b1 = torch.full((4, 21, 8, 8), float(20 + rank), dtype=torch.float16)
    h0 = shard.gather_to_rank0(b0, async_op=True, sizes=[4, 4])
    h1 = shard.gather_to_rank0(b1, async_op=True, sizes=[4, 4])
    r0, r1 = h0.wait(), h1.wait()
    if rank == 0:
        ok = ok and float(r0[1][0, 0, 0, 0]) == 11.0 and float(r1[1][3, 20, 7, 7]) == 21.0 and r1[0].shape[0] == 4
    else:
        ok = ok and r0 is None and r1 is None
    out[rank] = ok
    dist.destroy_process_group()


def test_gather_to_rank0_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    with ctx.Manager() as m:
        out = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert out[0] is True and out[1] is True


def test_gather_without_process_group_is_identity():
    import torch
    t = torch.zeros((2, 21, 4, 4), dtype=torch.float16)
    assert shard.gather_to_rank0(t)[0] is t
    assert shard.gather_to_rank0(t, async_op=True).wait()[0].data_ptr() == t.data_ptr()


def _worker_subgroup(rank, world, port, out):
    """gather_to_rank0 inside a sub-group that does not contain global rank 0."""
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    group = dist.new_group([1, 2])                    # every rank has to make the call
    ok = True
    if rank in (1, 2):
        local = torch.full((rank + 1, 21, 4, 4), float(rank), dtype=torch.float16)
        got = shard.gather_to_rank0(local, group=group)
        if rank == 1:                                 # group rank 0
            ok = len(got) == 2 and got[0].shape[0] == 2 and got[1].shape[0] == 3 and float(got[1][0, 0, 0, 0]) == 2.0
        else:
            ok = got is None
    out[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_gather_to_rank0_in_a_subgroup_without_global_rank0():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    with ctx.Manager() as m:
        out = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker_subgroup, args=(r, 3, port, out)) for r in range(3)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert all(out[r] is True for r in range(3))
