"""Sharded runner on the GPU: batched integrate == call-by-call integrate, chunked run == sequential run (bit for bit),
also with two processes (one chunk each) whose BEVs meet on rank 0 over gloo."""
import os
import socket

import numpy as np
import pytest

import sharded_common as sc

pytestmark = pytest.mark.gpu


def test_integrate_many_equals_integrate_call_by_call():
    Ts = sc.transforms(90)
    a, b = sc.make_accumulator(Ts), sc.make_accumulator(Ts)
    qa, qb = list(Ts), list(Ts)
    a.pose_provider = lambda pc: qa.pop(0)
    b.pose_provider = lambda pc: qb.pop(0)
    removed_a = [a.integrate(sc.observation(0, f)) for f in range(90)]
    removed_b = []
    for lo, hi in ((0, 1), (1, 30), (30, 31), (31, 73), (73, 90)):          # single frames, short and long batches
        removed_b += b.integrate_many([sc.observation(0, f) for f in range(lo, hi)], max_frames_per_launch=20)
    assert removed_a == removed_b and sum(removed_a) > 10                    # evictions happened inside batches
    assert np.array_equal(np.array(a.poses), np.array(b.poses))
    assert np.array_equal(np.array(a.seg_dists), np.array(b.seg_dists))
    ra, rb = a.sem_pcs, b.sem_pcs
    assert len(ra) == len(rb) and all(np.array_equal(x, y) for x, y in zip(ra, rb))
    assert len(a.rgbs) == len(b.rgbs) == len(a.poses)
    a.store.check_status()
    b.store.check_status()


def test_two_chunks_with_warm_up_equal_the_sequential_run():
    from pca_amd import sharded_run as sr
    Ts = sc.transforms(330)
    jobs, _, samples = sr.plan([Ts], 2, sc.ACCUM_H, sc.BEV_H, sc.SPACING)
    chunks = sorted((j for r in jobs for j in r), key=lambda j: j.start)
    assert len(chunks) == 2 and chunks[1].warm_start > 50 and len(samples[0]) > 150
    whole = sr.ChunkJob(0, 0, 0, 330, samples[0])
    ref = sc.run_job(whole, Ts)
    got = {}
    for j in chunks:
        got.update(sc.run_job(j, Ts))
    assert sorted(got) == sorted(ref) == [f for f, _ in samples[0]]
    for f in ref:
        assert np.array_equal(got[f].view(np.uint16), ref[f].view(np.uint16)), f


def test_two_chunks_on_two_concurrent_lanes_equal_the_sequential_run():
    """Two chunks at the SAME time on one GPU -- a lane each: own pca_ctx, own stream, own host thread (sharded_run.run_on_lanes)
    -- give, BEV for BEV, the bits of the sequential single-lane run; so do two whole sequences side by side."""
    import threading

    from pca_amd import _lib
    from pca_amd import sharded_run as sr
    Ts = sc.transforms(330)
    jobs, _, samples = sr.plan([Ts], 2, sc.ACCUM_H, sc.BEV_H, sc.SPACING)
    chunks = sorted((j for r in jobs for j in r), key=lambda j: j.start)
    assert len(chunks) == 2
    ref = sc.run_job(sr.ChunkJob(0, 0, 0, 330, samples[0]), Ts)
    seen_ctx = set()

    def run_job(job, lane):
        seen_ctx.add((threading.get_ident(), _lib.Context.get().h.value))
        return sc.run_job(job, Ts)
    per_lane = sr.run_on_lanes(chunks, run_job, n_lanes=2)
    assert len(per_lane) == 2 and len(seen_ctx) == 2                  # two threads, two contexts
    assert len({c for _, c in seen_ctx} | {_lib.Context.get().h.value}) == 3   # neither is the process-wide one
    got = {}
    for outs in per_lane:
        for o in outs:
            got.update(o)
    assert sorted(got) == sorted(ref) == [f for f, _ in samples[0]]
    for f in ref:
        assert np.array_equal(got[f].view(np.uint16), ref[f].view(np.uint16)), f
    # two whole, DIFFERENT sequences side by side against each one alone
    Tb = sc.transforms(200, seed=11)
    _, _, sb = sr.plan([Tb], 1, sc.ACCUM_H, sc.BEV_H, sc.SPACING)
    ja, jb = sr.ChunkJob(0, 0, 0, 330, samples[0]), sr.ChunkJob(1, 0, 0, 200, sb[0])
    alone_b = sc.run_job(jb, Tb, seq=1)
    side = sr.run_on_lanes([ja, jb], lambda job, lane: sc.run_job(job, Ts if job.seq == 0 else Tb, seq=job.seq), n_lanes=2)
    both = {}
    for outs in side:
        for o in outs:
            both[len(o)] = o
    a2, b2 = both[len(ref)], both[len(alone_b)]
    assert len(ref) != len(alone_b) and len(alone_b) > 50
    assert all(np.array_equal(a2[f].view(np.uint16), ref[f].view(np.uint16)) for f in ref)
    assert all(np.array_equal(b2[f].view(np.uint16), alone_b[f].view(np.uint16)) for f in alone_b)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out):
    import torch
    import torch.distributed as dist

    from pca_amd import shard
    from pca_amd import sharded_run as sr
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    Ts = sc.transforms(260)
    jobs, _, samples = sr.plan([Ts], world, sc.ACCUM_H, sc.BEV_H, sc.SPACING)
    mine = {}
    for j in jobs[rank]:
        mine.update(sc.run_job(j, Ts))
    frames = sorted(mine)
    local = torch.from_numpy(np.stack([mine[f] for f in frames])) if frames else torch.zeros((0, 21, 32, 32), dtype=torch.float16)
    ids = torch.tensor(frames, dtype=torch.int64)
    planes = shard.gather_to_rank0(local)
    which = shard.gather_to_rank0(ids)
    ok = True
    if rank == 0:
        whole = sr.ChunkJob(0, 0, 0, 260, samples[0])
        ref = sc.run_job(whole, Ts)
        seen = {}
        for p, w in zip(planes, which):
            for k, f in enumerate(w.tolist()):
                seen[f] = p[k].numpy()
        ok = sorted(seen) == sorted(ref) and len(ref) > 100
        ok = ok and all(np.array_equal(seen[f].view(np.uint16), ref[f].view(np.uint16)) for f in ref)
        ok = ok and all(len(w) > 0 for w in which)                             # both ranks contributed
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_chunked_run_gathered_on_rank0_equals_sequential():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    with ctx.Manager() as m:
        out = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        assert out[0] is True and out[1] is True
