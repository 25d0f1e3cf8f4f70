"""Sharded KITTI-360 BEV generation: one process per GPU, sequences cut into chunks with a warm-up prefix, no
collective during compute.

The reference's driver (run_kitti360_bev_gen.py:156-280) walks the frames of a sequence, integrates each one and, when
three pose-only conditions hold (:218-240), takes a BEV sample.  With the poses known up front (GT poses, or a file of
T_new_prev matrices) that loop shards:

  1. pose-only pass (`replay`): the sample jobs (frame, present frame) of every sequence -- the decisions of the
     sequential run, replayed with its own host expressions, whatever the cut;
  2. `plan`: sequences are cut BETWEEN frames into chunks balanced by frame count (`shard.plan_chunks`); a chunk's warm-up
     starts just before the oldest frame the sequential run still holds when it reaches the chunk's first frame, so from
     that frame on the chunk's accumulation window holds what the sequential one holds;
  3. `run_chunk`: warm-up frames go through `integrate_many` (batched K1, no samples), the chunk's own frames through
     `integrate` + `generate_bev` at the sample jobs; the stored points of a frame depend only on its own observation and
     on the transforms applied since, so the BEVs are bit-identical to the sequential run's (tested);
  4. every rank keeps / writes the BEVs it made (`AsyncBevWriter`, one file per sample as the reference) or streams them
     to rank 0 in asynchronous batches (`shard.gather_to_rank0`, RCCL over xGMI).
"""
from dataclasses import dataclass, field

import numpy as np

from . import host_logic as hl
from . import shard


def replay(Ts, accum_horizon=200., bev_horizon=80., min_spacing=1.):
    """Host replay of what the sequential run does with the poses: the accumulator's pose track (re-expression by every
    T_new_prev, segment distances, horizon eviction: sem_pc_accum.py:156-228) and the driver's three sample conditions
    (run_kitti360_bev_gen.py:218-240), with the very same numpy expressions -- so its decisions are the sequential run's,
    ties included.  Ts: (F,4,4) T_new_prev per frame.
    Returns (oldest, samples): oldest[f] = number of the oldest frame still stored after frame f was integrated;
    samples = [(frame, present frame number)] -- frame NUMBERS, not window indices: they mean the same in any chunk."""
    Ts = np.asarray(Ts, dtype=np.float64).reshape(-1, 4, 4)
    track = hl.PoseTrack()
    oldest, evicted, previous_idx, samples = [], 0, 0, []
    for f in range(Ts.shape[0]):
        removed, _ = track.step(Ts[f], accum_horizon)
        evicted += removed
        oldest.append(evicted)
        previous_idx -= removed
        present_idx = track.trigger(bev_horizon, previous_idx, min_spacing)
        if present_idx is None:
            continue
        previous_idx = present_idx
        samples.append((f, evicted + present_idx))
    return oldest, samples


@dataclass
class ChunkJob:
    seq: int
    warm_start: int     # first frame integrated; frames before `start` emit no samples
    start: int
    end: int
    samples: list = field(default_factory=list)      # (frame, present frame number) with start <= frame < end

    @property
    def cost(self):
        return self.end - self.warm_start


def plan(seq_Ts, n_ranks, accum_horizon=200., bev_horizon=80., min_spacing=1., max_imbalance=1.05):
    """seq_Ts: per sequence the (F,4,4) T_new_prev.  Returns (jobs per rank, load per rank [frames incl. warm-up],
    sample jobs per sequence).  A chunk's warm-up starts one frame before the oldest frame the sequential run still
    stores when it integrates the chunk's first frame: from `start` on the chunk then holds every frame the sequential
    run holds (a frame at exactly one horizon's distance may be kept by one and dropped by the other -- it lies 120 m
    or more outside any BEV view)."""
    lengths = [int(np.asarray(T).reshape(-1, 16).shape[0]) for T in seq_Ts]
    rep = [replay(T, accum_horizon, bev_horizon, min_spacing) for T in seq_Ts]
    def job_of(seq, start, end):
        oldest, samples = rep[seq]
        warm = 0 if start == 0 else max(0, oldest[start] - 1)
        return ChunkJob(seq, warm, start, end, [(f, p) for f, p in samples if start <= f < end])

    best = None
    # (a) LPT over equal pieces of every sequence, at a few granularities: few large chunks warm up rarely but balance badly
    #     over the ranks (nine sequences of 730 .. 14 384 frames), many small ones balance well and pay a horizon of warm-up each
    for pieces in (1, 2, 3, 4, 6, 8):
        per_rank, _ = shard.plan_chunks(lengths, n_ranks, warmup_frames=0, max_imbalance=max_imbalance, pieces_per_rank=pieces)
        jobs = [job_of(c.seq, c.start, c.end) for r in per_rank for c in r]
        items, loads = shard.lpt_assign([j.cost for j in jobs], n_ranks)   # balance with the real warm-up cost
        if best is None or max(loads) < max(best[2]):
            best = (jobs, items, loads)
        if n_ranks == 1:
            break
    # (b) contiguous lanes: the sequences laid end to end (longest first) and cut into n_ranks stretches of equal COST, a
    #     cut inside a sequence costing the next rank that sequence's warm-up: every rank warms up at most once more than it
    #     has sequence starts, and the cost per rank is found by bisection
    order = sorted(range(len(lengths)), key=lambda q: -lengths[q])

    def lanes(cap):
        jobs, items, loads, rank = [], [[] for _ in range(n_ranks)], [0.0] * n_ranks, 0
        for q in order:
            pos = 0
            while pos < lengths[q]:
                if rank >= n_ranks:
                    return None
                warm = 0 if pos == 0 else pos - job_of(q, pos, pos + 1).warm_start
                room = cap - loads[rank] - warm
                if room < 1 or (room < 32 and lengths[q] - pos > room and rank + 1 < n_ranks):
                    rank += 1                               # not worth a warm-up for a few frames: the next rank starts here
                    continue
                take = int(min(lengths[q] - pos, room))
                j = job_of(q, pos, pos + take)
                items[rank].append(len(jobs))
                jobs.append(j)
                loads[rank] += j.cost
                pos += take
        return jobs, items, loads
    if n_ranks > 1 and sum(lengths) > 0:
        lo, hi = float(sum(lengths)) / n_ranks, float(sum(lengths)) + 1.0
        found = None
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            got = lanes(mid)
            if got is None:
                lo = mid
            else:
                hi, found = mid, got
        if found is not None and max(found[2]) < max(best[2]):
            best = found
    jobs, items, loads = best
    out = [sorted((jobs[i] for i in it), key=lambda j: (j.seq, j.start)) for it in items]
    return out, loads, [r[1] for r in rep]


def run_chunk(acc, get_obs, job, on_sample, warm_batch=64):
    """Runs one chunk on a FRESH accumulator.  get_obs(frame) -> the observations list integrate() takes;
    on_sample(frame, present_idx) is called right after `frame` was integrated, present_idx being the index of the
    job's present frame in the accumulator's live window.  Returns the number of samples taken."""
    frames = list(range(job.warm_start, job.start))
    for b0 in range(0, len(frames), warm_batch):
        acc.integrate_many([get_obs(f) for f in frames[b0:b0 + warm_batch]])
    todo = dict(job.samples)
    taken = 0
    for f in range(job.start, job.end):
        acc.integrate(get_obs(f))
        if f in todo:
            oldest_here = f + 1 - len(acc.poses)          # number of the oldest frame this accumulator holds
            on_sample(f, todo[f] - oldest_here)
            taken += 1
    acc.store.check_status()          # end of the chunk: one synchronisation, every kernel of the chunk has been heard
    return taken


def run_on_lanes(jobs, run_job, n_lanes=1, device=None):
    """run_job(job, lane_index) for every job of this rank.  n_lanes > 1: the jobs are dealt to that many LANES (longest
    first onto the least loaded lane) -- a lane is a host thread with its own pca_ctx and stream (pca_amd._lib.Lane), so that
    the chunks of two lanes run on the GPU at the same time and one's kernels fill the CUs the other's tails leave idle.
    Chunks are independent (fresh accumulator each), so which lane runs which chunk changes no result.  Returns per lane the
    list of run_job's return values, in the order that lane ran its jobs."""
    jobs = list(jobs)
    if n_lanes <= 1 or len(jobs) <= 1:
        return [[run_job(j, 0) for j in jobs]]
    import threading

    from . import _lib
    n_lanes = min(n_lanes, len(jobs))
    items, _ = shard.lpt_assign([j.cost for j in jobs], n_lanes)
    lanes = [_lib.Lane(device) for _ in range(n_lanes)]
    out, errors = [[] for _ in range(n_lanes)], []

    def work(k):
        try:
            with lanes[k]:
                for i in sorted(items[k], key=lambda i: (jobs[i].seq, jobs[i].start)):
                    out[k].append(run_job(jobs[i], k))
                lanes[k].synchronize()
        except BaseException as e:                             # noqa: BLE001  (re-raised on the caller's thread)
            errors.append(e)
    threads = [threading.Thread(target=work, args=(k, )) for k in range(n_lanes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out


def run_rank(rank_jobs, make_accumulator, get_obs_for, on_bev, warm_batch=64, bev_num=1, lanes=1):
    """All chunks of one rank.  make_accumulator(seq) -> fresh accumulator; get_obs_for(seq) -> get_obs callable;
    on_bev(seq, frame, bevs) receives generate_bev's list (lanes > 1: from that many threads -- see run_on_lanes).
    Returns the number of samples."""
    def run_job(job, lane):
        acc = make_accumulator(job.seq)

        def on_sample(f, present_idx):
            on_bev(job.seq, f, acc.generate_bev(present_idx, bev_num, gen_future=True))
        return run_chunk(acc, get_obs_for(job.seq), job, on_sample, warm_batch)
    return sum(sum(per_lane) for per_lane in run_on_lanes(rank_jobs, run_job, lanes))
