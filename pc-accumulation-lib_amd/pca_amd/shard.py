"""Scene / sequence sharding across the GPUs of one node (one process per GPU) and the only collective of
the path: gathering finished BEV tensors to rank 0 (RCCL over xGMI; gloo on CPU for tests).

The reference is single-process; scenes and sequences are independent, and inside a
sequence the accumulation state can be rebuilt from any start frame after a warm-up of one memory
horizon.  So:
  * NuScenes: whole scenes (~40 frames each) are dealt out longest-first (`lpt_assign`);
  * KITTI-360: the nine sequences are too unequal for sequence granularity (best makespan 14384 of 74367
    frames = 5.2x on 8 GPUs), so long sequences are cut into contiguous chunks, each preceded by a warm-up
    prefix that is integrated but emits no samples (`plan_chunks`).  Which frames emit a BEV sample depends
    only on the poses (`sample_frames` replays the three trigger conditions of run_kitti360_bev_gen.py:
    218-240 on a pose track), so the sample list of the sharded run equals the sequential one when the
    poses are known up front (GT / oracle poses).
No collective runs during compute; `gather_to_rank0` moves [n_local, 21, px, px] float16 tensors.
"""
from dataclasses import dataclass

import numpy as np

from . import host_logic as hl


def lpt_assign(costs, n_ranks):
    """Longest-processing-time-first assignment.  Returns (items per rank, load per rank)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * n_ranks
    items = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda k: (loads[k], k))
        items[r].append(i)
        loads[r] += costs[i]
    return items, loads


@dataclass
class Chunk:
    seq: int          # sequence index
    warm_start: int   # first frame integrated (no samples emitted before `start`)
    start: int        # first frame whose samples belong to this chunk
    end: int          # one past the last frame


def plan_chunks(seq_lengths, n_ranks, warmup_frames, max_imbalance=1.05, pieces_per_rank=1):
    """Cuts sequences into chunks of about total / (n_ranks * pieces_per_rank) frames and assigns them longest first.
    Cost of a chunk = its frames + its warm-up prefix.  More pieces per rank balance better and warm up more often: the
    caller (sharded_run.plan) tries a few values and keeps the plan with the smallest busiest rank.
    Returns (chunks per rank, load per rank)."""
    total = float(sum(seq_lengths))
    target = max(total / (n_ranks * max(int(pieces_per_rank), 1)), 1.0)
    chunks = []
    for s, n in enumerate(seq_lengths):
        pieces = max(1, int(np.ceil(n / target)))
        edges = np.linspace(0, n, pieces + 1).round().astype(int)
        for a, b in zip(edges[:-1], edges[1:]):
            if b > a:
                chunks.append(Chunk(s, max(0, int(a) - warmup_frames), int(a), int(b)))
    costs = [c.end - c.warm_start for c in chunks]
    items, loads = lpt_assign(costs, n_ranks)
    return [[chunks[i] for i in it] for it in items], loads


def sample_frames(positions, accum_horizon, bev_horizon, min_spacing):
    """Replays the sample trigger of the KITTI driver on a sequence of ego positions (F,3): returns the
    list of (frame, present_idx) at which generate_bev would be called.  Uses the same PoseTrack logic
    (segment distances, lower-triangular path sums, horizon eviction) as the accumulator; distances are
    invariant under the per-frame rigid re-expression, so absolute positions can be used."""
    track = hl.PoseTrack()
    out = []
    previous_idx = 0
    for f, p in enumerate(np.asarray(positions, dtype=np.float64)):
        track.append(p)
        removed = 0
        if len(track) > 1:
            removed = track.evict_beyond(accum_horizon, track.push_segment())
        previous_idx -= removed
        if len(track) < 2:
            continue
        d = hl.incremental_path_dists(track.seg_array())
        if d[-1] < bev_horizon:
            continue
        present_idx = int(((d - bev_horizon) > 0).argmax())
        if d[-1] - d[present_idx] < bev_horizon:
            continue
        if hl.pose_dist(track.pose(previous_idx), track.pose(present_idx)) < min_spacing:
            continue
        previous_idx = present_idx
        out.append((f, present_idx))
    return out


class _PendingGather:
    """Handle of an asynchronous gather_to_rank0: wait() returns what the blocking call returns."""

    def __init__(self, work, bufs, sizes):
        self._work, self._bufs, self._sizes = work, bufs, sizes

    def wait(self):
        if self._work is not None:
            self._work.wait()
        if self._bufs is None:
            return None
        return [b[:k] for b, k in zip(self._bufs, self._sizes)]


def gather_to_rank0(local, group=None, async_op=False, sizes=None):
    """local: tensor [n_local, ...] (same trailing shape and dtype on every rank, n_local may differ).
    Returns on rank 0 the list of per-rank tensors (rank order), elsewhere None.  One size exchange plus one
    padded gather; with the nccl backend both run over RCCL / xGMI.
    async_op=True returns a handle at once (`.wait()` gives the result): the transfer of one batch of samples then
    overlaps the computation of the next (keep `local` untouched until then).  `sizes` (per-rank counts, known to
    every rank, e.g. a fixed batch size) skips the size exchange -- and with it the only synchronising step."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return _PendingGather(None, [local], [local.shape[0]]) if async_op else [local]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if sizes is None:
        n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        szs = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(szs, n, group=group)
        sizes = [int(s.item()) for s in szs]
    assert len(sizes) == world and sizes[rank] == local.shape[0]
    n_max = max(max(sizes), 1)
    padded = local
    if local.shape[0] != n_max:
        padded = torch.zeros((n_max, ) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    # `rank` is the rank inside `group`; dist.gather wants the GLOBAL rank of the destination
    dst = 0 if group is None else dist.get_global_rank(group, 0)
    work = dist.gather(padded.contiguous(), bufs, dst=dst, group=group, async_op=async_op)
    if async_op:
        return _PendingGather(work, bufs, sizes)
    if rank != 0:
        return None
    return [b[:k] for b, k in zip(bufs, sizes)]
