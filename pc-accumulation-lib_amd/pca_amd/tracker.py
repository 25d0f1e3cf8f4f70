"""Host bookkeeping of the NuScenes "fake detector + tracker" (ground-truth boxes).

O(#instances) per frame, stays on the host.  It decides WHICH (frame, instance index) pairs must have
their points flagged dynamic; the flagging itself is the K3 device kernel (pca_mark_dynamic).

Reference behaviour restated (nuscenes_oracle_sem_pc_accum.py):
  :191-250  per-frame tracking, dynamic decision (first vs last xy displacement > 1 m), retroactive marks
  :272-340  get_split_dyn_obj_trajs / get_dyn_obj_trajs
  :342-414  interval search + splitting into runs of consecutive time steps
"""
import numpy as np


class InstanceTracker:

    def __init__(self, track_classes=(0, 1, 2, 3, 5), trans_thresh=1.0):
        self.track_classes = list(track_classes)      # 'trailer' (4), bicycle, pedestrian are not tracked
        self.trans_thresh = trans_thresh
        self.observations = {}     # token -> [(world pose (3,), ts), ...] in arrival order
        self.dynamic = []          # tokens found to move, in detection order
        self.index_at = []         # per ts: {token: instance index used in that frame's inst column}

    def observe(self, ts, tokens, classes, world_centers):
        """Registers one frame's detections.  Returns the list of (frame_ts, inst_idx) pairs whose points
        must be flagged dynamic now (the newest frame for known movers; every frame that saw the object
        for a newly detected mover)."""
        marks = []
        seen = {}
        self.index_at.append(seen)
        for k, token in enumerate(tokens):
            if classes[k] not in self.track_classes:
                continue
            self.observations.setdefault(token, []).append((world_centers[k], ts))
            seen[token] = k
            if token in self.dynamic:
                marks.append((ts, k))
                continue
            hist = self.observations[token]
            if len(hist) < 2:
                continue
            moved = np.linalg.norm(hist[-1][0][:2] - hist[0][0][:2])
            if moved > self.trans_thresh:
                self.dynamic.append(token)
                for past_ts, table in enumerate(self.index_at):
                    if token in table:
                        marks.append((past_ts, table[token]))
        return marks

    # ---- trajectories of moving objects ---------------------------------------------------
    @staticmethod
    def _first_ge(values, target):
        for k, v in enumerate(values):
            if v >= target:
                return k
        raise ValueError(f'Value {target} not in array {values}')

    @staticmethod
    def _last_le(values, target):
        if values[0] > target:
            raise ValueError(f'Value {target} not in array {values}')
        for k in range(len(values) - 1):
            if values[k + 1] > target:
                return k
        return len(values) - 1

    @staticmethod
    def _consecutive_runs(tss):
        """Positions 0..len-1 grouped into runs whose time steps increase by exactly one.  A leading
        empty run is kept when the very first step breaks the chain -- it cannot here, t_prev = t0-1."""
        runs = [[]]
        prev = tss[0] - 1
        for pos, t in enumerate(tss):
            if t - prev != 1:
                runs.append([])
            runs[-1].append(pos)
            prev = t
        return runs

    def trajectories(self, ts_start=0, ts_end=None):
        """List of pose sequences (lists of [x,y,z]) of dynamic objects inside [ts_start, ts_end]."""
        out = []
        for token, hist in self.observations.items():
            if token not in self.dynamic:
                continue
            poses, tss = zip(*hist)
            try:
                lo = self._first_ge(tss, ts_start)
                hi = None if ts_end is None else self._last_le(tss, ts_end) + 1
            except ValueError:
                continue
            poses, tss = poses[lo:hi], tss[lo:hi]
            for run in self._consecutive_runs(tss):
                if len(run) < 2:
                    continue
                out.append([poses[pos].tolist() for pos in run])
        return out

    def split_trajectories(self, split_ts):
        return self.trajectories(ts_end=split_ts), self.trajectories(ts_start=split_ts), self.trajectories()
