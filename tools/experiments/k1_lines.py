"""Host-side count for the K1 floor argument: how many distinct 128-byte lines (and 32-byte sectors) of the class map and of the
colour image do the gathers of a frame touch?  Same frames as bench.py's k1_batched_distinct (synth_frame(7, k)) / k1_batched_ring
(ring_frame(7, k)); numpy f64 projection with the reference's expressions (sem_pc_accum.py:367-402).
usage: k1_lines.py [uniform|ring] [frames=64]   -> one JSON line"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'uniform'
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
fn = bench.ring_frame if kind == 'ring' else bench.synth_frame
H, W = bench.IMG_H, bench.IMG_W
tot = dict(points=0, in_frustum=0, kept=0, sem_lines=0, rgb_lines=0, sem_sectors=0, rgb_sectors=0,
           wave_sem_lines=0, wave_rgb_lines=0, waves=0)
filt = np.zeros(256, bool)
filt[bench.FILTERS] = True
for k in range(n_frames):
    pc, img, sem = fn(7, k)
    h = np.concatenate([pc[:, :3].astype(np.float64), np.ones((len(pc), 1))], 1)
    f = h @ bench.P_VELO_FRAME.T
    d = f[:, 2].copy()
    d[d == 0] = -1e-6
    with np.errstate(all='ignore'):
        u = np.round(f[:, 0] / np.abs(d))
        v = np.round(f[:, 1] / np.abs(d))
    ok = (u >= 0) & (u < W) & (v >= 0) & (v < H) & (d > 0) & np.isfinite(d)
    pix = (v[ok] * W + u[ok]).astype(np.int64)
    cls = sem.ravel()[pix]
    kept = ~filt[cls]
    sem_addr = pix                                   # 1 byte per pixel
    rgb_lo, rgb_hi = 3 * pix[kept], 3 * pix[kept] + 3  # the unaligned dword [3 pix, 3 pix + 4)
    tot['points'] += len(pc)
    tot['in_frustum'] += int(ok.sum())
    tot['kept'] += int(kept.sum())
    tot['sem_lines'] += len(np.unique(sem_addr // 128))
    tot['sem_sectors'] += len(np.unique(sem_addr // 32))
    tot['rgb_lines'] += len(np.unique(np.concatenate([rgb_lo // 128, rgb_hi // 128])))
    tot['rgb_sectors'] += len(np.unique(np.concatenate([rgb_lo // 32, rgb_hi // 32])))
    # per WAVE instruction (64 consecutive candidates / kept points in point order: what one gather instruction asks its L1 for)
    for w0 in range(0, len(pix), 64):
        tot['wave_sem_lines'] += len(np.unique(sem_addr[w0:w0 + 64] // 128))
        tot['waves'] += 1
    kp = pix[kept]
    for w0 in range(0, len(kp), 64):
        a = 3 * kp[w0:w0 + 64]
        tot['wave_rgb_lines'] += len(np.unique(np.concatenate([a // 128, (a + 3) // 128])))
out = dict(kind=kind, frames=n_frames, **tot,
           image_bytes_per_frame=H * W * 4,
           unique_line_MB=(tot['sem_lines'] + tot['rgb_lines']) * 128 / 1e6,
           unique_sector_MB=(tot['sem_sectors'] + tot['rgb_sectors']) * 32 / 1e6,
           per_wave_line_requests=tot['wave_sem_lines'] + tot['wave_rgb_lines'],
           per_wave_line_MB=(tot['wave_sem_lines'] + tot['wave_rgb_lines']) * 128 / 1e6,
           useful_gather_MB=(tot['in_frustum'] * 1 + tot['kept'] * 4) / 1e6,
           note='unique_*: distinct lines / sectors per FRAME (what an infinite cache in front of HBM would fetch); per_wave_*: distinct '
                'lines per 64-lane gather instruction, summed (what the L1s ask the L2 for when nothing is reused between instructions)')
print(json.dumps(out))
