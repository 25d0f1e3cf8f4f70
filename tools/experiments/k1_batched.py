"""Times only the batched K1 launch (B KITTI-shaped frames per launch): HIP events inside the library.
usage: k1_batched.py [pool=8] [B=64] [reps=20]      env: PCA_K1_CFG=BLKxPPT, PCA_K1_QUEUES=n"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import torch
import bench
from pca_amd import _lib
from pca_amd.device_store import DeviceStore
POOL = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
pool = []
for k in range(POOL):
    pc, img, sem = bench.synth_frame(0, k)
    pool.append((torch.from_numpy(img).cuda(), torch.from_numpy(pc).cuda(), torch.from_numpy(sem).cuda()))
tmp = DeviceStore(capacity=B * bench.N_PTS, max_frames=B + 1)
frames = [dict(pts=pool[k % POOL][1], rgb=pool[k % POOL][0], sem=pool[k % POOL][2]) for k in range(B)]
for _ in range(int(os.environ.get("WARM", "3"))):
    tmp.clear()
    tmp.append_kitti(frames, bench.P_VELO_FRAME, bench.IMG_H, bench.IMG_W, bench.FILTERS)
torch.cuda.synchronize()
ctx = _lib.Context.get()
ctx.profile(True)
for _ in range(REPS):
    tmp.clear()
    tmp.append_kitti(frames, bench.P_VELO_FRAME, bench.IMG_H, bench.IMG_W, bench.FILTERS)
ms, n = ctx.profile_read()['kitti_project_sample_filter']
ctx.profile(False)
tmp.check_status()
kept = int(tmp.offsets()[-1])
us = 1e3 * ms / n
alg = 16.0 * bench.N_PTS * B + 4.0 * (kept * 19.0 / 14.0 / 0.99) + 40.0 * kept
print(f'cfg={os.environ.get("PCA_K1_CFG", "default")} queues={os.environ.get("PCA_K1_QUEUES", "default")} pool={POOL} B={B} '
      f'kept={kept} us={us:.1f} GBps={alg / us / 1e3:.0f} frac={alg / us / 1e3 / 8000:.3f}')

if os.environ.get('PCA_K1_STAMPS'):
    import ctypes as C, numpy as np
    buf = np.zeros((65536, 8), np.uint64)
    n = ctx.lib.pca_debug_k1_stamps(ctx.h, buf.ctypes.data_as(C.c_void_p), 65536)
    b = buf[:n].astype(np.int64)
    t0 = b[:, 0].min()
    tile = b[:, 7] & 0xffffffff
    xcc = b[:, 7] >> 32
    ph = np.diff(b[:, :7], axis=1) / 100.0          # us (100 MHz realtime clock)
    names = ['start+frame', 'pts+phase1', 'cand compaction', 'phase2 (gathers)', 'scan (+lookback)', 'stores / list']
    print('blocks', n, 'kernel span us', (b[:, 6].max() - t0) / 100.0)
    for k, nm in enumerate(names):
        print(f'{nm:18s} mean {ph[:, k].mean():7.2f}  p50 {np.median(ph[:, k]):7.2f}  p95 {np.percentile(ph[:, k], 95):7.2f}')
    print('block lifetime mean', (b[:, 6] - b[:, 0]).mean() / 100.0)
    order = np.argsort(tile)
    st = (b[order, 0] - t0) / 100.0
    en = (b[order, 6] - t0) / 100.0
    idx = [i for i in (0, 100, 500, 1000, 2000, 3000, 5000, 7000) if i < n]
    print('start/end (us) by tile:', [(i, round(float(st[i]), 1), round(float(en[i]), 1)) for i in idx])
    print('tiles per xcc:', np.bincount(xcc, minlength=8).tolist())
    # concurrency: blocks alive at a few instants
    for tq in (10, 30, 60, 90):
        alive = int(((b[:, 0] - t0) / 100.0 <= tq).sum() - ((b[:, 6] - t0) / 100.0 <= tq).sum())
        print(f'alive at {tq} us: {alive}')
