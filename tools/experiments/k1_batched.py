"""Runs only the batched K1 launch (64 KITTI-shaped frames per launch) a few times -- for rocprofv3 counter passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import torch
import bench
from pca_amd.device_store import DeviceStore
pool = []
for k in range(8):
    pc, img, sem = bench.synth_frame(0, k)
    pool.append((torch.from_numpy(img).cuda(), torch.from_numpy(pc).cuda(), torch.from_numpy(sem).cuda()))
B = 64
tmp = DeviceStore(capacity=B * bench.N_PTS, max_frames=B + 1)
frames = [dict(pts=pool[k % 8][1], rgb=pool[k % 8][0], sem=pool[k % 8][2]) for k in range(B)]
for _ in range(6):
    tmp.clear()
    tmp.append_kitti(frames, bench.P_VELO_FRAME, bench.IMG_H, bench.IMG_W, bench.FILTERS)
torch.cuda.synchronize()
print('kept', int(tmp.offsets()[-1]))
if os.environ.get('PCA_K1_STAMPS'):
    import ctypes as C, numpy as np
    ctx = tmp.ctx
    buf = np.zeros((65536, 8), np.uint64)
    n = ctx.lib.pca_debug_k1_stamps(ctx.h, buf.ctypes.data_as(C.c_void_p), 65536)
    b = buf[:n].astype(np.int64)
    t0 = b[:, 0].min()
    order = np.argsort(b[:, 6])
    b = b[order]
    ph = np.diff(b[:, :6], axis=1) / 100.0          # us (100 MHz realtime clock)
    names = ['ticket', 'frame+pt loads', 'project+gather', 'scan+lookback', 'stores']
    print('blocks', n, 'kernel span us', (b[:, 5].max() - t0) / 100.0)
    for k, nm in enumerate(names):
        print(f'{nm:16s} mean {ph[:, k].mean():7.2f}  p50 {np.median(ph[:, k]):7.2f}  p95 {np.percentile(ph[:, k], 95):7.2f}')
    print('block lifetime mean', (b[:, 5] - b[:, 0]).mean() / 100.0)
    st = (b[:, 0] - t0) / 100.0
    print('start times (us) of tiles 0,100,500,1000,2000,3000:', [round(float(st[i]), 1) for i in (0, 100, 500, 1000, 2000, 3000) if i < n])
