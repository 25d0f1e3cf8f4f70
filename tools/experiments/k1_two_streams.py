"""Feasibility probe: does a batched K1 call on a second stream overlap with one on the first (front kernel of one call
beside the append kernel of the other)?  Timing only -- both calls share the context's staging workspace (same frames,
same values, so nothing invalid is ever read), the results are not checked here.
usage: k1_two_streams.py [distinct=64] [reps=20]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import torch
import bench
from pca_amd import _lib
from pca_amd.device_store import DeviceStore
ND = int(sys.argv[1]) if len(sys.argv) > 1 else 64
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = 64
pool = bench.device_pool(bench.synth_frame, 5, ND)
frames = [dict(pts=pool[k % ND][1], rgb=pool[k % ND][0], sem=pool[k % ND][2]) for k in range(B)]
descs = DeviceStore.kitti_descs(frames)
ctx = _lib.Context.get()
lib = ctx.lib
stores = [DeviceStore(capacity=B * bench.N_PTS, max_frames=B + 1) for _ in range(2)]
cst = [s.c_store() for s in stores]
Pc, fm = _lib.f64_array(bench.P_VELO_FRAME, 12), _lib.class_mask(bench.FILTERS)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def call(i, s):
    ctx.check(lib.pca_kitti_project_sample_filter(ctx.h, descs, B, Pc, bench.IMG_H, bench.IMG_W, fm, C.byref(cst[i]),
                                                  stores[i].frame_off.data_ptr(), 0, C.c_void_p(s.cuda_stream)))


for s in stores:
    s.clear()
torch.cuda.synchronize()
for _ in range(3):
    call(0, streams[0])
torch.cuda.synchronize()
for mode in ('one stream', 'two streams', 'one stream', 'two streams'):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(REPS):
        k = r & 1 if mode == 'two streams' else 0
        call(k, streams[k])
    torch.cuda.synchronize()
    print('%-12s %.1f us per call' % (mode, 1e6 * (time.perf_counter() - t0) / REPS), flush=True)
