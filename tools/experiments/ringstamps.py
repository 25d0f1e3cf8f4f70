"""Per-workgroup begin/end stamps of bev_tile_cells_heavy on the ring model (PCA_BEV_DBG=8)."""
import sys, os, ctypes as C
os.environ['PCA_BEV_DBG'] = os.environ.get('PCA_BEV_DBG', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import numpy as np, builtins, bench
rp = builtins.print
builtins.print = lambda *a, **k: None
if os.environ.get('SCENE') == 'uniform':
    import torch
    acc, pool, _ = bench.make_accumulator(bench.synth_frame, 0)
    st = bench.Stepper(acc, pool)
    st.fill()
    o = torch.empty((21, bench.PX, bench.PX), dtype=torch.float16, device='cuda')
    for _ in range(5):
        st.step(o)
else:
    r = bench.ring_model_pass(5)
builtins.print = rp
from pca_amd import _lib
lib = _lib.Context.get().lib
buf = (C.c_ulonglong * 8192)()
import torch
torch.cuda.synchronize()
print('rc', lib.pca_debug_bev_stamps(buf))
a = np.array(buf[:]).reshape(1024, 8)
h = a[a[:, 2] > 0]
light = os.environ['PCA_BEV_DBG'] == '16'
t0 = h[:, 0].min()
dur = (h[:, 1] - h[:, 0]) / 100.0
print('heavy tiles', len(h), 'span us', (h[:, 1].max() - t0) / 100.0)
print('dur us: min %.1f med %.1f max %.1f' % (dur.min(), np.median(dur), dur.max()))
print('start offsets us pct', np.percentile((h[:, 0] - t0) / 100.0, [0, 25, 50, 75, 100]))
o = np.argsort(-dur)[:10]
for i in o:
    t = h[i]
    if light:
        print('records', t[2], 'big cells', t[7] >> 32, 'smid', t[7] & 0xffffffff, 'total', dur[i], 'pass1', (t[3] - t[0]) / 100., 'sort', (t[4] - t[3]) / 100.,
              'small', (t[5] - t[4]) / 100., 'wavehist', (t[6] - t[5]) / 100., 'final', (t[1] - t[6]) / 100.)
        continue
    print('records', t[2], 'total', dur[i], 'start', (t[0] - t0) / 100.0, 'pass1', (t[3] - t[0]) / 100., 'med1', (t[4] - t[3]) / 100.,
          'pass2', (t[5] - t[4]) / 100., 'med2', (t[6] - t[5]) / 100., 'final', (t[1] - t[6]) / 100., 'smid', t[7])
print('distinct smid', len(np.unique(h[:, 7] & 0xffffffff)))
