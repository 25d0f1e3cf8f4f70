"""Per-workgroup phase stamps of bev_tile_bin on the headline workload (PCA_BEV_DBG=32): one line per call type."""
import sys, os, ctypes as C
os.environ['PCA_BEV_DBG'] = '32'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import numpy as np, builtins, bench, torch
rp = builtins.print
builtins.print = lambda *a, **k: None
acc, pool, _ = bench.make_accumulator(bench.synth_frame, 0)
st = bench.Stepper(acc, pool)
st.fill()
o = torch.empty((21, bench.PX, bench.PX), dtype=torch.float16, device='cuda')
for _ in range(8):
    st.step(o)
from pca_amd import _lib
lib = _lib.Context.get().lib
buf = (C.c_ulonglong * 8192)()
builtins.print = rp
for rep in range(8):
    builtins.print = lambda *a, **k: None
    st.step(o)
    builtins.print = rp
    torch.cuda.synchronize()
    lib.pca_debug_bev_stamps(buf)
    a = np.array(buf[:]).reshape(1024, 8).astype(np.int64)
    h = a[a[:, 0] > 0][:512]
    t0 = h[:, 0].min()
    ph = np.diff(h[:, :6], axis=1) / 100.0
    names = ['setup+passA', 'mem+barrier', 'scan', 'passB', 'memB']
    print('call: n_pend %d write_back %d | blocks %d span %.1f us | lifetime mean %.1f | ' % (h[0, 6] // 2, h[0, 6] % 2, len(h), (h[:, 5].max() - t0) / 100.0, ((h[:, 5] - h[:, 0]) / 100.0).mean())
          + ' '.join('%s %.1f' % (n, ph[:, k].mean()) for k, n in enumerate(names))
          + ' | start pct %s' % np.round(np.percentile((h[:, 0] - t0) / 100.0, [0, 50, 100]), 1).tolist())
