"""Level 1 on the headline: lifetime of a workgroup against its position in the window (PCA_BEV_DBG=32) -- what the chunks
whose frames lie outside the view cost."""
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
for rep in range(4):
    builtins.print = lambda *a, **k: None
    st.step(o)
    builtins.print = rp
    torch.cuda.synchronize()
    lib.pca_debug_bev_stamps(buf)
    a = np.array(buf[:]).reshape(1024, 8).astype(np.int64)
    live = a[:, 0] > 0
    idx = np.nonzero(live)[0]
    h = a[live]; idx = idx[h[:, 5] > 0]; h = h[h[:, 5] > 0]
    life = (h[:, 5] - h[:, 0]) / 100.0
    passA = (h[:, 2] - h[:, 0]) / 100.0
    print('call: n_pend %d write_back %d blocks %d total lifetime %.0f us (= %.1f us on 256 CUs)' % (h[-1, 6] // 2, h[-1, 6] % 2, len(h), life.sum(), life.sum() / 256))
    n = len(h)
    for d in range(10):
        s = slice(d * n // 10, (d + 1) * n // 10)
        print('   blocks %3d..%3d  lifetime %5.1f us  (pass A + barrier %5.1f)' % (idx[s][0], idx[s][-1], life[s].mean(), passA[s].mean()))
