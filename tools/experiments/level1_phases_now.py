import sys, os, ctypes as C
os.environ['PCA_BEV_DBG'] = '32'
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/pc-accumulation-lib_amd')
import numpy as np, builtins, bench, torch
rp = builtins.print
builtins.print = lambda *a, **k: None
acc, pool, _ = bench.make_accumulator(bench.synth_frame, 0)
st = bench.Stepper(acc, pool)
st.fill()
o = torch.empty((21, bench.PX, bench.PX), dtype=torch.float16, device='cuda')
for _ in range(40):
    st.step(o)
from pca_amd import _lib
lib = _lib.Context.get().lib
buf = (C.c_ulonglong * 8192)()
builtins.print = rp
for rep in range(5):
    # clear stamps by reading then stepping
    builtins.print = lambda *a, **k: None
    st.step(o)
    builtins.print = rp
    torch.cuda.synchronize()
    lib.pca_debug_bev_stamps(buf)
    a = np.array(buf[:]).reshape(1024, 8).astype(np.int64)
    tag = a[30, 6]
    live = (a[:, 0] > 0) & (a[:, 5] >= a[:, 0]) & (a[:, 6] == tag)
    h = a[live]
    t0 = h[:, 0].min()
    fresh = h[(h[:, 0] - t0) < 20000]          # this call's workgroups (stamps within 200 us of the first)
    ph = np.diff(fresh[:, :6], axis=1) / 100.0
    names = ['setup+passA(reg)', 'passA(mem)+barrier', 'scan', 'passB(reg)', 'passB(mem)']
    print('n_pend %d wb %d | workgroups %d | span %.1f us | start pct %s' % (tag // 2, tag % 2, len(fresh), (fresh[:, 5].max() - t0) / 100.0,
          np.round(np.percentile((fresh[:, 0] - t0) / 100.0, [0, 50, 90, 100]), 1).tolist())
          + ' | ' + ' '.join('%s %.1f' % (n, ph[:, k].mean()) for k, n in enumerate(names)))
