"""Per-workgroup phase stamps of level 1 on BASELINE config 4 (1e8 points, 512^2; PCA_BEV_DBG=32): the last call's phases."""
import sys, os, ctypes as C
os.environ['PCA_BEV_DBG'] = os.environ.get('DBG', '32')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import numpy as np, builtins, bench, torch
rp = builtins.print
builtins.print = lambda *a, **k: None
out = bench.config4_pass()
builtins.print = rp
from pca_amd import _lib
lib = _lib.Context.get().lib
buf = (C.c_ulonglong * 8192)()
torch.cuda.synchronize()
lib.pca_debug_bev_stamps(buf)
a = np.array(buf[:]).reshape(1024, 8).astype(np.int64)
h = a[a[:, 0] > 0][:512]
t0 = h[:, 0].min()
ph = np.diff(h[:, :6], axis=1) / 100.0
names = ['setup+passA(reg)', 'passA(mem)+barrier', 'scan', 'passB(reg)', 'passB(mem)']
print('kernels', {k: round(v, 1) for k, v in out['kernels_avg_us'].items()})
print('last call: n_pend %d write_back %d | blocks %d span %.1f us | lifetime mean %.1f' % (h[0, 6] // 2, h[0, 6] % 2, len(h), (h[:, 5].max() - t0) / 100.0, ((h[:, 5] - h[:, 0]) / 100.0).mean()))
for k, n in enumerate(names):
    print('  %-22s mean %8.1f us  min %8.1f  max %8.1f' % (n, ph[:, k].mean(), ph[:, k].min(), ph[:, k].max()))
print('start pct', np.round(np.percentile((h[:, 0] - t0) / 100.0, [0, 25, 50, 75, 100]), 1).tolist())
