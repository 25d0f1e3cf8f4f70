"""Phase stamps of the light tile kernel (bev_tile_cells) on the headline workload (PCA_BEV_DBG=16): means over all tiles."""
import sys, os, ctypes as C
os.environ['PCA_BEV_DBG'] = '16'
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
    h = a[a[:, 2] > 0]
    t0 = h[:, 0].min()
    seq = np.stack([h[:, 0], h[:, 3], h[:, 4], h[:, 5], h[:, 6], h[:, 1]], axis=1)
    ph = np.diff(seq, axis=1) / 100.0
    names = ['map+pass1', 'offsets+sort', 'small cells', 'wave hist', 'final+write']
    print('tiles %d records mean %.0f max %d | span %.1f us | lifetime mean %.1f max %.1f | ' % (len(h), h[:, 2].mean(), h[:, 2].max(), (h[:, 1].max() - t0) / 100.0,
          ((h[:, 1] - h[:, 0]) / 100.0).mean(), ((h[:, 1] - h[:, 0]) / 100.0).max())
          + ' '.join('%s %.1f' % (n, ph[:, k].mean()) for k, n in enumerate(names))
          + ' | start pct %s' % np.round(np.percentile((h[:, 0] - t0) / 100.0, [0, 50, 100]), 1).tolist()
          + ' | big cells per tile %.2f' % (h[:, 7] >> 32).mean())
    if rep == 3:
        life = (h[:, 1] - h[:, 0]) / 100.0
        print('corr(lifetime, records) %.2f' % np.corrcoef(life, h[:, 2])[0, 1])
        smid = h[:, 7] & 0xffffffff
        for i in np.argsort(-life)[:6]:
            same = smid == smid[i]
            print('  slow tile: records %d life %.1f phases %s | its CU holds %d tiles with %d records' % (h[i, 2], life[i], np.round(ph[i], 1).tolist(), same.sum(), h[same, 2].sum()))
        for i in np.argsort(life)[:3]:
            same = smid == smid[i]
            print('  fast tile: records %d life %.1f phases %s | its CU holds %d tiles with %d records' % (h[i, 2], life[i], np.round(ph[i], 1).tolist(), same.sum(), h[same, 2].sum()))
        cu = {}
        for sm, rec, lf in zip(smid, h[:, 2], life):
            cu.setdefault(sm, []).append((rec, lf))
        tot = np.array([sum(r for r, _ in v) for v in cu.values()]); mx = np.array([max(l for _, l in v) for v in cu.values()]); cnt = np.array([len(v) for v in cu.values()])
        print('CUs %d tiles per CU min %d max %d | corr(CU records, CU max life) %.2f' % (len(cu), cnt.min(), cnt.max(), np.corrcoef(tot, mx)[0, 1]))
