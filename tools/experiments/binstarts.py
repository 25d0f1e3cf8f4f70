"""Start / end time of every workgroup of bev_tile_bin against its number and its CU (PCA_BEV_DBG=32): is the
first round's staggered start a property of the dispatch (linear in the workgroup number) or of the CUs?"""
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
    h = a[:512]
    t0 = h[:, 0].min()
    s = (h[:, 0] - t0) / 100.0
    e = (h[:, 5] - t0) / 100.0
    smid = h[:, 7]
    print('call n_pend %d wb %d span %.1f' % (h[0, 6] // 2, h[0, 6] % 2, e.max()))
    print(' starts of workgroups 0..511 step 16:', np.round(s[::16], 1).tolist())
    print(' ends   of workgroups 0..511 step 16:', np.round(e[::16], 1).tolist())
    print(' distinct smid %d; smid of workgroups 0..31: %s' % (len(set(smid.tolist())), [hex(int(x)) for x in smid[:32]]))
    # per CU: the workgroups it ran, in order of start
    by = {}
    for b in range(512):
        by.setdefault(int(smid[b]), []).append((float(s[b]), float(e[b]), b))
    n_per = np.bincount([len(v) for v in by.values()])
    print(' workgroups per CU histogram:', n_per.tolist())
    gaps = []
    for v in by.values():
        v.sort()
        for i in range(1, len(v)):
            gaps.append(v[i][0] - v[i - 1][1])
    if gaps:
        print(' gap between a CU\'s workgroups: mean %.2f max %.2f us' % (np.mean(gaps), np.max(gaps)))
    first = sorted(v[0][0] for v in by.values())
    print(' first start per CU: percentiles 0/25/50/75/100', np.round(np.percentile(first, [0, 25, 50, 75, 100]), 1).tolist())
