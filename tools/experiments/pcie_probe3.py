import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import builtins
import numpy as np, torch
import bench
acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
real = builtins.print
builtins.print = lambda *a, **k: None
st = bench.Stepper(acc, pool); st.fill()
host_pool = [(f[0].cpu().numpy(), f[1].cpu().numpy(), f[2].cpu().numpy()) for f in pool]
cur = [0]
class S:
    def pred(self, rgb): return host_pool[cur[0] % 8][2][None, None]
acc.semseg_model = S()
def run(n, defer):
    prev = None
    tn = ti = tb = tm = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        cur[0] = k
        r, p, _ = host_pool[k % 8]
        b = time.perf_counter()
        acc.integrate([(r, p, None)]); c = time.perf_counter()
        bev = acc.generate_bev(bench.present_index(acc), 1, gen_future=True)[0]; d = time.perf_counter()
        if defer:
            if prev is not None: prev['rgb_full']
            prev = bev
        else:
            bev['rgb_full']
        e = time.perf_counter()
        ti += c - b; tb += d - c; tm += e - d
    if prev is not None: prev['rgb_full']
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    real('defer=%s: %.3f ms/step  (integrate %.3f, generate_bev %.3f, materialise %.3f)' % (defer, 1e3 * dt / n, 1e3 * ti / n, 1e3 * tb / n, 1e3 * tm / n))
run(40, False); run(40, False); run(40, True); run(40, True)
