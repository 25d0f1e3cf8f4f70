"""Host time of the PCIe-inclusive step by section (perf_counter, no profiler): integrate / trigger / generate_bev / first
access of the PREVIOUS sample (deferred form), and the C calls inside them."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import torch  # noqa: E402
import builtins  # noqa: E402

acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
real_print = builtins.print
builtins.print = lambda *a, **k: None
st = bench.Stepper(acc, pool)
st.fill()
host_pool = [(f[0].cpu().numpy(), f[1].cpu().numpy(), f[2].cpu().numpy()) for f in pool]
cur = {'k': 0}


class HostSemSeg:
    def pred(self, rgb):
        return host_pool[cur['k'] % len(pool)][2][None, None]


acc.semseg_model = HostSemSeg()
spent = {}
ctx = acc.store.ctx


class LibProxy:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, k):
        f = getattr(self._lib, k)

        def w(*a):
            t = time.perf_counter()
            r = f(*a)
            d = spent.setdefault('C:' + k, [0.0, 0])
            d[0] += time.perf_counter() - t
            d[1] += 1
            return r
        return w


def loop(n, timed):
    parked = None
    sec = [0.0, 0.0, 0.0, 0.0]
    for k in range(n):
        rgb_h, pc_h, _ = host_pool[k % len(host_pool)]
        cur['k'] = k
        t0 = time.perf_counter()
        acc.integrate([(rgb_h, pc_h, None)])
        t1 = time.perf_counter()
        idx = bench.present_index(acc)
        t2 = time.perf_counter()
        bev = acc.generate_bev(idx, 1, gen_future=True)[0]
        t3 = time.perf_counter()
        bev, parked = parked, bev
        if bev is not None:
            assert bev['rgb_full'].shape == (3, bench.PX, bench.PX)
        t4 = time.perf_counter()
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            sec[i] += d
    if parked is not None:
        assert parked['rgb_full'].shape == (3, bench.PX, bench.PX)
    torch.cuda.synchronize()
    return sec


loop(30, False)
N = 200
t0 = time.perf_counter()
sec = loop(N, True)
dt = time.perf_counter() - t0
real = builtins.print
builtins.print = real_print
print('deferred: %.1f us/step   integrate %.1f  trigger %.1f  generate_bev %.1f  first access of the previous sample %.1f' %
      ((1e6 * dt / N, ) + tuple(1e6 * s / N for s in sec)))
builtins.print = lambda *a, **k: None
ctx.lib = LibProxy(ctx.lib)
t0 = time.perf_counter()
loop(N, True)
dt = time.perf_counter() - t0
builtins.print = real_print
print('with C-call timers: %.1f us/step' % (1e6 * dt / N))
for k, v in sorted(spent.items(), key=lambda kv: -kv[1][0]):
    print('  %-40s %6.1f us/step (%d calls)' % (k, 1e6 * v[0] / N, v[1]))
