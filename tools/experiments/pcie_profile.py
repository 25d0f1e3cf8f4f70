"""cProfile of the PCIe-inclusive step (host numpy inputs, LazyBev outputs touched one step later)."""
import cProfile
import io
import pstats
import sys
import time

import os
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


def loop(n):
    parked = None
    for k in range(n):
        rgb_h, pc_h, _ = host_pool[k % len(host_pool)]
        cur['k'] = k
        acc.integrate([(rgb_h, pc_h, None)])
        bev = acc.generate_bev(bench.present_index(acc), 1, gen_future=True)[0]
        bev, parked = parked, bev
        if bev is not None:
            assert bev['rgb_full'].shape == (3, bench.PX, bench.PX)
    if parked is not None:
        assert parked['rgb_full'].shape == (3, bench.PX, bench.PX)
    torch.cuda.synchronize()


loop(30)
t0 = time.perf_counter()
loop(100)
dt = time.perf_counter() - t0
sys.stdout.write('deferred: %.1f us/step\n' % (1e6 * dt / 100))
pr = cProfile.Profile()
pr.enable()
loop(100)
pr.disable()
builtins.print = real_print
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
sys.stdout.write(s.getvalue())
