"""cProfile of the unchanged driver's step (host arrays in, LazyBev out, the sample touched one step later): where the
interpreter's time goes.  usage: pcie_cprofile.py [steps=300]"""
import builtins, cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rp, builtins.print = builtins.print, (lambda *a, **k: None)
acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
st = bench.Stepper(acc, pool)
st.fill()
host_pool = [(f[0].cpu().numpy(), f[1].cpu().numpy(), f[2].cpu().numpy()) for f in pool]
cur = {'k': 0}


class HostSemSeg:
    def pred(self, rgb):
        return host_pool[cur['k'] % len(pool)][2][None, None]


acc.semseg_model = HostSemSeg()


def run(n):
    parked = None
    for k in range(n):
        rgb_h, pc_h, _ = host_pool[k % len(host_pool)]
        cur['k'] = k
        acc.integrate([(rgb_h, pc_h, None)])
        bev = acc.generate_bev(bench.present_index(acc), 1, gen_future=True)[0]
        bev, parked = parked, bev
        if bev is not None:
            assert bev['rgb_full'].shape == (3, bench.PX, bench.PX)
    torch.cuda.synchronize()


run(30)
pr = cProfile.Profile()
pr.enable()
run(steps)
pr.disable()
builtins.print = rp
ps = pstats.Stats(pr)
ps.sort_stats('tottime')
print('steps', steps)
ps.print_stats(28)
