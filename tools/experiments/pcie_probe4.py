import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import builtins
import numpy as np, torch
import bench
from obs_dataloaders.obs_dataloader import ObservationDataloader
from pca_amd.ingest import PrefetchingLoader
acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
real = builtins.print
builtins.print = lambda *a, **k: None
st = bench.Stepper(acc, pool); st.fill()
host_pool = [(f[0].cpu().numpy(), f[1].cpu().numpy(), f[2].cpu().numpy()) for f in pool]
class M(ObservationDataloader):
    def __init__(self, n):
        super().__init__(None, 1); self.n, self.pc_paths = n, None
    def __len__(self): return self.n
    def read_obs(self, idx):
        r, p, _ = host_pool[idx % 8]; return (r, p, np.zeros((1, 1)))
class S:
    def pred(self, rgb): return pool[0][2][None, None]
acc.semseg_model = S()
# loader alone
t0 = time.perf_counter(); n = 0
for obs in PrefetchingLoader(M(60), depth=4): n += 1
torch.cuda.synchronize()
real('loader alone: %.3f ms per batch' % (1e3 * (time.perf_counter() - t0) / n))
it = iter(PrefetchingLoader(M(60), depth=4))
tn = ti = tb = tf = 0
parked = None
t00 = time.perf_counter()
for k in range(60):
    a = time.perf_counter(); obs = next(it); b = time.perf_counter()
    acc.integrate([(obs[0][0], obs[0][1], None)]); c = time.perf_counter()
    bev = acc.generate_bev(bench.present_index(acc), 1, gen_future=True)[0]; d = time.perf_counter()
    if parked is not None: parked['rgb_full']
    parked = bev
    e = time.perf_counter()
    tn += b - a; ti += c - b; tb += d - c; tf += e - d
torch.cuda.synchronize()
real('loop %.3f ms/step: next %.3f  integrate %.3f  generate_bev %.3f  fill %.3f' % (1e3 * (time.perf_counter() - t00) / 60, 1e3 * tn / 60, 1e3 * ti / 60, 1e3 * tb / 60, 1e3 * tf / 60))
it = iter(PrefetchingLoader(M(40), depth=4))
tn = ti = tb = tf = 0
for k in range(40):
    torch.cuda.synchronize(); a = time.perf_counter(); obs = next(it); torch.cuda.synchronize(); b = time.perf_counter()
    acc.integrate([(obs[0][0], obs[0][1], None)]); torch.cuda.synchronize(); c = time.perf_counter()
    bev = acc.generate_bev(bench.present_index(acc), 1, gen_future=True)[0]; torch.cuda.synchronize(); d = time.perf_counter()
    bev['rgb_full']; e = time.perf_counter()
    tn += b - a; ti += c - b; tb += d - c; tf += e - d
real('synced phases: next %.3f  integrate %.3f  generate_bev %.3f  fill %.3f ms' % (1e3 * tn / 40, 1e3 * ti / 40, 1e3 * tb / 40, 1e3 * tf / 40))
# the same with host inputs (no loader)
tn = ti = tb = tf = 0
class S2:
    def pred(self, rgb): return host_pool[0][2][None, None]
acc.semseg_model = S2()
for k in range(40):
    r, p, _ = host_pool[k % 8]
    torch.cuda.synchronize(); b = time.perf_counter()
    acc.integrate([(r, p, None)]); torch.cuda.synchronize(); c = time.perf_counter()
    bev = acc.generate_bev(bench.present_index(acc), 1, gen_future=True)[0]; torch.cuda.synchronize(); d = time.perf_counter()
    bev['rgb_full']; e = time.perf_counter()
    ti += c - b; tb += d - c; tf += e - d
real('host inputs, synced phases: integrate %.3f  generate_bev %.3f  fill %.3f ms' % (1e3 * ti / 40, 1e3 * tb / 40, 1e3 * tf / 40))
