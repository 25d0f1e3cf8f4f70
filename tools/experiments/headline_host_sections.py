"""Host time of the headline step by section (no waiting for the GPU: bursts of 6 steps after a synchronize)."""
import builtins, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import ctypes as C
rp, builtins.print = builtins.print, (lambda *a, **k: None)
acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
st = bench.Stepper(acc, pool)
st.fill()
out = torch.empty((21, bench.PX, bench.PX), dtype=torch.float16, device='cuda')
for _ in range(30):
    st.step(out)
pc = time.perf_counter
acc_t = {'integrate': 0.0, 'trigger': 0.0, 'generate_bev_device': 0.0}
# finer: wrap a few inner calls
inner = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = pc(); r = f(*a, **k); inner[key] = inner.get(key, 0.0) + pc() - t0; return r
    setattr(obj, name, g)
wrap(acc.store, 'append_kitti_obs', '  store.append_kitti_obs (C: pca_kitti_integrate inside)')
wrap(acc, '_obs_pointers', '  _obs_pointers')
wrap(acc, 'update_sem_pcs', '  update_sem_pcs')
wrap(acc.store, 'view_hint_into', '  store.view_hint_into')
wrap(acc.store, 'bev_pending', '  store.bev_pending')
wrap(acc.sem_bev_generator, '_raster_params', '  _raster_params')
lib = acc.store.ctx.lib
class L:  # time the two fat C calls
    pass
for nm in ('pca_kitti_integrate_v', 'pca_kitti_generate_bev_v'):
    f = getattr(lib, nm)
    def g(*a, f=f, nm=nm):
        t0 = pc(); r = f(*a); inner['    C ' + nm] = inner.get('    C ' + nm, 0.0) + pc() - t0; return r
    try:
        setattr(lib, nm, g)
    except Exception:
        pass
n = 0
for rep in range(40):
    torch.cuda.synchronize()
    for _ in range(6):
        t0 = pc(); st.integrate(); t1 = pc(); idx = bench.present_index(acc); t2 = pc(); acc.generate_bev_device(idx, out=out); t3 = pc()
        acc_t['integrate'] += t1 - t0; acc_t['trigger'] += t2 - t1; acc_t['generate_bev_device'] += t3 - t2
        n += 1
builtins.print = rp
tot = sum(acc_t.values())
print('host us per step: %.1f' % (1e6 * tot / n))
for k, v in acc_t.items():
    print('%-28s %6.1f us' % (k, 1e6 * v / n))
for k, v in sorted(inner.items()):
    print('%-60s %6.1f us' % (k, 1e6 * v / n))
