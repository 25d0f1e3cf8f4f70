"""bev_tile_bin with its arguments in the kernel-argument segment (store.bev) against the same raster with its arguments in
constant memory (store.bev_many of one job), no owed transforms: HIP-event time of the whole unit."""
import os, sys, time, builtins
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, torch
from pca_amd import _lib
rp = builtins.print
builtins.print = lambda *a, **k: None
acc, pool, _ = bench.make_accumulator(bench.synth_frame, 0)
st = bench.Stepper(acc, pool)
st.fill()
for _ in range(4):
    st.step()
builtins.print = rp
store = acc.store
store.flush_pending()
idx = bench.present_index(acc)
pcs, trajs = acc._window_inputs(idx, True)
w = pcs['pc_present'].window
gen = acc.sem_bev_generator
import numpy as np
prm = gen._raster_params(w.origin, np.eye(3), 0., 0., 80., False)
out = torch.empty((1, 21, 256, 256), dtype=torch.float16, device='cuda')
ctx = _lib.Context.get()
for name, fn in (('kernarg', lambda: store.bev(w.split, prm, out16=out[0])), ('constant', lambda: store.bev_many([(w.split, prm, 0, None)], out))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ctx.profile(2)
    for _ in range(50):
        fn()
    u = ctx.profile_read()['bev_unit']
    ctx.profile(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    print(name, 'unit us %.1f' % (1e3 * u[0] / u[1]), 'wall us %.1f' % (1e6 * (time.perf_counter() - t0) / 50))
