"""cProfile of the bench step loop (host side): where the Python time per step goes."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, '.'); sys.path.insert(0, 'pc-accumulation-lib_amd')
import builtins, torch, bench
rp = builtins.print
builtins.print = lambda *a, **k: None
acc, pool, _ = bench.make_accumulator(bench.synth_frame, 0)
n = [0]
out = torch.empty((21, bench.PX, bench.PX), dtype=torch.float16, device='cuda')
def step():
    rgb, pc, _ = pool[n[0] % bench.POOL]; n[0] += 1
    acc.integrate([(rgb, pc, None)])
    idx = bench.present_index(acc)
    if idx is None: return
    pcs, trajs = acc._window_inputs(idx, True)
    acc.sem_bev_generator.generate(pcs, trajs, device_only=True, out=out)
while bench.present_index(acc) is None or len(acc.poses) < 195: step()
for _ in range(20): step()
torch.cuda.synchronize()
# host-only cost: time the loop without waiting for the GPU (queue depth permitting)
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
builtins.print = rp
print('loop issue time per step %.1f us; incl. final sync %.1f us' % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35); print(s.getvalue()[:6000])
