"""cProfile of the headline step (device-resident inputs, generate_bev_device): where the interpreter's time goes, and the
host's time per step without waiting for the GPU.  usage: headline_cprofile.py [steps=400]"""
import builtins, cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rp, builtins.print = builtins.print, (lambda *a, **k: None)
acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
st = bench.Stepper(acc, pool)
st.fill()
out = torch.empty((21, bench.PX, bench.PX), dtype=torch.float16, device='cuda')
for _ in range(30):
    st.step(out)
torch.cuda.synchronize()
# host time per step: short bursts that the GPU's queue absorbs
t = []
for rep in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        st.step(out)
    t.append((time.perf_counter() - t0) / 8)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    st.step(out)
pr.disable()
torch.cuda.synchronize()
builtins.print = rp
print('host us per step (bursts of 8 after a sync):', [round(1e6 * x, 1) for x in t])
ps = pstats.Stats(pr)
ps.sort_stats('tottime')
print('steps', steps)
ps.print_stats(22)
