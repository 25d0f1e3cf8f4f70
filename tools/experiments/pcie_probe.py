"""Where does the PCIe-inclusive step spend its time?  plain vs reader-thread prefetch at two GIL switch intervals."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import builtins
import numpy as np, torch
import bench
acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
builtins.print = lambda *a, **k: None
st = bench.Stepper(acc, pool); st.fill()
for interval in (0.005, 0.0002):
    sys.setswitchinterval(interval)
    r = bench.pcie_inclusive_pass(acc, pool, 30)
    sys.stdout.write(f'switchinterval {interval}: plain {r["plain"]["ms_per_step"]:.3f} ms, pipelined {r["pipelined"]["ms_per_step"]:.3f} ms\n')
# pieces of the plain step
host_pool = [(f[0].cpu().numpy(), f[1].cpu().numpy(), f[2].cpu().numpy()) for f in pool]
def t(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
pin = torch.empty(host_pool[0][1].shape, dtype=torch.float32).pin_memory()
dev = torch.empty(host_pool[0][1].shape, dtype=torch.float32, device='cuda')
sys.stdout.write('pageable pc .to(cuda): %.3f ms\n' % t(lambda: torch.from_numpy(host_pool[0][1]).cuda()))
sys.stdout.write('pageable img .to(cuda): %.3f ms\n' % t(lambda: torch.from_numpy(host_pool[0][0]).cuda()))
sys.stdout.write('numpy -> pinned memcpy (1.9 MB): %.3f ms\n' % t(lambda: pin.copy_(torch.from_numpy(host_pool[0][1]))))
sys.stdout.write('pinned -> device async (1.9 MB): %.3f ms\n' % t(lambda: dev.copy_(pin, non_blocking=True)))
p16 = torch.empty((21, 256, 256), dtype=torch.float16, device='cuda')
sys.stdout.write('planes .cpu(): %.3f ms\n' % t(lambda: p16.cpu()))
hp = torch.empty((21, 256, 256), dtype=torch.float16, pin_memory=True)
sys.stdout.write('planes -> pinned async + sync: %.3f ms\n' % t(lambda: (hp.copy_(p16, non_blocking=True), torch.cuda.synchronize())))
sys.stdout.write('pinned alloc (cached) 2.75 MB: %.3f ms\n' % t(lambda: torch.empty((21, 256, 256), dtype=torch.float16, pin_memory=True)))
