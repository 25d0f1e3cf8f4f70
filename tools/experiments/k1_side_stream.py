"""Probe: the headline step with integrate() (K1 of the next frame) on a low-priority side stream, so that its 118 workgroups
run in the shadow of the previous step's tile kernel instead of alone on the GPU.  ms per step and a checksum of the planes,
plain against side-stream, alternating."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import builtins, numpy as np, torch, bench
rp = builtins.print
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def run(side_mode):
    builtins.print = lambda *a, **k: None
    acc, pool, _ = bench.make_accumulator(bench.synth_frame, 0)
    st = bench.Stepper(acc, pool)
    st.fill()
    outs = [torch.empty((21, bench.PX, bench.PX), dtype=torch.float16, device='cuda') for _ in range(2)]
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
    side = torch.cuda.Stream(priority=0) if side_mode == 'side' else None
    main = torch.cuda.current_stream()
    sums = []

    def step(k):
        if side is None:
            st.integrate()
        else:
            side.wait_stream(main) if k == 0 else None
            with torch.cuda.stream(side):
                st.integrate()
            main.wait_stream(side)
        idx = bench.present_index(acc)
        return acc.generate_bev_device(idx, out=outs[k & 1])
    for k in range(10):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(STEPS):
        step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    s = float(outs[(STEPS - 1) & 1].float().sum().item())
    builtins.print = rp
    return 1e3 * dt, s


for mode in ('plain', 'side', 'plain', 'side'):
    ms, s = run(mode)
    print('%-6s %.4f ms/step  checksum %.6f' % (mode, ms, s), flush=True)
