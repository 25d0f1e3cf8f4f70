"""Host cost of the benchmark step: enqueue time (no sync) vs wall time per step, pipelined or not, plus a cProfile."""
import cProfile
import io
import pstats
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import torch  # noqa: E402

acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
st = bench.Stepper(acc, pool)
st.fill()
out = torch.empty((200, 21, bench.PX, bench.PX), dtype=torch.float16, device='cuda')
for rep in range(3):
    for k in range(20):
        st.step(out[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(200):
        st.step(out[k])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('enqueue %.1f us/step, wall %.1f us/step' % (1e6 * (t1 - t0) / 200, 1e6 * (t2 - t0) / 200), flush=True)
pr = cProfile.Profile()
pr.enable()
for k in range(200):
    st.step(out[k])
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue())
