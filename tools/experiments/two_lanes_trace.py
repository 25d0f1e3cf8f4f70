"""Two independent sequences on one GPU, a lane each (own pca_ctx + stream + host thread): a short steady-state run for a kernel
trace.  Run under `rocprofv3 --kernel-trace`; tools/experiments/two_lanes_trace.sh turns the trace into the overlap figures.
usage: two_lanes_trace.py [steps=40] [lanes=2]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import builtins  # noqa: E402
import bench  # noqa: E402
import torch  # noqa: E402
from pca_amd import _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rp, builtins.print = builtins.print, (lambda *a, **k: None)
lanes = [_lib.Lane() for _ in range(n_lanes)]
st, outs = [], []
for k, lane in enumerate(lanes):
    with lane:
        acc, pool, _ = bench.make_accumulator(bench.synth_frame, 20 + k)
        s = bench.Stepper(acc, pool)
        s.fill()
        out = torch.empty((21, bench.PX, bench.PX), dtype=torch.float16, device='cuda')
        for _ in range(5):
            s.step(out)
        lane.synchronize()
        st.append(s)
        outs.append(out)
torch.cuda.synchronize()
go = threading.Barrier(n_lanes)
t = [0.0] * n_lanes


def run(k):
    with lanes[k]:
        go.wait()
        t0 = time.perf_counter()
        for _ in range(steps):
            st[k].step(outs[k])
        lanes[k].synchronize()
        t[k] = time.perf_counter() - t0


th = [threading.Thread(target=run, args=(k, )) for k in range(n_lanes)]
for x in th:
    x.start()
for x in th:
    x.join()
builtins.print = rp
print('lanes %d steps %d: ms per step per lane %s' % (n_lanes, steps, [round(1e3 * v / steps, 4) for v in t]))
