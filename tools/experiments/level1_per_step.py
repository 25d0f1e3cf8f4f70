"""Level 1 of the headline step, call by call: its HIP-event time against the owed chain of the call (1..4; 4 = write-back)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pc-accumulation-lib_amd'))
import numpy as np, builtins, bench, torch
from pca_amd import _lib
rp = builtins.print
builtins.print = lambda *a, **k: None
acc, pool, _ = bench.make_accumulator(bench.ring_frame if 'ring' in sys.argv else bench.synth_frame, 0)
st = bench.Stepper(acc, pool)
st.fill()
o = torch.empty((21, bench.PX, bench.PX), dtype=torch.float16, device='cuda')
for _ in range(8):
    st.step(o)
ctx = _lib.Context.get()
rows = []
for rep in range(16):
    n_owed = len(acc.store._pending) + 1
    ctx.profile(True)
    st.step(o)
    prof = ctx.profile_read()
    ctx.profile(False)
    rows.append((n_owed, {k: round(1e3 * v[0] / v[1], 1) for k, v in prof.items() if v[1]}))
builtins.print = rp
for r in rows:
    print(r)
