"""The unchanged driver's step (host arrays in, host dict out) on its own: bench.py's pcie_inclusive block.
usage: pcie_step.py [steps=30]        env: PCA_STAGING_THREADS, PCA_STAGING_NT, ..."""
import builtins
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rp, builtins.print = builtins.print, (lambda *a, **k: None)
acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
st = bench.Stepper(acc, pool)
st.fill()
for _ in range(5):
    st.step()
out = bench.pcie_inclusive_pass(acc, pool, steps)
builtins.print = rp
print(json.dumps({k: (v['ms_per_step'], v['ms_per_step_repeats']) for k, v in out.items() if isinstance(v, dict)}))
