"""bench.py's pcie_inclusive block, three times in a row."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import builtins
import bench
acc, pool, model = bench.make_accumulator(bench.synth_frame, 0)
rp = builtins.print
builtins.print = lambda *a, **k: None
st = bench.Stepper(acc, pool)
st.fill()
for _ in range(10):
    st.step()
for i in range(3):
    r = bench.pcie_inclusive_pass(acc, pool, 30)
    rp('run', i, 'plain %.3f ms  deferred %.3f ms' % (r['plain']['ms_per_step'], r['deferred']['ms_per_step']))
