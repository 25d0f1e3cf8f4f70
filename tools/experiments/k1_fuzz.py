"""Long randomised parity run of K1 + K2 against the oracle (generator of
tests/test_gpu_kernels.py::test_k1_k2_randomised_batches_match_oracle, more seeds)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'pc-accumulation-lib_amd')); sys.path.insert(0, os.path.join(R, 'tests'))
import torch as T
import test_gpu_kernels as t
from oracle import oracle as orc
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    if seed % 200 == 0:
        print('seed', seed, flush=True)                    # (a long silent run looks hung to the GPU runner)
    try:
        t.test_k1_k2_randomised_batches_match_oracle(T, orc, seed)
    except AssertionError as e:
        bad += 1
        print('SEED', seed, 'FAILED', str(e)[:300])
print('seeds %d..%d: %d failures' % (lo, hi, bad))
