"""Ceilings of the host -> device path on this box: pinned -> device copy rate (one stream, two streams), pageable -> pinned
staging rate of the library's pool, and both back to back / overlapped."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'pc-accumulation-lib_amd'))
from pca_amd import _lib  # noqa: E402

ctx = _lib.Context.get()
lib = ctx.lib
MB = 26
n = MB << 20
src = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
dev = torch.empty(n, dtype=torch.uint8, device='cuda')
s2 = torch.cuda.Stream()


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


dt = timeit(lambda: dev.copy_(pin, non_blocking=True))
print('pinned -> device, one stream: %.2f ms = %.1f GB/s' % (1e3 * dt, n / dt / 1e9))
half = n // 2


def two():
    dev[:half].copy_(pin[:half], non_blocking=True)
    with torch.cuda.stream(s2):
        dev[half:].copy_(pin[half:], non_blocking=True)


dt = timeit(two)
print('pinned -> device, two streams (halves): %.2f ms = %.1f GB/s' % (1e3 * dt, n / dt / 1e9))
vp = C.c_void_p * 1
t0 = time.perf_counter()
for _ in range(20):
    np.copyto(pin.numpy(), src)
dt = (time.perf_counter() - t0) / 20
print('numpy copy pageable -> pinned: %.2f ms = %.1f GB/s' % (1e3 * dt, n / dt / 1e9))
for thr in (os.environ.get('PCA_STAGING_THREADS', 'default'), ):
    def stage():
        lib.pca_host_stage_h2d(1, vp(src.ctypes.data), vp(pin.data_ptr()), vp(dev.data_ptr()), (C.c_int64 * 1)(n), ctx.stream())
    dt = timeit(stage)
    print('pca_host_stage_h2d (threads=%s): %.2f ms per 26 MB = %.1f GB/s end to end' % (thr, 1e3 * dt, n / dt / 1e9))
t0 = time.perf_counter()
for _ in range(20):
    lib.pca_host_stage_h2d(1, vp(src.ctypes.data), vp(pin.data_ptr()), vp(dev.data_ptr()), (C.c_int64 * 1)(0), ctx.stream())
