"""The staging pool of pca_host_stage_h2d / pca_kitti_integrate (csrc/pca_stage_pool.h) under ThreadSanitizer: back-to-back
jobs with three helper threads per core, slice tables freed between jobs (a helper left over from job N must
not copy or count a slice of job N + 1).  CPU only: the header is plain C++."""
import os
import shutil
import subprocess

import pytest

from conftest import PKG, ROOT


def test_back_to_back_jobs_with_more_threads_than_cores_under_tsan(tmp_path):
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('no g++')
    exe = str(tmp_path / 'stage_pool_tsan')
    src = os.path.join(ROOT, 'tests', 'native', 'stage_pool_tsan.cpp')
    build = subprocess.run([gxx, '-std=c++17', '-O1', '-g', '-fsanitize=thread', '-I', os.path.join(PKG, 'csrc'), src, '-o', exe,
                            '-pthread'], capture_output=True, text=True)
    if build.returncode != 0 and 'tsan' in (build.stderr or '').lower():
        pytest.skip('ThreadSanitizer runtime not installed: ' + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    helpers = min(3 * (os.cpu_count() or 4), 48)
    run = subprocess.run([exe, str(helpers), '1500'], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, TSAN_OPTIONS='halt_on_error=1 exitcode=66'))
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert 'ThreadSanitizer' not in run.stderr, run.stderr[-3000:]
    assert ', 0 bad' in run.stdout
