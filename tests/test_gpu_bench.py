"""bench.py end to end on the GPU box: the one-line JSON contract at N = 1, and the multi-rank control flow rehearsed
with two gloo ranks sharing the one GPU (PCA_BENCH_BACKEND=gloo; the real multi-GPU run uses nccl = RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline'}


def last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '1', '--no-ring'],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    d = last_json(r.stdout)
    assert REQUIRED <= set(d) and 'cpu_baseline' in d
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['unit'] == 'Mpoints/s' and d['value'] > 50 and d['higher_is_better'] is True and d['dtype'] == 'f64'
    assert 'workload' in d['config'] and 'model' not in d['config']
    rf = d['roofline']
    assert rf['bound'] == 'hbm' and rf['unit'] == 'GB/s' and rf['peak'] == 8000.0
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-12 and 0.05 < rf['frac'] < 1.0
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] == 1 and cb['value'] > 0 and 'sample' in cb


def test_bench_two_ranks_gloo_rehearsal():
    env = dict(os.environ, PCA_BENCH_BACKEND='gloo')
    for extra in ([], ['--gather']):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
               '127.0.0.1', '--master-port', '29541', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20',
               '--warmup', '1'] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        d = last_json(r.stdout)
        assert d['n_gpus'] == 2 and d['steps'] == 20 and d['value'] > 10
        assert d['gather_check']['in_timed_region'] is bool(extra) and d['gather_check']['samples_per_rank'] == 4
        assert 'cpu_baseline' not in d and 'ring_model' not in d            # reported at N = 1 only
