"""bench.py end to end on the GPU box: the one-line JSON contract at N = 1, and the multi-rank control flow rehearsed
with two gloo ranks sharing the one GPU (PCA_BENCH_BACKEND=gloo; the real multi-GPU run uses nccl = RCCL) -- started by
bench.py itself (`--gpus 2` with no launcher) and by torch.distributed.run as the driver does."""
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
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '1', '--no-ring',
                        '--config5-scale', '0.02'], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    d = last_json(r.stdout)
    assert REQUIRED <= set(d) and 'cpu_baseline' in d
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['unit'] == 'Mpoints/s' and d['value'] > 50 and d['higher_is_better'] is True and d['dtype'] == 'f64'
    assert d['repeats']['n'] >= 5 and d['repeats']['value_min'] <= d['value'] <= d['repeats']['value_max']
    assert 'workload' in d['config'] and 'model' not in d['config']
    rf = d['roofline']
    assert rf['bound'] == 'hbm' and rf['unit'] == 'GB/s' and rf['peak'] == 8000.0
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-12 and 0.05 < rf['frac'] < 1.0
    for k in ('k1_batched', 'k1_batched_distinct'):
        assert rf[k]['frames_per_call'] == 64 and 0.05 < rf[k]['frac'] < 1.0
    assert rf['k1_batched_distinct']['distinct_frames'] == 64
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] == 1 and cb['value'] > 0 and 'sample' in cb
    assert cb['numpy_shape']['value'] > 0 and cb['numpy_shape']['cores'] == os.cpu_count()
    assert d['config5']['scaling'] == 'strong' and d['config5']['bev_samples'] > 0 and d['config5']['seconds'] > 0
    assert d['nuscenes']['k1n']['us'] > 0 and d['config4']['ms_per_step'] > 0
    assert d['pcie_inclusive']['Mpoints_per_s'] > 10


def _check_two_ranks(d, gathered):
    assert d['n_gpus'] == 2 and d['steps'] == 20 and d['value'] > 10
    gc = d['gather_check']
    assert gc['in_timed_region'] is gathered and gc['samples_per_rank'] == 4 and gc['checksums_match'] is True
    assert 'cpu_baseline' not in d and 'ring_model' not in d            # reported at N = 1 only
    c5 = d['config5']
    assert c5['n_gpus'] == 2 and c5['gather_check']['checksums_match'] is True and len(c5['frames_incl_warmup_per_rank']) == 2


def test_bench_two_ranks_gloo_rehearsal():
    env = dict(os.environ, PCA_BENCH_BACKEND='gloo')
    for extra in ([], ['--gather']):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
               '127.0.0.1', '--master-port', '29541', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20',
               '--warmup', '1', '--config5-scale', '0.02'] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        _check_two_ranks(last_json(r.stdout), bool(extra))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: bench.py spawns the ranks itself and relays rank 0's line."""
    env = dict(os.environ, PCA_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '1',
                        '--config5-scale', '0.02'], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    _check_two_ranks(last_json(r.stdout), False)


def test_bench_config5_as_headline_workload():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'config5', '--config5-scale', '0.02'],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    d = last_json(r.stdout)
    assert d['scaling'] == 'strong' and d['n_gpus'] == 1 and d['value'] > 10 and d['config5']['frames'] == d['steps']


def test_bench_four_ranks_gloo_rehearsal_of_config5():
    """The strong-scaling job (BASELINE configs[4]) with FOUR ranks sharing the one GPU over gloo -- the most this box allows
    (eight ranks on one card are not): every rank runs chunks, the checksums of what was gathered match, per-rank wall times
    and the plan's loads are reported.  (The 8-rank PLAN is checked on the CPU: tests/test_sharded_plan.py.)"""
    env = dict(os.environ, PCA_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--workload', 'config5',
                        '--config5-scale', '0.02'], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = last_json(r.stdout)
    c5 = d['config5']
    assert d['n_gpus'] == 4 and d['scaling'] == 'strong' and c5['n_gpus'] == 4
    assert len(c5['frames_incl_warmup_per_rank']) == 4 and min(c5['frames_incl_warmup_per_rank']) > 0
    assert len(c5['seconds_per_rank']) == 4 and all(t > 0 for t in c5['seconds_per_rank'])
    assert c5['gather_check']['checksums_match'] is True and c5['ideal_speedup_of_this_plan'] > 2.5


def test_bench_one_rank_over_rccl():
    """The real transport on the one-GPU box: `torchrun --nproc-per-node 1 bench.py` initialises the `nccl` backend (RCCL on
    ROCm), every barrier / all-reduce of the timed region goes through it, the headline's chunk and config 5's last chunk
    are gathered as DEVICE tensors (shard.gather_to_rank0) and the checksums of what was sent and what arrived are compared.
    A multi-GPU node only adds peers to the same calls."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('PCA_BENCH_BACKEND', None)
    base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
            '--master-port', '29547', os.path.join(ROOT, 'bench.py'), '--gpus', '1']
    r = subprocess.run(base + ['--workload', 'config5', '--config5-scale', '0.01'], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = last_json(r.stdout)
    gc = d['config5']['gather_check']
    assert gc['backend'] == 'nccl' and gc['on_device_tensors'] is True and gc['checksums_match'] is True
    assert d['config5_ideal_speedup_of_this_plan'] == pytest.approx(1.0) and d['config5_seconds_per_rank_max'] > 0
    r = subprocess.run(base + ['--steps', '20', '--warmup', '1', '--no-extras', '--no-cpu-baseline', '--gather'],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = last_json(r.stdout)
    gc = d['gather_check']
    assert d['n_gpus'] == 1 and gc['backend'] == 'nccl' and gc['in_timed_region'] is True and gc['checksums_match'] is True


def test_bench_gather_in_the_timed_region_and_config5_in_one_rccl_process_group():
    """ONE process group over RCCL (world 1 on the one-GPU box) carries both: the headline with every finished BEV chunk
    streamed to rank 0 INSIDE the timed region (async gathers overlapping the next chunk's compute) and, afterwards, the
    sharded nine-sequence job with its own gather check -- the two runs of test_bench_one_rank_over_rccl as one."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('PCA_BENCH_BACKEND', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29549', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '1', '--gather',
           '--extras', 'config5', '--config5-scale', '0.03', '--no-cpu-baseline']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = last_json(r.stdout)
    assert d['n_gpus'] == 1 and d['gather_check']['backend'] == 'nccl' and d['gather_check']['in_timed_region'] is True
    assert d['gather_check']['checksums_match'] is True
    c5 = d['config5']
    assert c5['gather_check']['backend'] == 'nccl' and c5['gather_check']['on_device_tensors'] is True
    assert c5['gather_check']['checksums_match'] is True and c5['bev_samples'] > 0
    assert 'two_sequences' not in d and 'nuscenes' not in d                 # only the sharded job rode along


def test_bench_refuses_more_nccl_ranks_than_gpus_in_one_line():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('PCA_BENCH_BACKEND', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29551', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--no-extras']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert 'needs one GPU per rank' in r.stderr and 'Traceback' not in r.stderr.split('needs one GPU per rank')[0][-400:]
