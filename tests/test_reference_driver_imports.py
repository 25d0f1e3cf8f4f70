"""The reference's OWN driver script, unchanged, resolves every import against the drop-in root and gets as far
as the first device call (where, without a GPU, the product fails loudly).  Runs only where the reference
checkout exists (the build container); the GPU box has no /root/reference."""
import os
import subprocess
import sys

import pytest

from conftest import PKG
from fake_kitti import write_tree

REF_DRIVER = '/root/reference/run_kitti360_bev_gen.py'


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason='reference checkout not present')
def test_unchanged_kitti_driver_runs_up_to_the_device(tmp_path):
    import torch
    root = str(tmp_path / 'KITTI-360')
    write_tree(root, first_idx=130, n_frames=3)          # the driver starts sequence 0000 at frame 130
    env = dict(os.environ, PYTHONPATH=PKG, PCA_KITTI_T_FILE=os.path.join(root, 'T_new_prev.npy'),
               PYTHONDONTWRITEBYTECODE='1', PCA_PREFETCH='0')     # the plain loader: the first device call is integrate()
    r = subprocess.run([sys.executable, '-m', 'pca_amd.run', REF_DRIVER, root, 'none.onnx', '--use_gt_sem', '--bev_pixel_size', '32',
                        '--accum_horizon_dist', '20', '--bev_horizon_dist', '5'], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert 'ModuleNotFoundError' not in out and 'ImportError' not in out, out[-2000:]
    assert 'kitti360_sem_pc_accum.py' in out and 'integrate' in out, out[-2000:]
    if not torch.cuda.is_available():
        # no GPU here: the first integrate() must die loudly in the product, not fall back to anything
        assert 'no CPU fallback' in out, out[-2000:]
        assert PKG in out
        # the launcher's default (ingest pipeline on) dies just as loudly, in the loader
        env.pop('PCA_PREFETCH')
        r = subprocess.run([sys.executable, '-m', 'pca_amd.run', REF_DRIVER, root, 'none.onnx', '--use_gt_sem'],
                           cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
        assert 'no CPU fallback' in r.stdout + r.stderr and 'ingest.py' in r.stdout + r.stderr


REF_NUSC_DRIVER = '/root/reference/run_nuscenes_bev_gen.py'


@pytest.mark.skipif(not os.path.exists(REF_NUSC_DRIVER), reason='reference checkout not present')
def test_unchanged_nuscenes_driver_resolves_its_imports(tmp_path):
    """run_nuscenes_bev_gen.py:7-12 imports nuscenes_oracle_sem_pc_accum, nuscenes_sem_pc_accum and
    obs_dataloaders.nuscenes_obs_dataloader: all three must come from the drop-in root.  nuscenes-devkit is not
    installed here, so an empty stand-in package satisfies the driver's own `from nuscenes.nuscenes import NuScenes`;
    `--help` stops the script after its imports."""
    stub = tmp_path / 'stubs' / 'nuscenes'
    stub.mkdir(parents=True)
    (stub / '__init__.py').write_text('')
    (stub / 'nuscenes.py').write_text('class NuScenes:\n    pass\n')
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG, str(tmp_path / 'stubs')]), PYTHONDONTWRITEBYTECODE='1')
    probe = ('import sys, runpy\n'
             'sys.argv = [sys.argv[1], "--help"]\n'
             'try:\n'
             '    runpy.run_path(sys.argv[0], run_name="__main__")\n'
             'except SystemExit:\n'
             '    pass\n'
             'import nuscenes_oracle_sem_pc_accum as a, nuscenes_sem_pc_accum as b\n'
             'import obs_dataloaders.nuscenes_obs_dataloader as c\n'
             'print("FROM", a.__file__, b.__file__, c.__file__)\n')
    r = subprocess.run([sys.executable, '-c', probe, REF_NUSC_DRIVER], cwd=str(tmp_path), env=env, capture_output=True,
                       text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and 'ModuleNotFoundError' not in out and 'ImportError' not in out, out[-2000:]
    line = [ln for ln in out.splitlines() if ln.startswith('FROM')][0]
    assert line.count(PKG) == 3, line
    assert 'usage:' in out and 'nuscenes_path' in out
