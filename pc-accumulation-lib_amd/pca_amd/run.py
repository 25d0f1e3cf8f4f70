"""Launcher for the reference's driver scripts, unchanged:

    python -m pca_amd.run /path/to/reference/run_kitti360_bev_gen.py <KITTI-360> <semseg.onnx> [driver args]

`python script.py` puts the script's own directory first on sys.path, so a driver started from inside the
reference checkout would import the reference's numpy modules.  This launcher puts the drop-in root first and
executes the script file as __main__ (the file itself is not modified or copied).  It also switches the ingest pipeline on
(PCA_PREFETCH=1: the KITTI loader decodes and uploads ahead on a reader thread); BEV samples leave the device
asynchronously and are written by background threads either way (PCA_ASYNC_WRITE=0 / PCA_SYNC_BEV=1 switch that off)."""
import os
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    script = sys.argv[1]
    pkg_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:] = [pkg_root] + [p for p in sys.path if os.path.abspath(p or '.') != pkg_root]
    for m in [m for m in sys.modules if m == 'datasets' or m.startswith('datasets.')]:
        del sys.modules[m]
    os.environ.setdefault('PCA_PREFETCH', '1')      # decode + upload ahead of the GPU (pca_amd/ingest.py); 0 disables
    if '--use_gt_sem' in sys.argv:                  # the driver's flag: the images are never looked at
        os.environ.setdefault('PCA_INGEST_IMAGES', '0')
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
