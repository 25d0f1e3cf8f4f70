"""Where the host's time goes in the NuScenes scene from HOST arrays (bench.py nuscenes_scene_pass, pcie form): wall time of the
sections of integrate(), summed over the scene, and the scene's total (device included)."""
import builtins, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import nuscenes_oracle_sem_pc_accum as mod  # noqa: E402
from pca_amd import ingest  # noqa: E402

acc_t = {}


def timed(owner, name):
    f = getattr(owner, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc_t[name] = acc_t.get(name, 0.0) + time.perf_counter() - t0
    setattr(owner, name, g)


C = mod.NuScenesOracleSemanticPointCloudAccumulator
for n in ('integrate', '_frame_inputs', '_append_frame', 'generate_bev'):
    timed(C, n)
timed(ingest.PinnedUploader, 'upload_many')
timed(ingest.PinnedUploader, 'upload_stack')
rp, builtins.print = builtins.print, (lambda *a, **k: None)
out = bench.nuscenes_scene_pass(reps=2, forms=('pcie', ))
builtins.print = rp
runs = 3                                     # warm-up + 2 timed repetitions
print('pcie scene: %.1f ms' % out['pcie']['ms_per_scene'])
for k, v in sorted(acc_t.items(), key=lambda kv: -kv[1]):
    print('  %-16s %.2f ms per scene (host wall)' % (k, 1e3 * v / runs))
