"""NuScenes + ICP accumulator -- name kept importable for ``run_nuscenes_bev_gen.py``.

In the reference this variant is dead code: ``integrate`` raises NotImplementedError unconditionally
(nuscenes_sem_pc_accum.py:68).  The same behaviour is kept; use the oracle-pose accumulator.
"""
from sem_pc_accum import SemanticPointCloudAccumulator


class NuScenesSemanticPointCloudAccumulator(SemanticPointCloudAccumulator):

    def __init__(self, horizon_dist, icp_threshold, semseg_onnx_path=None, semseg_filters=None, sem_idxs=None,
                 use_gt_sem=None, bev_params=None, loc=None):
        super().__init__(horizon_dist, icp_threshold, semseg_onnx_path, semseg_filters, sem_idxs, use_gt_sem,
                         bev_params)
        if use_gt_sem:
            raise NotImplementedError()
        self.xyz_idx = 0
        self.dyn_idx = 8
        self.map = loc
        self.ego_global_xs = []
        self.ego_global_ys = []

    def integrate(self, observations: list):
        raise NotImplementedError('Check that new obs_dataloader works')
