"""Host-side helpers that have no golden fixture of their own (CPU, no GPU needed)."""
import numpy as np


def test_conv_semantic_ids_lookup_equals_sequential_masking():
    """datasets.kitti360_utils.conv_semantic_ids composes the sequential (old -> new) passes into one table for (N,1)
    integer arrays; the result must be that of the reference's in-order masked assignments (kitti360_utils.py:18-22 of the
    reference), chains and ids outside the table included."""
    from datasets.kitti360_utils import conv_semantic_ids
    from obs_dataloaders.kitti360_obs_dataloader import Kitti360Dataloader

    def sequential(a, d):
        for o, n in d.items():
            a[a[:, 0] == o] = n
        return a
    rng = np.random.default_rng(0)
    m = Kitti360Dataloader.gen_idx_mapping()
    for dt in (np.int16, np.int32, np.int64, np.uint8, np.float64):
        a = rng.integers(-1 if dt != np.uint8 else 0, 60, (20000, 1)).astype(dt)
        assert np.array_equal(conv_semantic_ids(a.copy(), m), sequential(a.copy(), m)), dt
    chain = {3: 7, 7: 100, 100: -5, -5: 3, 9: 9}
    a = rng.integers(-300, 300, (20000, 1)).astype(np.int16)
    assert np.array_equal(conv_semantic_ids(a.copy(), chain), sequential(a.copy(), chain))
    wide = rng.integers(0, 50, (1000, 3)).astype(np.int16)                 # not (N,1): whole rows are assigned
    assert np.array_equal(conv_semantic_ids(wide.copy(), m), sequential(wide.copy(), m))
