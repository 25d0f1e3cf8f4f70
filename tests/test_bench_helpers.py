"""CPU checks of bench.py's synthetic generators and report helpers (no GPU, no kernels)."""
import numpy as np

import bench


def test_sweep_ordered_nuscenes_rows_have_the_locality_of_a_real_sweep():
    rows, cam = bench.nusc_sweep_rows(3)
    again, cam2 = bench.nusc_sweep_rows(3)
    assert rows.shape == (34_720, 7) and cam.shape == (34_720, ) and np.array_equal(rows, again) and np.array_equal(cam, cam2)
    assert set(np.unique(cam)) <= set(range(-1, 6)) and 0.5 < (cam >= 0).mean() < 0.9
    on = cam >= 0
    u, v = rows[on, 4], rows[on, 5]
    assert u.min() > 1.0 and u.max() < 1599.0 and v.min() > 1.0 and v.max() < 899.0       # pts_feat_from_img's open box
    # the 32 beams of one azimuth step share the image column; the next step is a few pixels on: neighbours in the array
    # are neighbours in the image (SURVEY 8d's uniform draw gives a median column step of hundreds of pixels)
    same_cam = cam[1:] == cam[:-1]
    both = on[1:] & on[:-1] & same_cam
    assert np.median(np.abs(np.diff(rows[:, 4]))[both]) < 8.0
    uni = np.random.default_rng(0).uniform(1.01, 1598.99, 34_720)
    assert np.median(np.abs(np.diff(uni))) > 300.0


def test_config5_headline_fields():
    c5 = {'ideal_speedup_of_this_plan': 6.86, 'seconds_per_rank': [0.11, 0.09, 0.12]}
    out = bench.config5_headline(c5)
    assert out == {'config5_ideal_speedup_of_this_plan': 6.86, 'config5_seconds_per_rank_min': 0.09,
                   'config5_seconds_per_rank_max': 0.12}
    assert bench.config5_headline({'seconds_per_rank': {'error': 'x'}}) == {'config5_ideal_speedup_of_this_plan': None}
