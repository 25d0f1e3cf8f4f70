"""A dataset object with the few nuscenes-devkit calls the sweep merger uses, built from plain arrays (test fixture).

Shared by tools/make_golden.py (which runs the REFERENCE's inst_centric_get_sweeps on it) and the parity test of
datasets/nuscenes_sweeps.py.  Quaternion -> matrix is the textbook formula (pyquaternion is not in this image; parity at
that third-party call is unpinned)."""
import os
import types

import numpy as np


class FakeQuaternion:
    def __init__(self, elements):
        self.elements = np.asarray(elements, dtype=np.float64)

    @property
    def rotation_matrix(self):
        w, x, y, z = self.elements / np.linalg.norm(self.elements)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


CATEGORIES = ('vehicle.car', 'vehicle.truck', 'human.pedestrian.adult', 'animal', 'vehicle.bus.rigid',
              'movable_object.barrier', 'vehicle.bicycle')


def synth_tables(seed=5, n_records=3, n_pts=2500):
    """Arrays describing a keyframe with n_records - 1 predecessors (fewer than the sweeps asked for: the oldest repeats),
    moving ego, and boxes: wanted / unwanted classes, a box without lidar points, an empty box, overlapping boxes, tracks
    seen in several sweeps."""
    rng = np.random.default_rng(seed)
    t = {'n_records': n_records}
    t['timestamp'] = (1_600_000_000_000_000 + 50_000 * np.arange(n_records)).astype(np.int64)      # oldest first
    quat = lambda yaw, tilt: np.array([np.cos(yaw / 2), tilt, -tilt, np.sin(yaw / 2)])
    t['ego_t'] = np.stack([[400.0 + 0.6 * k, 1100.0 + 0.1 * k, 0.0] for k in range(n_records)])
    t['ego_q'] = np.stack([quat(0.3 + 0.01 * k, 0.002) for k in range(n_records)])
    t['cs_t'] = np.tile(np.array([0.94, 0.0, 1.84]), (n_records, 1))
    t['cs_q'] = np.tile(quat(-1.57, 0.004), (n_records, 1))
    pts = rng.uniform(-30, 30, (n_records, n_pts, 5)).astype(np.float32)
    pts[:, :, 2] = rng.uniform(-2, 2, (n_records, n_pts))
    pts[:, :, 3] = rng.integers(0, 256, (n_records, n_pts))
    pts[:, :50, :2] *= 0.05                                  # some points next to the sensor axis
    t['points'] = pts
    boxes = []                                               # (record, category, instance, n_lidar, centre, yaw, wlh, vel)
    for k in range(n_records):
        ego_xy = t['ego_t'][k, :2]
        spec = [(0, 0, 5, (8.0, 3.0), 0.4, (2.0, 4.6, 1.7)), (1, 1, 9, (-9.0, 6.0), 1.1, (2.6, 8.0, 3.2)),
                (2, 2, 3, (4.0, -7.0), 0.0, (0.8, 0.8, 1.8)), (3, 3, 4, (2.5, 9.0), 0.0, (0.6, 1.0, 0.7)),
                (0, 4, 0, (-14.0, -3.0), 0.2, (2.0, 4.4, 1.6)), (0, 5, 7, (300.0, 300.0), 0.0, (2.0, 4.5, 1.6)),
                (4, 6, 6, (8.5, 3.4), 0.5, (3.0, 11.0, 3.4)), (6, 7, 2, (-4.0, -12.0), 2.0, (0.7, 1.9, 1.4))]
        if k == 0:
            spec = spec[:5]                                  # tracks 5..7 appear later
        for cat, inst, nl, xy, yaw, wlh in spec:
            c = np.array([ego_xy[0] + xy[0] + 0.3 * k * (inst == 0), ego_xy[1] + xy[1], 0.9])
            boxes.append((k, cat, inst, nl, c, yaw + 0.3, wlh, rng.normal(0, 2, 3)))
    t['box_record'] = np.array([b[0] for b in boxes])
    t['box_category'] = np.array([b[1] for b in boxes])
    t['box_instance'] = np.array([b[2] for b in boxes])
    t['box_lidar_pts'] = np.array([b[3] for b in boxes])
    t['box_center'] = np.stack([b[4] for b in boxes])
    t['box_q'] = np.stack([quat(b[5], 0.0) for b in boxes])
    t['box_wlh'] = np.array([b[6] for b in boxes], dtype=np.float64)
    t['box_vel'] = np.stack([b[7] for b in boxes])
    return t


class FakeNuScenes:
    """nusc.get / get_sample_data_path / get_boxes / box_velocity over the arrays of synth_tables."""

    def __init__(self, tables, tmpdir, quaternion=FakeQuaternion):
        self.t = tables
        self.dir = str(tmpdir)
        self.q = quaternion
        n = int(tables['n_records'])
        self.sd = ['sd%d' % k for k in range(n)]
        for k in range(n):
            np.asarray(tables['points'][k], dtype=np.float32).tofile(os.path.join(self.dir, self.sd[k] + '.bin'))

    def get(self, table, token):
        t = self.t
        if table == 'sample':
            return {'data': {'LIDAR_TOP': self.sd[-1]}, 'scene_token': 'scene0', 'next': ''}
        if table == 'sample_data':
            k = self.sd.index(token)
            return {'timestamp': int(t['timestamp'][k]), 'prev': self.sd[k - 1] if k > 0 else '',
                    'calibrated_sensor_token': 'cs%d' % k, 'ego_pose_token': 'ego%d' % k}
        if table == 'calibrated_sensor':
            k = int(token[2:])
            return {'translation': t['cs_t'][k].tolist(), 'rotation': t['cs_q'][k].tolist()}
        if table == 'ego_pose':
            k = int(token[3:])
            return {'translation': t['ego_t'][k].tolist(), 'rotation': t['ego_q'][k].tolist()}
        if table == 'sample_annotation':
            b = int(token[3:])
            return {'token': token, 'num_lidar_pts': int(t['box_lidar_pts'][b]), 'sample_token': 'sample0',
                    'instance_token': 'inst%d' % int(t['box_instance'][b])}
        raise KeyError(table)

    def get_sample_data_path(self, token):
        return os.path.join(self.dir, token + '.bin')

    def get_boxes(self, sd_token):
        k, t = self.sd.index(sd_token), self.t
        return [types.SimpleNamespace(name=CATEGORIES[int(t['box_category'][b])], token='box%d' % b,
                                      center=np.array(t['box_center'][b]), orientation=self.q(t['box_q'][b]),
                                      wlh=np.array(t['box_wlh'][b]))
                for b in np.nonzero(t['box_record'] == k)[0]]

    def box_velocity(self, anno_token):
        return np.array(self.t['box_vel'][int(anno_token[3:])])


SWEEP_CFG = dict(n_sweeps=5, center_radius=2.0, in_box_tolerance=5e-2, return_instances_last_box=True,
                 point_cloud_range=[-1000, -1000, -1000, 1000, 1000, 1000],
                 detection_classes=('car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'motorcycle', 'bicycle',
                                    'pedestrian'),
                 map_point_feat2idx={'sweep_idx': 5, 'inst_idx': 6, 'cls_idx': 7})
