"""Merging the lidar sweeps of a NuScenes sample and labelling points with their ground-truth boxes.

What the reference does in ``inst_centric_get_sweeps`` (datasets/nuscenes_utils.py:246-531) followed by
``load_data_to_tensor`` (:533-545), restated for the observation loader (ingest):

  * the keyframe's LIDAR_TOP record and its ``n_sweeps - 1`` predecessors (a record without predecessor repeats), oldest
    first, each with its time lag [s] and sweep index (newest = n_sweeps - 1);
  * every sweep: f32 points of the .bin file, points within ``center_radius`` of the sensor axis dropped, moved into the
    keyframe's lidar frame (f64 product, stored back as f32), columns [x, y, z, intensity, lag, sweep, instance, class];
  * every annotated box of the sweep whose class is wanted and that has lidar points: the points inside the box (with
    tolerance) get the instance index of the box's track and its class index; later boxes overwrite earlier ones;
  * per labelled box occurrence: the track token and the box centre (global frame), in encounter order.

The dataset object only has to offer what the devkit's ``NuScenes`` offers: ``get(table, token)``, ``get_sample_data_path``,
``get_boxes`` (boxes with .name .token .center .orientation .wlh) and ``box_velocity``.  Quaternions are turned into
rotation matrices by ``rotation_matrix`` below (the textbook formula pyquaternion also implements; an ``orientation``
object that carries its own ``rotation_matrix`` is used as is).
"""
import numpy as np

# devkit category -> detection class (nuscenes-devkit eval/detection; the reference keeps a copy at :14-38)
DETECTION_NAME = {
    'human.pedestrian.adult': 'pedestrian', 'human.pedestrian.child': 'pedestrian',
    'human.pedestrian.police_officer': 'pedestrian', 'human.pedestrian.construction_worker': 'pedestrian',
    'vehicle.car': 'car', 'vehicle.motorcycle': 'motorcycle', 'vehicle.bicycle': 'bicycle',
    'vehicle.bus.bendy': 'bus', 'vehicle.bus.rigid': 'bus', 'vehicle.truck': 'truck',
    'vehicle.construction': 'construction_vehicle', 'vehicle.trailer': 'trailer',
    'movable_object.barrier': 'barrier', 'movable_object.trafficcone': 'traffic_cone',
}   # every other category is 'ignore'


def rotation_matrix(q):
    """3x3 rotation of a unit quaternion (w, x, y, z), or of an object with a ``rotation_matrix`` attribute."""
    if hasattr(q, 'rotation_matrix'):
        return np.asarray(q.rotation_matrix, dtype=np.float64)
    w, x, y, z = (float(v) for v in (q.elements if hasattr(q, 'elements') else q))
    n = np.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rigid(translation, rotation):
    T = np.eye(4)
    T[:3, :3] = rotation_matrix(rotation)
    T[:3, 3] = translation
    return T


def sensor_in_global(nusc, sd_token):
    """4x4 pose of a sample_data's sensor in the global frame: ego pose x calibrated sensor pose."""
    rec = nusc.get('sample_data', sd_token)
    cs = nusc.get('calibrated_sensor', rec['calibrated_sensor_token'])
    ego = nusc.get('ego_pose', rec['ego_pose_token'])
    return rigid(ego['translation'], ego['rotation']) @ rigid(cs['translation'], cs['rotation'])


def sweep_chain(nusc, keyframe_sd_token, n_sweeps):
    """[(token, time lag in s, sweep index)] oldest first; a record without 'prev' is repeated."""
    t_ref = nusc.get('sample_data', keyframe_sd_token)['timestamp'] * 1e-6
    chain, token = [], keyframe_sd_token
    for back in range(n_sweeps):
        rec = nusc.get('sample_data', token)
        chain.append((token, t_ref - rec['timestamp'] * 1e-6, n_sweeps - 1 - back))
        if rec['prev'] != '':
            token = rec['prev']
    return chain[::-1]


def _apply(T, xyz):
    """rows of [xyz 1] @ T.T, first three columns (f64 whatever the point dtype)."""
    h = np.concatenate([xyz, np.ones((xyz.shape[0], 1), dtype=xyz.dtype)], axis=1)
    return (h @ T.T)[:, :3]


def inst_centric_get_sweeps(nusc, sample_token, n_sweeps, center_radius, in_box_tolerance, return_instances_last_box,
                            point_cloud_range, detection_classes, map_point_feat2idx):
    """Returns {'points' (N,8) f32, 'instances_token' list, 'instances_center' list, and with
    return_instances_last_box 'instances_last_box' (n_inst,9) f32 + 'instances_name' (n_inst,) f32} -- the values the
    reference holds after load_data_to_tensor (f32 arrays instead of f32 tensors)."""
    key_sd = nusc.get('sample', sample_token)['data']['LIDAR_TOP']
    target_from_glob = np.linalg.inv(sensor_in_global(nusc, key_sd))
    col_inst, col_cls = map_point_feat2idx['inst_idx'], map_point_feat2idx['cls_idx']

    track_index = {}                      # instance token -> instance index (order of first labelled appearance)
    tracks = []                           # per instance: dict(poses, sweeps, size, cls, anno)
    tokens, centres, clouds = [], [], []
    for sd_token, lag, sweep in sweep_chain(nusc, key_sd, n_sweeps):
        raw = np.fromfile(nusc.get_sample_data_path(sd_token), dtype=np.float32).reshape(-1, 5)
        pts = np.full((raw.shape[0], 8), -1.0, dtype=np.float32)
        pts[:, :4] = raw[:, :4]
        pts[:, 4], pts[:, 5] = lag, sweep
        pts = pts[np.linalg.norm(pts[:, :2], axis=1) > center_radius]
        pts[:, :3] = _apply(target_from_glob @ sensor_in_global(nusc, sd_token), pts[:, :3])
        for box in nusc.get_boxes(sd_token):
            cls_name = DETECTION_NAME.get(box.name, 'ignore')
            if cls_name not in detection_classes:
                continue
            anno = nusc.get('sample_annotation', box.token)
            if anno['num_lidar_pts'] < 1:
                continue
            target_from_box = target_from_glob @ rigid(box.center, box.orientation)
            size = np.array([box.wlh[1], box.wlh[0], box.wlh[2]])            # length, width, height = dx, dy, dz
            local = _apply(np.linalg.inv(target_from_box), pts[:, :3])
            inside = np.all(np.abs(local / size) < (0.5 + in_box_tolerance), axis=1)
            if not inside.any():
                continue
            itok = anno['instance_token']
            if itok not in track_index:
                track_index[itok] = len(tracks)
                tracks.append({'poses': [], 'sweeps': [], 'size': size.tolist(),
                               'cls': detection_classes.index(cls_name), 'anno': None})
            tr = tracks[track_index[itok]]
            tr['poses'].append(target_from_box)
            tr['sweeps'].append(sweep)
            tr['anno'] = anno['token']
            pts[inside, col_inst] = track_index[itok]
            pts[inside, col_cls] = detection_classes.index(cls_name)
            tokens.append(itok)
            centres.append(box.center)
        clouds.append(pts)

    out = {'points': np.concatenate(clouds, axis=0) if clouds else np.zeros((0, 8), np.float32),
           'instances_token': tokens, 'instances_center': centres}
    if return_instances_last_box:
        assert point_cloud_range is not None
        rng = np.asarray(point_cloud_range, dtype=np.float64)
        last = np.zeros((len(tracks), 9))
        for k, tr in enumerate(tracks):
            # newest pose whose centre lies inside the range, else the oldest one
            pick = tr['poses'][0]
            for pose in reversed(tr['poses']):
                c = pose[:3, 3]
                if np.all((c >= rng[:3]) & (c < rng[3:] - 1e-2)):
                    pick = pose
                    break
            last[k, :3] = pick[:3, 3]
            last[k, 3:6] = tr['size']
            last[k, 6] = np.arctan2(pick[1, 0], pick[0, 0])
            vel = np.asarray(nusc.box_velocity(tr['anno']), dtype=np.float64).reshape(1, 3)     # global frame
            last[k, 7:9] = _apply(target_from_glob, vel).reshape(3)[:2]
        out['instances_last_box'] = last.astype(np.float32)
        out['instances_name'] = np.array([tr['cls'] for tr in tracks]).astype(np.float32)
    return out
