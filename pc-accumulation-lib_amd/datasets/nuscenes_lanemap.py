"""GT lane centrelines from the NuScenes map expansion (reference: datasets/nuscenes_lanemap.py).
Dataset plumbing; needs nuscenes-devkit and the map files."""
import numpy as np


def get_centerlines(dataroot: str, map_name: str, bbox: tuple = None, resolution_meters: float = 1.) -> list:
    from nuscenes.map_expansion.map_api import NuScenesMap
    nusc_map = NuScenesMap(dataroot=dataroot, map_name=map_name)
    poses = nusc_map.discretize_centerlines(resolution_meters)
    if bbox is not None:
        kept = []
        for p in poses:
            p = p[(p[:, 0] > bbox[0]) & (p[:, 0] < bbox[2])]
            kept.append(p[(p[:, 1] > bbox[1]) & (p[:, 1] < bbox[3])])
        poses = kept
    return [np.asarray(p) for p in poses]
