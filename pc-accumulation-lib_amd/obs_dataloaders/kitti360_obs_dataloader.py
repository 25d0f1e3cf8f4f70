"""KITTI-360 observation loader (host I/O; reference: obs_dataloaders/kitti360_obs_dataloader.py).
Yields [(PIL image, (N,4) f32 point cloud, (N,1) trainId labels)]."""
import os

import numpy as np
import PIL.Image as Image
from datasets.kitti360_utils import conv_semantic_ids, read_pc_bin_file, read_sem_gt_bin_file

from obs_dataloaders.obs_dataloader import ObservationDataloader

# KITTI-360 label id -> Cityscapes trainId (kitti360scripts/helpers/labels.py), ids 0..44 then -1.
_TRAIN_ID = (2, 255, 255, 255, 2, 2, 9, 0, 1, 9, 9, 2, 3, 4, 2, 2, 2, 5, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 14,
             14, 16, 17, 18, 2, 4, 2, 5, 5, 2, 2, 2, 2, 13, 2)


class Kitti360Dataloader(ObservationDataloader):

    def __init__(self, root_path: str, batch_size: int, sequences: list, start_idxs: list, end_idxs: list):
        super().__init__(root_path, batch_size)
        self.pc_paths, self.img_paths, self.sem_gt_paths = [], [], []
        for seq, lo, hi in zip(sequences, start_idxs, end_idxs):
            for idx in range(lo, hi):
                name = self.idx2str(idx)
                self.pc_paths.append(os.path.join('data_3d_raw', seq, 'velodyne_points', 'data', name + '.bin'))
                self.img_paths.append(os.path.join('data_2d_raw', seq, 'image_00', 'data_rect', name + '.png'))
                self.sem_gt_paths.append(os.path.join('data_3d_semantics', 'raw', seq, 'labels', name + '.bin'))
        self.idx2idx = self.gen_idx_mapping()

    def __len__(self):
        return len(self.pc_paths)

    def read_obs(self, idx):
        pc = read_pc_bin_file(os.path.join(self.root_path, self.pc_paths[idx]))
        img = Image.open(os.path.join(self.root_path, self.img_paths[idx]))
        sem_gt_path = os.path.join(self.root_path, self.sem_gt_paths[idx])
        sem_gt = read_sem_gt_bin_file(sem_gt_path)
        if sem_gt is None:
            print(f"Missing GT sem: {sem_gt_path}")
            sem_gt = np.zeros((pc.shape[0], 1))
        return (img, pc, conv_semantic_ids(sem_gt, self.idx2idx))

    @staticmethod
    def idx2str(idx: int):
        return f"{idx:010d}"

    @staticmethod
    def gen_idx_mapping():
        """{label id: trainId}, applied sequentially in this order by conv_semantic_ids."""
        mapping = {k: v for k, v in enumerate(_TRAIN_ID)}
        mapping[-1] = 13   # license plate
        return mapping
