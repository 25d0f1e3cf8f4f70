"""Iterator protocol of the observation loaders (reference: obs_dataloaders/obs_dataloader.py).

A loader is its own iterator: `for batch in loader` yields lists of `batch_size` consecutive observations
(`read_obs(i)`), a trailing partial batch is dropped, and iterating again restarts from observation 0.
"""
import os
from abc import ABC, abstractmethod


class ObservationDataloader(ABC):

    def __init__(self, root_path: str, batch_size: int):
        self.root_path, self.batch_size = root_path, batch_size
        self.idx = 0

    @abstractmethod
    def read_obs(self, idx):
        """One observation (KITTI-360: (PIL image, (N,4) points, labels | None); NuScenes: a dict)."""

    @abstractmethod
    def __len__(self):
        """Number of observations."""

    def __iter__(self):
        self.idx = 0
        if os.environ.get('PCA_PREFETCH', '0') not in ('', '0') and hasattr(self, 'pc_paths'):
            # ingest pipeline (pca_amd/ingest.py): a reader thread decodes ahead and uploads through pinned buffers; the
            # batches it yields hold device tensors, which the accumulators take in place of numpy / PIL inputs
            from pca_amd.ingest import PrefetchingLoader
            return iter(PrefetchingLoader(self, depth=int(os.environ.get('PCA_PREFETCH_DEPTH', '4'))))
        if os.environ.get('PCA_PREFETCH', '0') not in ('', '0') and hasattr(self, 'read_host'):
            # NuScenes: sweep merging and image decoding on reader threads, the six images through a pinned ring, projection
            # (K0n) and the point rows left on the device
            from pca_amd.ingest import NuScenesPrefetchingLoader
            return iter(NuScenesPrefetchingLoader(self, depth=int(os.environ.get('PCA_PREFETCH_DEPTH', '3'))))
        return self

    def __next__(self):
        first, stop = self.idx, self.idx + self.batch_size
        if stop > len(self):
            raise StopIteration
        self.idx = stop
        return [self.read_obs(k) for k in range(first, stop)]
