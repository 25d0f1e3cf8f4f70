"""Iterator protocol of the observation loaders (reference: obs_dataloaders/obs_dataloader.py)."""
from abc import ABC, abstractmethod


class ObservationDataloader(ABC):

    def __init__(self, root_path: str, batch_size: int):
        self.root_path = root_path
        self.batch_size = batch_size

    @abstractmethod
    def read_obs(self, idx):
        pass

    @abstractmethod
    def __len__(self):
        pass

    def __iter__(self):
        self.idx = 0
        return self

    def __next__(self):
        """Next list of `batch_size` observations; a trailing partial batch is dropped."""
        if self.idx + self.batch_size > len(self):
            raise StopIteration
        batch = [self.read_obs(self.idx + k) for k in range(self.batch_size)]
        self.idx += self.batch_size
        return batch
