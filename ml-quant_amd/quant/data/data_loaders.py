"""Data loaders for the driver scripts.

The reference's loaders (``quant/data/data_loaders.py:18-375``) are torchvision pipelines over MNIST / CIFAR /
ImageNet on disk.  torchvision is not part of this build's environment and the benchmark measures the forward
path on tensors resident in HBM, so the loaders here keep the reference's constructor
(``dataset_path, train_batch_size, test_batch_size, workers``) and interface (``get_train_loader``,
``get_test_loader``, ``cleanup``) and serve SYNTHETIC samples of each dataset's shape and class count, generated
once on the CPU from a fixed seed.  ``n_test`` bounds the synthetic test set (default: four test batches), ``n_train``
the synthetic training set (default: eight training batches).  The two sets are DIFFERENT samples (seeds ``seed`` and
``seed + 1``) and the training loader shuffles, as the reference's does (:95-101); there is no augmentation because there
are no images.  Anything a run on these loaders reports -- loss curves, top-k -- is a statement about noise: it exercises
the loop, the kernels and the checkpoints, not the accuracy of a scheme.
"""

from abc import ABC, abstractmethod
from typing import Optional, Tuple

import torch
from torch.utils.data import DataLoader, TensorDataset


class QuantDataLoader(ABC):
    """Constructor signature and methods of the reference's ``QuantDataLoader`` (:18-61)."""

    def __init__(self, train_batch_size: int, test_batch_size: int, dataset_path: str, workers: int,
                 download: bool = True, test_sampler=None, n_test: Optional[int] = None, seed: int = 0,
                 n_train: Optional[int] = None) -> None:
        self.train_batch_size = train_batch_size
        self.test_batch_size = test_batch_size
        self.dataset_path = dataset_path
        self.workers = workers
        self.n_test = n_test if n_test is not None else 4 * test_batch_size
        self.n_train = n_train if n_train is not None else 8 * train_batch_size
        self.seed = seed
        self.test_sampler = test_sampler

    @abstractmethod
    def sample_shape(self) -> Tuple[int, int, int]:
        raise NotImplementedError

    @abstractmethod
    def classes(self) -> int:
        raise NotImplementedError

    def _synthetic(self, n: int, batch: int, seed: int, shuffle: bool, sampler=None) -> DataLoader:
        g = torch.Generator().manual_seed(seed)
        data = torch.randn((n,) + self.sample_shape(), generator=g)
        target = torch.randint(0, self.classes(), (n,), generator=g)
        order = torch.Generator().manual_seed(seed + 7919) if shuffle else None      # (reproducible epochs)
        return DataLoader(TensorDataset(data, target), batch_size=batch, shuffle=shuffle and sampler is None, sampler=sampler,
                          generator=order, num_workers=0)

    def get_train_loader(self) -> DataLoader:
        """Synthetic training set: its own samples (not the test set's), reshuffled every epoch."""
        return self._synthetic(self.n_train, self.train_batch_size, self.seed + 1, shuffle=True)

    def get_test_loader(self) -> DataLoader:
        return self._synthetic(self.n_test, self.test_batch_size, self.seed, shuffle=False, sampler=self.test_sampler)

    def cleanup(self) -> None:
        """Nothing to release (the reference's ImageNet loader removes its temporary copy here)."""


class MNISTDataLoader(QuantDataLoader):
    def sample_shape(self):
        return (1, 28, 28)

    def classes(self):
        return 10


class CIFAR10DataLoader(QuantDataLoader):
    def sample_shape(self):
        return (3, 32, 32)

    def classes(self):
        return 10


class CIFAR100DataLoader(CIFAR10DataLoader):
    def classes(self):
        return 100


class ImageNetDataLoader(QuantDataLoader):
    def sample_shape(self):
        return (3, 224, 224)

    def classes(self):
        return 1000
