"""Device-resident batch feeding for the trainer.

``FastTensorDataLoader`` mirrors the reference's class of the same name (``data_loaders.py:486-551``): same
constructor, same batch boundaries, same permutation for a given ``torch`` seed (the permutation is drawn on the CPU
generator, exactly as the reference draws it, then applied on whatever device the tensors live on).  Keeping the
tensors on the GPU removes the per-iteration host-to-device copy of ``ClavaDDPMTrainer._train_step``."""
from __future__ import annotations

from collections.abc import Iterator

import torch
from torch import Tensor


class FastTensorDataLoader:
    """data_loaders.py:486.  ``tensors`` may live on any device; pass ``device=`` to move them once, up front."""

    def __init__(self, tensors: list[Tensor], batch_size: int = 32, shuffle: bool = False, device: torch.device | str | None = None):
        assert all(t.shape[0] == tensors[0].shape[0] for t in tensors), "All tensors must have the same amount of samples."
        self.tensors = [t.to(device) for t in tensors] if device is not None else list(tensors)
        self.dataset_len = self.tensors[0].shape[0]
        self.batch_size = batch_size
        self.shuffle = shuffle
        n_batches, remainder = divmod(self.dataset_len, self.batch_size)
        self.n_batches = n_batches + (1 if remainder > 0 else 0)

    def __iter__(self) -> "FastTensorDataLoader":
        if self.shuffle:
            r = torch.randperm(self.dataset_len)  # CPU generator: the reference's stream (data_loaders.py:527)
            self.tensors = [t[r.to(t.device)] for t in self.tensors]
        self.i = 0
        return self

    def __next__(self) -> list[Tensor]:
        if self.i >= self.dataset_len:
            raise StopIteration
        batch = [t[self.i:self.i + self.batch_size] for t in self.tensors]
        self.i += self.batch_size
        return batch

    def __len__(self) -> int:
        return self.n_batches


def endless(loader: FastTensorDataLoader) -> Iterator[list[Tensor]]:
    """The generator ``prepare_fast_dataloader`` returns (data_loaders.py:593-594): epochs back to back, reshuffled
    at every epoch start when the loader shuffles."""
    while True:
        yield from loader
