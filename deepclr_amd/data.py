"""Batch sources for the drop-in scripts.

The reference feeds its scripts through ``make_data_loader(cfg, is_train, batch_size=...)`` (tensorpack DataFlow over
LMDB, /root/reference/deepclr/data/build.py:205-241) whose batches are dicts ``{'x': (2B, N, C) float32 [templates...,
sources...], 'y': (B, label_dim), 'm': (2B, 4, 4), 'd': names, 't': (B, 2) timestamps}`` (data/build.py:62-98).
The LMDB / dataflow / pykitti readers are CPU I/O outside the MI355X hot path and their
packages are not installed here, so a configuration naming them raises. What this module does provide is a
tensor-backed source with the same batch contract, selected by ``data.dataset_type: synthetic_kitti`` or
``synthetic_modelnet`` in the configuration: seeded synthetic pairs (deepclr_amd.synthetic), so
that ``scripts/timing.py <config>`` runs end to end on the HIP path with nothing but this repository.
"""
from typing import Any, Dict, Iterator

import numpy as np
import torch

from . import synthetic
from .labels import LabelType

SYNTHETIC = {'synthetic_kitti': 'kitti', 'synthetic_modelnet': 'modelnet'}


class TensorDataLoader:
    """Iterable of reference-layout batches built from seeded synthetic scan pairs."""

    def __init__(self, cfg: Any, is_train: bool, batch_size: int = 1, **_kwargs: Any):
        data = cfg.data
        kind = str(data.get('dataset_type', '')).lower()
        if kind not in SYNTHETIC:
            raise RuntimeError("dataset_type '{}' needs the reference's LMDB / dataflow readers, which are outside the "
                               "MI355X forward hot path and not installed; use dataset_type: synthetic_kitti or "
                               "synthetic_modelnet, or feed tensors to ModelInferenceHelper directly".format(kind))
        self._kind = SYNTHETIC[kind]
        self._points = int(data.get('points', 16384 if self._kind == 'kitti' else 2048))
        self._pairs = int(data.get('pairs', 16))
        self._first = int(data.get('first_pair', 0)) + (0 if is_train else 100000)
        self._batch = int(batch_size)
        self._label_type = LabelType.create(cfg.model.label_type)
        self._cols = int(cfg.model.input_dim)

    def __len__(self) -> int:
        return (self._pairs + self._batch - 1) // self._batch

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        gen = {'kitti': synthetic.kitti_like_pair, 'modelnet': synthetic.modelnet_like_pair}[self._kind]
        for start in range(0, self._pairs, self._batch):
            ids = range(start, min(start + self._batch, self._pairs))
            pairs = [gen(self._first + i, self._points) for i in ids]
            x = np.stack([p[0][:, :self._cols] for p in pairs] + [p[1][:, :self._cols] for p in pairs], axis=0)
            y = np.stack([self._label_type.from_matrix(p[2]) for p in pairs]).astype(np.float32)
            m = np.tile(np.eye(4, dtype=np.float32), (2 * len(pairs), 1, 1))
            yield {'x': torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)), 'y': torch.from_numpy(y),
                   'm': torch.from_numpy(m), 'd': np.array(['synthetic_%s' % self._kind] * len(pairs)),
                   't': torch.tensor([[2 * i, 2 * i + 1] for i in ids], dtype=torch.int64)}


def make_data_loader(cfg: Any, is_train: bool, **kwargs: Any) -> TensorDataLoader:
    """Same call as the reference's (data/build.py:240-241)."""
    return TensorDataLoader(cfg, is_train, **kwargs)
