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
import os
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


class ArrayPairFlow:
    """Tensor-backed stand-in for the reference's per-file dataflow (`create_input_dataflow`,
    /root/reference/deepclr/data/datasets/build.py:97-130): one data point per scan pair with the reference's unified
    structure {dataset, idx [i, j], timestamps [t_i, t_j], clouds [template, source], transform (4, 4), augmentations
    [None, None]}, float arrays as float32 (ToFloat32, build.py:72-95), `reset_state()` / `len()` / iteration as
    scripts/inference.py:78-85 uses them. The source is a `.npz` file instead of an LMDB directory:

      KITTI_ODOMETRY_VELODYNE / GENERIC sequence file:  clouds (T, N, C), poses (T, 4, 4) [, timestamps (T,)]
          -> T - 1 pairs of consecutive frames, transform = inv(pose_i) pose_{i+1} (MergePairSequence + _get_motion,
          build.py:32-52)
      pair file (any dataset type):  templates (P, N, C), sources (P, N, C), transforms (P, 4, 4) [, timestamps (P, 2)]
      MODELNET40 object file:  clouds (T, N, C)  ->  every cloud with a copy of itself, identity transform, timestamps =
          the index (DuplicateCloud, build.py:55-69)
    """

    def __init__(self, dataset_type: Any, filename: str, shuffle: bool = False):
        if shuffle:
            raise RuntimeError("shuffled reading belongs to the training pipeline, which is outside this build")
        with np.load(filename, allow_pickle=False) as f:
            arrays = {k: f[k] for k in f.files}
        self._name = os.path.splitext(os.path.basename(filename))[0]
        kind = getattr(dataset_type, 'name', str(dataset_type)).upper()
        if 'templates' in arrays:
            self._t, self._s = arrays['templates'], arrays['sources']
            self._m = arrays['transforms']
            n = len(self._t)
            self._stamps = arrays.get('timestamps', np.stack([2.0 * np.arange(n), 2.0 * np.arange(n) + 1.0], axis=1))
            self._idx = np.stack([np.arange(n), np.arange(n)], axis=1)
        elif kind == 'MODELNET40':
            c = arrays['clouds']
            self._t, self._s = c, c
            self._m = np.tile(np.eye(4), (len(c), 1, 1))
            self._idx = np.stack([np.arange(len(c))] * 2, axis=1)
            self._stamps = self._idx.astype(np.float64)
        else:
            c, poses = arrays['clouds'], arrays['poses'].astype(np.float64)
            if len(c) < 2 or len(poses) != len(c):
                raise RuntimeError("sequence file needs clouds (T >= 2, N, C) and poses (T, 4, 4)")
            self._t, self._s = c[:-1], c[1:]
            self._m = np.stack([np.linalg.inv(poses[i]).dot(poses[i + 1]) for i in range(len(c) - 1)])
            stamps = arrays.get('timestamps', np.arange(len(c), dtype=np.float64))
            self._stamps = np.stack([stamps[:-1], stamps[1:]], axis=1)
            self._idx = np.stack([np.arange(len(c) - 1), np.arange(1, len(c))], axis=1)

    def reset_state(self) -> None:
        pass

    def __len__(self) -> int:
        return len(self._t)

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        for i in range(len(self._t)):
            yield {'dataset': self._name, 'idx': [int(self._idx[i, 0]), int(self._idx[i, 1])],
                   'timestamps': [float(self._stamps[i, 0]), float(self._stamps[i, 1])],
                   'clouds': [np.ascontiguousarray(self._t[i], dtype=np.float32),
                              np.array(self._s[i], dtype=np.float32, copy=True)],
                   'transform': self._m[i].astype(np.float32), 'augmentations': [None, None]}


def create_input_dataflow(dataset_type: Any, filename: str, shuffle: bool = False) -> ArrayPairFlow:
    """Same call as the reference's (data/datasets/build.py:97). `.npz` files are read here (ArrayPairFlow); the
    reference's LMDB directories need its dataflow / lmdb readers, which are not part of this build."""
    if not str(filename).endswith('.npz'):
        raise RuntimeError("'{}': the reference's LMDB datasets are read through dataflow + lmdb (not installed, outside "
                           "the MI355X forward hot path); convert the sequence to a .npz file (clouds, poses[, timestamps]) "
                           "or feed tensors to ModelInferenceHelper directly".format(filename))
    return ArrayPairFlow(dataset_type, filename, shuffle)
