"""Result files and odometry error metrics for the poses the hot path produces.

File format = the reference's (`/root/reference/deepclr/evaluation/data.py:104-137`): one text row per scan pair,
26 columns `[stamp, predicted 3x4 row-major (12), ground-truth 3x4 row-major (12), inference time in ms]`, written
with `numpy.savetxt` defaults, one file `<sequence name>.txt` per sequence (`evaluation/evaluator.py:142-144`), so
files written here load in the reference's `Evaluator.read` and the other way round.

Metrics = the KITTI odometry errors the reference reports (`evaluation/metrics.py:8-47`,
`evaluation/evaluator.py:22-65`): per pair, and over every 100..800 m segment starting at every 10th frame,
normalised by the segment length. Arrays instead of per-element objects; plots and the euler-angle error vectors
(transforms3d, absent here) are not part of this module.
"""
import os
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import yaml

from .config import Config, ConfigEnum

STEP_SIZE = 10                                            # evaluator.py:18 (one start frame per second at 10 Hz)
SEGMENT_LENGTHS = (100, 200, 300, 400, 500, 600, 700, 800)   # evaluator.py:19, metres


def _as_4x4(rows12: np.ndarray) -> np.ndarray:
    """(n, 12) row-major 3x4 -> (n, 4, 4)."""
    rows12 = np.asarray(rows12, dtype=np.float64).reshape(-1, 3, 4)
    out = np.tile(np.eye(4), (rows12.shape[0], 1, 1))
    out[:, :3, :] = rows12
    return out


def chain_poses(transforms: np.ndarray) -> np.ndarray:
    """Relative transforms (n, 4, 4) -> absolute poses (n + 1, 4, 4): P0 = I, P[i+1] = P[i] @ T[i] (data.py:31-35)."""
    poses = np.empty((len(transforms) + 1, 4, 4))
    poses[0] = np.eye(4)
    for i, t in enumerate(transforms):
        poses[i + 1] = poses[i] @ t
    return poses


def travelled(transforms: np.ndarray) -> np.ndarray:
    """Cumulative path length (n + 1,) from the translation norms of the steps (data.py:37-39)."""
    steps = np.linalg.norm(np.asarray(transforms, dtype=np.float64).reshape(-1, 4, 4)[:, :3, 3], axis=1)
    return np.concatenate(([0.0], np.cumsum(steps)))


class Sequence:
    """Predicted and ground-truth relative transforms of one scan sequence with stamps and inference times."""

    def __init__(self) -> None:
        self.stamps: List[float] = []
        self.times: List[float] = []
        self._pred: List[np.ndarray] = []
        self._gt: List[np.ndarray] = []

    def __len__(self) -> int:
        return len(self.stamps)

    def add_transforms(self, stamp: float, pred: np.ndarray, gt: np.ndarray, time: float = 0.0) -> None:
        self.stamps.append(float(stamp))
        self._pred.append(np.asarray(pred, dtype=np.float64).reshape(4, 4))
        self._gt.append(np.asarray(gt, dtype=np.float64).reshape(4, 4))
        self.times.append(float(time))

    @property
    def prediction(self) -> np.ndarray:
        return np.stack(self._pred) if self._pred else np.zeros((0, 4, 4))

    @property
    def ground_truth(self) -> np.ndarray:
        return np.stack(self._gt) if self._gt else np.zeros((0, 4, 4))

    def table(self) -> np.ndarray:
        """(n, 26) array in file column order."""
        n = len(self)
        out = np.empty((n, 26))
        out[:, 0] = self.stamps
        out[:, 1:13] = self.prediction[:, :3, :].reshape(n, 12)
        out[:, 13:25] = self.ground_truth[:, :3, :].reshape(n, 12)
        out[:, 25] = self.times
        return out

    def write(self, filename: str) -> None:
        np.savetxt(filename, self.table())

    @classmethod
    def from_table(cls, data: np.ndarray) -> 'Sequence':
        data = np.atleast_2d(np.asarray(data, dtype=np.float64))
        if data.shape[1] != 26:
            raise RuntimeError("a result row has 26 columns: stamp, 12 predicted, 12 ground truth, time")
        seq = cls()
        pred, gt = _as_4x4(data[:, 1:13]), _as_4x4(data[:, 13:25])
        for i in range(data.shape[0]):
            seq.add_transforms(data[i, 0], pred[i], gt[i], data[i, 25])
        return seq

    @classmethod
    def read(cls, filename: str) -> 'Sequence':
        return cls.from_table(np.loadtxt(filename))


def _angle(diff: np.ndarray) -> np.ndarray:
    """Rotation angle of (n, 4, 4) transforms from the trace, clamped as the KITTI devkit does (metrics.py:30-36)."""
    d = 0.5 * (diff[:, 0, 0] + diff[:, 1, 1] + diff[:, 2, 2] - 1.0)
    return np.arccos(np.clip(d, -1.0, 1.0))


def kitti_errors(a: np.ndarray, b: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Translation [m] and rotation [rad] error between transforms a and b, (n, 4, 4) each: the error transform is
    evaluated in both orders and the smaller value kept, separately per quantity (metrics.py:16-20, 45-49)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4, 4)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4, 4)
    ab = a @ np.linalg.inv(b)
    ba = b @ np.linalg.inv(a)
    trans = np.minimum(np.linalg.norm(ab[:, :3, 3], axis=1), np.linalg.norm(ba[:, :3, 3], axis=1))
    rot = np.minimum(_angle(ab), _angle(ba))
    return trans, rot


def chordal_error(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """metrics.py:59-64 (including its double division by sqrt(8))."""
    diff = np.asarray(a)[..., :3, :3] - np.asarray(b)[..., :3, :3]
    fro = np.sqrt((diff ** 2).sum(axis=(-2, -1))) / np.sqrt(8)
    return 2 * np.arcsin(fro / np.sqrt(8))


def step_errors(seq: Sequence) -> Dict[str, np.ndarray]:
    """Per-pair errors (evaluator.py:22-28): translation [m], rotation [rad], rmse translation, chordal, time [ms]."""
    pred, gt = seq.prediction, seq.ground_truth
    trans, rot = kitti_errors(pred, gt)
    rmse = np.sqrt(((pred[:, :3, 3] - gt[:, :3, 3]) ** 2).sum(axis=1) / 3.0)
    return {'translation': trans, 'rotation': rot, 'translation_rmse': rmse,
            'rotation_chordal': chordal_error(pred, gt), 'time': np.asarray(seq.times, dtype=np.float64)}


def segment_errors(seq: Sequence, step_size: int = STEP_SIZE,
                   segment_lengths: Iterable[float] = SEGMENT_LENGTHS) -> Dict[str, np.ndarray]:
    """KITTI segment errors (evaluator.py:31-65): for every `step_size`-th start frame and every segment length, the
    first later frame whose ground-truth path length exceeds start + length closes the segment; both errors are
    divided by the segment length ([m/m], [rad/m]); speed = length / (0.1 s x frames)."""
    p_pred, p_gt = chain_poses(seq.prediction), chain_poses(seq.ground_truth)
    dist = travelled(seq.ground_truth)
    first, last, length = [], [], []
    for f in range(0, len(dist), step_size):
        for seg in segment_lengths:
            beyond = np.nonzero(dist[f:] > dist[f] + seg)[0]
            if beyond.size:
                first.append(f)
                last.append(f + int(beyond[0]))
                length.append(float(seg))
    first_a, last_a, length_a = np.asarray(first, dtype=int), np.asarray(last, dtype=int), np.asarray(length)
    if not first:
        empty = np.zeros(0)
        return {'translation': empty, 'rotation': empty, 'first_frame': first_a, 'segment_length': length_a,
                'speed': empty}
    d_pred = np.linalg.inv(p_pred[first_a]) @ p_pred[last_a]
    d_gt = np.linalg.inv(p_gt[first_a]) @ p_gt[last_a]
    trans, rot = kitti_errors(d_pred, d_gt)
    return {'translation': trans / length_a, 'rotation': rot / length_a, 'first_frame': first_a,
            'segment_length': length_a, 'speed': length_a / (0.1 * (last_a - first_a + 1))}


class Evaluator:
    """Collects transforms per sequence name, writes / reads the per-sequence result files, reports errors."""

    def __init__(self) -> None:
        self._sequences: 'OrderedDict[str, Sequence]' = OrderedDict()

    def reset(self) -> None:
        self._sequences.clear()

    def add_transforms(self, name: str, stamp: float, pred: Optional[np.ndarray], gt: np.ndarray,
                       time: float = 0.0) -> None:
        """pred None (first frame of a sequential run, scripts/inference.py:113-117) is skipped."""
        if pred is None:
            return
        self._sequences.setdefault(name, Sequence()).add_transforms(stamp, pred, gt, time)

    def has_sequence(self, name: str) -> bool:
        return name in self._sequences

    def get_sequence(self, name: str) -> Sequence:
        return self._sequences[name]

    def get_sequences(self) -> 'OrderedDict[str, Sequence]':
        return self._sequences

    def write(self, path: str) -> None:
        for name, seq in self._sequences.items():
            seq.write(os.path.join(path, name + '.txt'))

    @classmethod
    def read(cls, path: str, filenames: Optional[List[str]] = None) -> 'Evaluator':
        if filenames is None:
            filenames = sorted(f for f in os.listdir(path)
                               if f.endswith('.txt') and os.path.isfile(os.path.join(path, f)))
        ev = cls()
        for f in filenames:
            ev._sequences[os.path.splitext(f)[0]] = Sequence.read(os.path.join(path, f))
        return ev

    def get_step_errors(self) -> 'OrderedDict[str, Dict[str, np.ndarray]]':
        return OrderedDict((n, step_errors(s)) for n, s in self._sequences.items())

    def get_segment_errors(self) -> 'OrderedDict[str, Dict[str, np.ndarray]]':
        return OrderedDict((n, segment_errors(s)) for n, s in self._sequences.items())

    def summary(self) -> Dict[str, float]:
        """Means over all sequences: the numbers `scripts/evaluation.py:37-80` tabulates (step errors, time mean,
        KITTI translation [%] and rotation [deg/m])."""
        steps, segs = list(self.get_step_errors().values()), list(self.get_segment_errors().values())

        def mean(parts, key, scale=1.0):
            arr = np.concatenate([p[key] for p in parts]) if parts else np.zeros(0)
            return float(arr.mean() * scale) if arr.size else float('nan')
        return {'step_translation_mean [m]': mean(steps, 'translation'),
                'step_rotation_mean [deg]': mean(steps, 'rotation', 180.0 / np.pi),
                'time_mean [ms]': mean(steps, 'time'),
                'kitti_translation [%]': mean(segs, 'translation', 100.0),
                'kitti_rotation [deg/m]': mean(segs, 'rotation', 180.0 / np.pi)}


class DatasetType(ConfigEnum):
    """Dataset kinds a scenario may name (/root/reference/deepclr/data/datasets/build.py:13-17)."""
    GENERIC = 1
    KITTI_ODOMETRY_VELODYNE = 2
    MODELNET40 = 3


def load_scenario(filename: str, with_method: bool = False) -> Config:
    """Scenario file of scripts/inference.py (/root/reference/deepclr/evaluation/scenario.py:6-33): `name`,
    `dataset_type`, `sequential`, `data` {sequence name: path, environment variables expanded}, optional `method`
    {name, params}."""
    with open(filename, 'r') as stream:
        cfg = Config.from_dict(yaml.safe_load(stream) or {})
    missing = [k for k in ('name', 'dataset_type', 'sequential', 'data') if cfg.get(k) is None]
    method = cfg.get('method') or Config()
    if with_method and method.get('name') is None:
        missing.append('method.name')
    if missing:
        raise RuntimeError("Configuration is not valid, missing required parameters.")
    cfg.method = Config.from_dict({'name': method.get('name'), 'params': dict(method.get('params') or {})})
    cfg.dataset_type = DatasetType.create(cfg.dataset_type)
    for name, path in cfg.data.items():
        full = os.path.realpath(os.path.expandvars(os.path.expanduser(path)))
        if '%' in full or '$' in full:
            raise RuntimeError("Could not replace a variable in path '{}'".format(full))
        cfg.data[name] = full
    return cfg
