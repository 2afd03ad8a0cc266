"""Result files and odometry error metrics for the poses the hot path produces.

File format = the reference's (`/root/reference/deepclr/evaluation/data.py:104-137`): one text row per scan pair,
26 columns `[stamp, predicted 3x4 row-major (12), ground-truth 3x4 row-major (12), inference time in ms]`, written
with `numpy.savetxt` defaults, one file `<sequence name>.txt` per sequence (`evaluation/evaluator.py:142-144`), so
files written here load in the reference's `Evaluator.read` and the other way round.

Metrics = the KITTI odometry errors the reference reports (`evaluation/metrics.py:8-47`,
`evaluation/evaluator.py:22-65`): per pair, and over every 100..800 m segment starting at every 10th frame,
normalised by the segment length. Arrays instead of per-element objects; `MetricsContainer` presents them with the
reference's attribute layout (`errors.mean.translation.kitti`, `errors.std.time`, iteration over per-item records) so
`scripts/evaluation.py:37-140` and the `scripts/paper/*_table.py` readers run on it unchanged. Figures: `plots.py`.

Euler angles: the reference takes them from transforms3d (`metrics.py:38-40,52-56`), which this image lacks;
`euler_sxyz` restates the published static-xyz decomposition and is checked by recomposition only, so the `vec`
fields and the per-pair `rotation.rmse` are PARITY UNPINNED; every other field is pinned by
tests/golden/eval_expected.npz, written by the reference's own classes.
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
    transforms = np.asarray(transforms, dtype=np.float64).reshape(-1, 4, 4)
    poses = np.empty((len(transforms) + 1, 4, 4))
    poses[0] = np.eye(4)
    for i, t in enumerate(transforms):
        poses[i + 1] = poses[i] @ t
    return poses


def travelled(transforms: np.ndarray) -> np.ndarray:
    """Cumulative path length (n + 1,) from the translation norms of the steps (data.py:37-39)."""
    steps = np.linalg.norm(np.asarray(transforms, dtype=np.float64).reshape(-1, 4, 4)[:, :3, 3], axis=1)
    return np.concatenate(([0.0], np.cumsum(steps)))


class Motion:
    """Relative transforms (n, 4, 4) with the poses and path lengths they chain to (data.py:16-58). Converts to the
    transform array under `numpy.asarray`."""

    def __init__(self, transforms: Optional[np.ndarray] = None) -> None:
        self._rows: List[np.ndarray] = [] if transforms is None else [
            np.asarray(t, dtype=np.float64).reshape(4, 4) for t in transforms]
        self._cache: Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]] = None

    def __len__(self) -> int:
        return len(self._rows)

    def __array__(self, dtype=None, copy=None):
        return self.transforms if dtype is None else self.transforms.astype(dtype)

    def add_transform(self, m: np.ndarray) -> None:
        self._rows.append(np.asarray(m, dtype=np.float64).reshape(4, 4))
        self._cache = None

    def _derived(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        if self._cache is None:
            t = np.stack(self._rows) if self._rows else np.zeros((0, 4, 4))
            self._cache = (t, chain_poses(t), travelled(t))
        return self._cache

    @property
    def transforms(self) -> np.ndarray:
        return self._derived()[0]

    @property
    def poses(self) -> np.ndarray:
        """(n + 1, 4, 4), the first one the identity."""
        return self._derived()[1]

    @property
    def distances(self) -> np.ndarray:
        return self._derived()[2]

    def get_path(self) -> np.ndarray:
        return self.poses[:, :3, 3]

    def get_frame_by_distance(self, first_frame: int, distance: float) -> int:
        """First frame whose path length exceeds that of `first_frame` by more than `distance`, else -1."""
        d = self.distances
        beyond = np.nonzero(d[first_frame:] > d[first_frame] + distance)[0]
        return first_frame + int(beyond[0]) if beyond.size else -1

    def write(self, filename: str, use_poses: bool) -> None:
        """12 columns per row, the chained poses or the relative transforms (data.py:72-77)."""
        rows = self.poses if use_poses else self.transforms
        np.savetxt(filename, rows[:, :3, :].reshape(len(rows), 12))


class Sequence:
    """Predicted and ground-truth relative transforms of one scan sequence with stamps and inference times."""

    def __init__(self) -> None:
        self.stamps: List[float] = []
        self.times: List[float] = []
        self.prediction = Motion()
        self.ground_truth = Motion()

    def __len__(self) -> int:
        return len(self.stamps)

    def add_transforms(self, stamp: float, pred: np.ndarray, gt: np.ndarray, time: float = 0.0) -> None:
        self.stamps.append(float(stamp))
        self.prediction.add_transform(pred)
        self.ground_truth.add_transform(gt)
        self.times.append(float(time))

    def table(self) -> np.ndarray:
        """(n, 26) array in file column order."""
        n = len(self)
        out = np.empty((n, 26))
        out[:, 0] = self.stamps
        out[:, 1:13] = self.prediction.transforms[:, :3, :].reshape(n, 12)
        out[:, 13:25] = self.ground_truth.transforms[:, :3, :].reshape(n, 12)
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
        seq.stamps = [float(v) for v in data[:, 0]]
        seq.times = [float(v) for v in data[:, 25]]
        seq.prediction = Motion(_as_4x4(data[:, 1:13]))
        seq.ground_truth = Motion(_as_4x4(data[:, 13:25]))
        return seq

    @classmethod
    def read(cls, filename: str) -> 'Sequence':
        return cls.from_table(np.loadtxt(filename))


def _angle(diff: np.ndarray) -> np.ndarray:
    """Rotation angle of (n, 4, 4) transforms from the trace, clamped as the KITTI devkit does (metrics.py:30-36)."""
    d = 0.5 * (diff[:, 0, 0] + diff[:, 1, 1] + diff[:, 2, 2] - 1.0)
    return np.arccos(np.clip(d, -1.0, 1.0))


def euler_sxyz(rot: np.ndarray) -> np.ndarray:
    """(n, 3, 3) rotation matrices -> (n, 3) roll, pitch, yaw about the static x, y, z axes (R = Rz Ry Rx), the
    convention the reference asks transforms3d for (`mat2euler(R, axes='sxyz')`, metrics.py:39,54-55). Near the
    pitch = +-90 deg singularity yaw is set to 0 and roll absorbs the rest, as that library does."""
    r = np.asarray(rot, dtype=np.float64).reshape(-1, 3, 3)
    cy = np.hypot(r[:, 0, 0], r[:, 1, 0])
    regular = cy > 4.0 * np.finfo(np.float64).eps
    roll = np.where(regular, np.arctan2(r[:, 2, 1], r[:, 2, 2]), np.arctan2(-r[:, 1, 2], r[:, 1, 1]))
    pitch = np.arctan2(-r[:, 2, 0], cy)
    yaw = np.where(regular, np.arctan2(r[:, 1, 0], r[:, 0, 0]), 0.0)
    return np.stack([roll, pitch, yaw], axis=1)


def _kitti_full(a: np.ndarray, b: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Translation / rotation error and their vectors: the error transform is evaluated in both orders and the
    smaller value kept with its vector, separately per quantity (metrics.py:16-20, 45-49). The rotation vector is
    the euler angles of the error transform's rotation block (the reference first strips scale and shear with
    `affines.decompose`, the identity on rigid transforms)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4, 4)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4, 4)
    ab = a @ np.linalg.inv(b)
    ba = b @ np.linalg.inv(a)
    t_ab, t_ba = np.linalg.norm(ab[:, :3, 3], axis=1), np.linalg.norm(ba[:, :3, 3], axis=1)
    first = t_ab < t_ba
    trans = np.where(first, t_ab, t_ba)
    trans_vec = np.where(first[:, None], ab[:, :3, 3], ba[:, :3, 3])
    r_ab, r_ba = _angle(ab), _angle(ba)
    first = r_ab < r_ba
    rot = np.where(first, r_ab, r_ba)
    rot_vec = euler_sxyz(np.where(first[:, None, None], ab[:, :3, :3], ba[:, :3, :3]))
    return trans, trans_vec, rot, rot_vec


def kitti_errors(a: np.ndarray, b: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Translation [m] and rotation [rad] error between transforms a and b, (n, 4, 4) each."""
    trans, _, rot, _ = _kitti_full(a, b)
    return trans, rot


def chordal_error(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """metrics.py:59-64 (including its double division by sqrt(8))."""
    diff = np.asarray(a)[..., :3, :3] - np.asarray(b)[..., :3, :3]
    fro = np.sqrt((diff ** 2).sum(axis=(-2, -1))) / np.sqrt(8)
    return 2 * np.arcsin(fro / np.sqrt(8))


def euler_rmse(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """RMS difference of the two transforms' euler angles (metrics.py:52-56)."""
    d = euler_sxyz(np.asarray(a)[..., :3, :3]) - euler_sxyz(np.asarray(b)[..., :3, :3])
    return np.sqrt((d ** 2).sum(axis=1) / 3.0)


def step_errors(seq: Sequence) -> Dict[str, np.ndarray]:
    """Per-pair errors (evaluator.py:22-28): translation [m], rotation [rad], rmse translation, chordal and euler
    rmse rotation, the error vectors, time [ms]."""
    pred, gt = seq.prediction.transforms, seq.ground_truth.transforms
    trans, trans_vec, rot, rot_vec = _kitti_full(pred, gt)
    rmse = np.sqrt(((pred[:, :3, 3] - gt[:, :3, 3]) ** 2).sum(axis=1) / 3.0)
    return {'translation': trans, 'rotation': rot, 'translation_rmse': rmse, 'translation_vec': trans_vec,
            'rotation_rmse': euler_rmse(pred, gt), 'rotation_chordal': chordal_error(pred, gt),
            'rotation_vec': rot_vec, 'time': np.asarray(seq.times, dtype=np.float64)}


def segment_errors(seq: Sequence, step_size: int = STEP_SIZE,
                   segment_lengths: Iterable[float] = SEGMENT_LENGTHS) -> Dict[str, np.ndarray]:
    """KITTI segment errors (evaluator.py:31-65): for every `step_size`-th start frame and every segment length, the
    first later frame whose ground-truth path length exceeds start + length closes the segment; both errors are
    divided by the segment length ([m/m], [rad/m]); speed = length / (0.1 s x frames).

    The secondary fields follow the reference's `divide` to the letter (metrics.py:80-83, 104-108): after the KITTI
    value is divided by the length, `translation_rmse`, `rotation_rmse` and `rotation_chordal` are set to that
    quotient divided by the length AGAIN (kitti / length^2), not to their own metric; the segment tables of
    `scripts/evaluation.py` print them, so they are reproduced as they are."""
    p_pred, p_gt = seq.prediction.poses, seq.ground_truth.poses
    dist = seq.ground_truth.distances
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
        e1, e3 = np.zeros(0), np.zeros((0, 3))
        return {'translation': e1, 'rotation': e1, 'translation_rmse': e1, 'translation_vec': e3,
                'rotation_rmse': e1, 'rotation_chordal': e1, 'rotation_vec': e3,
                'first_frame': first_a, 'segment_length': length_a, 'speed': e1}
    d_pred = np.linalg.inv(p_pred[first_a]) @ p_pred[last_a]
    d_gt = np.linalg.inv(p_gt[first_a]) @ p_gt[last_a]
    trans, trans_vec, rot, rot_vec = _kitti_full(d_pred, d_gt)
    trans, rot = trans / length_a, rot / length_a
    return {'translation': trans, 'rotation': rot, 'translation_rmse': trans / length_a,
            'translation_vec': trans_vec / length_a[:, None], 'rotation_rmse': rot / length_a,
            'rotation_chordal': rot / length_a, 'rotation_vec': rot_vec / length_a[:, None],
            'first_frame': first_a, 'segment_length': length_a,
            'speed': length_a / (0.1 * (last_a - first_a + 1))}


class _Record:
    """Plain attribute bag (`x.translation.kitti`, `x.time`, `x.segment_length` ...)."""

    def __init__(self, **fields) -> None:
        self.__dict__.update(fields)

    def __repr__(self) -> str:
        return 'Record({})'.format(', '.join('{}={!r}'.format(k, v) for k, v in self.__dict__.items()))


_TRANSLATION = (('kitti', 'translation'), ('rmse', 'translation_rmse'), ('vec', 'translation_vec'))
_ROTATION = (('kitti', 'rotation'), ('rmse', 'rotation_rmse'), ('chordal', 'rotation_chordal'),
             ('vec', 'rotation_vec'))


def _stat(func, col: np.ndarray):
    """Column statistic; NaN (of the row shape) for no items, where the reference raises on the empty array."""
    if len(col) == 0:
        return np.full(col.shape[1:], np.nan) if col.ndim > 1 else float('nan')
    return func(col, axis=0)


class MetricsContainer:
    """The errors of a run as the reference hands them to its scripts (metrics.py:158-203): `min`, `max`, `mean`,
    `median`, `std` records with `.translation.{kitti,rmse,vec}`, `.rotation.{kitti,rmse,chordal,vec}` and `.time`
    (0 for segment errors); `len`, indexing and iteration give the per-item records (with `first_frame`,
    `segment_length`, `speed` on segment errors). `arrays` holds the columns the records are cut from."""

    def __init__(self, arrays: Dict[str, np.ndarray]) -> None:
        self.arrays = dict(arrays)
        self.is_segments = 'segment_length' in self.arrays
        n = len(self.arrays['translation'])
        if 'time' not in self.arrays:
            self.arrays['time'] = np.zeros(n)
        for name, func in (('min', np.min), ('max', np.max), ('mean', np.mean), ('median', np.median),
                           ('std', np.std)):
            setattr(self, name, self._record(lambda col, f=func: _stat(f, col)))

    def _record(self, pick) -> _Record:
        rec = _Record(translation=_Record(**{k: pick(self.arrays[col]) for k, col in _TRANSLATION}),
                      rotation=_Record(**{k: pick(self.arrays[col]) for k, col in _ROTATION}))
        if self.is_segments:
            rec.__dict__.update(first_frame=pick(self.arrays['first_frame']),
                                segment_length=pick(self.arrays['segment_length']), speed=pick(self.arrays['speed']))
        rec.time = pick(self.arrays['time'])
        return rec

    @classmethod
    def merge(cls, parts: Iterable['MetricsContainer']) -> 'MetricsContainer':
        """All items of several containers in order (evaluator.py:68-70)."""
        parts = list(parts)
        if not parts:
            raise ValueError("nothing to merge")
        return cls({k: np.concatenate([p.arrays[k] for p in parts]) for k in parts[0].arrays})

    def __len__(self) -> int:
        return len(self.arrays['translation'])

    def __getitem__(self, i: int) -> _Record:
        if not -len(self) <= i < len(self):
            raise IndexError(i)
        return self._record(lambda col: col[i])

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Evaluator:
    """Collects transforms per sequence name, writes / reads the per-sequence result files, reports errors
    (evaluator.py:73-204)."""

    def __init__(self) -> None:
        self._sequences: 'OrderedDict[str, Sequence]' = OrderedDict()
        self._errors: Dict[str, object] = {}

    def reset(self) -> None:
        self._sequences.clear()
        self.reset_errors()

    def reset_errors(self) -> None:
        self._errors.clear()

    def add_transforms(self, name: str, stamp: float, pred: Optional[np.ndarray], gt: np.ndarray,
                       time: float = 0.0) -> None:
        """pred None (first frame of a sequential run, scripts/inference.py:113-117) is skipped."""
        if pred is None:
            return
        self._sequences.setdefault(name, Sequence()).add_transforms(stamp, pred, gt, time)
        self.reset_errors()

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

    def _cached(self, key: str, make):
        if key not in self._errors:
            self._errors[key] = make()
        return self._errors[key]

    def get_step_errors(self) -> 'OrderedDict[str, MetricsContainer]':
        return self._cached('step', lambda: OrderedDict(
            (n, MetricsContainer(step_errors(s))) for n, s in self._sequences.items()))

    def get_total_step_errors(self) -> MetricsContainer:
        return self._cached('step_total', lambda: MetricsContainer.merge(self.get_step_errors().values()))

    def get_segment_errors(self) -> 'OrderedDict[str, MetricsContainer]':
        return self._cached('segment', lambda: OrderedDict(
            (n, MetricsContainer(segment_errors(s))) for n, s in self._sequences.items()))

    def get_total_segment_errors(self) -> MetricsContainer:
        return self._cached('segment_total', lambda: MetricsContainer.merge(self.get_segment_errors().values()))

    def summary(self) -> Dict[str, float]:
        """Means over all sequences: the headline columns of `scripts/evaluation.py:37-80` (step errors, time mean,
        KITTI translation [%] and rotation [deg/m])."""
        step, seg = self.get_total_step_errors().mean, self.get_total_segment_errors().mean
        return {'step_translation_mean [m]': float(step.translation.kitti),
                'step_rotation_mean [deg]': float(np.rad2deg(step.rotation.kitti)),
                'time_mean [ms]': float(step.time),
                'kitti_translation [%]': float(seg.translation.kitti * 100.0),
                'kitti_rotation [deg/m]': float(np.rad2deg(seg.rotation.kitti))}

    # figures (evaluator.py:170-204); matplotlib is imported on first use
    def plot_error_over_time(self):
        from . import plots
        return OrderedDict((n, plots.plot_error_over_time(e)) for n, e in self.get_step_errors().items())

    def plot_kitti_errors(self):
        from . import plots
        return OrderedDict((n, plots.plot_kitti_errors(e)) for n, e in self.get_segment_errors().items())

    def plot_total_kitti_errors(self):
        from . import plots
        return plots.plot_kitti_errors(self.get_total_segment_errors())

    def plot_segment_error_bars(self):
        from . import plots
        return plots.plot_segment_error_bars(self.get_segment_errors())

    def plot_sequences(self):
        from . import plots
        return OrderedDict((n, plots.plot_sequence(s, title=n)) for n, s in self._sequences.items())

    def plot_sequences_2d(self):
        from . import plots
        return OrderedDict((n, plots.plot_sequence_2d(s, title=n)) for n, s in self._sequences.items())


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
