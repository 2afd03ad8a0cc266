"""Pose label types and the label -> 4x4 conversion the metric is defined on.

Host-side mirror of ``LabelType`` (/root/reference/deepclr/data/labels.py:10-101):
``dim`` (16-24), ``names`` (26-34), ``bias`` (36-44), ``_dqnormalize`` (46-51),
``to_matrix`` (78-101) and ``from_matrix`` (53-76) for the quaternion label
types. The reference delegates quaternion algebra to transforms3d==0.3.1, which
is not in this image; the few closed-form helpers needed are written out here
in numpy float64 -- including, since round 6, the static-xyz euler pair of
POSE3D_EULER (labels.py:54-58, 82-86; degrees in the label): **parity
unpinned** (no transforms3d here, no fixture in the reference tree), checked
by recomposition and against the dual-quaternion branch in tests/test_host.py.
`affines.decompose` (which also strips scale and shear) is the identity split
on the rigid transforms this path produces and is restated as such.
"""
from enum import auto
from typing import List, Optional, Tuple

import numpy as np

from .config import ConfigEnum

_EPS64 = np.finfo(np.float64).eps


def _quat2mat(q: np.ndarray) -> np.ndarray:
    w, x, y, z = (float(v) for v in q)
    n = w * w + x * x + y * y + z * z
    if n < _EPS64:
        return np.eye(3)
    s = 2.0 / n
    xs, ys, zs = x * s, y * s, z * s
    return np.array([[1.0 - (y * ys + z * zs), x * ys - w * zs, x * zs + w * ys],
                     [x * ys + w * zs, 1.0 - (x * xs + z * zs), y * zs - w * xs],
                     [x * zs - w * ys, y * zs + w * xs, 1.0 - (x * xs + y * ys)]])


def _mat2quat(m: np.ndarray) -> np.ndarray:
    """Rotation matrix -> unit quaternion (w >= 0), via the symmetric 4x4 eigen form."""
    qxx, qyx, qzx, qxy, qyy, qzy, qxz, qyz, qzz = np.asarray(m, dtype=np.float64).flat
    k = np.array([[qxx - qyy - qzz, 0, 0, 0],
                  [qyx + qxy, qyy - qxx - qzz, 0, 0],
                  [qzx + qxz, qzy + qyz, qzz - qxx - qyy, 0],
                  [qyz - qzy, qzx - qxz, qxy - qyx, qxx + qyy + qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(k)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


def _qmult(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                     w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                     w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2])


def _euler2mat_sxyz(roll: float, pitch: float, yaw: float) -> np.ndarray:
    """transforms3d.euler.euler2mat(roll, pitch, yaw, axes='sxyz'): rotations about the STATIC x, then y, then z axes,
    R = Rz(yaw) Ry(pitch) Rx(roll); radians."""
    ci, si, cj, sj, ck, sk = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    return np.array([[cj * ck, si * sj * ck - ci * sk, ci * sj * ck + si * sk],
                     [cj * sk, si * sj * sk + ci * ck, ci * sj * sk - si * ck],
                     [-sj, si * cj, ci * cj]])


def _mat2euler_sxyz(r: np.ndarray) -> Tuple[float, float, float]:
    """transforms3d.euler.mat2euler(R, axes='sxyz'): (roll, pitch, yaw) in radians; at the pitch = +-90 deg singularity yaw
    is 0 and roll takes the rest, as that library does (deepclr_amd/evaluation.py euler_sxyz is the batched twin)."""
    r = np.asarray(r, dtype=np.float64)
    cy = np.hypot(r[0, 0], r[1, 0])
    if cy > 4.0 * _EPS64:
        return float(np.arctan2(r[2, 1], r[2, 2])), float(np.arctan2(-r[2, 0], cy)), float(np.arctan2(r[1, 0], r[0, 0]))
    return float(np.arctan2(-r[1, 2], r[1, 1])), float(np.arctan2(-r[2, 0], cy)), 0.0


class LabelType(ConfigEnum):
    POSE3D_EULER = auto()
    POSE3D_QUAT = auto()
    POSE3D_DUAL_QUAT = auto()

    @property
    def dim(self) -> int:
        return {LabelType.POSE3D_EULER: 6, LabelType.POSE3D_QUAT: 7, LabelType.POSE3D_DUAL_QUAT: 8}[self]

    @property
    def names(self) -> List[str]:
        if self == LabelType.POSE3D_EULER:
            return ['x', 'y', 'z', 'roll', 'pitch', 'yaw']
        if self == LabelType.POSE3D_QUAT:
            return ['pos_x', 'pos_y', 'pos_z', 'rot_w', 'rot_x', 'rot_y', 'rot_z']
        return ['real_w', 'real_x', 'real_y', 'real_z', 'dual_w', 'dual_x', 'dual_y', 'dual_z']

    @property
    def bias(self) -> Optional[List[float]]:
        if self == LabelType.POSE3D_EULER:
            return None
        if self == LabelType.POSE3D_QUAT:
            return [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
        return [1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]

    @staticmethod
    def _dqnormalize(real: np.ndarray, dual: np.ndarray, eps: float = 1e-8) -> Tuple[np.ndarray, np.ndarray]:
        norm = np.sqrt(np.dot(real, real)) + eps
        return real / norm, dual / norm

    def to_matrix(self, label: np.ndarray, scale: Optional[float] = None) -> np.ndarray:
        label = np.asarray(label, dtype=np.float64)
        if scale is not None:
            label = label / scale
        if self == LabelType.POSE3D_EULER:                   # reference labels.py:82-86: angles in degrees
            m = np.eye(4)
            m[:3, :3] = _euler2mat_sxyz(*np.deg2rad(label[3:6]))
            m[:3, 3] = label[:3]
            return m
        if self == LabelType.POSE3D_QUAT:
            m = np.eye(4)
            m[:3, :3] = _quat2mat(label[3:])
            m[:3, 3] = label[:3]
            return m
        if self == LabelType.POSE3D_DUAL_QUAT:
            real, dual = self._dqnormalize(label[:4], label[4:])
            m = np.eye(4)
            m[:3, :3] = _quat2mat(real)
            conj = real * np.array([1.0, -1.0, -1.0, -1.0])
            m[:3, 3] = (2.0 * _qmult(dual, conj))[1:]
            return m
        raise NotImplementedError("LabelType '{}' not implemented".format(self))

    def from_matrix(self, data: np.ndarray, scale: Optional[float] = None) -> np.ndarray:
        data = np.asarray(data, dtype=np.float64)
        t, r = data[:3, 3], data[:3, :3]
        if self == LabelType.POSE3D_EULER:                   # reference labels.py:54-58
            roll, pitch, yaw = _mat2euler_sxyz(r)
            label = np.array([t[0], t[1], t[2], np.rad2deg(roll), np.rad2deg(pitch), np.rad2deg(yaw)])
        elif self == LabelType.POSE3D_QUAT:
            q = _mat2quat(r)
            label = np.array([t[0], t[1], t[2], q[0], q[1], q[2], q[3]])
        elif self == LabelType.POSE3D_DUAL_QUAT:
            real = _mat2quat(r)
            dual = 0.5 * _qmult(np.array([0.0, t[0], t[1], t[2]]), real)
            label = np.concatenate((real, dual))
        else:
            raise NotImplementedError("LabelType '{}' not implemented".format(self))
        if scale is not None:
            label = label * scale
        return label
