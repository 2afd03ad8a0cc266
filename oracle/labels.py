"""Dual-quaternion label -> 4x4 pose, restated -- TEST INFRASTRUCTURE ONLY.

Follows ``LabelType.to_matrix`` (POSE3D_DUAL_QUAT branch) and ``_dqnormalize``
(/root/reference/deepclr/data/labels.py:46-51, 93-99). The quaternion helpers it
calls live in transforms3d==0.3.1 (/root/reference/requirements.txt:27), absent
from this image -> restated from that library's published formulas
(``quat2mat``, ``qmult``, ``qconjugate``); PARITY UNPINNED for those three.
All arithmetic is numpy float64 on the host, as in the reference.
"""
import numpy as np

_FLOAT_EPS = np.finfo(np.float64).eps


def quat2mat(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q
    nq = w * w + x * x + y * y + z * z
    if nq < _FLOAT_EPS:
        return np.eye(3)
    s = 2.0 / nq
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    return np.array([[1.0 - (yy + zz), xy - wz, xz + wy],
                     [xy + wz, 1.0 - (xx + zz), yz - wx],
                     [xz - wy, yz + wx, 1.0 - (xx + yy)]])


def qmult(q1: np.ndarray, q2: np.ndarray) -> np.ndarray:
    w1, x1, y1, z1 = q1
    w2, x2, y2, z2 = q2
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                     w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                     w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2])


def qconjugate(q: np.ndarray) -> np.ndarray:
    return np.array(q) * np.array([1.0, -1.0, -1.0, -1.0])


def dual_quat_to_matrix(label: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """(8,) [real wxyz, dual wxyz] -> (4,4)."""
    label = np.asarray(label)
    real, dual = label[:4], label[4:]
    norm = np.sqrt(np.dot(real, real)) + eps
    real, dual = real / norm, dual / norm
    m = np.eye(4)
    m[:3, :3] = quat2mat(real)
    m[:3, 3] = (2.0 * qmult(dual, qconjugate(real)))[1:]
    return m
