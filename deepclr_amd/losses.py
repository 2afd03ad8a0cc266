"""Pose losses of the reference, forward values only (SURVEY.md section 8f row 4 lists them with the training ops).

`forward(x, y=labels)` of a model configured with a loss returns its value, as the reference does
(/root/reference/deepclr/models/deepclr.py:500-506); used for validation figures. Plain torch on the GPU tensors the
kernels return: these are (B, 8)-sized reductions, not a hot path. No backward through the HIP kernels exists.

Formulas: /root/reference/deepclr/utils/metrics.py:41-52 (normalisation by the norm of the rotation quaternion,
+eps), 55-73 (translation part), 129-151 (rotation part); weighting: deepclr.py:352-389.
"""
from typing import Optional

import torch

from .labels import LabelType


def _normalize(x: torch.Tensor, label_type: LabelType, eps: float) -> torch.Tensor:
    if label_type == LabelType.POSE3D_QUAT:
        return torch.cat((x[:, :3], x[:, 3:] / (torch.norm(x[:, 3:], p=2, dim=1, keepdim=True) + eps)), dim=1)
    if label_type == LabelType.POSE3D_DUAL_QUAT:
        return x / (torch.norm(x[:, :4], p=2, dim=1, keepdim=True) + eps)
    raise RuntimeError("Unsupported label type for normalization")


def _reduce(x: torch.Tensor, reduction: Optional[str]) -> torch.Tensor:
    if reduction in (None, 'none'):
        return x
    if reduction == 'mean':
        return x.mean()
    if reduction == 'sum':
        return x.sum()
    raise RuntimeError(f"Unsupported reduction '{reduction}'")


def trans_loss(source: torch.Tensor, target: torch.Tensor, label_type: LabelType, p: int = 2,
               reduction: Optional[str] = 'mean', eps: float = 1e-8) -> torch.Tensor:
    """p-norm between the translation parts: columns 0..2 of euler / quaternion labels, the dual part (columns 4..7)
    of normalised dual quaternions."""
    if label_type in (LabelType.POSE3D_EULER, LabelType.POSE3D_QUAT):
        a, b = source[:, :3], target[:, :3]
    elif label_type == LabelType.POSE3D_DUAL_QUAT:
        a, b = _normalize(source, label_type, eps)[:, 4:], _normalize(target, label_type, eps)[:, 4:]
    else:
        raise RuntimeError("Unsupported label type for this loss type.")
    return _reduce(torch.norm(a - b, dim=1, p=p, keepdim=True), reduction)


def rot_loss(source: torch.Tensor, target: torch.Tensor, label_type: LabelType, p: int = 2,
             reduction: Optional[str] = 'mean', eps: float = 1e-8) -> torch.Tensor:
    """p-norm between the rotation parts: euler angles, or the (normalised) rotation quaternion."""
    if label_type == LabelType.POSE3D_EULER:
        a, b = source[:, 3:], target[:, 3:]
    elif label_type == LabelType.POSE3D_QUAT:
        a, b = _normalize(source, label_type, eps)[:, 3:], _normalize(target, label_type, eps)[:, 3:]
    elif label_type == LabelType.POSE3D_DUAL_QUAT:
        a, b = _normalize(source, label_type, eps)[:, :4], _normalize(target, label_type, eps)[:, :4]
    else:
        raise RuntimeError("Unsupported label type for this loss type")
    return _reduce(torch.norm(a - b, dim=1, p=p, keepdim=True), reduction)


def transform_losses(y_pred: torch.Tensor, y: torch.Tensor, label_type: LabelType, p: int):
    """(mean translation loss, mean rotation loss) over the batch; raises on nan / inf like the reference
    (deepclr.py:311-325)."""
    t, r = trans_loss(y_pred, y, label_type, p=p, reduction='mean'), rot_loss(y_pred, y, label_type, p=p, reduction='mean')
    for name, v in (('translation', t), ('rotation', r)):
        if torch.isnan(v) or torch.isinf(v):
            raise RuntimeError("TransformLoss: {} loss is nan or inf:\ny_pred = \n{}\ny = \n{}".format(name, y_pred, y))
    return t, r
