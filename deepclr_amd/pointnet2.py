"""Drop-in for the ``pointnet2`` package symbols DeepCLR imports.

The reference does ``from pointnet2 import PointnetSAModuleMSG``
(/root/reference/deepclr/models/deepclr.py:9) and constructs it with
``npoint, radii, nsamples, mlps, use_xyz=True, bn=batch_norm`` (deepclr.py:63-70).
That package (sshaoshuai/Pointnet2.PyTorch) is an un-vendored submodule; this module
offers the same constructor and ``forward(xyz, features) -> (new_xyz, new_features)``
contract and the same parameter names (``mlps.{scale}.layer{j}.conv.{weight,bias}``,
(out, in, 1, 1) weights, kaiming-normal init, zero bias) on top of the fused HIP
set-abstraction kernel. The level-1 functions are re-exported under their upstream names.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .ops import ball_query, furthest_point_sample, gather_operation, grouping_operation  # noqa: F401
from .models.helper import PackedCache

__all__ = ['PointnetSAModuleMSG', 'furthest_point_sample', 'gather_operation', 'ball_query',
           'grouping_operation']

_FUSED_MLP = [16, 16, 32]


class _ConvUnit(nn.Sequential):
    def __init__(self, c_in: int, c_out: int):
        super().__init__()
        conv = nn.Conv2d(c_in, c_out, kernel_size=(1, 1), bias=True)
        nn.init.kaiming_normal_(conv.weight)
        nn.init.constant_(conv.bias, 0)
        self.add_module('conv', conv)
        self.add_module('activation', nn.ReLU(inplace=True))


class _SharedMLP(nn.Sequential):
    def __init__(self, spec: List[int]):
        super().__init__()
        for j in range(len(spec) - 1):
            self.add_module('layer{}'.format(j), _ConvUnit(spec[j], spec[j + 1]))


class PointnetSAModuleMSG(nn.Module):
    """Set abstraction with multi-scale grouping: FPS -> per scale [ball query, shared MLP, max]."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = True, use_xyz: bool = True, pool_method: str = 'max_pool',
                 instance_norm: bool = False):
        super().__init__()
        if bn or instance_norm:
            raise NotImplementedError("normalisation layers are outside the MI355X hot path "
                                      "(DeepCLR passes bn=batch_norm=False)")
        if not use_xyz or pool_method != 'max_pool':
            raise NotImplementedError("DeepCLR uses use_xyz=True with max pooling")
        if not (len(radii) == len(nsamples) == len(mlps)) or not 1 <= len(radii) <= 2:
            raise NotImplementedError("the fused kernel handles one or two grouping scales")
        self.npoint = npoint
        self.radii = [float(r) for r in radii]
        self.nsamples = [int(s) for s in nsamples]
        self.mlps = nn.ModuleList()
        self._in_feat = None
        for spec in mlps:
            spec = list(spec)
            if spec[1:] != _FUSED_MLP:
                raise NotImplementedError("the fused kernel is built for mlp widths {} (got {})"
                                          .format(_FUSED_MLP, spec[1:]))
            self._in_feat = spec[0]
            spec[0] += 3
            self.mlps.append(_SharedMLP(spec))
        if self._in_feat not in (0, 1):
            raise NotImplementedError("fused set abstraction takes xyz or xyz + 1 feature per point")
        self._cache = PackedCache()

    def out_features(self) -> int:
        return 32 * len(self.mlps)

    def packed_mlps(self) -> List[torch.Tensor]:
        def build():
            return [ops.pack_sa_mlp([u.conv.weight for u in stack], [u.conv.bias for u in stack])
                    for stack in self.mlps]
        return self._cache.get(list(self.parameters()), build)

    def sample(self, clouds: torch.Tensor):
        """Furthest point sampling only (the serial stage; the pipelined runner issues it batches ahead
        on side streams): clouds (B, N, C) -> (idx (B, npoint) int32, group_pts, group_box); the group
        tensors are the sampling kernel's spatial partition, or None when it has none for this N."""
        if clouds.shape[2] != 3 + self._in_feat:
            raise RuntimeError("expected {} columns per point, got {}".format(3 + self._in_feat, clouds.shape[2]))
        return ops.fps_clouds_grouped(clouds, self.npoint)

    def forward_rows(self, clouds: torch.Tensor, sample=None) -> torch.Tensor:
        """clouds (B, N, 3 + in_feat) interleaved -> feature rows F (B*npoint, 68); sample = self.sample(clouds)."""
        if sample is None:
            sample = self.sample(clouds)
        idx, gpts, gbox = sample
        groups = None if gpts is None else (gpts, gbox)
        return ops.sa_msg_fused(clouds, idx, self.radii, self.nsamples, self.packed_mlps(), groups=groups)

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor] = None,
                new_xyz: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """xyz (B, N, 3), features (B, C, N) or None -> new_xyz (B, npoint, 3), new_features (B, 32*scales, npoint)."""
        if new_xyz is not None:
            raise NotImplementedError("caller-supplied centroids are not used by DeepCLR")
        clouds = xyz if features is None else torch.cat((xyz, features.transpose(1, 2)), dim=2)
        rows = self.forward_rows(clouds.contiguous())
        b = xyz.shape[0]
        ch = ops.rows_to_channels(rows, b, self.npoint, self.out_features())
        return ch[:, :3, :].transpose(1, 2).contiguous(), ch[:, 3:, :].contiguous()
