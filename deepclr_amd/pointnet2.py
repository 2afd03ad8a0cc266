"""Drop-in for the ``pointnet2`` package symbols DeepCLR imports.

The reference does ``from pointnet2 import PointnetSAModuleMSG``
(/root/reference/deepclr/models/deepclr.py:9) and constructs it with
``npoint, radii, nsamples, mlps, use_xyz=True, bn=batch_norm`` (deepclr.py:63-70).
That package (sshaoshuai/Pointnet2.PyTorch) is an un-vendored submodule; this module
offers the same constructor and ``forward(xyz, features) -> (new_xyz, new_features)``
contract and the same parameter names (``mlps.{scale}.layer{j}.conv.{weight,bias}``,
(out, in, 1, 1) weights, kaiming-normal init, zero bias) on top of the fused HIP
set-abstraction kernel. The level-1 functions are re-exported under their upstream names.

Two execution paths. The shape every shipped model uses for its first level -- xyz (+ <= 1 feature) in, shared MLP
c -> 16 -> 16 -> 32 per scale -- runs in ONE fused kernel (csrc/sa.hip). Any other shape (a second level over 64
feature channels, other MLP widths: reference deepclr.py:72-83) is composed from the level-1 HIP operators the
reference itself composes (furthest_point_sample, gather_operation, ball_query, grouping_operation) plus the
f32-MFMA ``dclr_linear`` for the shared MLP with the max over nsample folded into its last layer; torch only
concatenates, pads and reshapes in between. That path is unfused and materialises the grouped tensor, as the
reference does; it exists for coverage, not for speed.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .ops import ball_query, furthest_point_sample  # noqa: F401  (index outputs: nothing to differentiate)
from .models.helper import PackedCache, flat_parameters, fold_batch_norm

__all__ = ['PointnetSAModuleMSG', 'furthest_point_sample', 'gather_operation', 'ball_query',
           'grouping_operation', 'GatherOperation', 'GroupingOperation']


class GatherOperation(torch.autograd.Function):
    """features (B, C, N), idx (B, npoint) int32 -> (B, C, npoint); backward scatters the gradient back
    (upstream pointnet2_utils.GatherOperation over gather_points_wrapper_fast / gather_points_grad_wrapper_fast,
    /root/reference/extern/pointnet2.patch:275-304)."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return ops.gather_operation(features, idx)

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        (idx,) = ctx.saved_tensors
        return ops.gather_operation_grad(grad_out.contiguous(), idx, ctx.n), None


class GroupingOperation(torch.autograd.Function):
    """features (B, C, N), idx (B, npoint, nsample) int32 -> (B, C, npoint, nsample); backward sums the gradient of
    every slot into its source point (upstream pointnet2_utils.GroupingOperation over group_points_wrapper_fast /
    group_points_grad_wrapper_fast, /root/reference/extern/pointnet2.patch:144-174)."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        return ops.grouping_operation(features, idx)

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        (idx,) = ctx.saved_tensors
        return ops.grouping_operation_grad(grad_out.contiguous(), idx, ctx.n), None


gather_operation = GatherOperation.apply
grouping_operation = GroupingOperation.apply

_FUSED_MLP = [16, 16, 32]


class _BatchNorm2d(nn.Sequential):
    """The published wrapper (pointnet2 pytorch_utils._BNBase): a module holding the norm layer under the name `bn`
    (state_dict keys `layerJ.bn.bn.*`), scale 1 and shift 0 at start."""

    def __init__(self, width: int):
        super().__init__()
        self.add_module('bn', nn.BatchNorm2d(width))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class _ConvUnit(nn.Sequential):
    """conv -> [batch norm] -> ReLU (published pytorch_utils._ConvBase: with batch norm the conv has no bias)."""

    def __init__(self, c_in: int, c_out: int, bn: bool = False):
        super().__init__()
        conv = nn.Conv2d(c_in, c_out, kernel_size=(1, 1), bias=not bn)
        nn.init.kaiming_normal_(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        self.add_module('conv', conv)
        if bn:
            self.add_module('bn', _BatchNorm2d(c_out))
        self.add_module('activation', nn.ReLU(inplace=True))

    def folded(self):
        """Weight and bias the kernels pack: eval-mode batch norm folded into the conv (models/helper.py)."""
        norm = self.bn.bn if hasattr(self, 'bn') else None
        return fold_batch_norm(self.conv.weight, self.conv.bias, norm)


class _SharedMLP(nn.Sequential):
    def __init__(self, spec: List[int], bn: bool = False):
        super().__init__()
        for j in range(len(spec) - 1):
            self.add_module('layer{}'.format(j), _ConvUnit(spec[j], spec[j + 1], bn))


class PointnetSAModuleMSG(nn.Module):
    """Set abstraction with multi-scale grouping: FPS -> per scale [ball query, shared MLP, max]."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = True, use_xyz: bool = True, pool_method: str = 'max_pool',
                 instance_norm: bool = False):
        super().__init__()
        if instance_norm:
            raise NotImplementedError("instance norm is not used by DeepCLR (it passes bn=batch_norm only, deepclr.py:63-70)")
        if not use_xyz or pool_method != 'max_pool':
            raise NotImplementedError("DeepCLR uses use_xyz=True with max pooling")
        if not (len(radii) == len(nsamples) == len(mlps)) or len(radii) < 1:
            raise ValueError("radii, nsamples and mlps must list the same (non-zero) number of scales")
        self.npoint = npoint
        self.radii = [float(r) for r in radii]
        self.nsamples = [int(s) for s in nsamples]
        self.mlps = nn.ModuleList()
        self._in_feat = None
        self._out = []
        fused = len(radii) <= 2
        for spec in mlps:
            spec = list(spec)
            if len(spec) < 2 or (self._in_feat is not None and spec[0] != self._in_feat):
                raise ValueError("every scale needs [in_features, width, ...] with the same in_features")
            fused = fused and spec[1:] == _FUSED_MLP and spec[0] in (0, 1)
            self._in_feat = spec[0]
            self._out.append(spec[-1])
            spec[0] += 3
            self.mlps.append(_SharedMLP(spec, bn))
        self.differentiable = False              # True: forward() takes the composed, differentiable path whenever a
                                                 # gradient is wanted, also for shapes the fused kernel covers
        self.fused = fused                       # one-kernel path (csrc/sa.hip); otherwise level-1 operators + dclr_linear
        self._cache = PackedCache()
        self._cache_composed = PackedCache()
        self._range_ok = None                    # weights key of the last checked split-f16 pass (ops.CHECK_RANGE)
        self.overflow_ptr = None                 # device address of the owning model's range flag (lib.MappedFlag), or None

    def out_features(self) -> int:
        return sum(self._out)

    def packed_mlps(self):
        """Fused path: one flat [W1 b1 W2 b2 W3 b3] buffer per scale. Composed path: per scale a list of
        (packed weight, bias, n_out, padded K) for dclr_linear."""
        if not self.fused:
            return self.packed_mlps_composed()

        def build():
            return [ops.pack_sa_mlp(*zip(*[u.folded() for u in stack])) for stack in self.mlps]
        return self._cache.get(flat_parameters(self), build)

    def packed_mlps_composed(self):
        """Per scale a list of (packed weight, bias, n_out, padded K) for dclr_linear: the composed path's weights (the
        composed path also serves fused-shape modules on clouds beyond the fused kernels' 65536 points)."""
        def build():
            packed = []
            for stack in self.mlps:
                layers = []
                for u in stack:
                    w, bias = u.folded()
                    w = w.detach().reshape(w.shape[0], -1)
                    kp = (w.shape[1] + 7) // 8 * 8
                    layers.append((ops.pack_weight(w.contiguous(), kp), bias.detach().contiguous(), w.shape[0], kp))
                packed.append(layers)
            return packed
        return self._cache_composed.get(flat_parameters(self), build)

    def sample(self, clouds: torch.Tensor, view=None):
        """Furthest point sampling only (the serial stage; the pipelined runner issues it batches ahead
        on side streams): clouds (B, N, C) -> (idx (B, npoint) int32, group_pts, group_box, slice_box); the group
        tensors are the sampling kernel's spatial partition, or None when it has none for this N.
        view = ops.batch_view(batches): `clouds` is the first of several batches read where they lie."""
        if clouds.shape[2] != 3 + self._in_feat:
            raise RuntimeError("expected {} columns per point, got {}".format(3 + self._in_feat, clouds.shape[2]))
        if not self.fused:
            raise NotImplementedError("sample()/forward_rows() belong to the fused kernel; this module runs composed")
        return ops.fps_clouds_grouped(clouds, self.npoint, view)

    def _range_key(self):
        return tuple((p.data_ptr(), p._version) for p in flat_parameters(self))

    def range_unchecked(self) -> bool:
        """True until a range-checked split-f16 pass has succeeded for the current weights (forward_rows)."""
        return self._range_ok != self._range_key()

    def forward_rows(self, clouds: torch.Tensor, sample=None, view=None) -> torch.Tensor:
        """clouds (B, N, 3 + in_feat) interleaved -> feature rows F (B*npoint, 68); sample = self.sample(clouds)."""
        if clouds.shape[1] > ops.FUSED_MAX_POINTS:
            # more points per cloud than the fused kernels index (16 bits): the same module composed from the level-1
            # operators, which take any n; rows F from its channel-layout result
            if view is not None:
                raise RuntimeError("clouds of more than {} points are not read in place across batches".format(ops.FUSED_MAX_POINTS))
            xyz = clouds[:, :, :3].contiguous()
            feats = clouds[:, :, 3:].transpose(1, 2).contiguous() if clouds.shape[2] > 3 else None
            # (a sample computed ahead -- the pipelined runner's sampling stage -- is used, not recomputed: the serial
            # stage is paid once)
            new_xyz, new_feats = self._forward_composed(xyz, feats, train=False, idx=None if sample is None else sample[0])
            return ops.channels_to_rows(torch.cat((new_xyz.transpose(1, 2), new_feats), dim=1).contiguous(), ops.F_STRIDE)
        if sample is None:
            sample = self.sample(clouds, view)
        idx, gpts, gbox = sample[:3]
        groups = None if gpts is None else (gpts, gbox) + tuple(sample[3:4])      # (+ slice boxes where the sampler exports them)
        mlps = self.packed_mlps()
        rows = ops.sa_msg_fused(clouds, idx, self.radii, self.nsamples, mlps, groups=groups, view=view,
                                overflow=self.overflow_ptr if ops.PRECISION == 'f16x2' else None)
        if ops.PRECISION == 'f16x2' and ops.CHECK_RANGE != 'never':
            # split-f16 operands clamp at +-65504: the first call after the weights changed (or every call with
            # CHECK_RANGE = 'always') also runs the f32 matrix instructions and compares (two host syncs, once)
            key = self._range_key()
            if ops.CHECK_RANGE == 'always' or key != self._range_ok:
                want = ops.sa_msg_fused(clouds, idx, self.radii, self.nsamples, mlps, groups=groups, precision='f32', view=view)
                err, scale = float((rows - want).abs().max()), float(want.abs().max())
                if not err <= 1e-4 * max(1.0, scale) or not scale < ops.F16_MAX:
                    raise RuntimeError("split-f16 matrix path out of range in set abstraction: features reach {:.4g} "
                                       "(limit 65504) and differ from the f32 matrix path by {:.3g}; run this checkpoint "
                                       "with DCLR_PRECISION=f32".format(scale, err))
                self._range_ok = key
        return rows

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor] = None,
                new_xyz: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """xyz (B, N, 3), features (B, C, N) or None -> new_xyz (B, npoint, 3), new_features (B, 32*scales, npoint)."""
        if new_xyz is not None:
            raise NotImplementedError("caller-supplied centroids are not used by DeepCLR")
        if (0 if features is None else features.shape[1]) != self._in_feat:
            raise RuntimeError("expected {} feature channels, got {}".format(
                self._in_feat, 0 if features is None else features.shape[1]))
        if not self.fused or (torch.is_grad_enabled() and (xyz.requires_grad or (features is not None and features.requires_grad)
                                                           or any(p.requires_grad for p in flat_parameters(self)))
                              and self.differentiable):
            return self._forward_composed(xyz, features)
        clouds = xyz if features is None else torch.cat((xyz, features.transpose(1, 2)), dim=2)
        rows = self.forward_rows(clouds.contiguous())
        b = xyz.shape[0]
        ch = ops.rows_to_channels(rows, b, self.npoint, self.out_features())
        return ch[:, :3, :].transpose(1, 2).contiguous(), ch[:, 3:, :].contiguous()

    def _forward_composed(self, xyz: torch.Tensor, features: Optional[torch.Tensor],
                          train: Optional[bool] = None, idx: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """The reference's own composition (QueryAndGroup + SharedMLP + max_pool2d) on the level-1 HIP operators.
        idx: the furthest-point sample of `xyz` if the caller has it already."""
        xyz = xyz.contiguous()
        b = xyz.shape[0]
        if idx is None:
            idx = ops.furthest_point_sample(xyz.detach(), self.npoint)
        xyz_t = xyz.transpose(1, 2).contiguous()
        new_xyz_t = gather_operation(xyz_t, idx)                                   # (B, 3, npoint), differentiable
        new_xyz = new_xyz_t.transpose(1, 2).contiguous()
        feats = None if features is None else features.contiguous()
        # Training (/root/reference/deepclr/engine/engines.py:57-84 differentiates through the module): gather / group
        # go through the HIP operators and their HIP backward; the shared MLP and the max then stay in torch, whose
        # autograd has their backward (rocBLAS GEMMs). Inference keeps dclr_linear with the max folded into the last layer.
        if train is None:
            train = torch.is_grad_enabled() and (xyz.requires_grad or (feats is not None and feats.requires_grad)
                                                 or any(p.requires_grad for p in flat_parameters(self)))
        outs = []
        for radius, nsample, layers, stack in zip(self.radii, self.nsamples, self.packed_mlps_composed() if not train else
                                                  [None] * len(self.radii), self.mlps):
            bq = ops.ball_query(radius, nsample, xyz.detach(), new_xyz.detach())   # (B, npoint, nsample) int32
            if train:
                grouped = grouping_operation(xyz_t, bq) - new_xyz_t.unsqueeze(-1)  # (B, 3, npoint, nsample)
                if feats is not None:
                    grouped = torch.cat((grouped, grouping_operation(feats, bq)), dim=1)
                outs.append(stack(grouped).max(dim=3).values)                      # 1x1 convs + ReLU, max over nsample
                continue
            ns_p = (nsample + 63) // 64 * 64                                       # dclr_linear folds the max over blocks of 64 rows:
            if ns_p != nsample:                                                    # pad with repeats of the first neighbour (max unchanged)
                bq = torch.cat((bq, bq[:, :, :1].expand(-1, -1, ns_p - nsample)), dim=2).contiguous()
            grouped = ops.grouping_operation(xyz_t, bq) - new_xyz_t.unsqueeze(-1)  # (B, 3, npoint, ns_p)
            if feats is not None:
                grouped = torch.cat((grouped, ops.grouping_operation(feats, bq)), dim=1)
            c_in, kp0 = grouped.shape[1], layers[0][3]
            rows = torch.zeros(b * self.npoint * ns_p, kp0, dtype=torch.float32, device=xyz.device)
            rows[:, :c_in] = grouped.permute(0, 2, 3, 1).reshape(-1, c_in)
            for j, (wp, bias, n_out, kp) in enumerate(layers):
                if j + 1 < len(layers):
                    rows = ops.linear(rows, wp, bias, n_out, kp, relu=True, ldy=layers[j + 1][3])
                else:
                    pooled = ops.linear(rows, wp, bias, n_out, kp, relu=True, colmax_groups=b * self.npoint)
            outs.append(pooled.view(b, self.npoint, -1).transpose(1, 2))
        return new_xyz, torch.cat(outs, dim=1).contiguous()
