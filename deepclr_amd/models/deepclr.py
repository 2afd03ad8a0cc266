"""DeepCLR forward pass on the fused HIP kernels.

Mirrors the module tree, constructor arguments, ``state_dict`` keys and the
``forward(x, is_feat, m, y, debug)`` / ``cloud_features(x, m)`` contract of
/root/reference/deepclr/models/deepclr.py (SetAbstraction 48-94, MotionEmbedding 176-246,
OutputSimple 249-294, DeepCLR 442-521, name-based factory 412-427). Internally the three stages
exchange point-major rows (F: 68 floats, E: 264 floats per point, include/deepclr_amd.h);
reference-layout tensors exist only at the API edge (``cloud_features`` output, ``is_feat`` input,
and the per-module ``forward`` methods the reference's layer tests call).

Out of scope here: losses (a ground truth ``y`` raises), backward, batch norm.
"""
import abc
import ctypes
import os
import time
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import lib, losses, ops
from ..config import Config
from ..labels import LabelType
from ..pointnet2 import PointnetSAModuleMSG, grouping_operation
from .base import BaseModel
from .helper import Conv1dMultiLayer, LinearMultiLayer, PackedCache, flat_parameters

FEAT = 64          # feature columns of a cloud-feature row (two 32-channel scales)


class DeepCLRModule(nn.Module, metaclass=abc.ABCMeta):
    """Base of the configurable sub-networks; looked up by class name (reference: deepclr.py:412-414)."""

    @abc.abstractmethod
    def output_dim(self) -> int:
        raise NotImplementedError


def _subclass_by_name(base: type, name: str) -> Optional[type]:
    for sub in base.__subclasses__():
        if sub.__name__ == name:
            return sub
        found = _subclass_by_name(sub, name)
        if found is not None:
            return found
    return None


def _init_module(cfg: Config, *args: Any, **kwargs: Any) -> DeepCLRModule:
    cls = _subclass_by_name(DeepCLRModule, cfg.name)
    if cls is None:
        raise NotImplementedError("Class '{}' not found as subclass of '{}'".format(cfg.name, DeepCLRModule.__name__))
    return cls(*args, **cfg.params, **kwargs)


# --------------------------------------------------------------------------------------------------
# set abstraction
# --------------------------------------------------------------------------------------------------
class SetAbstraction(DeepCLRModule):
    """Per-cloud feature extraction (reference: deepclr.py:48-94). Level 0 is the fused kernel; an optional second
    level (deepclr.py:72-83,92-93; no shipped config has one) runs composed from the level-1 operators
    (deepclr_amd/pointnet2.py) on the level-0 centroids and features."""

    def __init__(self, input_dim: int, point_dim: int, mlps: List[List[List[int]]], npoint: List[int],
                 radii: List[List[float]], nsamples: List[List[int]], batch_norm: bool = False, **_kwargs: Any):
        super().__init__()
        assert point_dim == 3
        assert len(mlps) == len(npoint) == len(radii) == len(nsamples)
        assert 0 < len(mlps) <= 2
        feat_in = input_dim - point_dim
        self._input_dim = input_dim
        self._output_feat_dim = int(np.sum([spec[-1] for spec in mlps[-1]]))
        self._sa0 = PointnetSAModuleMSG(npoint=npoint[0], radii=radii[0], nsamples=nsamples[0],
                                        mlps=[[feat_in, *spec] for spec in mlps[0]], use_xyz=True, bn=batch_norm)
        if len(npoint) == 2:
            self._sa1 = PointnetSAModuleMSG(npoint=npoint[1], radii=radii[1], nsamples=nsamples[1],
                                            mlps=[[*spec] for spec in mlps[1]], use_xyz=True, bn=batch_norm)
        else:
            self._sa1 = None
        self.npoint = npoint[-1]
        # Row pipeline (rows F, 64 feature columns): level 0 on the fused kernel -- xyz + <= 1 feature, mlp widths
        # [16, 16, 32], every shipped configuration -- and at most 64 output features. Any other shape the reference
        # accepts (deepclr.py:50-70: any `mlps`, any number of input features) runs composed from the level-1 HIP
        # operators in the reference's channel layout (forward()).
        self.rows_path = self._sa0.fused and self._output_feat_dim <= FEAT

    def output_dim(self) -> int:
        return 3 + self._output_feat_dim

    def sample(self, clouds: torch.Tensor, view=None):
        return self._sa0.sample(clouds, view)

    def forward_rows(self, clouds: torch.Tensor, sample=None, view=None) -> torch.Tensor:
        """(2B, N, C) point-major clouds -> rows F."""
        if not self.rows_path:
            raise NotImplementedError("this set-abstraction shape runs composed (forward()); the row pipeline needs level 0 "
                                      "as xyz + <= 1 feature with mlp widths [16, 16, 32] and <= 64 output features")
        rows = self._sa0.forward_rows(clouds, sample, view)
        if self._sa1 is None:
            return rows
        b = clouds.shape[0] if view is None else 2 * view[0] * view[1]
        ch = ops.rows_to_channels(rows, b, self._sa0.npoint, self._sa0.out_features())
        xyz, feat = self._sa1(ch[:, :3, :].transpose(1, 2).contiguous(), ch[:, 3:, :].contiguous())
        return ops.channels_to_rows(torch.cat((xyz.transpose(1, 2), feat), dim=1).contiguous(), ops.F_STRIDE)

    def forward_train(self, clouds: torch.Tensor) -> torch.Tensor:
        """forward() with a gradient: every level composed from the level-1 HIP operators and their HIP backward
        (gather / group), the shared MLP and the max in torch (PointnetSAModuleMSG._forward_composed, train branch)."""
        xyz = clouds[:, :3, :].transpose(1, 2).contiguous()
        feats = clouds[:, 3:, :].contiguous() if clouds.shape[1] > 3 else None
        for level in (self._sa0, self._sa1):
            if level is not None:
                xyz, feats = level._forward_composed(xyz, feats, train=True)
        return torch.cat((xyz.transpose(1, 2), feats), dim=1).contiguous()

    def forward(self, clouds: torch.Tensor, *_args: Any) -> torch.Tensor:
        """(2B, C, N) channel-major clouds -> (2B, 3 + feat, npoint), as the reference module."""
        if self.rows_path:
            rows = self.forward_rows(clouds.transpose(1, 2).contiguous())
            return ops.rows_to_channels(rows, clouds.shape[0], self.npoint, self._output_feat_dim)
        # composed (reference: split_features -> _sa0 [-> _sa1] -> merge_features, deepclr.py:88-94)
        xyz = clouds[:, :3, :].transpose(1, 2).contiguous()
        feats = clouds[:, 3:, :].contiguous() if clouds.shape[1] > 3 else None
        xyz, feats = self._sa0(xyz, feats)
        if self._sa1 is not None:
            xyz, feats = self._sa1(xyz, feats)
        return torch.cat((xyz.transpose(1, 2), feats), dim=1).contiguous()


# --------------------------------------------------------------------------------------------------
# flow embedding
# --------------------------------------------------------------------------------------------------
class MotionEmbeddingBase(nn.Module):
    """kNN grouping (or, with k == 0, GlobalGrouping: every point of the pair's source cloud, reference
    deepclr.py:108-139,186-187) + shared MLP + radius mask + max (reference: deepclr.py:176-231)."""

    def __init__(self, input_dim: int, point_dim: int, k: int, radius: float, mlp: List[int],
                 append_features: bool = True, batch_norm: bool = False, **_kwargs: Any):
        super().__init__()
        if k < 0:
            raise ValueError("k must be >= 0 (0: every point of the source cloud)")
        if point_dim != 3:
            raise NotImplementedError("three-dimensional points only")
        self._point_dim = point_dim
        self._feat_dim = input_dim - point_dim
        # Row pipeline (fused kernel, rows F -> rows E): the shipped shape -- mlp [128, 128, 256], k <= 32, <= 64 features.
        # Any other `mlp` / k / feature width the reference accepts (deepclr.py:180-199) runs composed from the
        # level-1 HIP operators (forward()).
        self.rows_path = k <= 32 and list(mlp) == [128, 128, 256] and self._feat_dim <= FEAT
        self._append_features = append_features
        self._k, self._radius = int(k), float(radius)
        c_in = point_dim + (2 if append_features else 1) * self._feat_dim
        self._conv = Conv1dMultiLayer([c_in, *mlp], batch_norm=batch_norm)
        self._cache = PackedCache()
        self._cache_composed = PackedCache()

    def output_dim(self) -> int:
        return self._point_dim + self._conv.output_dim()

    def _packed(self):
        def build():
            (w1, b1), (w2, b2), (w3, b3) = self._conv.affine_params()
            w1 = w1.detach().reshape(w1.shape[0], -1)
            d, f = self._point_dim, self._feat_dim
            dev = w1.device
            kmap = torch.full((FEAT,), -1, dtype=torch.int32, device=dev)
            kmap[:f] = torch.arange(f, dtype=torch.int32, device=dev)
            if self._append_features:
                w_t, w_s = w1[:, d:d + f], w1[:, d + f:d + 2 * f]
            else:                                   # merged = [pos_diff, feat_s - feat_t] (deepclr.py:212-213)
                w_t, w_s = -w1[:, d:d + f], w1[:, d:d + f]
            return {
                'w1a': w1[:, :d].contiguous(), 'b1': b1.detach().contiguous(),
                'wt': ops.pack_weight(w_t.contiguous(), FEAT, kmap), 'ws': ops.pack_weight(w_s.contiguous(), FEAT, kmap),
                'w2p': ops.pack_weight(w2, 128, tile16=True), 'b2': b2.detach().contiguous(),
                'w3p': ops.pack_weight(w3, 128, tile16=True), 'b3': b3.detach().contiguous(),
                # the tile the kernel runs this k on (k == 0: slices of 32 neighbours, see forward_rows)
                'w2h': ops.pack_weight_f16(w2, 128, ops.flow_f16_tile(self._k or 32)),
                'w3h': ops.pack_weight_f16(w3, 128, ops.flow_f16_tile(self._k or 32)),
            }
        return self._cache.get(flat_parameters(self), build)

    def forward_rows(self, f_rows: torch.Tensor, pairs: int, npoint: int, precision: Optional[str] = None) -> torch.Tensor:
        """rows F of [templates..., sources...] -> rows E (pairs*npoint, 264). precision: matrix path of this call
        ('f16x2' / 'f32'); None = ops.PRECISION."""
        if not self.rows_path:
            raise NotImplementedError("this flow-embedding shape runs composed (forward()); the row pipeline needs "
                                      "mlp [128, 128, 256], k <= 32 and <= 64 feature channels")
        precision = precision or ops.PRECISION
        p = self._packed()
        half = pairs * npoint
        pt = ops.linear(f_rows[:half], p['wt'], None, 128, FEAT, relu=False)
        ps = ops.linear(f_rows[half:], p['ws'], None, 128, FEAT, relu=False)

        def embed(idx):
            if precision == 'f16x2':
                return ops.flow_embedding_fused_f16(f_rows, idx, pt, ps, p['w1a'], p['b1'], p['w2h'], p['b2'],
                                                    p['w3h'], p['b3'], self._radius)
            return ops.flow_embedding_fused(f_rows, idx, pt, ps, p['w1a'], p['b1'], p['w2p'], p['b2'],
                                            p['w3p'], p['b3'], self._radius)
        if self._k > 0:
            return embed(ops.knn_rows(f_rows, pairs, npoint, self._k))
        # GlobalGrouping: the neighbourhood of every template point is the whole source cloud of its pair. The fused
        # kernel takes up to 32 neighbours per point; the maximum over all of them is the maximum over 32-point slices
        # of the maxima within each (post-ReLU values and the 0 of radius-masked rows: the same floor in every slice).
        e_rows = None
        for j0 in range(0, npoint, 32):
            # always 32 slots (one weight packing, ops.flow_f16_tile): a short last slice ends in -1, the search's own
            # mark for "no neighbour", which the kernels leave out of the maximum
            col = torch.arange(j0, j0 + 32, dtype=torch.int32, device=f_rows.device)
            col = torch.where(col < npoint, col, torch.full_like(col, -1))
            idx = col.view(1, 1, 32).expand(pairs, npoint, 32).contiguous()
            part = embed(idx)
            if e_rows is None:
                e_rows = part
            else:
                torch.maximum(e_rows[:, :256], part[:, :256], out=e_rows[:, :256])
        return e_rows

    def forward(self, clouds0: torch.Tensor, clouds1: torch.Tensor) -> torch.Tensor:
        """(B, 3+F, P) template / source feature clouds -> (B, 3+mlp[-1], P), as the reference module."""
        if not self.rows_path:
            return self._forward_composed(clouds0.contiguous(), clouds1.contiguous())
        b, _, npoint = clouds0.shape
        f_rows = ops.channels_to_rows(torch.cat((clouds0, clouds1), dim=0).contiguous(), ops.F_STRIDE)
        e_rows = self.forward_rows(f_rows, b, npoint)
        return ops.rows_to_channels(e_rows, b, npoint, 256)

    def forward_train(self, clouds0: torch.Tensor, clouds1: torch.Tensor) -> torch.Tensor:
        """forward() with a gradient (reference: deepclr.py:201-231 under autograd): the neighbour lists come from the HIP
        search (indices: nothing to differentiate), the source rows are gathered through GroupingOperation -- dclr_group_points
        forward, dclr_group_points_grad backward, what the reference's `pts1[group_index]` gets from torch's index kernels --
        and the shared MLP, the radius mask and the max run in torch, whose autograd has their backward."""
        b, c, p0 = clouds0.shape
        p1 = clouds1.shape[2]
        d, k = self._point_dim, self._k
        dev = clouds0.device
        clouds0, clouds1 = clouds0.contiguous(), clouds1.contiguous()
        if k > 0:
            xyz0 = clouds0[:, :d, :].detach().transpose(1, 2).contiguous()
            xyz1 = clouds1[:, :d, :].detach().transpose(1, 2).contiguous()
            row = torch.empty(b * p0 * k, dtype=torch.int64, device=dev)
            col = torch.empty(b * p0 * k, dtype=torch.int64, device=dev)
            ops._call('dclr_knn', 'knn', b, p1, p0, k, xyz1.data_ptr(), xyz0.data_ptr(), row.data_ptr(), col.data_ptr(),
                      lib.stream_ptr())
            if bool((col < 0).any()):
                raise RuntimeError("kNN grouping: a template point has fewer than k = {} source points".format(k))
            idx = (col.view(b, p0, k) - (torch.arange(b, device=dev) * p1).view(b, 1, 1)).to(torch.int32).contiguous()
        else:
            k = p1
            idx = torch.arange(p1, dtype=torch.int32, device=dev).view(1, 1, p1).expand(b, p0, p1).contiguous()
        grouped = grouping_operation(clouds1, idx)                                  # (B, C, P0, k), differentiable
        pos_diff = grouped[:, :d] - clouds0[:, :d, :].unsqueeze(-1)
        feat_t = clouds0[:, d:, :].unsqueeze(-1)
        if self._append_features:
            h = torch.cat((pos_diff, feat_t.expand(-1, -1, -1, k), grouped[:, d:]), dim=1)
        else:
            h = torch.cat((pos_diff, grouped[:, d:] - feat_t), dim=1)
        # the reference's Conv1dMultiLayer on (groups, C, k) rows (deepclr.py:212-217): the torch modules themselves, so that
        # batch norm (statistics over every group and neighbour, per channel) and dropout act as they do there
        h = self._conv.forward_torch(h.reshape(b, h.shape[1], p0 * k)).reshape(b, -1, p0, k)
        if self._radius > 0.0:
            h = h.masked_fill((torch.norm(pos_diff, dim=1) >= self._radius).unsqueeze(1), 0.0)
        return torch.cat((clouds0[:, :d, :], h.max(dim=3)[0]), dim=1).contiguous()

    def _packed_composed(self):
        """dclr_linear weights of the composed path. The LAST layer gets one extra input column with weight -1e30: the
        rows of neighbours beyond the radius (and the rows that pad a neighbourhood to a multiple of 64) carry 1 there,
        so their outputs are exactly 0 after the ReLU -- what the reference's masked_scatter_ writes (deepclr.py:220-223)
        -- and the max over the neighbourhood folds into that layer's launch (ops.linear(colmax_groups=...))."""
        def build():
            layers = []
            params = self._conv.affine_params()
            for j, (w, b) in enumerate(params):
                w = w.detach().reshape(w.shape[0], -1)
                last = j == len(params) - 1
                k_in = w.shape[1] + (1 if last else 0)
                kp = (k_in + 7) // 8 * 8
                if last:
                    w = torch.cat((w, torch.full((w.shape[0], 1), -1.0e30, device=w.device)), dim=1)
                layers.append((ops.pack_weight(w.contiguous(), kp), b.detach().contiguous(), w.shape[0], kp))
            return layers
        return self._cache_composed.get(flat_parameters(self), build)

    def _forward_composed(self, clouds0: torch.Tensor, clouds1: torch.Tensor) -> torch.Tensor:
        """The reference's own composition (deepclr.py:142-173 or 108-139, then 201-231) on the level-1 HIP operators:
        dclr_knn -> dclr_group_points -> dclr_linear chain with the radius mask and the max folded into its last launch."""
        b, c, p0 = clouds0.shape
        p1 = clouds1.shape[2]
        d, k = self._point_dim, self._k
        dev = clouds0.device
        layers = self._packed_composed()
        xyz0 = clouds0[:, :d, :].transpose(1, 2).contiguous()                       # (B, P0, 3)
        if k > 0:
            if p1 < k:
                raise RuntimeError("kNN grouping: {} source points, k = {}".format(p1, k))
            xyz1 = clouds1[:, :d, :].transpose(1, 2).contiguous()
            row = torch.empty(b * p0 * k, dtype=torch.int64, device=dev)
            col = torch.empty(b * p0 * k, dtype=torch.int64, device=dev)
            ops._call('dclr_knn', 'knn', b, p1, p0, k, xyz1.data_ptr(), xyz0.data_ptr(), row.data_ptr(), col.data_ptr(),
                      lib.stream_ptr())
            if bool((col < 0).any()):
                # upstream: torch_cluster.knn returns fewer than k neighbours and .view(2, G, k) fails (deepclr.py:167)
                raise RuntimeError("kNN grouping: a template point has fewer than k = {} source points within the "
                                   "search's start distance, or non-finite coordinates".format(k))
            idx = (col.view(b, p0, k) - (torch.arange(b, device=dev) * p1).view(b, 1, 1)).to(torch.int32)
        else:                                                                      # GlobalGrouping: all source points, in order
            k = p1
            idx = torch.arange(p1, dtype=torch.int32, device=dev).view(1, 1, p1).expand(b, p0, p1)
        r = (k + 63) // 64 * 64                                                    # rows per neighbourhood (dclr_linear: blocks of 64)
        c_in = d + (2 if self._append_features else 1) * (c - d)
        kp0, n_last = layers[0][3], layers[-1][2]
        out = torch.empty(b, n_last, p0, dtype=torch.float32, device=dev)
        # template points in chunks that keep the WIDEST materialised row set -- the input rows or any layer's output rows
        # (ldy of its consumer), two of which are alive at once -- below ~256 MB (a 3-column input in front of a 256-wide
        # layer is 32 times wider after the first launch than before it)
        widest = max([kp0, 8] + [layers[j + 1][3] for j in range(len(layers) - 1)])
        chunk = max(1, min(p0, (1 << 26) // (r * widest * b)))
        for q0 in range(0, p0, chunk):
            q1 = min(p0, q0 + chunk)
            q = q1 - q0
            sel = torch.zeros(b, q, r, dtype=torch.int32, device=dev)
            sel[:, :, :k] = idx[:, q0:q1, :]
            grouped = ops.grouping_operation(clouds1, sel.contiguous())            # (B, C, q, r)
            pos_diff = grouped[:, :d] - clouds0[:, :d, q0:q1].unsqueeze(-1)
            feat_t = clouds0[:, d:, q0:q1].unsqueeze(-1)
            if self._append_features:
                merged = torch.cat((pos_diff, feat_t.expand(-1, -1, -1, r), grouped[:, d:]), dim=1)
            else:
                merged = torch.cat((pos_diff, grouped[:, d:] - feat_t), dim=1)
            rows = torch.zeros(b * q * r, kp0, dtype=torch.float32, device=dev)
            rows[:, :c_in] = merged.permute(0, 2, 3, 1).reshape(-1, c_in)
            masked = torch.zeros(b, q, r, dtype=torch.bool, device=dev)
            masked[:, :, k:] = True
            if self._radius > 0.0:
                masked[:, :, :k] = torch.norm(pos_diff[:, :, :, :k], dim=1) >= self._radius
            h = rows
            for j, (wp, bias, n_out, kp) in enumerate(layers):
                if j + 1 < len(layers):
                    ldy = layers[j + 1][3]
                    h = ops.linear(h, wp, bias, n_out, kp, relu=True, ldy=ldy)
                    if j + 2 == len(layers):                                       # the mask column of the last layer's input
                        h[:, n_out] = masked.reshape(-1).to(torch.float32)
                else:
                    if len(layers) == 1:
                        h[:, c_in] = masked.reshape(-1).to(torch.float32)
                    pooled = ops.linear(h, wp, bias, n_out, kp, relu=True, colmax_groups=b * q)
            out[:, :, q0:q1] = pooled.view(b, q, n_last).transpose(1, 2)
        return torch.cat((clouds0[:, :d, :], out), dim=1).contiguous()


class MotionEmbedding(DeepCLRModule):
    """Batch layout [T0..TB-1, S0..SB-1] (reference: deepclr.py:234-246)."""

    def __init__(self, **kwargs: Any):
        super().__init__()
        self._embedding = MotionEmbeddingBase(**kwargs)

    @property
    def rows_path(self) -> bool:
        return self._embedding.rows_path

    def output_dim(self) -> int:
        return self._embedding.output_dim()

    def forward_rows(self, f_rows: torch.Tensor, pairs: int, npoint: int, precision: Optional[str] = None) -> torch.Tensor:
        return self._embedding.forward_rows(f_rows, pairs, npoint, precision)

    def forward(self, clouds: torch.Tensor) -> torch.Tensor:
        half = clouds.shape[0] // 2
        return self._embedding(clouds[:half], clouds[half:])

    def forward_train(self, clouds: torch.Tensor) -> torch.Tensor:
        half = clouds.shape[0] // 2
        return self._embedding.forward_train(clouds[:half], clouds[half:])


# --------------------------------------------------------------------------------------------------
# pose head
# --------------------------------------------------------------------------------------------------
class OutputSimple(DeepCLRModule):
    """Mini-PointNet + fully connected regression head (reference: deepclr.py:249-294)."""

    def __init__(self, input_dim: int, label_type: LabelType, mlp: List[int], linear: List[int],
                 batch_norm: bool = False, dropout: float = 1.0, **_kwargs: Any):
        super().__init__()
        self.rows_path = input_dim == 3 + 256        # rows E of the fused flow embedding; other widths run composed (forward())
        self._input_dim = input_dim
        self._label_type = label_type
        self.conv = Conv1dMultiLayer([input_dim, *mlp], batch_norm=batch_norm)
        self.linear = LinearMultiLayer(linear, batch_norm=batch_norm, dropout_keep=dropout, dropout_last=True)
        self.output = nn.Linear(linear[-1], label_type.dim, bias=True)
        nn.init.xavier_uniform_(self.output.weight)
        if label_type.bias is not None:
            with torch.no_grad():
                self.output.bias.copy_(torch.tensor(label_type.bias, dtype=torch.float32))
        self._act = {LabelType.POSE3D_DUAL_QUAT: 2, LabelType.POSE3D_QUAT: 3}.get(label_type, 0)
        self._cache = PackedCache()
        self._cache16 = PackedCache()
        self._cache_plain = PackedCache()

    def output_dim(self) -> int:
        return self._label_type.dim

    def _packed(self):
        def build():
            layers = []
            for i, (w, b) in enumerate(self.conv.affine_params()):
                n, k_in = w.shape[0], w.shape[1]
                if i == 0:      # reference column order [xyz | feat] -> row order [feat | xyz | pad]
                    kp = ops.E_STRIDE
                    kmap = torch.full((kp,), -1, dtype=torch.int32, device=w.device)
                    kmap[:256] = torch.arange(3, 259, dtype=torch.int32, device=w.device)
                    kmap[256:259] = torch.arange(0, 3, dtype=torch.int32, device=w.device)
                    wp = ops.pack_weight(w, kp, kmap)
                else:
                    kp = (k_in + 7) // 8 * 8
                    wp = ops.pack_weight(w, kp)
                layers.append((wp, b.detach().contiguous(), n, kp))
            return layers
        return self._cache.get(flat_parameters(self.conv), build)

    def _packed_f16(self):
        def build():
            layers = []
            for i, (w, b) in enumerate(self.conv.affine_params()):
                n, k_in = w.shape[0], w.shape[1]
                if i == 0:      # rows E: [feat | xyz | pad] -> 272 columns (k-steps of 16), the rest zero weights
                    kp = (ops.E_STRIDE + 15) // 16 * 16
                    kmap = torch.full((kp,), -1, dtype=torch.int32, device=w.device)
                    kmap[:256] = torch.arange(3, 259, dtype=torch.int32, device=w.device)
                    kmap[256:259] = torch.arange(0, 3, dtype=torch.int32, device=w.device)
                    wp = ops.pack_weight_f16(w, kp, 32, kmap)
                else:
                    kp = (k_in + 15) // 16 * 16
                    wp = ops.pack_weight_f16(w, kp, 32)
                layers.append((wp, b.detach().contiguous(), n, kp))
            return layers
        return self._cache16.get(flat_parameters(self.conv), build)

    def _fusable(self, layers, rows: int, pairs: int, precision: Optional[str] = None) -> bool:
        """The one-launch conv chain needs 32-row tiles inside one pair and hidden widths <= 512. Its grid is
        rows / 32 workgroups, one per CU. On the f32 matrix path, below half the chip (single pairs, the reference's
        own batch size) the per-layer kernels, whose grids also split the output columns, finish sooner (84 vs
        142 us at one pair); the split-fp16 chain (65 us) is used at every batch size, which also keeps the
        result of a pair independent of the batch it travels in."""
        return ((rows >= 4096 or (precision or ops.PRECISION) == 'f16x2') and rows % 32 == 0 and (rows // pairs) % 32 == 0 and len(layers) <= 8
                and all(n % 32 == 0 for _, _, n, _ in layers) and all(kp <= 512 for _, _, _, kp in layers)
                and all(n <= 512 for _, _, n, _ in layers[:-1]))

    def forward_rows(self, e_rows: torch.Tensor, pairs: int, precision: Optional[str] = None) -> torch.Tensor:
        precision = precision or ops.PRECISION
        layers = self._packed()
        if self._fusable(layers, e_rows.shape[0], pairs, precision) and os.environ.get('DCLR_HEAD_FUSED', '1') != '0':
            if precision == 'f16x2':
                g = ops.head_conv_fused_f16(e_rows, ops.E_STRIDE, self._packed_f16(), pairs)
            else:
                g = ops.head_conv_fused(e_rows, layers, pairs)               # conv chain + max over points
            g = self.linear(g)
            return ops.fc(g, self.output.weight, self.output.bias, act=self._act)
        h = e_rows
        for wp, b, n, kp in layers[:-1]:
            h = ops.linear(h, wp, b, n, kp, relu=True, ldy=(n + 7) // 8 * 8)
        wp, b, n, kp = layers[-1]
        g = ops.linear(h, wp, b, n, kp, relu=True, colmax_groups=pairs)      # conv + max over points
        g = self.linear(g)
        return ops.fc(g, self.output.weight, self.output.bias, act=self._act)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, C, P) -> (B, label_dim)."""
        if self.rows_path:
            return self.forward_rows(ops.channels_to_rows(x.contiguous(), ops.E_STRIDE), x.shape[0])
        # any input width: the conv chain layer by layer on dclr_linear (reference column order, no remap), the max over
        # points folded into the last launch; a pair's rows are padded to a multiple of 64 with repeats of its first point
        b, c, p = x.shape
        layers = self._packed_plain()
        pp = (p + 63) // 64 * 64
        pts = x.transpose(1, 2)                                                     # (B, P, C)
        if pp != p:
            pts = torch.cat((pts, pts[:, :1, :].expand(-1, pp - p, -1)), dim=1)
        h = torch.zeros(b * pp, layers[0][3], dtype=torch.float32, device=x.device)
        h[:, :c] = pts.reshape(-1, c)
        for j, (wp, bias, n, kp) in enumerate(layers):
            if j + 1 < len(layers):
                h = ops.linear(h, wp, bias, n, kp, relu=True, ldy=layers[j + 1][3])
            else:
                g = ops.linear(h, wp, bias, n, kp, relu=True, colmax_groups=b)
        g = self.linear(g)
        return ops.fc(g, self.output.weight, self.output.bias, act=self._act)

    def forward_train(self, x: torch.Tensor) -> torch.Tensor:
        """forward() with a gradient (reference: deepclr.py:284-294 under autograd; the output activation is applied out of
        place, the reference's in-place writes into y give the same values)."""
        h = self.conv.forward_torch(x).max(dim=2)[0]        # the torch modules themselves: batch norm and dropout as the
        h = self.linear.forward_torch(h)                    # reference applies them (helper.py:92-93,122-123)
        y = F.linear(h, self.output.weight, self.output.bias)
        if self._act == 2:                          # dual quaternion: sigmoid on column 0, tanh on 1..3
            y = torch.cat((torch.sigmoid(y[:, :1]), torch.tanh(y[:, 1:4]), y[:, 4:]), dim=1)
        elif self._act == 3:                        # quaternion + translation: sigmoid on column 3, tanh on 4..6
            y = torch.cat((y[:, :3], torch.sigmoid(y[:, 3:4]), torch.tanh(y[:, 4:])), dim=1)
        return y

    def _packed_plain(self):
        def build():
            layers = []
            for w, b in self.conv.affine_params():
                w = w.detach().reshape(w.shape[0], -1)
                kp = (w.shape[1] + 7) // 8 * 8
                layers.append((ops.pack_weight(w.contiguous(), kp), b.detach().contiguous(), w.shape[0], kp))
            return layers
        return self._cache_plain.get(flat_parameters(self.conv), build)


# --------------------------------------------------------------------------------------------------
# losses (forward values only: validation figures; there is no backward through the HIP kernels)
# --------------------------------------------------------------------------------------------------
class DeepCLRLoss(DeepCLRModule, metaclass=abc.ABCMeta):
    def output_dim(self) -> int:
        return 1

    def get_weights(self) -> Dict:
        return {}


class TransformLoss(DeepCLRLoss):
    """Fixed weights: sx * translation + sq * rotation (reference: deepclr.py:352-369)."""

    def __init__(self, label_type: LabelType, p: int, sx: float, sq: float, **_kwargs: Any):
        super().__init__()
        self._label_type, self._p = label_type, p
        self._sx, self._sq = sx, sq

    def forward(self, y_pred: torch.Tensor, y: torch.Tensor, **_kwargs: Any) -> torch.Tensor:
        t, r = losses.transform_losses(y_pred, y, self._label_type, self._p)
        return t * self._sx + r * self._sq


class TransformUncertaintyLoss(DeepCLRLoss):
    """Learned weights: t * exp(-sx) + sx + r * exp(-sq) + sq (reference: deepclr.py:372-389)."""

    def __init__(self, label_type: LabelType, p: int, sx: float, sq: float, **_kwargs: Any):
        super().__init__()
        self._label_type, self._p = label_type, p
        self._sx = torch.nn.Parameter(torch.Tensor([sx]))
        self._sq = torch.nn.Parameter(torch.Tensor([sq]))

    def get_weights(self) -> Dict:
        return {'sx': self._sx.item(), 'sq': self._sq.item()}

    def forward(self, y_pred: torch.Tensor, y: torch.Tensor, **_kwargs: Any) -> torch.Tensor:
        t, r = losses.transform_losses(y_pred, y, self._label_type, self._p)
        return t * torch.exp(-self._sx) + self._sx + r * torch.exp(-self._sq) + self._sq


class AccumulatedLoss(DeepCLRLoss):
    def __init__(self, modules: List[torch.nn.Module]):
        super().__init__()
        self.loss_list = torch.nn.ModuleList(modules)

    def get_weights(self) -> Dict:
        weights: Dict = {}
        for loss in self.loss_list:
            for key, value in loss.get_weights().items():
                if key in weights:
                    raise RuntimeError("Duplicate loss keys")
                weights[key] = value
        return weights

    def forward(self, *args: Any) -> torch.Tensor:
        return torch.stack([loss(*args) for loss in self.loss_list], dim=0).sum()


def _init_loss(cfg: Config, label_type: LabelType, **kwargs: Any) -> DeepCLRLoss:
    cls = _subclass_by_name(DeepCLRLoss, cfg['name'])
    if cls is None:
        raise NotImplementedError("Class '{}' not found as subclass of 'DeepCLRLoss'".format(cfg['name']))
    return cls(label_type=label_type, **cfg['params'], **kwargs)


# --------------------------------------------------------------------------------------------------
# network
class _Prep(tuple):
    """(pt, ps, knn_idx) of one stage-1 run, out of a plan's small ring. `done`: event the consumer records once the
    stage-2 launches that read the buffers are enqueued; the plan waits for it before it hands the slot out again."""
    done: Optional[torch.cuda.Event] = None

    def release(self) -> None:
        self.done = torch.cuda.Event()
        self.done.record()


class _MergePlan:
    """Arguments + workspace of dclr_merge_forward for one (device, pairs, npoint, matrix path): built once,
    reused by every batch of that shape (the workspace belongs to the stream the calls are enqueued on)."""

    def __init__(self, args, keep, versions, pairs: int, n_out: int, device):
        self.args, self._keep, self._versions = args, keep, versions
        self._pairs, self._n_out, self._device = pairs, n_out, device
        self._ring, self._next = [], 0

    @staticmethod
    def _version_key(mods):
        return tuple((p.data_ptr(), p._version) for m in mods for p in flat_parameters(m))

    def current(self) -> bool:
        return self._versions == self._version_key(self._keep['mods'])

    @classmethod
    def build(cls, flow, head, device, pairs: int, npoint: int, overflow_ptr: Optional[int] = None):
        rows = pairs * npoint
        f16 = ops.PRECISION == 'f16x2'
        layers = head._packed_f16() if f16 else head._packed()
        if not head._fusable(head._packed(), rows, pairs) or (not f16 and rows < 4096) or flow._k == 0:
            return None                             # (GlobalGrouping runs slice by slice through forward_rows)
        fcs = [(*m.folded(), 1) for m in head.linear.layers()]         # eval-mode batch norm folded in; dropout = identity
        fcs.append((head.output.weight, head.output.bias, head._act))
        if len(layers) > lib.MERGE_MAX_LAYERS or len(fcs) > lib.MERGE_MAX_FC or any(b is None for _, b, _ in fcs):
            return None
        p = flow._packed()
        a = lib.MergeArgs()
        a.pairs, a.npoint, a.k, a.precision, a.radius = pairs, npoint, flow._k, int(f16), flow._radius
        a.n_head_layers, a.head_k_in, a.n_fc = len(layers), ops.E_STRIDE, len(fcs)
        a.overflow = overflow_ptr                   # the split-f16 kernels report a clamped activation there
        keep = {'mods': [flow, head], 'tensors': [p, layers]}

        def dev(t):
            t = t.detach().contiguous()
            keep['tensors'].append(t)
            return t.data_ptr()
        for i, (wp, b, n, kp) in enumerate(layers):
            a.head_k[i], a.head_n[i], a.head_w[i], a.head_b[i] = kp, n, wp.data_ptr(), b.data_ptr()
        for i, (w, b, act) in enumerate(fcs):
            a.fc_k[i], a.fc_n[i], a.fc_act[i], a.fc_w[i], a.fc_b[i] = w.shape[1], w.shape[0], act, dev(w), dev(b)
        a.wt, a.ws, a.w1a, a.b1 = p['wt'].data_ptr(), p['ws'].data_ptr(), p['w1a'].data_ptr(), p['b1'].data_ptr()
        a.w2, a.w3 = (p['w2h'] if f16 else p['w2p']).data_ptr(), (p['w3h'] if f16 else p['w3p']).data_ptr()
        a.b2, a.b3 = p['b2'].data_ptr(), p['b3'].data_ptr()
        width = max(w.shape[0] for w, _, _ in fcs)
        ws = {'pt': torch.empty(rows, 128, device=device), 'ps': torch.empty(rows, 128, device=device),
              'knn': torch.empty(pairs, npoint, flow._k, dtype=torch.int32, device=device),
              'e': torch.empty(rows, ops.E_STRIDE, device=device),
              'colmax': torch.empty(pairs, layers[-1][2], device=device),
              'tmp': torch.empty(2, pairs, width, device=device)}
        keep['ws'] = ws
        a.pt, a.ps, a.knn_idx, a.e_rows = ws['pt'].data_ptr(), ws['ps'].data_ptr(), ws['knn'].data_ptr(), ws['e'].data_ptr()
        a.colmax = ws['colmax'].data_ptr()
        a.fc_tmp[0], a.fc_tmp[1] = ws['tmp'][0].data_ptr(), ws['tmp'][1].data_ptr()
        return cls(a, keep, cls._version_key(keep['mods']), pairs, fcs[-1][0].shape[0], device)

    PREP_RING = 3

    def _check(self, f_rows: torch.Tensor) -> torch.Tensor:
        f_rows = lib.dev_f32(f_rows, 'f_rows')
        if f_rows.shape != (2 * self._pairs * self.args.npoint, ops.F_STRIDE):
            raise RuntimeError("feature rows do not match the planned batch shape")
        return f_rows

    def prep(self, f_rows: torch.Tensor, events=None):
        """Stage 1 alone (layer-1 halves + kNN) into buffers of its own -- what a side stream can run ahead."""
        f_rows = self._check(f_rows)
        rows, a = self._pairs * self.args.npoint, self.args
        # a ring of PREP_RING buffer sets instead of three allocations per batch (a long run otherwise parks gigabytes
        # in the caching allocator behind record_stream): a slot is reused only after its consumer released it
        if len(self._ring) < self.PREP_RING:
            out = _Prep((torch.empty(rows, 128, device=self._device), torch.empty(rows, 128, device=self._device),
                         torch.empty(self._pairs, a.npoint, a.k, dtype=torch.int32, device=self._device)))
            self._ring.append(out)
        else:
            old = self._ring[self._next]
            if old.done is None:                        # never released by its consumer: leave it alone, take fresh memory
                out = _Prep((torch.empty(rows, 128, device=self._device), torch.empty(rows, 128, device=self._device),
                             torch.empty(self._pairs, a.npoint, a.k, dtype=torch.int32, device=self._device)))
            else:
                torch.cuda.current_stream().wait_event(old.done)
                out = _Prep(tuple(old))
            self._ring[self._next] = out
            self._next = (self._next + 1) % self.PREP_RING
        ws = self._keep['ws']
        a.f_rows, a.stages = f_rows.data_ptr(), 1
        a.pt, a.ps, a.knn_idx = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()
        try:
            lib.check(lib.load().dclr_merge_forward(ctypes.byref(a), events, lib.stream_ptr()), 'merge_forward')
        finally:
            a.pt, a.ps, a.knn_idx = ws['pt'].data_ptr(), ws['ps'].data_ptr(), ws['knn'].data_ptr()
        return out

    def run(self, f_rows: torch.Tensor, events=None, prep=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        f_rows = self._check(f_rows)
        a, ws = self.args, self._keep['ws']
        if out is not None and (out.shape != (self._pairs, self._n_out) or not out.is_contiguous()
                                or out.dtype != torch.float32 or out.device != f_rows.device):
            raise RuntimeError("out must be a contiguous float32 (pairs, label_dim) tensor on the same device")
        y = out if out is not None else torch.empty(self._pairs, self._n_out, device=self._device)
        a.f_rows, a.y, a.stages = f_rows.data_ptr(), y.data_ptr(), 3 if prep is None else 2
        if prep is not None:
            a.pt, a.ps, a.knn_idx = prep[0].data_ptr(), prep[1].data_ptr(), prep[2].data_ptr()
        try:
            lib.check(lib.load().dclr_merge_forward(ctypes.byref(a), events, lib.stream_ptr()), 'merge_forward')
        finally:
            a.pt, a.ps, a.knn_idx = ws['pt'].data_ptr(), ws['ps'].data_ptr(), ws['knn'].data_ptr()
        return y


class _CloudPlan:
    """Arguments + scratch of dclr_cloud_forward (sampling -> set abstraction -> layer-1 halves + kNN in ONE foreign call)
    for one (device, stream, launch shape, matrix path). The sampler's index list and spatial groups are scratch of the
    plan (reused by the next call on the same stream); the products -- rows F and the stage-1 buffers of the dense stages --
    come out of a small ring whose slots are handed out again once their consumer has released them (_Prep.release)."""

    RING = 3

    def __init__(self, args, keep, versions, merge_plan: _MergePlan, shapes, device):
        self.args, self._keep, self._versions, self._merge = args, keep, versions, merge_plan
        self._shapes, self._device = shapes, device
        self._ring, self._next = [], 0

    def current(self) -> bool:
        return self._versions == _MergePlan._version_key(self._keep['mods']) and self._merge.current()

    @classmethod
    def build(cls, sa, merge_plan: _MergePlan, device, per: int, nb: int, n: int, c: int):
        layout = ops.fps_group_layout(n)
        npoint, ns = sa.npoint, len(sa.radii)
        b = 2 * per * nb
        if layout is None or ns > lib.CLOUD_MAX_SCALES or (n > 16384 and npoint * 4 > 32 * 1024):
            return None
        ng, gs = layout
        a = lib.CloudArgs()
        a.b, a.n, a.c, a.npoint, a.pairs_per_batch, a.n_batches = b, n, c, npoint, per, nb
        a.f16, a.n_scales = int(ops.PRECISION == 'f16x2'), ns
        mlps = sa.packed_mlps()
        for i in range(ns):
            a.radii[i], a.nsamples[i], a.mlp[i] = float(sa.radii[i]), int(sa.nsamples[i]), mlps[i].data_ptr()
        scratch = {'idx': torch.empty(b, npoint, dtype=torch.int32, device=device),
                   'gpts': torch.empty(b, ng * gs, 4, device=device), 'gbox': torch.empty(b, ng, 8, device=device)}
        if n <= 16384 and gs > 64 and ops.SLICE_BOXES:
            scratch['sbox'] = torch.empty(b, ng * (gs // 64), 8, device=device)
        need = lib.load().dclr_fps_workspace_bytes(b, n) if n > 16384 else 0
        if need:
            scratch['ws'] = torch.empty((need + 3) // 4, dtype=torch.int32, device=device)
        a.fps_idx, a.group_pts, a.group_box = scratch['idx'].data_ptr(), scratch['gpts'].data_ptr(), scratch['gbox'].data_ptr()
        a.slice_box, a.workspace, a.workspace_bytes = lib.ptr(scratch.get('sbox')), lib.ptr(scratch.get('ws')), need
        a.merge = ctypes.addressof(merge_plan.args)
        keep = {'mods': [sa], 'scratch': scratch, 'mlps': mlps}
        k = merge_plan.args.k
        shapes = ((b * npoint, ops.F_STRIDE), (per * nb * npoint, 128), (per * nb, npoint, k))
        return cls(a, keep, _MergePlan._version_key(keep['mods']), merge_plan, shapes, device)

    def _fresh(self):
        rows_s, half_s, knn_s = self._shapes
        return _Prep((torch.empty(half_s, device=self._device), torch.empty(half_s, device=self._device),
                      torch.empty(knn_s, dtype=torch.int32, device=self._device), torch.empty(rows_s, device=self._device)))

    def fill_ring(self) -> None:
        """Allocate every slot of the ring now (released: free to hand out)."""
        while len(self._ring) < self.RING:
            slot = self._fresh()
            slot.release()
            self._ring.append(slot)

    def _slot(self):
        fresh = self._fresh
        if len(self._ring) < self.RING:
            out = fresh()
            self._ring.append(out)
            return out
        old = self._ring[self._next]
        if old.done is None:                            # never released by its consumer: leave it alone, take fresh memory
            out = fresh()
        else:
            torch.cuda.current_stream().wait_event(old.done)
            out = _Prep(tuple(old))
        self._ring[self._next] = out
        self._next = (self._next + 1) % self.RING
        return out

    def run(self, clouds: torch.Tensor, stride: int, events=None, merge_events=None):
        """clouds: the first batch (2 per, n, c) of the launch, the others `stride` floats apart -> (rows F, prep)."""
        a, m = self.args, self._merge.args
        if clouds.shape != (2 * a.pairs_per_batch, a.n, a.c):
            raise RuntimeError("clouds do not match the planned launch shape")
        out = self._slot()
        a.clouds, a.batch_stride, a.f_rows = lib.dev_f32(clouds, 'clouds').data_ptr(), stride, out[3].data_ptr()
        saved = (m.pt, m.ps, m.knn_idx)
        m.pt, m.ps, m.knn_idx = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()
        try:
            lib.check(lib.load().dclr_cloud_forward(ctypes.byref(a), events, merge_events, lib.stream_ptr()), 'cloud_forward')
        finally:
            m.pt, m.ps, m.knn_idx = saved
        return out[3], out


# --------------------------------------------------------------------------------------------------
class DeepCLR(BaseModel):
    """Set abstraction over all 2B clouds -> flow embedding per pair -> pose head."""

    def __init__(self, input_dim: int, label_type: LabelType, cloud_features: Config, merge: Config,
                 output: Config, transform: Optional[Config] = None, loss: Optional[Any] = None, **kwargs: Any):
        super().__init__()
        self._input_dim = input_dim
        cloud_features, merge, output = (Config.from_dict(c) if not isinstance(c, Config) else c
                                         for c in (cloud_features, merge, output))
        # optional module in front of the per-cloud features (reference deepclr.py:453-454,460-464: `_cloud_layers` is then
        # [transform, cloud features] and the state_dict keys shift by one). No shipped configuration names one and the
        # only per-cloud module class the reference has is SetAbstraction: such a model runs module by module.
        front = None
        if transform is not None:
            front = _init_module(transform if isinstance(transform, Config) else Config.from_dict(transform),
                                 input_dim=input_dim, **kwargs)
        cloud = _init_module(cloud_features, input_dim=input_dim if front is None else front.output_dim(), **kwargs)
        merge_layer = _init_module(merge, input_dim=cloud.output_dim(), **kwargs)
        head = _init_module(output, input_dim=merge_layer.output_dim(), label_type=label_type, **kwargs)
        self._cloud_layers = nn.Sequential(cloud) if front is None else nn.Sequential(front, cloud)
        self._merge_layers = nn.Sequential(merge_layer, head)
        # The fused row pipeline (rows F -> rows E -> pose) covers every shipped configuration; a configuration with other
        # layer widths / k / feature counts runs module by module in the reference's channel layout, every module composed
        # from the level-1 HIP operators where its own shape is not the fused one.
        self._rows_path = front is None and all(getattr(mod, 'rows_path', False) for mod in (cloud, merge_layer, head))
        # batch norm / dropout (`batch_norm: true`, `dropout` < 1): identity-like in eval mode (folded / skipped), but in
        # training mode they need batch statistics / random masks, which only the differentiable torch path provides
        self._train_only = any(isinstance(m, (nn.Dropout, nn.modules.batchnorm._BatchNorm)) for m in self.modules())
        self._plans: Dict[Any, Any] = {}
        self._range_ok = None                       # weights key of the last checked forward that passed (ops.CHECK_RANGE)
        self._range_flag = None                     # lib.MappedFlag: set by the split-f16 kernels when a clamp engages
        if loss is None:
            self._loss_layer = None
        elif isinstance(loss, list):
            self._loss_layer = AccumulatedLoss([_init_loss(c, label_type, **kwargs) for c in loss])
        else:
            self._loss_layer = _init_loss(loss, label_type, **kwargs)

    def get_input_dim(self) -> int:
        return self._input_dim

    def has_loss(self) -> bool:
        return self._loss_layer is not None

    def get_loss_weights(self) -> Dict:
        return self._loss_layer.get_weights() if self._loss_layer is not None else {}

    @property
    def npoint(self) -> int:
        return self._cloud_layers[-1].npoint

    def prepare(self) -> None:
        """Build every kernel-side (packed) weight buffer now, on the current stream. Optional: they are built on
        first use otherwise (PackedCache orders other streams behind that build)."""
        if not next(self.parameters()).is_cuda:
            return
        for mod in self.modules():
            names = ('packed_mlps', '_packed', '_packed_f16') if getattr(mod, 'rows_path', True) and self._rows_path \
                else ('packed_mlps', '_packed_composed', '_packed_plain')
            for name in names:
                fn = getattr(mod, name, None)
                if callable(fn) and mod is not self:
                    fn()

    # -- row-level pipeline (what bench.py and the sharded runner drive) ---------------------------
    def sample(self, x: torch.Tensor, view=None):
        """(2B, N, C) -> furthest-point sample (indices (2B, npoint) int32 + the kernel's spatial groups).
        view = ops.batch_view(batches): x is the first of several batches that are read where they lie; the result covers
        all of them in the order [templates of every batch | sources of every batch]."""
        return self._cloud_layers[0].sample(x, view)

    def cloud_feature_rows(self, x: torch.Tensor, sample=None, view=None) -> torch.Tensor:
        """(2B, N, C) -> rows F ((2B)*npoint, 68); sample: precomputed self.sample(x), else computed here."""
        sa0 = getattr(self._cloud_layers[0], '_sa0', None)
        if sa0 is not None and x.is_cuda:
            sa0.overflow_ptr = self._range_flag_ptr()      # the split-f16 set-abstraction layers report a clamp there too
        return self._cloud_layers[0].forward_rows(x, sample, view)

    def merge_prep(self, f_rows: torch.Tensor, pairs: int):
        """The part of merge_rows that needs nothing but the feature rows (per-point halves of flow layer 1, kNN),
        for callers that run it ahead on another stream; None where the one-call path does not apply."""
        plan = self._merge_plan(f_rows, pairs)
        if plan is None:
            return None
        events = ops.TIMER.merge_events(pairs, self.npoint, plan.args.k, plan.args.n_fc, 1) if ops.TIMER is not None else None
        return plan.prep(f_rows, events)

    def cloud_merge_prep(self, x: torch.Tensor, view=None):
        """cloud_feature_rows + merge_prep behind ONE foreign call (dclr_cloud_forward): (2B, N, C) [x view] -> (rows F,
        prep), or None where the one-call path does not apply (a second set-abstraction level, shapes without a grouped
        sampler or a merge plan, the first -- range-checked -- forward after the weights changed): callers then take the
        two methods one after the other. The pipelined runner's sampling chain: ~0.3 ms of host time per launch otherwise."""
        sa = self._cloud_layers[0]
        sa0 = getattr(sa, '_sa0', None)
        if not self._rows_path or sa0 is None or getattr(sa, '_sa1', None) is not None or not sa0.fused or not x.is_cuda \
                or x.shape[0] % 2 or x.shape[2] != self._input_dim or not x.is_contiguous() or x.dtype != torch.float32 \
                or os.environ.get('DCLR_CLOUD_FUSED', '1') == '0':
            return None
        if ops.PRECISION == 'f16x2' and ops.CHECK_RANGE != 'never' and (
                ops.CHECK_RANGE == 'always' or self._range_unchecked() or sa0.range_unchecked()):
            return None
        self.check_range()
        per, nb, stride = view if view is not None else (x.shape[0] // 2, 1, 0)
        plan = self._cloud_plan(sa0, x, per, nb)
        if plan is None:
            return None
        events = merge_events = None
        if ops.TIMER is not None:
            events = ops.TIMER.cloud_events(2 * per * nb, x.shape[1])
            merge_events = ops.TIMER.merge_events(per * nb, self.npoint, plan._merge.args.k, plan._merge.args.n_fc, 1)
        return plan.run(x, stride, events, merge_events)

    def _cloud_plan(self, sa0, x: torch.Tensor, per: int, nb: int):
        key = ('cloud', x.device, per, nb, x.shape[1], ops.PRECISION, ops.CHECK_RANGE == 'never', lib.stream_ptr())
        plan = self._plan_lookup(key)
        if plan is None or not plan.current():
            merge_plan = self._merge_plan(x, per * nb)
            if merge_plan is None:
                return None
            plan = _CloudPlan.build(sa0, merge_plan, x.device, per, nb, x.shape[1], x.shape[2])
            if plan is None:
                return None
            self._plan_store(key, plan)
        return plan

    PLAN_CACHE = 48                                    # launch plans kept: (side streams x 2 + dense streams) of a runner, with room

    def _plan_lookup(self, key):
        plan = self._plans.get(key)
        if plan is not None:
            self._plans[key] = self._plans.pop(key)    # most recently used last (dicts keep insertion order)
        return plan

    def _plan_store(self, key, plan) -> None:
        """Keep the plan; beyond PLAN_CACHE entries the LEAST recently used one goes (its scratch stays alive while a caller
        still holds it). Clearing the whole cache, as rounds 3-4 did at 16 entries, made every launch of a runner with more
        streams than that rebuild its arguments, scratch and output ring inside the timed window."""
        self._plans.pop(key, None)
        self._plans[key] = plan
        while len(self._plans) > self.PLAN_CACHE:
            self._plans.pop(next(iter(self._plans)))

    def plan_cloud_forward(self, x: torch.Tensor, view=None) -> bool:
        """Build now, for the CURRENT stream, what cloud_merge_prep(x, view) would build on its first use there (arguments,
        scratch, every slot of the output ring: ~0.3 ms of host time and a dozen allocations). The pipelined runner calls
        it for each of its side streams when it sees the first batch, so that no launch inside a timed window pays it."""
        sa = self._cloud_layers[0]
        sa0 = getattr(sa, '_sa0', None)
        if not self._rows_path or sa0 is None or getattr(sa, '_sa1', None) is not None or not sa0.fused or not x.is_cuda \
                or x.shape[0] % 2 or x.shape[2] != self._input_dim or os.environ.get('DCLR_CLOUD_FUSED', '1') == '0':
            return False
        per, nb, _ = view if view is not None else (x.shape[0] // 2, 1, 0)
        plan = self._cloud_plan(sa0, x, per, nb)
        if plan is None:
            return False
        plan.fill_ring()
        return True

    def merge_rows(self, f_rows: torch.Tensor, pairs: int, events=None, prep=None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Rows F -> pose outputs (pairs, label_dim). Shapes the one-call path covers (MotionEmbedding +
        OutputSimple, fusable head) go through dclr_merge_forward: one foreign call and one allocation per batch
        instead of ten and a dozen -- at ~0.3 ms per step the host would otherwise set the pace."""
        self._refuse_training_modules()
        self.check_range()
        if ops.PRECISION == 'f16x2' and ops.CHECK_RANGE != 'never' \
                and (ops.CHECK_RANGE == 'always' or self._range_unchecked()):
            return self._merge_rows_checked(f_rows, pairs, out)
        plan = self._merge_plan(f_rows, pairs)
        if plan is not None:
            if events is None and ops.TIMER is not None:
                events = ops.TIMER.merge_events(pairs, self.npoint, plan.args.k, plan.args.n_fc,
                                                3 if prep is None else 2)               # per-stage HIP events
            return plan.run(f_rows, events, prep, out)
        e_rows = self._merge_layers[0].forward_rows(f_rows, pairs, self.npoint)
        y = self._merge_layers[1].forward_rows(e_rows, pairs)
        return y if out is None else out.copy_(y)

    def _refuse_training_modules(self) -> None:
        if self.training and self._train_only:
            raise RuntimeError("this model has batch norm / dropout layers and is in training mode: the inference kernels fold "
                               "the running statistics and skip dropout, which is what eval() means -- call model.eval() "
                               "(ModelInferenceHelper does); a training step runs with gradients enabled")

    def _range_flag_ptr(self) -> Optional[int]:
        """Device address of the word the split-f16 kernels report a clamped activation to -- and read back: while it is set,
        the fused dense stages write their poses as NaN (csrc/gemm.hip fc_kernel). None on the f32 matrix path and with
        CHECK_RANGE = 'never' (the caller opted out: nothing is tracked, nothing poisoned)."""
        if ops.PRECISION != 'f16x2' or ops.CHECK_RANGE == 'never':
            return None
        if self._range_flag is None:
            self._range_flag = lib.MappedFlag()
        return self._range_flag.dev_ptr

    def check_range(self, synchronize=False) -> None:
        """Raise if a split-f16 forward that has COMPLETED since the last check clamped an activation at 65504 (its poses
        are wrong). The fused kernels set one word of mapped host memory when a clamp engages (csrc/mma16f.h
        dclr_report_overflow); reading it costs no device synchronisation, so every entry into the model looks at it --
        forward(), merge_rows(), the pipelined runner's steps -- and a caller that wants the verdict for work still in
        flight passes synchronize=True (the whole device) or 'stream' (the current stream). CHECK_RANGE='first' covers the first forward of a checkpoint (f32 re-run with
        a message that names the peak); this flag covers every later input."""
        flag = self._range_flag
        if flag is None:
            return
        if ops.CHECK_RANGE == 'never':              # the caller opted out of range checks: nothing is reported, nothing kept
            flag.clear()
            return
        if synchronize == 'stream':
            # the CURRENT stream only (ModelInferenceHelper.finish: a predict call's work is all there), polled for up to
            # ~2 ms before blocking: a blocking wait returns ~20 us after the work is done, which a caller that times single
            # pairs (scripts/timing.py) would see in every call
            ev = torch.cuda.Event()
            ev.record()
            t_end = time.perf_counter() + 2e-3
            while not ev.query():
                if time.perf_counter() > t_end:
                    ev.synchronize()
                    break
        elif synchronize:
            torch.cuda.synchronize()
        if flag.is_set():
            flag.clear()
            raise RuntimeError("split-f16 matrix path out of range: an activation exceeded 65504 and was clamped in a forward "
                               "pass completed since the last check -- the poses of that pass are wrong. Run this "
                               "checkpoint with DCLR_PRECISION=f32.")

    def _range_key(self):
        return tuple((p.data_ptr(), p._version) for m in self._merge_layers for p in flat_parameters(m))

    def _range_unchecked(self) -> bool:
        """True until a checked forward has passed for the current weights (any in-place change bumps a version)."""
        return self._range_ok != self._range_key()

    def _merge_rows_checked(self, f_rows: torch.Tensor, pairs: int, out: Optional[torch.Tensor]) -> torch.Tensor:
        """The dense stages on the split-f16 path AND on the f32 matrix instructions; raises when an operand left the f16
        range (the split path clamps at +-65504 and would return wrong poses silently). Runs on the first forward after
        the weights changed (ops.CHECK_RANGE = 'first', the default: two host syncs, once) or on every forward ('always')."""
        flow, head = self._merge_layers[0], self._merge_layers[1]
        k = getattr(getattr(flow, '_embedding', None), '_k', 0)
        if k > 0 and bool((ops.knn_rows(f_rows, pairs, self.npoint, k) < 0).any()):
            # upstream fails here too: torch_cluster.knn returns fewer than k neighbours for such a query and
            # KnnGrouping's .view(2, G, k) raises (reference deepclr.py:164-167). The unchecked forwards mask the
            # unfilled slots instead (csrc/flow16.hip) -- memory-safe, but not a result upstream would have produced.
            raise RuntimeError("kNN grouping: a template point has fewer than k = {} source points within the search's "
                               "start distance (1e5 m), or non-finite coordinates reached the flow embedding".format(k))
        e16 = flow.forward_rows(f_rows, pairs, self.npoint, 'f16x2')
        y16 = head.forward_rows(e16, pairs, 'f16x2')
        e32 = flow.forward_rows(f_rows, pairs, self.npoint, 'f32')
        y32 = head.forward_rows(e32, pairs, 'f32')
        peak = max(float(f_rows.abs().max()), float(e32.abs().max()))
        err = float((y16 - y32).abs().max())
        if not (err <= 1e-4 * max(1.0, float(y32.abs().max()))) or not peak < ops.F16_MAX:
            raise RuntimeError("split-f16 matrix path out of range: activations reach {:.4g} (limit 65504) and the pose "
                               "outputs differ from the f32 matrix path by {:.3g}; run this checkpoint with "
                               "DCLR_PRECISION=f32".format(peak, err))
        self._range_ok = self._range_key()
        self._merge_plan(f_rows, pairs)             # the next forward of this shape takes the one-call path: build its plan now
        return y16 if out is None else out.copy_(y16)

    def _merge_plan(self, f_rows: torch.Tensor, pairs: int):
        flow, head = self._merge_layers[0], self._merge_layers[1]
        if os.environ.get('DCLR_MERGE_FUSED', '1') == '0' or not isinstance(flow, MotionEmbedding) \
                or not isinstance(head, OutputSimple) or not self._rows_path:
            return None
        # the workspace belongs to one stream: calls enqueued on different streams may run side by side
        key = (f_rows.device, pairs, self.npoint, ops.PRECISION, ops.CHECK_RANGE == 'never', lib.stream_ptr())
        plan = self._plan_lookup(key)
        if plan is None or not plan.current():
            plan = _MergePlan.build(flow._embedding, head, f_rows.device, pairs, self.npoint, self._range_flag_ptr())
            if plan is None:
                return None
            self._plan_store(key, plan)
        return plan

    @property
    def label_dim(self) -> int:
        return self._merge_layers[1].output_dim()

    def sequence_rows(self, f_rows: torch.Tensor, frames: int, carry: Optional[torch.Tensor] = None):
        """Rows F of `frames` consecutive clouds (+ the rows of the frame before them, if any) -> the pair
        layout merge_rows() takes: templates = every frame but the last, sources = every frame but the first.
        Returns (rows, pairs, rows of the last frame)."""
        v = f_rows.view(frames, self.npoint, f_rows.shape[-1])
        if carry is not None:
            v = torch.cat((carry.view(1, self.npoint, -1), v))
        pairs = v.shape[0] - 1
        rows = torch.cat((v[:-1], v[1:])).view(2 * pairs * self.npoint, -1) if pairs > 0 else None
        return rows, pairs, v[-1].clone()

    @staticmethod
    def _augment(x: torch.Tensor, m: torch.Tensor) -> None:
        """In-place homogeneous transform of the point columns (reference: deepclr.py:512-514)."""
        dim = m.shape[-1] - 1
        pts = x[:, :, :dim]
        x[:, :, :dim] = torch.baddbmm(m[:, :dim, dim].unsqueeze(1), pts, m[:, :dim, :dim].transpose(1, 2))

    def cloud_features(self, x: torch.Tensor, m: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(2B, N, C) -> (2B, 3 + feat, npoint) in the reference's channel-major layout."""
        if m is not None:
            self._augment(x, m)
        if not self._rows_path:                       # reference: x.transpose(1, 2) -> cloud layers (deepclr.py:516-520)
            return self._cloud_layers(x.transpose(1, 2).contiguous())
        rows = self.cloud_feature_rows(x.contiguous())
        return ops.rows_to_channels(rows, x.shape[0], self.npoint, self._cloud_layers[0].output_dim() - 3)

    def forward(self, x: torch.Tensor, is_feat: bool = False, m: Optional[torch.Tensor] = None,
                y: Optional[torch.Tensor] = None, debug: bool = False)\
            -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[Dict]]:
        if x.shape[0] % 2 != 0:
            raise RuntimeError("batch must hold templates followed by the same number of sources")
        pairs = x.shape[0] // 2
        f_rows = None
        if not torch.is_grad_enabled():
            self._refuse_training_modules()
        if self.training and torch.is_grad_enabled():
            # Training step (reference: engine/engines.py:57-84 runs forward with m and y, then loss.backward()): module by
            # module in the reference's channel layout with a gradient -- sampling / ball query / kNN on the HIP operators
            # (indices), gather and group through the HIP operators and their HIP backward, MLPs and reductions in torch.
            # The fused inference kernels have no backward; model.eval() (ModelInferenceHelper does it) selects them.
            if is_feat:
                feat = x
            else:
                if m is not None:
                    self._augment(x, m)
                feat = x.transpose(1, 2).contiguous()
                for layer in self._cloud_layers:
                    feat = layer.forward_train(feat)
            y_pred = self._merge_layers[1].forward_train(self._merge_layers[0].forward_train(feat))
        elif not self._rows_path:
            # module by module in the reference's channel layout (deepclr.py:494-499), each module on the HIP operators
            feat = x if is_feat else self.cloud_features(x, m)
            y_pred = self._merge_layers[1](self._merge_layers[0](feat))
        else:
            if is_feat:
                f_rows = ops.channels_to_rows(x.contiguous(), ops.F_STRIDE)
            else:
                if m is not None:
                    self._augment(x, m)
                f_rows = self.cloud_feature_rows(x.contiguous())
            y_pred = self.merge_rows(f_rows, pairs)
        if self._loss_layer is None or y is None:
            return y_pred, None, None
        # loss value as the reference returns it (deepclr.py:500-503); debug carries the (augmented, set-abstracted)
        # clouds in the reference's channel layout
        loss = self._loss_layer(y_pred, y)
        aux = None
        if debug:
            nfeat = self._cloud_layers[-1].output_dim() - 3
            aux = {'x_aug': x if is_feat else (feat if f_rows is None else
                                               ops.rows_to_channels(f_rows, x.shape[0], self.npoint, nfeat))}
        return y_pred, loss, aux
