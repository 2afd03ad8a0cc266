"""CPU restatement of DeepCLR's forward pass -- TEST INFRASTRUCTURE ONLY.

Restates, in functional form over a reference-layout ``state_dict``:

  * ``PointnetSAModuleMSG`` of the absent pointnet2 package (published module:
    FPS -> gather -> per scale [ball query -> group -> subtract centroid ->
    cat(xyz, feat) -> SharedMLP(1x1 conv + ReLU) -> max over nsample] -> cat);
    constructed by the reference at /root/reference/deepclr/models/deepclr.py:63-70
    with ``use_xyz=True`` and ``bn=batch_norm``.
  * ``split_features`` / ``merge_features`` / ``SetAbstraction.forward``
    (/root/reference/deepclr/models/deepclr.py:30-45, 88-94).
  * ``KnnGrouping`` + ``MotionEmbeddingBase.forward`` (deepclr.py:142-173, 201-231).
  * ``OutputSimple.forward`` + ``_output_activation`` (deepclr.py:275-294).
  * ``Conv1d``/``Linear`` = affine [+ batch norm] + ReLU after EVERY layer
    (/root/reference/deepclr/models/helper.py:27-38, 57-65); inference (eval mode):
    batch norm applies its running statistics, dropout (helper.py:77-85,107-113) is
    the identity and only shifts the layers' indices in the state_dict.
  * ``DeepCLR.forward`` / ``cloud_features`` (deepclr.py:488-521), inference
    branch only (``m is None``, ``y is None``).

The composition is pinned against the reference's own Python (see
tests/golden/make_golden.py); the primitives it calls are "parity unpinned"
(oracle/primitives.c header).
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import primitives as P


class OracleSAModuleMSG(nn.Module):
    """Restated multi-scale-grouping set abstraction with the upstream parameter
    naming ``mlps.{scale}.layer{j}.conv.{weight,bias}`` (weights (out,in,1,1),
    kaiming-normal, zero bias)."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = False, use_xyz: bool = True):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint, self.radii, self.nsamples, self.use_xyz = npoint, list(radii), list(nsamples), use_xyz
        self.mlps = nn.ModuleList()
        for spec in mlps:
            spec = list(spec)
            if use_xyz:
                spec[0] += 3
            stack = nn.Sequential()
            for j in range(len(spec) - 1):
                # published pytorch_utils.Conv2d: conv (no bias when a norm layer follows) -> [bn.bn] -> ReLU
                unit = nn.Sequential()
                conv = nn.Conv2d(spec[j], spec[j + 1], kernel_size=(1, 1), bias=not bn)
                nn.init.kaiming_normal_(conv.weight)
                if not bn:
                    nn.init.constant_(conv.bias, 0)
                unit.add_module('conv', conv)
                if bn:
                    wrap = nn.Sequential()
                    wrap.add_module('bn', nn.BatchNorm2d(spec[j + 1]))
                    unit.add_module('bn', wrap)
                unit.add_module('activation', nn.ReLU(inplace=True))
                stack.add_module('layer{}'.format(j), unit)
            self.mlps.append(stack)

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor] = None)\
            -> Tuple[torch.Tensor, torch.Tensor]:
        weights = [[(u.conv.weight, u.conv.bias, u.bn.bn if hasattr(u, 'bn') else None) for u in stack] for stack in self.mlps]
        return sa_msg_forward(xyz, features, self.npoint, self.radii, self.nsamples, weights, self.use_xyz)


def sa_msg_forward(xyz, features, npoint, radii, nsamples, weights, use_xyz=True):
    """xyz (B,N,3), features (B,C,N)|None -> new_xyz (B,npoint,3), new_features (B,sum(out),npoint)."""
    xyz = xyz.contiguous()
    xyz_flipped = xyz.transpose(1, 2).contiguous()
    fps_idx = P.furthest_point_sample(xyz, npoint)
    new_xyz = P.gather_operation(xyz_flipped, fps_idx).transpose(1, 2).contiguous()
    outs = []
    for radius, nsample, layers in zip(radii, nsamples, weights):
        idx = P.ball_query(radius, nsample, xyz, new_xyz)
        grouped = P.grouping_operation(xyz_flipped, idx)               # (B,3,npoint,nsample)
        grouped = grouped - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            gf = P.grouping_operation(features.contiguous(), idx)
            grouped = torch.cat([grouped, gf], dim=1) if use_xyz else gf
        h = grouped
        for w, bias, *norm in layers:                                  # norm: a BatchNorm2d module, a dict of its tensors, or absent
            h = _batch_norm(F.conv2d(h, w, bias), norm[0] if norm else None)
            h = F.relu(h)
        outs.append(h.max(dim=3)[0])                                   # max_pool2d over nsample
    return new_xyz, torch.cat(outs, dim=1)


def _batch_norm(h: torch.Tensor, norm) -> torch.Tensor:
    """Eval-mode batch norm (running statistics) from a module or from {'weight', 'bias', 'running_mean', 'running_var'}."""
    if norm is None:
        return h
    if isinstance(norm, nn.Module):
        assert not norm.training, "the oracle restates inference"
        norm = {'weight': norm.weight, 'bias': norm.bias, 'running_mean': norm.running_mean, 'running_var': norm.running_var}
    return F.batch_norm(h, norm['running_mean'], norm['running_var'], norm['weight'], norm['bias'], training=False, eps=1e-5)


def _norm_of(sd: Dict[str, torch.Tensor], base: str):
    return {k: sd['{}.{}'.format(base, k)] for k in ('weight', 'bias', 'running_mean', 'running_var')} \
        if base + '.running_mean' in sd else None


def _mlp1d(h: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, n_layers: int, step: int = 1) -> torch.Tensor:
    """step 2: Dropout modules sit between the layers (identity at inference; the layers keep the even indices)."""
    for j in range(n_layers):
        base = '{}._sequential.{}._sequential'.format(prefix, j * step)
        w, b = sd[base + '.0.weight'], sd[base + '.0.bias']
        h = F.conv1d(h, w, b) if w.dim() == 3 else F.linear(h, w, b)
        h = F.relu(_batch_norm(h, _norm_of(sd, base + '.1')))
    return h


class OracleDeepCLR:
    """Functional restatement of the reference network over its state_dict."""

    def __init__(self, model_cfg: dict, state_dict: Dict[str, torch.Tensor]):
        self.cfg = model_cfg
        self.sd = {k: v.detach().to('cpu', torch.float32) for k, v in state_dict.items()}
        prm = model_cfg['params']
        self.lin_step = 2 if float(prm.get('dropout', 1.0)) < 1.0 else 1     # helper.py:107-113: Dropout behind every Linear
        self.input_dim = int(model_cfg['input_dim'])
        self.point_dim = int(model_cfg['point_dim'])
        # per-cloud modules in `_cloud_layers` order: an optional `transform` module, then the cloud features
        # (deepclr.py:453-464); the reference's only per-cloud module class is SetAbstraction
        mods = ([prm['transform']] if prm.get('transform') else []) + [prm['cloud_features']]
        # one entry per (module, set-abstraction level) (deepclr.py:63-83): level 1's mlp specs START with their input
        # width, level 0's do not (deepclr.py:61 vs 73)
        self.sa_levels = []
        for mi, mod in enumerate(mods):
            sa = mod['params']
            assert mod['name'] == 'SetAbstraction' and 1 <= len(sa['npoint']) <= 2
            self.sa_levels += [{'module': mi, 'level': lv, 'npoint': int(sa['npoint'][lv]), 'radii': [float(r) for r in sa['radii'][lv]],
                                'nsamples': [int(s) for s in sa['nsamples'][lv]],
                                'layers': [len(m) - (1 if lv == 1 else 0) for m in sa['mlps'][lv]]}
                               for lv in range(len(sa['npoint']))]
        self.npoint = self.sa_levels[-1]['npoint']
        me = prm['merge']['params']
        assert prm['merge']['name'] == 'MotionEmbedding'
        self.k, self.radius = int(me['k']), float(me['radius'])
        self.me_layers = len(me['mlp'])
        self.append_features = bool(me.get('append_features', True))
        out = prm['output']['params']
        assert prm['output']['name'] == 'OutputSimple'
        self.head_conv_layers, self.head_lin_layers = len(out['mlp']), len(out['linear']) - 1
        self.label_type = str(model_cfg['label_type']).upper()

    # -- deepclr.py:510-521 (m is None) + SetAbstraction.forward 88-94 -------------------------
    def cloud_features(self, x: torch.Tensor) -> torch.Tensor:
        x = x.to(torch.float32).transpose(1, 2)                        # (2B, C, N)
        xyz = x[:, :3, :].transpose(1, 2).contiguous()
        feats = x[:, 3:, :].contiguous() if x.size(1) > 3 else None
        for level in self.sa_levels:                                   # deepclr.py:90-93, module after module (516-520)
            def layer(s, j):
                base = '_cloud_layers.{}._sa{}.mlps.{}.layer{}'.format(level['module'], level['level'], s, j)
                return self.sd[base + '.conv.weight'], self.sd.get(base + '.conv.bias'), _norm_of(self.sd, base + '.bn.bn')
            weights = [[layer(s, j) for j in range(n)] for s, n in enumerate(level['layers'])]
            xyz, feats = sa_msg_forward(xyz, feats, level['npoint'], level['radii'], level['nsamples'], weights)
        return torch.cat((xyz.transpose(1, 2), feats), dim=1)          # (2B, 3+F, npoint)

    # -- deepclr.py:142-173 ---------------------------------------------------------------------
    def knn_groups(self, cloud0: torch.Tensor, cloud1: torch.Tensor):
        def flat(c):
            pts = c.transpose(1, 2).contiguous().view(-1, c.shape[1])
            batch = torch.arange(c.shape[0]).view(-1, 1).repeat(1, c.shape[2]).view(-1)
            return pts, batch
        pts0, batch0 = flat(cloud0)
        pts1, batch1 = flat(cloud1)
        gi = P.knn(pts1[:, :self.point_dim].contiguous(), pts0[:, :self.point_dim].contiguous(),
                   self.k, batch1, batch0).view(2, pts0.shape[0], self.k)
        return pts0, pts1, gi

    # -- deepclr.py:201-231, 243-246 ------------------------------------------------------------
    def flow_embedding(self, clouds: torch.Tensor) -> torch.Tensor:
        half = clouds.shape[0] // 2
        c0, c1 = clouds[:half], clouds[half:]
        if self.k == 0:
            # GlobalGrouping (deepclr.py:108-139, selected at 186-187): every template point is grouped with ALL points
            # of its pair's source cloud, in index order
            pts0 = c0.transpose(1, 2).contiguous().view(-1, c0.shape[1])
            pts1 = c1.transpose(1, 2).contiguous().view(-1, c1.shape[1])
            p0, p1 = c0.shape[2], c1.shape[2]
            idx0 = torch.arange(pts0.shape[0]).view(-1, 1).repeat(1, p1)
            idx1 = torch.arange(pts1.shape[0]).view(c1.shape[0], -1).repeat(1, p0).view(idx0.shape)
            g0, g1 = pts0[idx0], pts1[idx1]
        else:
            pts0, pts1, gi = self.knn_groups(c0, c1)
            g0, g1 = pts0[gi[0]], pts1[gi[1]]
        d = self.point_dim
        pos_diff = g1[:, :, :d] - g0[:, :, :d]
        if self.append_features:
            merged = torch.cat((pos_diff, g0[:, :, d:], g1[:, :, d:]), dim=2)
        else:
            merged = torch.cat((pos_diff, g1[:, :, d:] - g0[:, :, d:]), dim=2)
        feat = _mlp1d(merged.transpose(1, 2), self.sd, '_merge_layers.0._embedding._conv', self.me_layers)
        if self.radius > 0.0:
            mask = torch.norm(pos_diff, dim=2) >= self.radius
            feat = feat.masked_fill(mask.unsqueeze(1), 0.0)
        feat = feat.max(dim=2)[0]
        out = torch.cat((pts0[:, :d], feat), dim=1)
        return out.view(c0.shape[0], -1, out.shape[1]).transpose(1, 2).contiguous()

    # -- deepclr.py:275-294 ---------------------------------------------------------------------
    def pose_head(self, x: torch.Tensor) -> torch.Tensor:
        h = _mlp1d(x, self.sd, '_merge_layers.1.conv', self.head_conv_layers).max(dim=2)[0]
        h = _mlp1d(h, self.sd, '_merge_layers.1.linear', self.head_lin_layers, self.lin_step)
        y = F.linear(h, self.sd['_merge_layers.1.output.weight'], self.sd['_merge_layers.1.output.bias'])
        y = y.clone()
        if self.label_type == 'POSE3D_QUAT':
            y[:, 3] = torch.sigmoid(y[:, 3])
            y[:, 4:] = torch.tanh(y[:, 4:])
        elif self.label_type == 'POSE3D_DUAL_QUAT':
            y[:, 0] = torch.sigmoid(y[:, 0])
            y[:, 1:4] = torch.tanh(y[:, 1:4])
        return y

    # -- deepclr.py:488-508 ---------------------------------------------------------------------
    def forward(self, x: torch.Tensor, is_feat: bool = False) -> torch.Tensor:
        with torch.no_grad():
            if not is_feat:
                x = self.cloud_features(x)
            return self.pose_head(self.flow_embedding(x))

    __call__ = forward


def build_oracle_model(model_cfg: dict, state_dict: Dict[str, torch.Tensor]) -> OracleDeepCLR:
    return OracleDeepCLR(model_cfg, state_dict)
