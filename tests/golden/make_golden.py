#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own composition code.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box). It loads the reference's leaf modules by path --
deepclr/models/{base,helper,deepclr}.py, deepclr/data/labels.py,
deepclr/config/config.py, deepclr/utils/{factory,quaternion,metrics}.py --
registers the oracle's restated primitives under the names of the third-party
packages that are absent from this image (pointnet2.PointnetSAModuleMSG,
torch_cluster.knn, transforms3d.quaternions.*; torchgeometry and ignite are
inert placeholders that the inference branch never calls), runs the reference
``DeepCLR.forward`` / ``cloud_features`` / ``LabelType.to_matrix`` on seeded
inputs, checks that ``oracle/model.py`` reproduces every intermediate, and
writes the vectors as fixtures. What this pins: the oracle's restatement of the
reference-owned composition. What it cannot pin: the third-party primitives
themselves (their sources are not in the reference tree).

Usage:  python tests/golden/make_golden.py  [--reference /root/reference]
"""
import argparse
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

sys.path.insert(0, os.path.dirname(HERE))
import oracle                                    # noqa: E402
import helpers                                   # noqa: E402  (tests/helpers.py: configurations shared with the tests)
from oracle import labels as olabels             # noqa: E402
from deepclr_amd import synthetic                # noqa: E402


def _load_reference(ref_root: str):
    def pkg(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    for name in ('deepclr', 'deepclr.config', 'deepclr.data', 'deepclr.utils', 'deepclr.models'):
        pkg(name)

    tc = types.ModuleType('torch_cluster')
    tc.knn = oracle.knn
    sys.modules['torch_cluster'] = tc

    pn = types.ModuleType('pointnet2')
    pn.PointnetSAModuleMSG = oracle.OracleSAModuleMSG
    sys.modules['pointnet2'] = pn

    tgm = types.ModuleType('torchgeometry')       # only reached when m is not None (training)
    sys.modules['torchgeometry'] = tgm

    t3d = types.ModuleType('transforms3d')
    t3d.quaternions = types.ModuleType('transforms3d.quaternions')
    t3d.quaternions.quat2mat = olabels.quat2mat
    t3d.quaternions.qmult = olabels.qmult
    t3d.quaternions.qconjugate = olabels.qconjugate
    sys.modules['transforms3d'] = t3d
    sys.modules['transforms3d.quaternions'] = t3d.quaternions

    ign = pkg('ignite')
    ign_utils = types.ModuleType('ignite._utils')
    ign_utils.convert_tensor = lambda x, device=None, non_blocking=False: x.to(device) if device else x
    sys.modules['ignite._utils'] = ign_utils
    ign._utils = ign_utils

    def leaf(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(ref_root, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    leaf('deepclr.config.config', 'deepclr/config/config.py')
    leaf('deepclr.data.labels', 'deepclr/data/labels.py')
    leaf('deepclr.utils.factory', 'deepclr/utils/factory.py')
    leaf('deepclr.utils.tensor', 'deepclr/utils/tensor.py')
    leaf('deepclr.utils.quaternion', 'deepclr/utils/quaternion.py')
    leaf('deepclr.utils.metrics', 'deepclr/utils/metrics.py')
    leaf('deepclr.models.base', 'deepclr/models/base.py')
    leaf('deepclr.models.helper', 'deepclr/models/helper.py')
    return leaf('deepclr.models.deepclr', 'deepclr/models/deepclr.py')


def _ref_model(ref, cfg: dict, sd):
    Config = sys.modules['deepclr.config.config'].Config
    LabelType = sys.modules['deepclr.data.labels'].LabelType

    def sub(d):
        c = Config(allow_dynamic_params=True)
        c.read_dict(d)
        return c

    prm = cfg['params']
    model = ref.DeepCLR(input_dim=cfg['input_dim'], label_type=LabelType.create(cfg['label_type']),
                        cloud_features=sub(prm['cloud_features']), merge=sub(prm['merge']),
                        output=sub(prm['output']), transform=sub(prm['transform']) if prm.get('transform') else None,
                        point_dim=cfg['point_dim'],
                        batch_norm=prm['batch_norm'], dropout=prm['dropout'])
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.eval()
    return model, LabelType.create(cfg['label_type'])


def loss_golden(ref, out_dir: str) -> None:
    """Loss values of the reference's own loss modules (deepclr/models/deepclr.py:297-412 on utils/metrics.py)."""
    LabelType = sys.modules['deepclr.data.labels'].LabelType
    lt = LabelType.create('POSE3D_DUAL_QUAT')
    rng = np.random.default_rng(5)
    y_pred = torch.from_numpy(rng.normal(size=(8, 8)).astype(np.float32))
    y = torch.from_numpy(rng.normal(size=(8, 8)).astype(np.float32))
    out = {'y_pred': y_pred.numpy(), 'y': y.numpy()}
    for p in (1, 2):
        fixed = ref.TransformLoss(lt, p=p, sx=1.5, sq=40.0)
        learned = ref.TransformUncertaintyLoss(lt, p=p, sx=0.3, sq=-2.5)
        both = ref.AccumulatedLoss([fixed, ref.TransformLoss(lt, p=p, sx=0.5, sq=2.0)])   # same-shaped terms only
        t, r = ref.TransformLossCalculation(lt, p)(y_pred, y)
        out['p%d' % p] = np.array([t.item(), r.item(), fixed(y_pred, y).item(), learned(y_pred, y).item(),
                                   both(y_pred, y).item()], dtype=np.float64)
    np.savez_compressed(os.path.join(out_dir, 'losses.npz'), **out)
    print('wrote losses.npz')


def _sha(t: torch.Tensor) -> str:
    return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()


def _sample(t: torch.Tensor, n: int, seed: int):
    flat = t.contiguous().view(-1)
    pos = np.random.default_rng(seed).choice(flat.numel(), size=min(n, flat.numel()), replace=False)
    pos.sort()
    return pos.astype(np.int64), flat[torch.from_numpy(pos)].numpy()


def small_cfg() -> dict:
    cfg = synthetic.model_cfg('kitti')
    sa = cfg['params']['cloud_features']['params']
    sa['npoint'], sa['radii'], sa['nsamples'] = [64], [[2.0, 4.0]], [[8, 16]]
    cfg['params']['merge']['params'].update(k=8, radius=6.0)
    return cfg


def degenerate_batch(n_pairs: int, n_points: int, c: int, seed: int) -> np.ndarray:
    """torch.rand-style clouds in [0,1]^c (the reference test's input, tests/model/test_deepclr.py:19,42)
    with every 7th point duplicated, so FPS / ball-query / kNN ties are exercised."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.0, 1.0, size=(2 * n_pairs, n_points, c)).astype(np.float32)
    x[:, 7::7, :] = x[:, 6:-1:7, :][:, :x[:, 7::7, :].shape[1], :]
    return x


def small_global_cfg() -> dict:
    """k == 0: GlobalGrouping instead of kNN (reference: deepclr.py:186-187), on the reduced sizes."""
    cfg = small_cfg()
    cfg['params']['merge']['params'].update(k=0, radius=6.0)
    return cfg


def small_two_level_cfg() -> dict:
    """Two set-abstraction levels (reference: deepclr.py:72-83,92-93): 512 points -> 128 centroids x 64 features ->
    64 centroids x (32 + 32) features."""
    cfg = small_cfg()
    sa = cfg['params']['cloud_features']['params']
    sa['npoint'], sa['radii'], sa['nsamples'] = [128, 64], [[2.0, 4.0], [4.0, 8.0]], [[8, 16], [8, 24]]
    sa['mlps'] = [[[16, 16, 32], [16, 16, 32]], [[64, 32, 32], [64, 48, 32]]]
    return cfg


CASES = [
    # name,              cfg factory,                        batch factory,                              weight seed, full
    ('small_kitti_n512_b2', small_cfg, lambda: synthetic.make_batch('kitti', 2, 512), 11, True),
    ('kitti_rand_n96_b2', lambda: synthetic.model_cfg('kitti'), lambda: degenerate_batch(2, 96, 4, 5), 12, False),
    ('kitti_n2048_b1', lambda: synthetic.model_cfg('kitti'), lambda: synthetic.make_batch('kitti', 1, 2048), 13, False),
    ('modelnet_n1024_b1', lambda: synthetic.model_cfg('modelnet'),
     lambda: synthetic.make_batch('modelnet', 1, 1024), 14, False),
    ('small_global_n256_b2', small_global_cfg, lambda: synthetic.make_batch('kitti', 2, 256, first_pair=7), 15, True),
    ('small_two_level_n512_b2', small_two_level_cfg, lambda: synthetic.make_batch('kitti', 2, 512, first_pair=9), 16, True),
    # configurations no shipped model uses (other widths, k = 40, extra input features, append_features = False)
    ('custom_widths_n512_b2', helpers.custom_widths_cfg, lambda: synthetic.make_batch('kitti', 2, 512, first_pair=17), 17, True),
    ('custom_features_n384_b2', helpers.custom_features_cfg, helpers.custom_features_batch, 18, True),
    # batch norm everywhere (non-trivial running statistics from random_state_dict) + dropout 0.7, eval mode
    ('small_bn_n512_b2', helpers.small_bn_cfg, lambda: synthetic.make_batch('kitti', 2, 512, first_pair=23), 19, True),
    # more neighbours than the search's rank selection (and the fused flow kernel) take: k = 70 of 128 source centroids
    ('small_k70_n512_b2', helpers.small_k70_cfg, lambda: synthetic.make_batch('kitti', 2, 512, first_pair=29), 20, True),
    # a `transform` module (a SetAbstraction of its own) in front of the cloud features (deepclr.py:447,453-464)
    ('small_transform_n512_b2', helpers.small_transform_cfg, lambda: synthetic.make_batch('kitti', 2, 512, first_pair=31), 21, True),
]


def run_case(ref, name, cfg, x_np, wseed, full):
    sd = synthetic.random_state_dict(cfg, seed=wseed)
    model, label_type = _ref_model(ref, cfg, sd)
    orc = oracle.build_oracle_model(cfg, sd)
    x = torch.from_numpy(x_np)

    with torch.no_grad():
        feat_ref = model.cloud_features(x.clone())
        emb_ref = model._merge_layers[0](feat_ref)
        y_ref, loss, dbg = model(x.clone())
        y_feat_ref, _, _ = model(feat_ref, is_feat=True)
    assert loss is None and dbg is None
    assert torch.equal(y_ref, y_feat_ref)

    feat_or = orc.cloud_features(x)
    emb_or = orc.flow_embedding(feat_or)
    y_or = orc(x)
    for a, b, what in ((feat_ref, feat_or, 'cloud_features'), (emb_ref, emb_or, 'flow_embedding'), (y_ref, y_or, 'y')):
        err = (a - b).abs().max().item()
        assert a.shape == b.shape and err <= 1e-6 * max(1.0, a.abs().max().item()), (name, what, err)

    mats_ref = np.stack([label_type.to_matrix(v.numpy().copy()) for v in y_ref])
    mats_or = np.stack([olabels.dual_quat_to_matrix(v.numpy()) for v in y_or])
    assert np.abs(mats_ref - mats_or).max() < 1e-6, name

    # primitive-level records from the oracle (what the reference composition consumed); the sampling levels in order: those
    # of an optional `transform` module, then those of the cloud features
    prm = cfg['params']
    levels = [(m['params']['npoint'][lv], m['params']['radii'][lv], m['params']['nsamples'][lv])
              for m in ([prm['transform']] if prm.get('transform') else []) + [prm['cloud_features']]
              for lv in range(len(m['params']['npoint']))]
    assert 1 <= len(levels) <= 2
    xyz = x[:, :, :3].contiguous()
    fps_idx = oracle.furthest_point_sample(xyz, levels[0][0])
    new_xyz = oracle.gather_operation(xyz.transpose(1, 2).contiguous(), fps_idx).transpose(1, 2).contiguous()
    extra = {}
    if len(levels) == 1:
        assert torch.equal(new_xyz.transpose(1, 2), feat_ref[:, :3, :])
    else:                                        # the second level samples the first one's centroids (deepclr.py:92-93, 516-520)
        fps_idx1 = oracle.furthest_point_sample(new_xyz, levels[1][0])
        new_xyz1 = oracle.gather_operation(new_xyz.transpose(1, 2).contiguous(), fps_idx1)
        assert torch.equal(new_xyz1, feat_ref[:, :3, :])
        extra['fps_idx1'] = fps_idx1.numpy().astype(np.int16)
    bq = [oracle.ball_query(r, s, xyz, new_xyz) for r, s in zip(levels[0][1], levels[0][2])]
    half = feat_ref.shape[0] // 2
    npoint = feat_ref.shape[2]
    if cfg['params']['merge']['params']['k'] > 0:
        _, _, gi = orc.knn_groups(feat_ref[:half], feat_ref[half:])
        knn_local = (gi[1] - (torch.arange(half).repeat_interleave(npoint) * npoint).view(-1, 1)).to(torch.int32)
        knn_local = knn_local.view(half, npoint, -1)
    else:                                        # GlobalGrouping: every source point of the pair, in index order
        knn_local = torch.arange(npoint, dtype=torch.int32).view(1, 1, -1).expand(half, npoint, -1).contiguous()

    out = {
        'x': x_np, 'weight_seed': np.int64(wseed), 'fps_idx': fps_idx.numpy().astype(np.int16),
        'bq_sha256': np.array([_sha(t) for t in bq]),
        'knn_sha256': np.array(_sha(knn_local)),
        'y': y_ref.numpy(), 'mat': mats_ref, **extra,
    }
    if full:
        out.update(**{'bq%d' % i: t.numpy().astype(np.int16) for i, t in enumerate(bq)},
                   knn=knn_local.numpy().astype(np.int16),
                   cloud_features=feat_ref.numpy(), flow_embedding=emb_ref.numpy())
    else:
        for key, t in (('cloud_features', feat_ref), ('flow_embedding', emb_ref)):
            pos, val = _sample(t, 4096, 99)
            out[key + '_pos'], out[key + '_val'] = pos, val
            out[key + '_shape'] = np.array(t.shape, dtype=np.int64)
            out[key + '_abs_mean'] = np.float64(t.double().abs().mean().item())
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('{:24s} y[0]={}  -> {} ({} KiB)'.format(name, np.round(y_ref[0].numpy(), 4), os.path.basename(path),
                                                os.path.getsize(path) // 1024))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference')
    ap.add_argument('--only-losses', action='store_true', help='regenerate losses.npz only')
    ap.add_argument('--only', default=None, help='regenerate one case only')
    args = ap.parse_args()
    torch.set_num_threads(8)
    ref = _load_reference(args.reference)
    if args.only is None:
        loss_golden(ref, HERE)
    if args.only_losses:
        return
    for name, cfg_fn, x_fn, wseed, full in CASES:
        if args.only is None or args.only == name:
            run_case(ref, name, cfg_fn(), x_fn(), wseed, full)


if __name__ == '__main__':
    main()
