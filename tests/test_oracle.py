"""CPU tests: the oracle against the committed goldens (reference composition run in the build
container, tests/golden/make_golden.py) and against independent restatements of its primitives."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import labels as olabels
from deepclr_amd import synthetic
from helpers import GOLDEN_CASES, load_golden, sha, degenerate_batch


@pytest.mark.parametrize('name', list(GOLDEN_CASES))
def test_oracle_reproduces_golden(name):
    g, cfg, sd = load_golden(name)
    orc = oracle.build_oracle_model(cfg, sd)
    x = torch.from_numpy(g['x'])
    prm = cfg['params']
    sa = (prm.get('transform') or prm['cloud_features'])['params']    # the module that samples the raw clouds
    xyz = x[:, :, :3].contiguous()

    fps = oracle.furthest_point_sample(xyz, sa['npoint'][0])
    assert np.array_equal(fps.numpy(), g['fps_idx'].astype(np.int32))
    new_xyz = oracle.gather_operation(xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    for s, (r, ns) in enumerate(zip(sa['radii'][0], sa['nsamples'][0])):
        bq = oracle.ball_query(r, ns, xyz, new_xyz)
        assert sha(bq) == str(g['bq_sha256'][s])
        if GOLDEN_CASES[name][1]:
            assert np.array_equal(bq.numpy(), g['bq%d' % s].astype(np.int32))

    feat = orc.cloud_features(x)
    emb = orc.flow_embedding(feat)
    y = orc(x)
    if GOLDEN_CASES[name][1]:
        np.testing.assert_allclose(feat.numpy(), g['cloud_features'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(emb.numpy(), g['flow_embedding'], rtol=1e-4, atol=1e-5)
    else:
        for key, t in (('cloud_features', feat), ('flow_embedding', emb)):
            assert tuple(t.shape) == tuple(g[key + '_shape'])
            got = t.contiguous().view(-1)[torch.from_numpy(g[key + '_pos'])].numpy()
            np.testing.assert_allclose(got, g[key + '_val'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y.numpy(), g['y'], rtol=1e-4, atol=1e-5)
    mats = np.stack([olabels.dual_quat_to_matrix(v) for v in y.numpy()])
    assert np.abs(mats - g['mat']).max() < 1e-4


def _fps_reference_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, 'fps_reference.npz'))
    names = sorted({k.split('/')[0] for k in g.files})
    assert len(names) >= 6 and {'kitti_n16384_m1024', 'kitti_n20000_m256'} <= set(names)
    return [(n, g[n + '/points'], g[n + '/picks'], int(g[n + '/m'])) for n in names]


def test_fps_reproduces_the_reference_numpy_sampler(golden_dir):
    """tests/golden/fps_reference.npz holds the rows picked by the reference's own FarthestPointSampling._fps
    (deepclr/data/transforms/transforms.py:47-59, run by tests/golden/make_fps_golden.py) on clouds whose best /
    runner-up gap exceeds float32 rounding at every step: the float32 squared-distance rule of the oracle must pick
    the same rows in the same order."""
    for name, pts, picks, m in _fps_reference_cases(golden_dir):
        if m >= len(pts):            # the transform returns the cloud itself; the CUDA rule repeats index 0 afterwards
            assert np.array_equal(picks, np.arange(len(pts)))
            continue
        got = oracle.furthest_point_sample(torch.from_numpy(pts)[None], m).numpy()[0]
        assert np.array_equal(got, picks), name


def test_fps_more_samples_than_points():
    """tests/model/test_deepclr.py:19-25 feeds 96 points with npoint=1024: after exhaustion index 0 repeats."""
    pts = torch.rand(1, 96, 3, generator=torch.Generator().manual_seed(0))
    idx = oracle.furthest_point_sample(pts, 200).numpy()[0]
    assert sorted(idx[:96].tolist()) == list(range(96))
    assert (idx[96:] == 0).all()


def test_ball_query_padding_and_empty():
    xyz = torch.tensor([[[0., 0, 0], [0.1, 0, 0], [5, 5, 5], [0.2, 0, 0]]])
    new_xyz = torch.tensor([[[0., 0, 0], [100, 100, 100]]])
    idx = oracle.ball_query(0.5, 4, xyz, new_xyz).numpy()[0]
    assert idx[0].tolist() == [0, 1, 3, 0]          # hits in index order, tail padded with the first hit
    assert idx[1].tolist() == [0, 0, 0, 0]          # no hit -> the pre-zeroed buffer is left alone
    idx = oracle.ball_query(0.5, 2, xyz, new_xyz).numpy()[0]
    assert idx[0].tolist() == [0, 1]                # stops at nsample


def test_knn_sorted_and_stable():
    x = torch.tensor([[0., 0, 0], [1, 0, 0], [1, 0, 0], [3, 0, 0]])
    y = torch.tensor([[0.9, 0, 0]])
    out = oracle.knn(x, y, 3, torch.zeros(4, dtype=torch.long), torch.zeros(1, dtype=torch.long))
    assert out[0].tolist() == [0, 0, 0] and out[1].tolist() == [1, 2, 0]   # ties keep the lower index first


def test_knn_against_bruteforce():
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(size=(2 * 300, 3)).astype(np.float32))
    y = torch.from_numpy(rng.normal(size=(2 * 200, 3)).astype(np.float32))
    bx, by = torch.arange(2).repeat_interleave(300), torch.arange(2).repeat_interleave(200)
    col = oracle.knn(x, y, 7, bx, by)[1].view(400, 7)
    for b in range(2):
        d = torch.cdist(y[b * 200:(b + 1) * 200].double(), x[b * 300:(b + 1) * 300].double())
        ref = d.topk(7, dim=1, largest=False).indices + b * 300
        assert torch.equal(col[b * 200:(b + 1) * 200].sort(dim=1).values, ref.sort(dim=1).values)


def test_duplicates_do_not_break_oracle():
    x = torch.from_numpy(degenerate_batch(1, 96, 4, 7))
    cfg = synthetic.model_cfg('kitti')
    y = oracle.build_oracle_model(cfg, synthetic.random_state_dict(cfg, 1))(x)
    assert y.shape == (1, 8) and torch.isfinite(y).all()


def test_dual_quat_roundtrip():
    from deepclr_amd.labels import LabelType
    _, _, m = synthetic.kitti_like_pair(0, 16)
    lt = LabelType.POSE3D_DUAL_QUAT
    label = lt.from_matrix(m)
    assert np.abs(lt.to_matrix(label) - m).max() < 1e-7   # _dqnormalize adds eps=1e-8 to the norm
    assert np.abs(olabels.dual_quat_to_matrix(label) - m).max() < 1e-7


def test_preprocess_oracle_reproduces_the_reference_transform_outputs(golden_dir):
    """tests/golden/preprocess.npz was written by the reference's own transforms.py (make_preprocess_golden.py):
    the numpy restatement the GPU preparation kernel is checked against must reproduce every case bit for bit."""
    from oracle import preprocess as opre
    g = np.load(os.path.join(golden_dir, 'preprocess.npz'))
    names = sorted({k.split('/')[0] for k in g.files})
    assert len(names) == 11
    for name in names:
        nth, start, lo, hi, dim = g[name + '/params']
        kw = dict(nth=int(nth), start=int(start), min_range=float(lo), max_range=float(hi))
        if dim >= 0:
            kw['input_dim'] = int(dim)
        got = opre.prepare_cloud(g[name + '/raw'], **kw)
        assert got.shape == g[name + '/want'].shape and np.array_equal(got, g[name + '/want'], equal_nan=True), name
