"""Shared helpers for the parity tests (oracle = checker; see oracle/__init__.py)."""
import hashlib
import os

import numpy as np
import torch

from deepclr_amd import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

GOLDEN_CASES = {
    # name -> (model-config factory, stores full intermediates)
    'small_kitti_n512_b2': ('small', True),
    'kitti_rand_n96_b2': ('kitti', False),
    'kitti_n2048_b1': ('kitti', False),
    'modelnet_n1024_b1': ('modelnet', False),
    'small_global_n256_b2': ('small_global', True),      # k == 0: GlobalGrouping
    'small_two_level_n512_b2': ('small_two_level', True),   # second set-abstraction level (deepclr.py:72-83)
    # configurations no shipped model uses: other layer widths, k = 40, more input features, append_features = False
    # (reference deepclr.py:50-70,180-199 accept them; here they run composed from the level-1 HIP operators)
    'custom_widths_n512_b2': ('custom_widths', True),
    'custom_features_n384_b2': ('custom_features', True),
    # batch_norm: true with non-trivial running statistics and dropout 0.7 (reference helper.py:27-36,57-63,107-113): eval mode,
    # so the norm layers apply their running statistics (folded into the packed weights here) and dropout is the identity
    'small_bn_n512_b2': ('small_bn', True),
    # k = 70 neighbours of 128 source centroids (reference deepclr.py:180-199 takes any k; the search's rank selection stops at 40)
    'small_k70_n512_b2': ('small_k70', True),
    # a `transform` module in front of the cloud features (reference deepclr.py:447,453-464; no shipped config has one):
    # 512 points -> 128 centroids x 64 features (module 0) -> 64 centroids x (32 + 32) features (module 1)
    'small_transform_n512_b2': ('small_transform', True),
}


def small_cfg() -> dict:
    """Reduced set-abstraction / kNN sizes so a golden can hold every intermediate tensor."""
    cfg = synthetic.model_cfg('kitti')
    sa = cfg['params']['cloud_features']['params']
    sa['npoint'], sa['radii'], sa['nsamples'] = [64], [[2.0, 4.0]], [[8, 16]]
    cfg['params']['merge']['params'].update(k=8, radius=6.0)
    return cfg


def small_global_cfg() -> dict:
    cfg = small_cfg()
    cfg['params']['merge']['params'].update(k=0, radius=6.0)
    return cfg


def small_transform_cfg() -> dict:
    """`transform`: a SetAbstraction of its own in front of the cloud features (`_cloud_layers` = [transform, features]; the
    reference's only per-cloud module class). The feature module's level 0 then takes 64 input features."""
    cfg = small_cfg()
    cfg['params']['transform'] = {'name': 'SetAbstraction', 'params': {
        'npoint': [128], 'radii': [[2.0, 4.0]], 'nsamples': [[8, 16]], 'mlps': [[[16, 16, 32], [16, 16, 32]]]}}
    sa = cfg['params']['cloud_features']['params']
    sa['npoint'], sa['radii'], sa['nsamples'] = [64], [[4.0, 8.0]], [[8, 24]]
    sa['mlps'] = [[[32, 32], [48, 32]]]
    return cfg


def small_two_level_cfg() -> dict:
    """Two set-abstraction levels: 512 points -> 128 centroids x 64 features -> 64 centroids x (32 + 32) features."""
    cfg = small_cfg()
    sa = cfg['params']['cloud_features']['params']
    sa['npoint'], sa['radii'], sa['nsamples'] = [128, 64], [[2.0, 4.0], [4.0, 8.0]], [[8, 16], [8, 24]]
    sa['mlps'] = [[[16, 16, 32], [16, 16, 32]], [[64, 32, 32], [64, 48, 32]]]
    return cfg


def custom_widths_cfg() -> dict:
    """One scale of widths [32, 32, 64] at level 0, flow MLP [64, 64, 128] over k = 40 neighbours, head 128-128-256 /
    256-128-64."""
    cfg = small_cfg()
    sa = cfg['params']['cloud_features']['params']
    sa['npoint'], sa['radii'], sa['nsamples'], sa['mlps'] = [128], [[3.0]], [[16]], [[[32, 32, 64]]]
    cfg['params']['merge']['params'].update(k=40, radius=6.0, mlp=[64, 64, 128])
    cfg['params']['output']['params'].update(mlp=[128, 128, 256], linear=[256, 128, 64])
    return cfg


def custom_features_cfg() -> dict:
    """Six input columns (xyz + 3 features), two scales of different depth (80 output features), feature DIFFERENCES in the
    flow embedding (append_features = False), no radius mask, 96 centroids (not a multiple of 64)."""
    cfg = small_cfg()
    cfg['input_dim'] = 6
    sa = cfg['params']['cloud_features']['params']
    sa['npoint'], sa['radii'], sa['nsamples'], sa['mlps'] = [96], [[2.0, 4.0]], [[8, 16]], [[[16, 32], [16, 16, 48]]]
    cfg['params']['merge']['params'].update(k=12, radius=0.0, mlp=[96, 64], append_features=False)
    cfg['params']['output']['params'].update(mlp=[64, 128], linear=[128, 32])
    return cfg


def small_bn_cfg() -> dict:
    cfg = small_cfg()
    cfg['params'].update(batch_norm=True, dropout=0.7)
    return cfg


def small_k70_cfg() -> dict:
    cfg = small_cfg()
    sa = cfg['params']['cloud_features']['params']
    sa['npoint'], sa['radii'], sa['nsamples'] = [128], [[2.0, 4.0]], [[8, 16]]
    cfg['params']['merge']['params'].update(k=70, radius=20.0)
    return cfg


def custom_features_batch(n_pairs: int = 2, n_points: int = 384) -> np.ndarray:
    x = synthetic.make_batch('kitti', n_pairs, n_points, first_pair=21)
    extra = np.random.default_rng(33).uniform(-1.0, 1.0, size=x.shape[:2] + (2,)).astype(np.float32)
    return np.concatenate((x, extra), axis=2)


def case_cfg(name: str) -> dict:
    kind = GOLDEN_CASES[name][0]
    if kind == 'custom_widths':
        return custom_widths_cfg()
    if kind == 'custom_features':
        return custom_features_cfg()
    if kind == 'small_global':
        return small_global_cfg()
    if kind == 'small_bn':
        return small_bn_cfg()
    if kind == 'small_k70':
        return small_k70_cfg()
    if kind == 'small_two_level':
        return small_two_level_cfg()
    if kind == 'small_transform':
        return small_transform_cfg()
    return small_cfg() if kind == 'small' else synthetic.model_cfg(kind)


def load_golden(name: str):
    g = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    cfg = case_cfg(name)
    sd = synthetic.random_state_dict(cfg, seed=int(g['weight_seed']))
    return g, cfg, sd


def sha(t) -> str:
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest()


def degenerate_batch(n_pairs: int, n_points: int, c: int, seed: int) -> np.ndarray:
    """U[0,1]^c clouds with every 7th point a duplicate of its predecessor (forces exact ties)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.0, 1.0, size=(2 * n_pairs, n_points, c)).astype(np.float32)
    x[:, 7::7, :] = x[:, 6:-1:7, :][:, :x[:, 7::7, :].shape[1], :]
    return x


def pose_delta(mats_a: np.ndarray, mats_b: np.ndarray) -> float:
    """Mean over pairs of max|M_a - M_b| on the 4x4 (BASELINE.json's 'mean pose delta')."""
    return float(np.mean(np.abs(np.asarray(mats_a) - np.asarray(mats_b)).reshape(len(mats_a), -1).max(axis=1)))
