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


def small_two_level_cfg() -> dict:
    """Two set-abstraction levels: 512 points -> 128 centroids x 64 features -> 64 centroids x (32 + 32) features."""
    cfg = small_cfg()
    sa = cfg['params']['cloud_features']['params']
    sa['npoint'], sa['radii'], sa['nsamples'] = [128, 64], [[2.0, 4.0], [4.0, 8.0]], [[8, 16], [8, 24]]
    sa['mlps'] = [[[16, 16, 32], [16, 16, 32]], [[64, 32, 32], [64, 48, 32]]]
    return cfg


def case_cfg(name: str) -> dict:
    kind = GOLDEN_CASES[name][0]
    if kind == 'small_global':
        return small_global_cfg()
    if kind == 'small_two_level':
        return small_two_level_cfg()
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
