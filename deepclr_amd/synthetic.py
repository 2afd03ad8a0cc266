"""Fixed synthetic inputs for parity tests and ``bench.py``.

No dataset or checkpoint ships with the reference (weights are Git-LFS
pointers, /root/reference/models/kitti_00-06/weights.tar:1-3), so every
measurement uses:

  * clouds from ``numpy.random.default_rng(1234 + i)`` per scan pair ``i``:
    KITTI-like (x,y ~ N(0,20^2) m, z ~ N(-1,0.5^2), intensity ~ U[0,1]; source =
    small rigid motion + 1 cm noise + permutation, mirroring
    /root/reference/configs/training/kitti_00-06.yaml:20-29) or ModelNet-like
    (points on a sphere shell; mirrors configs/training/modelnet40.yaml:11-21);
  * weights in the reference's ``state_dict`` key layout,
    drawn from a numpy generator so that tests, goldens and the bench rebuild
    the identical tensors without shipping a 7 MB checkpoint.

The two architecture dictionaries restate the hyper-parameters of
/root/reference/models/kitti_00-06/model_config.yaml and
/root/reference/models/modelnet40/model_config.yaml.
"""
import copy
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch


def _arch(input_dim, npoint, radii, nsamples, k, radius):
    return {
        'weights': None, 'input_dim': input_dim, 'point_dim': 3,
        'label_type': 'POSE3D_DUAL_QUAT', 'model_type': 'DEEPCLR',
        'params': {
            'batch_norm': False, 'dropout': 1.0,
            'cloud_features': {'name': 'SetAbstraction', 'params': {
                'npoint': [npoint], 'radii': [list(radii)], 'nsamples': [list(nsamples)],
                'mlps': [[[16, 16, 32], [16, 16, 32]]]}},
            'merge': {'name': 'MotionEmbedding', 'params': {'k': k, 'radius': radius, 'mlp': [128, 128, 256]}},
            'output': {'name': 'OutputSimple', 'params': {
                'mlp': [256, 256, 512, 512, 1024], 'linear': [1024, 512, 256]}},
        },
    }


KITTI_MODEL_CFG = _arch(4, 1024, (0.5, 1.0), (512, 1024), 20, 10.0)
MODELNET_MODEL_CFG = _arch(3, 512, (0.1, 0.2), (256, 512), 30, 0.2)


def model_cfg(kind: str) -> dict:
    return copy.deepcopy({'kitti': KITTI_MODEL_CFG, 'modelnet': MODELNET_MODEL_CFG}[kind])


def _euler_to_mat(rx: float, ry: float, rz: float) -> np.ndarray:
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    mx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    my = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    mz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return mz @ my @ mx


def kitti_like_pair(i: int, n: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Returns template (n,4), source (n,4), ground-truth 4x4 (source = M * template)."""
    rng = np.random.default_rng(1234 + i)
    tmpl = np.empty((n, 4), dtype=np.float64)
    tmpl[:, 0:2] = rng.normal(0.0, 20.0, size=(n, 2))
    tmpl[:, 2] = rng.normal(-1.0, 0.5, size=n)
    tmpl[:, 3] = rng.uniform(0.0, 1.0, size=n)
    rot = _euler_to_mat(*np.deg2rad(rng.normal(0.0, [0.1, 0.1, 1.0])))
    trans = rng.normal(0.0, [0.2, 0.02, 0.02])
    src = tmpl.copy()
    src[:, :3] = tmpl[:, :3] @ rot.T + trans + rng.normal(0.0, 0.01, size=(n, 3))
    src = src[rng.permutation(n)]
    m = np.eye(4)
    m[:3, :3], m[:3, 3] = rot, trans
    return tmpl.astype(np.float32), src.astype(np.float32), m


def modelnet_like_pair(i: int, n: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    rng = np.random.default_rng(1234 + i)
    p = rng.uniform(-1.0, 1.0, size=(n, 3))
    p /= np.maximum(np.linalg.norm(p, axis=1, keepdims=True), 1e-9)
    p *= rng.uniform(0.5, 1.0)
    rot = _euler_to_mat(*np.deg2rad(rng.uniform(-5.0, 5.0, size=3)))
    trans = rng.uniform(-0.1, 0.1, size=3)
    src = p @ rot.T + trans + rng.normal(0.0, 0.02, size=(n, 3))
    src = src[rng.permutation(n)]
    m = np.eye(4)
    m[:3, :3], m[:3, 3] = rot, trans
    return p.astype(np.float32), src.astype(np.float32), m


def ring_scan(rng: np.random.Generator, n: int, rings: int = 64) -> np.ndarray:
    """A spinning-LiDAR scan with KITTI's sampling density and no dataset: `rings` beams between +2 and -24.8 degrees
    of elevation (HDL-64E), sensor 1.73 m above a flat ground, 2 n / rings azimuth steps per revolution of which
    every second one is kept (the reference's converter drops every second point of a raw scan,
    /root/reference/scripts/converter/kitti_odometry.py:14,22) -> exactly n points (n, 4) in scan order, ring by
    ring. A beam ends on the ground (range = height / sin(-elevation): the dense near field), on one of two facades
    of a street canyon, on one of a few dozen car-sized boxes, or at the 80 m range limit. Near the sensor a 1 m
    ball holds several hundred points -- the regime where ball query reaches its nsample caps; the Gaussian clouds
    of kitti_like_pair hold ~20."""
    per_ring = 2 * n // rings
    assert per_ring * rings == 2 * n, "n must be a multiple of rings / 2"
    elev = np.deg2rad(np.linspace(2.0, -24.8, rings))[:, None]                       # (rings, 1)
    azim = (2.0 * np.pi / per_ring) * np.arange(per_ring)[None, :] + rng.uniform(0, 2 * np.pi)
    dx, dy, dz = np.cos(elev) * np.cos(azim), np.cos(elev) * np.sin(azim), np.sin(elev) * np.ones_like(azim)
    height, far = 1.73, 80.0
    rng_ground = np.where(dz < -1e-3, height / np.maximum(-dz, 1e-3), far)
    half_width = rng.uniform(7.0, 14.0, size=2)                                      # facades at y = +w0, y = -w1
    rng_wall = np.where(dy > 1e-3, half_width[0] / np.maximum(dy, 1e-3),
                        np.where(dy < -1e-3, half_width[1] / np.maximum(-dy, 1e-3), far))
    dist = np.minimum(np.minimum(rng_ground, rng_wall), far)
    for _ in range(24):                                                              # parked cars: 4.2 x 1.8 x 1.5 m boxes
        cx, cy = rng.uniform(-40, 40), rng.choice([-1, 1]) * rng.uniform(2.5, 6.0)
        lo = np.array([cx - 2.1, cy - 0.9, -height]); hi = np.array([cx + 2.1, cy + 0.9, -height + 1.5])
        with np.errstate(divide='ignore', invalid='ignore'):
            t0 = np.stack([lo[0] / dx, lo[1] / dy, lo[2] / dz]); t1 = np.stack([hi[0] / dx, hi[1] / dy, hi[2] / dz])
        tn, tf = np.nanmax(np.minimum(t0, t1), axis=0), np.nanmin(np.maximum(t0, t1), axis=0)
        hit = (tn <= tf) & (tn > 0.5)
        dist = np.where(hit, np.minimum(dist, tn), dist)
    dist = dist * (1.0 + rng.normal(0.0, 0.002, size=dist.shape))                     # range noise, 2 mm per metre
    pts = np.stack([dx * dist, dy * dist, dz * dist, rng.uniform(0.0, 1.0, size=dist.shape)], axis=-1)
    return pts[:, ::2, :].reshape(-1, 4)                                             # every 2nd point, ring-major order


def ring_pair(i: int, n: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """LiDAR-density counterpart of kitti_like_pair: template = ring_scan, source = the same small rigid motion +
    1 cm noise + permutation."""
    rng = np.random.default_rng(4321 + i)
    tmpl = ring_scan(rng, n)
    rot = _euler_to_mat(*np.deg2rad(rng.normal(0.0, [0.1, 0.1, 1.0])))
    trans = rng.normal(0.0, [0.2, 0.02, 0.02])
    src = tmpl.copy()
    src[:, :3] = tmpl[:, :3] @ rot.T + trans + rng.normal(0.0, 0.01, size=(n, 3))
    src = src[rng.permutation(n)]
    m = np.eye(4)
    m[:3, :3], m[:3, 3] = rot, trans
    return tmpl.astype(np.float32), src.astype(np.float32), m


def make_batch(kind: str, n_pairs: int, n_points: int, first_pair: int = 0) -> np.ndarray:
    """(2B, N, C) float32 in the reference batch layout [T0..TB-1, S0..SB-1]
    (/root/reference/deepclr/data/build.py:82). kind 'ring': KITTI architecture on ring_pair clouds."""
    gen = {'kitti': kitti_like_pair, 'modelnet': modelnet_like_pair, 'ring': ring_pair}[kind]
    pairs = [gen(first_pair + i, n_points) for i in range(n_pairs)]
    return np.stack([p[0] for p in pairs] + [p[1] for p in pairs], axis=0)


def state_dict_shapes(cfg: dict) -> 'OrderedDict[str, Tuple[int, ...]]':
    """Reference state_dict layout for a model-config mapping. With `batch_norm: true` every
    affine layer is followed by a BatchNorm module (/root/reference/deepclr/models/helper.py:27-30,57-60: index 1 of the
    layer's `_sequential`; set abstraction: the published SharedMLP drops the conv bias and adds `bn.bn`), and with
    `dropout` < 1 the head's fully connected stack interleaves Dropout modules (helper.py:107-113: the Linear layers then sit
    at the even indices of its `_sequential`)."""
    prm = cfg['params']
    bn = bool(prm.get('batch_norm', False))
    shapes: 'OrderedDict[str, Tuple[int, ...]]' = OrderedDict()

    def norm(base: str, width: int) -> None:
        shapes[base + '.weight'] = (width,)
        shapes[base + '.bias'] = (width,)
        shapes[base + '.running_mean'] = (width,)
        shapes[base + '.running_var'] = (width,)
        shapes[base + '.num_batches_tracked'] = ()

    # per-cloud modules in `_cloud_layers` order: an optional `transform` module first (reference deepclr.py:453-464)
    cloud_mods = ([prm['transform']] if prm.get('transform') else []) + [prm['cloud_features']]
    feat_in = cfg['input_dim'] - cfg['point_dim']
    sa_out = 0
    for mi, mod in enumerate(cloud_mods):
        sa = mod['params']
        for lv in range(len(sa['mlps'])):            # level 1 specs start with their input width (deepclr.py:61 vs 73)
            sa_out = 0
            for s, spec in enumerate(sa['mlps'][lv]):
                chans = [feat_in + 3, *spec] if lv == 0 else [spec[0] + 3, *spec[1:]]
                for j in range(len(chans) - 1):
                    base = '_cloud_layers.{}._sa{}.mlps.{}.layer{}'.format(mi, lv, s, j)
                    shapes[base + '.conv.weight'] = (chans[j + 1], chans[j], 1, 1)
                    if bn:
                        norm(base + '.bn.bn', chans[j + 1])
                    else:
                        shapes[base + '.conv.bias'] = (chans[j + 1],)
                sa_out += spec[-1]
        feat_in = sa_out                             # the next module's input features

    def stack(prefix: str, chans, conv: bool, step: int = 1) -> None:
        for j in range(len(chans) - 1):
            base = '{}._sequential.{}._sequential'.format(prefix, j * step)
            shapes[base + '.0.weight'] = (chans[j + 1], chans[j], 1) if conv else (chans[j + 1], chans[j])
            shapes[base + '.0.bias'] = (chans[j + 1],)
            if bn:
                norm(base + '.1', chans[j + 1])

    me = prm['merge']['params']
    stack('_merge_layers.0._embedding._conv', [3 + (2 if me.get('append_features', True) else 1) * sa_out, *me['mlp']], True)
    out = prm['output']['params']
    stack('_merge_layers.1.conv', [3 + me['mlp'][-1], *out['mlp']], True)
    lin = out['linear']
    stack('_merge_layers.1.linear', lin, False, step=2 if float(prm.get('dropout', 1.0)) < 1.0 else 1)
    shapes['_merge_layers.1.output.weight'] = (8, lin[-1])
    shapes['_merge_layers.1.output.bias'] = (8,)
    return shapes


def random_state_dict(cfg: dict, seed: int = 0, bias_scale: float = 0.05) -> Dict[str, torch.Tensor]:
    """Xavier-uniform-scaled weights and small non-zero biases from numpy's PCG64.

    Non-zero biases (unlike the reference's zero init, helper.py:23-25) so that
    every bias path of the kernels is exercised by the parity tests.
    """
    rng = np.random.default_rng(seed)
    sd: Dict[str, torch.Tensor] = OrderedDict()
    for name, shape in state_dict_shapes(cfg).items():
        if name.endswith('.num_batches_tracked'):
            sd[name] = torch.tensor(100, dtype=torch.int64)
            continue
        if len(shape) == 1 and not name.endswith('.bias'):
            # batch-norm scale and running statistics: away from (1, 0, 1) so that a fold that drops a term shows
            lo, hi = {'weight': (0.5, 1.5), 'running_mean': (-0.3, 0.3), 'running_var': (0.4, 2.5)}[name.rsplit('.', 1)[1]]
            sd[name] = torch.from_numpy(rng.uniform(lo, hi, size=shape).astype(np.float32))
            continue
        if name.endswith('.weight'):
            fan_out, fan_in = shape[0], shape[1]
            bound = np.sqrt(6.0 / (fan_in + fan_out))
            arr = rng.uniform(-bound, bound, size=shape)
        else:
            arr = rng.uniform(-bias_scale, bias_scale, size=shape)
            if name == '_merge_layers.1.output.bias':
                arr[0] += 1.0
        sd[name] = torch.from_numpy(arr.astype(np.float32))
    return sd
