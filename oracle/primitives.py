"""ctypes front-end of ``oracle/primitives.c`` -- TEST INFRASTRUCTURE ONLY.

Function names and argument order mirror the Python API of the absent
third-party libraries the reference calls (pointnet2 ``pointnet2_utils`` and
``torch_cluster.knn``; call sites /root/reference/deepclr/models/deepclr.py:63-70,
164-166), so the restated composition in ``oracle/model.py`` reads like the
upstream modules. CPU float32 tensors only.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'liboracle.so')
_lib = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (idempotent)."""
    src = os.path.join(_HERE, 'primitives.c')
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(['make', '-C', _HERE, '-B'], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        i, f, p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
        lib.dclr_oracle_fps.argtypes = [i, i, i, p, p, p]
        lib.dclr_oracle_gather_points.argtypes = [i, i, i, i, p, p, p]
        lib.dclr_oracle_ball_query.argtypes = [i, i, i, f, i, p, p, p]
        lib.dclr_oracle_group_points.argtypes = [i, i, i, i, i, p, p, p]
        lib.dclr_oracle_knn.argtypes = [i, i, i, i, p, p, p, p]
        lib.dclr_oracle_fps_block.argtypes = [i]
        lib.dclr_oracle_fps_block.restype = i
        lib.dclr_oracle_num_threads.restype = i
        lib.dclr_oracle_set_threads.argtypes = [i]
        lib.dclr_oracle_set_threads.restype = None
        for name in ('dclr_oracle_fps', 'dclr_oracle_gather_points', 'dclr_oracle_ball_query',
                     'dclr_oracle_group_points', 'dclr_oracle_knn'):
            getattr(lib, name).restype = None
        _lib = lib
    return _lib


def _f32(t: torch.Tensor) -> torch.Tensor:
    assert t.device.type == 'cpu' and t.dtype == torch.float32, "oracle runs on CPU float32"
    return t.contiguous()


def num_threads() -> int:
    return int(_load().dclr_oracle_num_threads())


def set_threads(n: int) -> None:
    _load().dclr_oracle_set_threads(int(n))


def furthest_point_sample(xyz: torch.Tensor, npoint: int) -> torch.Tensor:
    """xyz (B, N, 3) -> idx (B, npoint) int32; upstream ``furthest_point_sample``."""
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    idx = torch.zeros(b, npoint, dtype=torch.int32)
    temp = torch.full((b, n), 1e10, dtype=torch.float32)
    _load().dclr_oracle_fps(b, n, npoint, xyz.data_ptr(), temp.data_ptr(), idx.data_ptr())
    return idx


def gather_operation(features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """features (B, C, N), idx (B, npoint) int32 -> (B, C, npoint); upstream ``gather_operation``."""
    features = _f32(features)
    idx = idx.contiguous()
    b, c, n = features.shape
    npoint = idx.shape[1]
    out = torch.empty(b, c, npoint, dtype=torch.float32)
    _load().dclr_oracle_gather_points(b, c, n, npoint, features.data_ptr(), idx.data_ptr(), out.data_ptr())
    return out


def ball_query(radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    """xyz (B, N, 3), new_xyz (B, npoint, 3) -> idx (B, npoint, nsample) int32; upstream ``ball_query``."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.zeros(b, m, nsample, dtype=torch.int32)
    _load().dclr_oracle_ball_query(b, n, m, float(radius), nsample, new_xyz.data_ptr(), xyz.data_ptr(),
                                   idx.data_ptr())
    return idx


def grouping_operation(features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """features (B, C, N), idx (B, npoint, nsample) -> (B, C, npoint, nsample); upstream ``grouping_operation``."""
    features = _f32(features)
    idx = idx.contiguous()
    b, c, n = features.shape
    _, npoint, nsample = idx.shape
    out = torch.empty(b, c, npoint, nsample, dtype=torch.float32)
    _load().dclr_oracle_group_points(b, c, n, npoint, nsample, features.data_ptr(), idx.data_ptr(),
                                     out.data_ptr())
    return out


def knn(x: torch.Tensor, y: torch.Tensor, k: int, batch_x: torch.Tensor, batch_y: torch.Tensor) -> torch.Tensor:
    """``torch_cluster.knn`` for equally sized sorted batches: (2, M*k) int64 = [query(y) index; x index]."""
    x, y = _f32(x), _f32(y)
    b = int(batch_x.max().item()) + 1 if batch_x.numel() else 0
    assert b == (int(batch_y.max().item()) + 1 if batch_y.numel() else 0)
    nx, ny = x.shape[0] // b, y.shape[0] // b
    assert nx * b == x.shape[0] and ny * b == y.shape[0] and x.shape[1] == 3 and k <= 4096
    assert torch.equal(batch_x, torch.arange(b).repeat_interleave(nx))
    assert torch.equal(batch_y, torch.arange(b).repeat_interleave(ny))
    row = torch.empty(b * ny * k, dtype=torch.int64)
    col = torch.empty(b * ny * k, dtype=torch.int64)
    _load().dclr_oracle_knn(b, nx, ny, k, x.data_ptr(), y.data_ptr(), row.data_ptr(), col.data_ptr())
    mask = col != -1
    return torch.stack((row[mask], col[mask]), dim=0)
