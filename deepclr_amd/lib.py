"""ctypes binding of libdeepclr_amd.so (include/deepclr_amd.h).

The product path has no CPU fallback: if the library is missing or a tensor is
not on a HIP device the call raises, exactly as the reference's wrappers raise
through TORCH_CHECK for non-CUDA tensors (/root/reference/extern/pointnet2.patch:97-99).
"""
import ctypes
import os
from typing import Optional

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB_PATH = os.environ.get('DCLR_LIB', os.path.join(_CSRC, 'libdeepclr_amd.so'))      # DCLR_LIB: A/B builds (scratch/)

_i, _f, _p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> argtypes; every entry point declared in include/deepclr_amd.h
SIGNATURES = {
    'dclr_version': [],
    'dclr_error_string': [_i],
    'dclr_host_device_pointer': [_p, ctypes.POINTER(ctypes.c_void_p)],
    'dclr_furthest_point_sampling': [_i, _i, _i, _p, _p, _p, _p],
    'dclr_gather_points': [_i, _i, _i, _i, _p, _p, _p, _p],
    'dclr_ball_query': [_i, _i, _i, _f, _i, _p, _p, _p, _p],
    'dclr_group_points': [_i, _i, _i, _i, _i, _p, _p, _p, _p],
    'dclr_gather_points_grad': [_i, _i, _i, _i, _p, _p, _p, _p],
    'dclr_group_points_grad': [_i, _i, _i, _i, _i, _p, _p, _p, _p],
    'dclr_knn': [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    'dclr_fps_clouds': [_i, _i, _i, _i, _p, _p, _p],
    'dclr_fps_group_layout': [_i, _p, _p],
    'dclr_fps_workspace_bytes': [_i, _i],
    'dclr_fps_clouds_ws': [_i, _i, _i, _i, _p, _p, _p, ctypes.c_longlong, _p],
    'dclr_fps_clouds_grouped_ws': [_i, _i, _i, _i, _p, _p, _p, _p, _p, ctypes.c_longlong, _p],
    'dclr_fps_clouds_grouped': [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    'dclr_sa_msg_fused': [_i, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    'dclr_sa_msg_fused_f16': [_i, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    'dclr_fps_clouds_grouped_batched': [_i, _i, _i, _i, _p, _i, _i, ctypes.c_longlong, _p, _p, _p, _p, _p, ctypes.c_longlong, _p],
    'dclr_sa_msg_fused_batched': [_i, _i, _i, _i, _i, _p, _i, _i, ctypes.c_longlong, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    'dclr_sa_msg_fused_batched_ov': [_i, _i, _i, _i, _i, _p, _i, _i, ctypes.c_longlong, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    'dclr_rows_to_channels': [_i, _i, _i, _i, _i, _p, _p, _p],
    'dclr_channels_to_rows': [_i, _i, _i, _i, _i, _p, _p, _p],
    'dclr_pack_weight': [_i, _i, _p, _p, _i, _i, _p, _p],
    'dclr_pack_weight16': [_i, _i, _p, _p, _i, _i, _p, _p],
    'dclr_linear': [_i, _i, _i, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p],
    'dclr_linear_pair': [_i, _i, _i, _p, _i, _p, _p, _p, _p, _i, _p],
    'dclr_head_conv_fused': [_i, _i, _p, _p, _p, _p, _p, _i, _p, _i, _p],
    'dclr_knn_rows': [_i, _i, _i, _p, _p, _p],
    'dclr_flow_embedding_fused': [_i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    'dclr_fc': [_i, _i, _i, _p, _p, _p, _i, _p, _p],
    'dclr_pack_weight_f16': [_i, _i, _p, _p, _i, _i, _p, _p],
    'dclr_head_conv_fused_f16': [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _i, _p],
    'dclr_flow_embedding_fused_f16': [_i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    'dclr_flow_f16_tile': [_i],
    'dclr_merge_forward': [_p, _p, _p],
    'dclr_cloud_forward': [_p, _p, _p, _p],
    'dclr_prepare_cloud_blocks': [_i, _i, _i],
    'dclr_prepare_cloud': [_i, _i, _p, _i, _i, _f, _f, _i, _p, _p, _p, _p],
}

MERGE_MAX_LAYERS, MERGE_MAX_FC = 8, 4
MERGE_EVENTS = 6 + MERGE_MAX_FC


class MergeArgs(ctypes.Structure):
    """DclrMergeArgs (include/deepclr_amd.h)."""
    _fields_ = [
        ('struct_size', ctypes.c_uint32),
        ('pairs', _i), ('npoint', _i), ('k', _i), ('precision', _i), ('stages', _i), ('radius', _f),
        ('n_head_layers', _i), ('head_k_in', _i), ('n_fc', _i),
        ('head_k', _i * MERGE_MAX_LAYERS), ('head_n', _i * MERGE_MAX_LAYERS),
        ('fc_k', _i * MERGE_MAX_FC), ('fc_n', _i * MERGE_MAX_FC), ('fc_act', _i * MERGE_MAX_FC),
        ('f_rows', _p), ('wt', _p), ('ws', _p), ('w1a', _p), ('b1', _p), ('w2', _p), ('w3', _p), ('b2', _p), ('b3', _p),
        ('head_w', _p * MERGE_MAX_LAYERS), ('head_b', _p * MERGE_MAX_LAYERS),
        ('fc_w', _p * MERGE_MAX_FC), ('fc_b', _p * MERGE_MAX_FC),
        ('pt', _p), ('ps', _p), ('knn_idx', _p), ('e_rows', _p), ('colmax', _p), ('fc_tmp', _p * 2), ('y', _p),
        ('overflow', _p),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(type(self))


CLOUD_MAX_SCALES, CLOUD_EVENTS = 4, 3


class CloudArgs(ctypes.Structure):
    """DclrCloudArgs (include/deepclr_amd.h)."""
    _fields_ = [
        ('struct_size', ctypes.c_uint32),
        ('b', _i), ('n', _i), ('c', _i), ('npoint', _i), ('pairs_per_batch', _i), ('n_batches', _i),
        ('batch_stride', ctypes.c_longlong), ('f16', _i), ('n_scales', _i),
        ('radii', _f * CLOUD_MAX_SCALES), ('nsamples', _i * CLOUD_MAX_SCALES), ('mlp', _p * CLOUD_MAX_SCALES),
        ('clouds', _p), ('fps_idx', _p), ('group_pts', _p), ('group_box', _p), ('slice_box', _p),
        ('workspace', _p), ('workspace_bytes', ctypes.c_longlong), ('f_rows', _p), ('merge', _p), ('overflow', _p),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(type(self))


_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "deepclr_amd: {} is missing. Build it with `python -m deepclr_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback.".format(LIB_PATH))
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = (ctypes.c_char_p if name == 'dclr_error_string'
                          else ctypes.c_longlong if name == 'dclr_fps_workspace_bytes'
                          else _i)
        _lib = lib
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().dclr_error_string(code).decode()
        raise RuntimeError("{} failed: {} (code {})".format(what, msg, code))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device. torch.cuda.current_stream() builds a Stream object and
    resolves the device through several Python layers (~10 us per call; the pipelined runner asks several times per launch):
    the raw accessors take ~0.3 us."""
    if _raw_stream is not None and _raw_device is not None:
        try:
            return _raw_stream(_raw_device())
        except Exception:                         # no device (CPU-only unit tests): the public accessor says so properly
            pass
    return torch.cuda.current_stream().cuda_stream


def dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    """Contract of the reference wrappers: GPU tensor, contiguous (pointnet2.patch:8-10 CHECK_INPUT)."""
    if not t.is_cuda:
        raise RuntimeError("{} must be a GPU tensor (deepclr_amd has no CPU path)".format(name))
    if t.dtype != torch.float32:
        raise RuntimeError("{} must be float32".format(name))
    if not t.is_contiguous():
        raise RuntimeError("{} must be contiguous".format(name))
    return t


class MappedFlag:
    """One int32 of pinned host memory that kernels can write through `dev_ptr` (the host allocation mapped into the
    device's address space) and the host can read at any time without a device synchronisation -- the sticky
    activation-range flag of the split-f16 kernels (DclrMergeArgs.overflow, include/deepclr_amd.h)."""

    def __init__(self):
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.view = self.host.numpy()
        dev = ctypes.c_void_p()
        check(load().dclr_host_device_pointer(self.host.data_ptr(), ctypes.byref(dev)), 'dclr_host_device_pointer')
        self.dev_ptr = dev.value

    def is_set(self) -> bool:
        return bool(self.view[0])

    def clear(self) -> None:
        self.view[0] = 0


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()
