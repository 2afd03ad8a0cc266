"""Operator-level API: the functions the reference imports from its native extensions.

Level 1 mirrors, name for name and argument for argument, what
``deepclr/models/deepclr.py`` obtains from the absent third-party packages:
``pointnet2`` (``furthest_point_sample``, ``gather_operation``, ``ball_query``,
``grouping_operation``; exported by /root/reference/extern/pointnet2.patch:37-40) and
``torch_cluster.knn`` (call site deepclr.py:164-166). Level 2 exposes the fused
kernels the model forward uses. Everything runs on the current torch stream.
"""
import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import lib

F_STRIDE = 68
E_STRIDE = 264
FUSED_MAX_POINTS = 65536        # points per cloud the fused sampler / set-abstraction kernels take (16-bit point indices)

# Optional launch timer (bench.py): an object with begin(name) -> token and end(token), called on the
# stream the kernel is enqueued on. None in normal operation.
TIMER = None

# Matrix path of the fused flow-embedding and pose-head kernels: 'f16x2' = f32 operands split into f16 hi/lo
# halves on the f16 matrix instructions (f32-accurate, csrc/mma16f.h), 'f32' = the f32 matrix instructions.
PRECISION = os.environ.get('DCLR_PRECISION', 'f16x2')

# Split-f16 operands saturate at +-65504 (csrc/mma16f.h clamps instead of producing inf). Weights are checked when they
# are packed; activations cannot be checked inside the fused kernels for free. CHECK_RANGE (DCLR_CHECK_RANGE):
#   'first' (default)  the FIRST forward after the weights changed (load_state_dict, .to(), an optimizer step) also runs
#                      the dense stages on the f32 matrix instructions, on that call's own activations, and raises if the
#                      two disagree or an activation leaves the range: a new checkpoint cannot clamp silently on first use;
#   'always' (or '1')  every forward does (debugging; halves throughput);
#   'never' (or '0')   no check.
SLICE_BOXES = os.environ.get('DCLR_SLICE_BOXES', '1') != '0'    # A/B: 0 = set abstraction tests whole 256-point groups only
CHECK_RANGE = {'1': 'always', '0': 'never'}.get(os.environ.get('DCLR_CHECK_RANGE', 'first'),
                                                os.environ.get('DCLR_CHECK_RANGE', 'first'))
if CHECK_RANGE not in ('first', 'always', 'never'):
    raise RuntimeError("DCLR_CHECK_RANGE must be one of first, always, never, 1, 0 (got '{}')".format(CHECK_RANGE))
if PRECISION not in ('f16x2', 'f32'):
    raise RuntimeError("DCLR_PRECISION must be f16x2 or f32 (got '{}')".format(PRECISION))
F16_MAX = 65504.0


def check_f16_range(t: torch.Tensor, what: str) -> None:
    """Raise if a tensor about to be packed as split-f16 operands leaves the f16 range (one host sync, pack time only)."""
    peak = float(t.detach().abs().max()) if t.numel() else 0.0
    if not peak < F16_MAX:
        raise RuntimeError("{}: |value| reaches {:.4g}, outside the split-f16 operand range (+-65504); run with "
                           "DCLR_PRECISION=f32 (f32 matrix instructions) for this checkpoint".format(what, peak))



def _call(name: str, what: str, *args) -> None:
    fn = getattr(lib.load(), name)
    if TIMER is None:
        lib.check(fn(*args), what)
        return
    token = TIMER.begin(what)
    code = fn(*args)
    TIMER.end(token)
    lib.check(code, what)


# ------------------------------------------------------------------------------------------------
# level 1
# ------------------------------------------------------------------------------------------------
def furthest_point_sample(xyz: torch.Tensor, npoint: int) -> torch.Tensor:
    """xyz (B, N, 3) -> (B, npoint) int32 indices."""
    xyz = lib.dev_f32(xyz, 'xyz')
    b, n, c = xyz.shape
    assert c == 3
    idx = torch.empty(b, npoint, dtype=torch.int32, device=xyz.device)
    temp = torch.full((b, n), 1e10, dtype=torch.float32, device=xyz.device)
    _call('dclr_furthest_point_sampling', 'furthest_point_sample', b, n, npoint, xyz.data_ptr(), temp.data_ptr(),
                                                      idx.data_ptr(), lib.stream_ptr())
    return idx


def gather_operation(features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """features (B, C, N), idx (B, npoint) int32 -> (B, C, npoint)."""
    features = lib.dev_f32(features, 'features')
    b, c, n = features.shape
    assert idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous()
    npoint = idx.shape[1]
    out = torch.empty(b, c, npoint, dtype=torch.float32, device=features.device)
    _call('dclr_gather_points', 'gather_operation', b, c, n, npoint, features.data_ptr(), idx.data_ptr(), out.data_ptr(),
                                            lib.stream_ptr())
    return out


def ball_query(radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    """xyz (B, N, 3), new_xyz (B, npoint, 3) -> (B, npoint, nsample) int32."""
    xyz, new_xyz = lib.dev_f32(xyz, 'xyz'), lib.dev_f32(new_xyz, 'new_xyz')
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.zeros(b, m, nsample, dtype=torch.int32, device=xyz.device)
    _call('dclr_ball_query', 'ball_query', b, n, m, float(radius), nsample, new_xyz.data_ptr(), xyz.data_ptr(),
                                         idx.data_ptr(), lib.stream_ptr())
    return idx


def grouping_operation(features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """features (B, C, N), idx (B, npoint, nsample) int32 -> (B, C, npoint, nsample)."""
    features = lib.dev_f32(features, 'features')
    b, c, n = features.shape
    assert idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous()
    _, npoint, nsample = idx.shape
    out = torch.empty(b, c, npoint, nsample, dtype=torch.float32, device=features.device)
    _call('dclr_group_points', 'grouping_operation', b, c, n, npoint, nsample, features.data_ptr(), idx.data_ptr(),
                                           out.data_ptr(), lib.stream_ptr())
    return out


def gather_operation_grad(grad_out: torch.Tensor, idx: torch.Tensor, n: int) -> torch.Tensor:
    """Backward of gather_operation: grad_out (B, C, npoint), idx (B, npoint) int32 -> grad_features (B, C, n)."""
    grad_out = lib.dev_f32(grad_out, 'grad_out')
    b, c, npoint = grad_out.shape
    assert idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous() and idx.shape == (b, npoint)
    grad = torch.zeros(b, c, n, dtype=torch.float32, device=grad_out.device)
    _call('dclr_gather_points_grad', 'gather_operation_grad', b, c, n, npoint, grad_out.data_ptr(), idx.data_ptr(),
          grad.data_ptr(), lib.stream_ptr())
    return grad


def grouping_operation_grad(grad_out: torch.Tensor, idx: torch.Tensor, n: int) -> torch.Tensor:
    """Backward of grouping_operation: grad_out (B, C, npoint, nsample), idx (B, npoint, nsample) int32 ->
    grad_features (B, C, n)."""
    grad_out = lib.dev_f32(grad_out, 'grad_out')
    b, c, npoint, nsample = grad_out.shape
    assert idx.is_cuda and idx.dtype == torch.int32 and idx.is_contiguous() and idx.shape == (b, npoint, nsample)
    grad = torch.zeros(b, c, n, dtype=torch.float32, device=grad_out.device)
    _call('dclr_group_points_grad', 'grouping_operation_grad', b, c, n, npoint, nsample, grad_out.data_ptr(), idx.data_ptr(),
          grad.data_ptr(), lib.stream_ptr())
    return grad


def knn(x: torch.Tensor, y: torch.Tensor, k: int, batch_x: Optional[torch.Tensor] = None,
        batch_y: Optional[torch.Tensor] = None, batch_size: Optional[int] = None) -> torch.Tensor:
    """``torch_cluster.knn`` for equally sized sorted batches: (2, len(y)*k) int64 = [y index; x index] (fewer columns when
    a query has fewer than k candidates within a squared distance of 1e10: upstream's slots start there and stay -1).
    The batch count has to be known on the host (it sizes the launch): one device-to-host read of the two batch
    vectors' last entries, none when the caller passes ``batch_size`` (torch_cluster >= 1.6 takes the same argument)."""
    x, y = lib.dev_f32(x, 'x'), lib.dev_f32(y, 'y')
    if x.dim() != 2 or x.shape[1] != 3 or y.shape[1] != 3:
        raise RuntimeError("knn: only 3-d points are supported")
    b = 1
    if batch_x is not None or batch_y is not None:
        if batch_x is None or batch_y is None:
            raise RuntimeError("knn: give both batch vectors or neither")
        if batch_size is not None:
            b = int(batch_size)
        else:
            last = torch.stack((batch_x[-1], batch_y[-1])).cpu()         # ONE host synchronisation
            b = int(last[0]) + 1
            if int(last[1]) + 1 != b:
                raise RuntimeError("knn: batch_x and batch_y disagree on the batch size")
    nx, ny = x.shape[0] // b, y.shape[0] // b
    if nx * b != x.shape[0] or ny * b != y.shape[0]:
        raise RuntimeError("knn: only equally sized batch items are supported")
    if nx < k:
        raise RuntimeError("knn: fewer candidates per batch item than k")
    row = torch.empty(b * ny * k, dtype=torch.int64, device=x.device)
    col = torch.empty(b * ny * k, dtype=torch.int64, device=x.device)
    _call('dclr_knn', 'knn', b, nx, ny, k, x.data_ptr(), y.data_ptr(), row.data_ptr(), col.data_ptr(),
                                  lib.stream_ptr())
    # upstream (torch-cluster 1.5.9, knn_cuda.cu) drops the slots no candidate was inserted into (squared distance
    # >= 1e10, the slots' initial value): `mask = col != -1` -- a host synchronisation there as here
    mask = col != -1
    return torch.stack((row[mask], col[mask]), dim=0)


# ------------------------------------------------------------------------------------------------
# level 2
# ------------------------------------------------------------------------------------------------
def fps_clouds(clouds: torch.Tensor, npoint: int) -> torch.Tensor:
    """clouds (B, N, C>=3) interleaved -> (B, npoint) int32. Clouds of 16385..65536 points go through the
    workspace variant (sorted points + running minima in a scratch buffer, only ~10 % of it touched per round)."""
    clouds = lib.dev_f32(clouds, 'clouds')
    b, n, c = clouds.shape
    if n > FUSED_MAX_POINTS:
        # beyond the fused samplers (65536 points per cloud): the level-1 operator, which takes any n (running minima in
        # global memory), on a packed copy of the coordinates
        return furthest_point_sample(clouds[:, :, :3].contiguous(), npoint)
    idx = torch.empty(b, npoint, dtype=torch.int32, device=clouds.device)
    need = lib.load().dclr_fps_workspace_bytes(b, n)
    if need > 0 and npoint * 4 <= 32 * 1024:
        ws = torch.empty((need + 3) // 4, dtype=torch.int32, device=clouds.device)
        _call('dclr_fps_clouds_ws', 'fps_clouds[%dx%d]' % (b, n), b, n, c, npoint, clouds.data_ptr(), idx.data_ptr(), ws.data_ptr(),
              need, lib.stream_ptr())
        return idx
    _call('dclr_fps_clouds', 'fps_clouds[%dx%d]' % (b, n), b, n, c, npoint, clouds.data_ptr(), idx.data_ptr(), lib.stream_ptr())
    return idx


def fps_group_layout(n: int):
    """(n_groups, group_size) of the spatial partition the sampling kernel can export, or None."""
    ng, gs = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib.load().dclr_fps_group_layout(n, ctypes.addressof(ng), ctypes.addressof(gs))
    return (ng.value, gs.value) if rc == 0 else None


def batch_view(batches) -> Optional[Tuple[int, int, int]]:
    """(pairs per batch, number of batches, stride in floats) when the (2B, N, C) batches of one launch lie at a constant
    stride in memory (the same tensor every time: stride 0; views of one staging chunk; a ring) -- then the grouped sampler
    and set abstraction read them in place (dclr_*_batched) -- else None (the caller concatenates)."""
    first = batches[0]
    if len(batches) < 2 or not first.is_contiguous() or first.dtype != torch.float32 or first.shape[0] % 2:
        return None
    if first.dim() != 3 or fps_group_layout(first.shape[1]) is None:
        return None                 # no grouped sampler at this cloud size (n <= 1024, n > 65536): those launches concatenate
    stride = batches[1].data_ptr() - first.data_ptr()
    if stride < 0 or stride % 4 or (0 < stride < first.numel() * 4):
        return None
    for i, t in enumerate(batches):
        if t.shape != first.shape or t.dtype != first.dtype or t.device != first.device or not t.is_contiguous() \
                or t.data_ptr() != first.data_ptr() + i * stride:
            return None
    return first.shape[0] // 2, len(batches), stride // 4


def fps_clouds_grouped(clouds: torch.Tensor, npoint: int, view: Optional[Tuple[int, int, int]] = None):
    """Sampling plus the kernel's spatial groups: (idx, group_pts, group_box, slice_box); the last three are None when
    the cloud size has no grouped kernel (then set abstraction sweeps exhaustively), slice_box alone where the groups are
    single 64-point slices or come from the workspace kernel (n > 16384).
    view = batch_view([...]): `clouds` is the FIRST of several batches read in place; results cover all of them in the
    concatenated order [templates of every batch | sources of every batch]."""
    clouds = lib.dev_f32(clouds, 'clouds')
    b, n, c = clouds.shape
    layout = fps_group_layout(n)
    if view is not None:
        per, nb, stride = view
        if layout is None or b != 2 * per or (n > 16384 and npoint * 4 > 32 * 1024):
            raise RuntimeError("batch view: no grouped sampler for these clouds (concatenate the batches instead)")
        b = 2 * per * nb
    elif layout is None or (n > 16384 and npoint * 4 > 32 * 1024):
        return fps_clouds(clouds, npoint), None, None, None
    else:
        # one batch: the plain call is the batched one with a single batch (odd cloud counts: no template / source halves)
        per, nb, stride = (b // 2, 1, 0) if b % 2 == 0 else (0, 0, 0)
    ng, gs = layout
    idx = torch.empty(b, npoint, dtype=torch.int32, device=clouds.device)
    gpts = torch.empty(b, ng * gs, 4, dtype=torch.float32, device=clouds.device)
    gbox = torch.empty(b, ng, 8, dtype=torch.float32, device=clouds.device)
    what = 'fps_clouds[%dx%d]' % (b, n)
    if nb == 0:                           # odd number of clouds: the older entry points (no slice boxes)
        if n > 16384:
            need = lib.load().dclr_fps_workspace_bytes(b, n)
            ws = torch.empty((need + 3) // 4, dtype=torch.int32, device=clouds.device)
            _call('dclr_fps_clouds_grouped_ws', what, b, n, c, npoint, clouds.data_ptr(), idx.data_ptr(), gpts.data_ptr(),
                  gbox.data_ptr(), ws.data_ptr(), need, lib.stream_ptr())
        else:
            _call('dclr_fps_clouds_grouped', what, b, n, c, npoint, clouds.data_ptr(), idx.data_ptr(), gpts.data_ptr(),
                  gbox.data_ptr(), lib.stream_ptr())
        return idx, gpts, gbox, None
    sbox = None
    if n <= 16384 and gs > 64 and SLICE_BOXES:
        sbox = torch.empty(b, ng * (gs // 64), 8, dtype=torch.float32, device=clouds.device)
    need = lib.load().dclr_fps_workspace_bytes(b, n) if n > 16384 else 0
    ws = torch.empty((need + 3) // 4, dtype=torch.int32, device=clouds.device) if need else None
    _call('dclr_fps_clouds_grouped_batched', what, b, n, c, npoint, clouds.data_ptr(), per, nb, stride, idx.data_ptr(),
          gpts.data_ptr(), gbox.data_ptr(), lib.ptr(sbox), lib.ptr(ws), need, lib.stream_ptr())
    return idx, gpts, gbox, sbox


def pack_sa_mlp(weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]) -> torch.Tensor:
    """[W1 b1 W2 b2 W3 b3] flat f32 buffer for dclr_sa_msg_fused (1x1 conv weights (out,in,1,1))."""
    parts = []
    for w, bias in zip(weights, biases):
        parts += [w.detach().reshape(w.shape[0], -1).reshape(-1), bias.detach().reshape(-1)]
    return torch.cat(parts).to(torch.float32).contiguous()


def sa_msg_fused(clouds: torch.Tensor, fps_idx: torch.Tensor, radii: Sequence[float], nsamples: Sequence[int],
                 mlps: List[torch.Tensor], want_counts: bool = False, groups=None, precision: Optional[str] = None,
                 view: Optional[Tuple[int, int, int]] = None, overflow: Optional[int] = None):
    """clouds (B,N,C), fps_idx (B,npoint) -> rows F (B*npoint, 68) [, counts (B,npoint,scales)].
    groups: (group_pts, group_box[, slice_box]) from fps_clouds_grouped for the same clouds, or None.
    precision: 'f16x2' (layers 2, 3 of the shared MLP on split-f16 operands) or 'f32'; default ops.PRECISION.
    view: as fps_clouds_grouped (clouds = the first batch, fps_idx / groups cover all batches).
    overflow: device address of the word the split-f16 layers set when an activation is clamped (lib.MappedFlag.dev_ptr;
    an even number of clouds only -- the entry that takes it is the batched one)."""
    clouds = lib.dev_f32(clouds, 'clouds')
    b, n, c = clouds.shape
    if view is not None:
        b = 2 * view[0] * view[1]
        if fps_idx.shape[0] != b or clouds.shape[0] != 2 * view[0]:
            raise RuntimeError("batch view: fps_idx must cover all batches of the view")
    npoint = fps_idx.shape[1]
    ns = len(radii)
    out = torch.empty(b * npoint, F_STRIDE, dtype=torch.float32, device=clouds.device)
    counts = torch.empty(b, npoint, ns, dtype=torch.int32, device=clouds.device) if want_counts else None
    radii_h = (ctypes.c_float * ns)(*[float(r) for r in radii])
    nsamp_h = (ctypes.c_int * ns)(*[int(s) for s in nsamples])
    mlp_h = (ctypes.c_void_p * ns)(*[lib.dev_f32(m, 'mlp').data_ptr() for m in mlps])
    sbox = groups[2] if groups is not None and len(groups) > 2 else None
    if view is None and (sbox is not None or overflow is not None) and b % 2 == 0:
        view = (b // 2, 1, 0)                                     # the batched entry with one batch = the plain call
    if view is not None:
        _call('dclr_sa_msg_fused_batched_ov', 'sa_msg_fused[%dx%d]' % (b, n), int((precision or PRECISION) == 'f16x2'), b, n, c,
              npoint, clouds.data_ptr(), view[0], view[1], view[2], fps_idx.data_ptr(), ns,
              ctypes.cast(radii_h, ctypes.c_void_p), ctypes.cast(nsamp_h, ctypes.c_void_p),
              ctypes.cast(mlp_h, ctypes.c_void_p), out.data_ptr(), lib.ptr(counts),
              None if groups is None else groups[0].data_ptr(), None if groups is None else groups[1].data_ptr(),
              lib.ptr(sbox), overflow, lib.stream_ptr())
        return (out, counts) if want_counts else out
    entry = 'dclr_sa_msg_fused_f16' if (precision or PRECISION) == 'f16x2' else 'dclr_sa_msg_fused'
    _call(entry, 'sa_msg_fused[%dx%d]' % (b, n), b, n, c, npoint, clouds.data_ptr(), fps_idx.data_ptr(), ns,
                                           ctypes.cast(radii_h, ctypes.c_void_p), ctypes.cast(nsamp_h, ctypes.c_void_p),
                                           ctypes.cast(mlp_h, ctypes.c_void_p), out.data_ptr(), lib.ptr(counts),
          None if groups is None else groups[0].data_ptr(), None if groups is None else groups[1].data_ptr(),
          lib.stream_ptr())
    return (out, counts) if want_counts else out


def _xyz_col(stride: int) -> int:
    return {F_STRIDE: 64, E_STRIDE: 256}[stride]


def rows_to_channels(rows: torch.Tensor, b: int, npoint: int, nfeat: int) -> torch.Tensor:
    """rows F / E (b*npoint, stride) -> reference layout (b, 3 + nfeat, npoint) [xyz | first nfeat features]."""
    rows = lib.dev_f32(rows, 'rows')
    out = torch.empty(b, 3 + nfeat, npoint, dtype=torch.float32, device=rows.device)
    _call('dclr_rows_to_channels', 'rows_to_channels', b, npoint, nfeat, _xyz_col(rows.shape[1]), rows.shape[1],
                                               rows.data_ptr(), out.data_ptr(), lib.stream_ptr())
    return out


def channels_to_rows(channels: torch.Tensor, stride: int) -> torch.Tensor:
    """reference layout (b, 3 + nfeat, npoint) -> rows (b*npoint, stride)."""
    channels = lib.dev_f32(channels, 'channels')
    b, ch, npoint = channels.shape
    rows = torch.empty(b * npoint, stride, dtype=torch.float32, device=channels.device)
    _call('dclr_channels_to_rows', 'channels_to_rows', b, npoint, ch - 3, _xyz_col(stride), stride, channels.data_ptr(),
                                               rows.data_ptr(), lib.stream_ptr())
    return rows


def pack_weight(w: torch.Tensor, kp: int, kmap: Optional[torch.Tensor] = None, tile16: bool = False) -> torch.Tensor:
    """Row-major (n_out, k_in) -> MFMA fragment order, K padded to kp, N padded to a multiple of 32
    (tile16: the 16-column layout of the fused flow-embedding kernel, N padded to a multiple of 16)."""
    w = lib.dev_f32(w.detach().reshape(w.shape[0], -1).contiguous(), 'w')
    n_out, k_in = w.shape
    if tile16:
        np_ = (n_out + 15) // 16 * 16
        packed = torch.empty(np_ * kp, dtype=torch.float32, device=w.device)
        _call('dclr_pack_weight16', 'pack_weight16', n_out, k_in, w.data_ptr(), lib.ptr(kmap), kp, np_,
              packed.data_ptr(), lib.stream_ptr())
        return packed
    np_ = (n_out + 31) // 32 * 32
    packed = torch.empty(np_ * kp, dtype=torch.float32, device=w.device)
    if kmap is not None:
        assert kmap.dtype == torch.int32 and kmap.numel() == kp and kmap.is_cuda
    _call('dclr_pack_weight', 'pack_weight', n_out, k_in, w.data_ptr(), lib.ptr(kmap), kp, np_, packed.data_ptr(),
                                          lib.stream_ptr())
    return packed


def linear(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], n: int, kp: int, relu: bool,
           ldy: Optional[int] = None, colmax_groups: Optional[int] = None) -> torch.Tensor:
    """rows x (m, ldx) -> rows y (m, ldy) = act(x[:, :kp] W^T + b), or, with colmax_groups = G,
    the (G, n) column maxima over each block of m/G rows."""
    x = lib.dev_f32(x, 'x')
    m, ldx = x.shape
    if colmax_groups is not None:
        out = torch.zeros(colmax_groups, n, dtype=torch.float32, device=x.device)
        _call('dclr_linear', 'linear+colmax[%dx%dx%d]' % (m, n, kp), m, n, kp, x.data_ptr(), ldx, w_packed.data_ptr(), lib.ptr(bias), 1, None, 0,
                                         out.data_ptr(), m // colmax_groups, lib.stream_ptr())
        return out
    ldy = n if ldy is None else ldy
    alloc = torch.zeros if ldy > n else torch.empty     # padding columns feed the next layer's zero weights
    y = alloc(m, ldy, dtype=torch.float32, device=x.device)
    _call('dclr_linear', 'linear[%dx%dx%d]' % (m, n, kp), m, n, kp, x.data_ptr(), ldx, w_packed.data_ptr(), lib.ptr(bias), int(relu),
                                     y.data_ptr(), ldy, None, 0, lib.stream_ptr())
    return y


def head_conv_fused(x: torch.Tensor, layers, groups: int) -> torch.Tensor:
    """x rows (m, ldx); layers = [(w_packed, bias, n, kp), ...] -> (groups, n_last) column maxima of the
    ReLU'd last layer over each block of m / groups rows (the conv chain of the pose head in one launch)."""
    x = lib.dev_f32(x, 'x')
    m, ldx = x.shape
    nl = len(layers)
    k_h = (ctypes.c_int * nl)(*[int(l[3]) for l in layers])
    n_h = (ctypes.c_int * nl)(*[int(l[2]) for l in layers])
    w_h = (ctypes.c_void_p * nl)(*[l[0].data_ptr() for l in layers])
    b_h = (ctypes.c_void_p * nl)(*[l[1].data_ptr() for l in layers])
    out = torch.zeros(groups, layers[-1][2], dtype=torch.float32, device=x.device)
    _call('dclr_head_conv_fused', 'head_conv_fused[%dx%d]' % (groups, m // groups), m, nl, ctypes.cast(k_h, ctypes.c_void_p), ctypes.cast(n_h, ctypes.c_void_p),
          ctypes.cast(w_h, ctypes.c_void_p), ctypes.cast(b_h, ctypes.c_void_p), x.data_ptr(), ldx, out.data_ptr(),
          m // groups, lib.stream_ptr())
    return out


def pack_weight_f16(w: torch.Tensor, kp: int, width: int, kmap: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row-major (n_out, k_in) -> split-fp16 fragments (hi plane | lo plane) for `width`-column MFMA tiles."""
    w = lib.dev_f32(w.detach().reshape(w.shape[0], -1).contiguous(), 'w')
    check_f16_range(w, 'pack_weight_f16')
    n_out, k_in = w.shape
    np_ = (n_out + width - 1) // width * width
    packed = torch.empty(np_ * kp, dtype=torch.float32, device=w.device)          # 2 planes of np*kp halves
    if kmap is not None:
        assert kmap.dtype == torch.int32 and kmap.numel() == kp and kmap.is_cuda
    _call('dclr_pack_weight_f16', 'pack_weight_f16', n_out, k_in, w.data_ptr(), lib.ptr(kmap), kp, width,
          packed.data_ptr(), lib.stream_ptr())
    return packed


def head_conv_fused_f16(x: torch.Tensor, k_in: int, layers, groups: int) -> torch.Tensor:
    """head_conv_fused on the split-fp16 matrix path; layers = [(w_packed_f16, bias, n, kp), ...], kp % 16 == 0."""
    x = lib.dev_f32(x, 'x')
    m, ldx = x.shape
    nl = len(layers)
    k_h = (ctypes.c_int * nl)(*[int(l[3]) for l in layers])
    n_h = (ctypes.c_int * nl)(*[int(l[2]) for l in layers])
    w_h = (ctypes.c_void_p * nl)(*[l[0].data_ptr() for l in layers])
    b_h = (ctypes.c_void_p * nl)(*[l[1].data_ptr() for l in layers])
    out = torch.zeros(groups, layers[-1][2], dtype=torch.float32, device=x.device)
    _call('dclr_head_conv_fused_f16', 'head_conv_fused[%dx%d]' % (groups, m // groups), m, nl, int(k_in), ctypes.cast(k_h, ctypes.c_void_p),
          ctypes.cast(n_h, ctypes.c_void_p), ctypes.cast(w_h, ctypes.c_void_p), ctypes.cast(b_h, ctypes.c_void_p),
          x.data_ptr(), ldx, out.data_ptr(), m // groups, lib.stream_ptr())
    return out


def flow_f16_tile(k: int) -> int:
    """Tile width the library's split-f16 flow kernel wants its layer-2 / layer-3 weights packed with when it runs k
    neighbours per point (32 from k = 29 up, 16 below; A/B builds force one)."""
    return int(lib.load().dclr_flow_f16_tile(int(k)))


def flow_embedding_fused_f16(f_rows: torch.Tensor, knn_idx: torch.Tensor, pt: torch.Tensor, ps: torch.Tensor,
                             w1a: torch.Tensor, b1: torch.Tensor, w2p: torch.Tensor, b2: torch.Tensor,
                             w3p: torch.Tensor, b3: torch.Tensor, radius: float) -> torch.Tensor:
    pairs, npoint, k = knn_idx.shape
    e = torch.empty(pairs * npoint, E_STRIDE, dtype=torch.float32, device=f_rows.device)
    _call('dclr_flow_embedding_fused_f16', 'flow_embedding[%dx%dx%d]' % (pairs, npoint, k), pairs, npoint, k, float(radius), f_rows.data_ptr(),
          knn_idx.data_ptr(), pt.data_ptr(), ps.data_ptr(), w1a.data_ptr(), b1.data_ptr(), w2p.data_ptr(),
          b2.data_ptr(), w3p.data_ptr(), b3.data_ptr(), e.data_ptr(), lib.stream_ptr())
    return e


def knn_rows(f_rows: torch.Tensor, pairs: int, npoint: int, k: int) -> torch.Tensor:
    f_rows = lib.dev_f32(f_rows, 'f_rows')
    idx = torch.empty(pairs, npoint, k, dtype=torch.int32, device=f_rows.device)
    _call('dclr_knn_rows', 'knn_rows[%dx%dx%d]' % (pairs, npoint, k), pairs, npoint, k, f_rows.data_ptr(), idx.data_ptr(), lib.stream_ptr())
    return idx


def flow_embedding_fused(f_rows: torch.Tensor, knn_idx: torch.Tensor, pt: torch.Tensor, ps: torch.Tensor,
                         w1a: torch.Tensor, b1: torch.Tensor, w2p: torch.Tensor, b2: torch.Tensor,
                         w3p: torch.Tensor, b3: torch.Tensor, radius: float) -> torch.Tensor:
    pairs, npoint, k = knn_idx.shape
    e = torch.empty(pairs * npoint, E_STRIDE, dtype=torch.float32, device=f_rows.device)
    _call('dclr_flow_embedding_fused', 'flow_embedding[%dx%dx%d]' % (pairs, npoint, k), pairs, npoint, k, float(radius), f_rows.data_ptr(),
                                                   knn_idx.data_ptr(), pt.data_ptr(), ps.data_ptr(), w1a.data_ptr(),
                                                   b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), w3p.data_ptr(),
                                                   b3.data_ptr(), e.data_ptr(), lib.stream_ptr())
    return e


def fc(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: int) -> torch.Tensor:
    x, w = lib.dev_f32(x, 'x'), lib.dev_f32(w, 'w')
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty(m, n, dtype=torch.float32, device=x.device)
    _call('dclr_fc', 'fc[%d]' % m, m, n, k, x.data_ptr(), w.data_ptr(), lib.ptr(bias), act, y.data_ptr(),
                                 lib.stream_ptr())
    return y
