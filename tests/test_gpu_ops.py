"""GPU parity, level 1: every operator of the C ABI against the CPU oracle, bit for bit.

Integer/index results must be identical (fixed seeds, tie-heavy inputs included); float
gathers must be identical too (pure copies)."""
import numpy as np
import pytest
import torch

import oracle
from deepclr_amd import ops, synthetic
from helpers import degenerate_batch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _cloud(kind, b, n, seed):
    rng = np.random.default_rng(seed)
    if kind == 'normal':
        x = rng.normal(size=(b, n, 3)).astype(np.float32)
    elif kind == 'kitti':
        x = synthetic.make_batch('kitti', (b + 1) // 2, n, first_pair=seed)[:b, :, :3]
    elif kind == 'dup':                       # duplicates force exact distance ties
        x = degenerate_batch((b + 1) // 2, n, 3, seed)[:b]
    elif kind == 'grid':                      # lattice: many equal distances between distinct points
        x = rng.integers(0, 6, size=(b, n, 3)).astype(np.float32)
    elif kind == 'line':                      # stretched along one axis: all 12 sorting-cell bits go to x (no key table)
        x = (rng.normal(size=(b, n, 3)) * np.array([500.0, 0.01, 0.002])).astype(np.float32)
    elif kind == 'sheet':                     # a thin sheet: 6 + 6 + 0 bits (the LiDAR case), far from the origin
        x = (rng.uniform(-1, 1, size=(b, n, 3)) * np.array([80.0, 60.0, 0.05]) + np.array([1000.0, -2000.0, 3.0])).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.mark.parametrize('kind,b,n,m', [
    ('normal', 2, 1, 4), ('normal', 3, 37, 20), ('dup', 2, 96, 200), ('grid', 2, 300, 128),
    ('normal', 2, 1000, 256), ('kitti', 2, 1024, 512), ('dup', 2, 1500, 512), ('grid', 2, 2048, 300),
    ('kitti', 2, 4096, 1024), ('normal', 2, 5000, 700), ('kitti', 3, 16384, 1024), ('dup', 2, 12000, 512),
    ('normal', 1, 20000, 200), ('kitti', 1, 65536, 150), ('grid', 1, 40000, 100),
    ('normal', 2, 1025, 300), ('dup', 2, 2049, 400), ('grid', 3, 16383, 600), ('kitti', 2, 8193, 1100),
    ('grid', 2, 9000, 9000),
    ('line', 2, 3000, 200), ('line', 2, 16384, 300), ('line', 1, 40000, 200), ('sheet', 2, 5000, 300), ('sheet', 1, 30000, 200),
    # a last group of one or two points that is not the first group of its wave (its box lane holds no point)
    ('kitti', 2, 1089, 64), ('kitti', 2, 2113, 64), ('kitti', 2, 4225, 64), ('kitti', 2, 8449, 64), ('kitti', 2, 8192 + 512 + 2, 64),
    ('normal', 2, 12288 + 768 + 1, 64),
])
def test_fps_bit_exact(kind, b, n, m):
    xyz = _cloud(kind, b, n, seed=n + m)
    want = oracle.furthest_point_sample(xyz, m)
    got = ops.furthest_point_sample(xyz.to(DEV), m).cpu()
    assert torch.equal(got, want), 'first mismatch at {}'.format((got != want).nonzero()[:3].tolist())


def test_fps_reproduces_the_reference_numpy_sampler(golden_dir):
    """The rows the reference's own numpy sampler picks (tests/golden/fps_reference.npz, written by
    tests/golden/make_fps_golden.py from deepclr/data/transforms/transforms.py:47-59) on tie-free clouds, against both
    sampler entry points of the library (level 1 and the fused path's `fps_clouds`, single clouds and all in one batch)."""
    import os
    g = np.load(os.path.join(golden_dir, 'fps_reference.npz'))
    names = sorted({k.split('/')[0] for k in g.files})
    checked = 0
    for name in names:
        pts, picks, m = g[name + '/points'], g[name + '/picks'], int(g[name + '/m'])
        if m >= len(pts):
            continue
        x = torch.from_numpy(pts)[None].to(DEV)
        assert np.array_equal(ops.furthest_point_sample(x, m).cpu().numpy()[0], picks), name
        assert np.array_equal(ops.fps_clouds(x, m).cpu().numpy()[0], picks), name
        checked += 1
    assert checked >= 5
    # round 5: the sizes of the headline kernels -- fps_pruned_kernel<1024,16,4,3> (N = 16384, the c2 sampler) and the
    # workspace kernel fps_paged_kernel (N = 20000) -- are pinned by the reference's sampler too, not by the oracle alone
    assert {'kitti_n16384_m1024', 'kitti_n20000_m256'} <= set(names)
    same = [n for n in names if g[n + '/points'].shape[0] == 2048 and int(g[n + '/m']) == 512]
    if len(same) > 1:                                              # several reference clouds in one launch
        x = torch.from_numpy(np.stack([g[n + '/points'] for n in same])).to(DEV)
        got = ops.fps_clouds(x, 512).cpu().numpy()
        for row, n in zip(got, same):
            assert np.array_equal(row, g[n + '/picks']), n


@pytest.mark.parametrize('kind, b, n, m', [('normal', 2, 20000, 200), ('kitti', 1, 40000, 150), ('grid', 1, 65536, 64),
                                           ('dup', 2, 17000, 300), ('line', 1, 32768, 100)])
def test_fps_fallback_kernels_without_a_workspace(kind, b, n, m):
    """dclr_fps_clouds on 16385..65536 points: no workspace in the signature, so the running minima stay in registers and the
    coordinates are re-read every round (fps_stream_kernel<32> / <64>; also what dclr_furthest_point_sampling falls back to
    when the stream-ordered allocator refuses). Round 6 rewrote their prologue and load scheduling (they spilled 252 / 904
    bytes): same samples as the oracle, bit for bit, ties and duplicates included; with `temp`, the level-1 side effect too."""
    from deepclr_amd import lib
    xyz = _cloud(kind, b, n, 23)
    want = oracle.furthest_point_sample(xyz, m)
    x = xyz.to(DEV).contiguous()
    idx = torch.full((b, m), -7, dtype=torch.int32, device=DEV)
    lib.check(lib.load().dclr_fps_clouds(b, n, 3, m, x.data_ptr(), idx.data_ptr(), lib.stream_ptr()), 'dclr_fps_clouds')
    assert torch.equal(idx.cpu(), want)
    assert torch.equal(ops.fps_clouds(x, m).cpu(), want)             # (the workspace kernel the Python path takes: same samples)


def test_fps_level1_large_cloud_and_temp_side_effect():
    """Level 1 through the C symbol: n > 65536 takes the global-temp kernel, 16385..65536 the workspace kernel on a
    stream-ordered scratch allocation, smaller clouds the register kernel; temp must hold the running minima over the
    first m - 1 samples whichever kernel ran (several samples per round included)."""
    from deepclr_amd import lib
    xyz = _cloud('normal', 1, 70000, 5)
    m = 48
    want = oracle.furthest_point_sample(xyz, m)
    x = xyz.to(DEV)
    for n_use in (70000, 40000, 20000, 16384, 3000):
        xs = x[:, :n_use].contiguous()
        temp = torch.full((1, n_use), 1e10, device=DEV)
        idx = torch.empty(1, m, dtype=torch.int32, device=DEV)
        lib.check(lib.load().dclr_furthest_point_sampling(1, n_use, m, xs.data_ptr(), temp.data_ptr(), idx.data_ptr(),
                                                          lib.stream_ptr()), 'fps')
        ref_idx = want if n_use == 70000 else oracle.furthest_point_sample(xyz[:, :n_use].contiguous(), m)
        assert torch.equal(idx.cpu(), ref_idx)
        sel = xs[0, idx[0, :-1].long()]                              # temp = min distance to the first m-1 picks
        ref_temp = ((xs[0, :, None, :] - sel[None]) ** 2).sum(-1).min(dim=1).values
        torch.testing.assert_close(temp[0], ref_temp, rtol=1e-5, atol=1e-6)


def test_fps_level1_concurrent_streams_with_stream_ordered_scratch():
    """dclr_furthest_point_sampling allocates its workspace in stream order (hipMallocAsync) for 16384 < n <= 65536 -- the
    upstream signature has no workspace argument. Two streams calling it at the same time, several times over, must each
    get the oracle's indices and running minima: their allocations, kernels and frees interleave on the device."""
    from deepclr_amd import lib
    m = 96
    clouds = [_cloud('normal', 2, 20000, 11), _cloud('kitti', 2, 20000, 12)]
    want = [oracle.furthest_point_sample(c, m) for c in clouds]
    xs = [c.to(DEV) for c in clouds]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    torch.cuda.synchronize()
    for rep in range(4):
        for i, st in enumerate(streams):                      # enqueue on both streams back to back: the launches overlap
            with torch.cuda.stream(st):
                temp = torch.full((2, 20000), 1e10, device=DEV)
                idx = torch.empty(2, m, dtype=torch.int32, device=DEV)
                lib.check(lib.load().dclr_furthest_point_sampling(2, 20000, m, xs[i].data_ptr(), temp.data_ptr(), idx.data_ptr(),
                                                                  st.cuda_stream), 'fps')
                outs[i].append((idx, temp))
    torch.cuda.synchronize()
    for i in range(2):
        sel = xs[i][:, :, None, :]
        for idx, temp in outs[i]:
            assert torch.equal(idx.cpu(), want[i])
            picks = torch.gather(xs[i], 1, idx[:, :-1].long()[:, :, None].expand(-1, -1, 3))          # (2, m-1, 3)
            ref_temp = ((sel - picks[:, None, :, :]) ** 2).sum(-1).min(dim=2).values
            torch.testing.assert_close(temp, ref_temp, rtol=1e-5, atol=1e-6)


def test_fps_clouds_matches_level1_on_interleaved_input():
    x = torch.from_numpy(synthetic.make_batch('kitti', 1, 3000)).to(DEV)       # (2, 3000, 4)
    a = ops.fps_clouds(x, 333)
    b = ops.furthest_point_sample(x[:, :, :3].contiguous(), 333)
    assert torch.equal(a, b)


@pytest.mark.parametrize('kind,b,n,m,radius,nsample', [
    ('kitti', 2, 2048, 256, 1.0, 32), ('kitti', 2, 16384, 1024, 0.5, 512), ('dup', 2, 96, 50, 0.3, 16),
    ('grid', 1, 500, 100, 1.0, 8), ('normal', 2, 777, 130, 0.05, 4), ('normal', 1, 4096, 64, 10.0, 1024),
])
def test_ball_query_gather_group_bit_exact(kind, b, n, m, radius, nsample):
    xyz = _cloud(kind, b, n, seed=n)
    fps = oracle.furthest_point_sample(xyz, m)
    xyz_t = xyz.transpose(1, 2).contiguous()
    new_xyz_o = oracle.gather_operation(xyz_t, fps)
    new_xyz_g = ops.gather_operation(xyz_t.to(DEV), fps.to(DEV))
    assert torch.equal(new_xyz_g.cpu(), new_xyz_o)
    new_xyz = new_xyz_o.transpose(1, 2).contiguous()
    if kind == 'normal' and radius < 0.1:
        new_xyz[0, 0] = 100.0                                       # a centroid with no point in range
    want = oracle.ball_query(radius, nsample, xyz, new_xyz)
    got = ops.ball_query(radius, nsample, xyz.to(DEV), new_xyz.to(DEV))
    assert torch.equal(got.cpu(), want)
    feats = torch.from_numpy(np.random.default_rng(1).normal(size=(b, 5, n)).astype(np.float32))
    assert torch.equal(ops.grouping_operation(feats.to(DEV), got).cpu(), oracle.grouping_operation(feats, want))


@pytest.mark.parametrize('kind,b,nx,ny,k', [
    ('normal', 2, 64, 64, 5), ('kitti', 2, 1024, 1024, 20), ('dup', 2, 512, 512, 30), ('grid', 2, 300, 200, 16),
    ('normal', 1, 3000, 100, 64), ('normal', 3, 100, 257, 7),
    # more neighbours than the rank selection takes (k > 40) and than round 4 accepted (k > 64): one arg-min round per neighbour
    ('normal', 2, 300, 50, 100), ('dup', 2, 200, 40, 128), ('grid', 1, 1024, 64, 200), ('kitti', 1, 128, 128, 128),
])
def test_knn_bit_exact(kind, b, nx, ny, k):
    x = _cloud(kind, b, nx, seed=1).reshape(-1, 3)
    y = _cloud(kind, b, ny, seed=2).reshape(-1, 3)
    bx, by = torch.arange(b).repeat_interleave(nx), torch.arange(b).repeat_interleave(ny)
    want = oracle.knn(x, y, k, bx, by)
    got = ops.knn(x.to(DEV), y.to(DEV), k, bx.to(DEV), by.to(DEV)).cpu()
    assert torch.equal(got, want)


def test_knn_drops_candidates_beyond_the_initial_slot_distance():
    """torch-cluster 1.5.9 starts its k slots at distance 1e10 / index -1 and inserts on "slot > distance": a candidate at a
    squared distance of 1e10 or more is never taken and the Python side drops the -1 slots. Queries with 4 near and 4 far
    candidates and k = 6 come back with 4 neighbours each (DeepCLR's own .view(2, G, k) could not survive that: its clouds
    never get there); one candidate exactly AT 1e10 is dropped too."""
    near = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]])
    far = torch.tensor([[2.0e5, 0, 0], [0, -3.0e5, 0], [1.0e5, 0, 0], [4.0e5, 4.0e5, 0]])       # (1e5)^2 = 1e10 exactly
    x = torch.cat([near, far, near + 0.5, far * 2]).reshape(-1, 3)                              # two batch items of 8
    y = torch.tensor([[0.0, 0, 0], [0.1, 0.1, 0.1], [0.0, 0, 0], [0.2, 0, 0]])
    bx, by = torch.arange(2).repeat_interleave(8), torch.arange(2).repeat_interleave(2)
    want = oracle.knn(x, y, 6, bx, by)
    got = ops.knn(x.to(DEV), y.to(DEV), 6, bx.to(DEV), by.to(DEV)).cpu()
    assert want.shape[1] < 4 * 6 and torch.equal(got, want), (got, want)


def test_ops_reject_cpu_tensors_and_bad_sizes():
    with pytest.raises(RuntimeError):
        ops.furthest_point_sample(torch.zeros(1, 8, 3), 4)
    with pytest.raises(RuntimeError):
        ops.knn(torch.zeros(4, 3, device=DEV), torch.zeros(4, 3, device=DEV), 8)     # fewer candidates than k
    with pytest.raises(RuntimeError):
        ops.ball_query(0.1, 4, torch.zeros(1, 8, 3, device=DEV).transpose(1, 2), torch.zeros(1, 2, 3, device=DEV))


@pytest.mark.parametrize('kind,b,n,m', [
    ('kitti', 2, 65536, 1024), ('normal', 2, 20000, 300), ('grid', 2, 40000, 500), ('dup', 2, 32768, 400),
    ('kitti', 1, 16385, 200), ('normal', 1, 32769, 257), ('grid', 1, 65535, 64),
    ('kitti', 1, 16386, 60), ('kitti', 1, 16384 + 256 + 1, 60), ('kitti', 1, 32768 + 512 + 3, 60),   # tiny last group
])
def test_fps_large_clouds_workspace_kernel_bit_exact(kind, b, n, m):
    """16384 < n <= 65536: the spatially pruned kernel with its points in a workspace (ops.fps_clouds) against
    the oracle and against the level-1 entry point (same kernel on a stream-ordered scratch allocation)."""
    xyz = _cloud(kind, b, n, seed=n + m)
    want = oracle.furthest_point_sample(xyz, m)
    x = xyz.to(DEV)
    got = ops.fps_clouds(x, m).cpu()
    assert torch.equal(got, want), 'first mismatch at {}'.format((got != want).nonzero()[:3].tolist())
    assert torch.equal(ops.furthest_point_sample(x, m).cpu(), want)


# ---- randomised shapes (hypothesis): every kernel variant boundary gets crossed sooner or later -------------------
from hypothesis import HealthCheck, given, settings, strategies as st      # noqa: E402

_KINDS = st.sampled_from(['normal', 'kitti', 'dup', 'grid'])
_SETTINGS = dict(max_examples=40, deadline=None, derandomize=True,
                 suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


@settings(**_SETTINGS)
@given(kind=_KINDS, b=st.integers(1, 3), n=st.integers(1, 40000), m=st.integers(1, 700), seed=st.integers(0, 999))
def test_fps_bit_exact_random_shapes(kind, b, n, m, seed):
    xyz = _cloud(kind, b, n, seed)
    want = oracle.furthest_point_sample(xyz, m)
    got = ops.fps_clouds(xyz.to(DEV), m).cpu()                        # register / pruned / paged kernels by n
    assert torch.equal(got, want), (kind, b, n, m, (got != want).nonzero()[:3].tolist())


@settings(**_SETTINGS)
@given(kind=_KINDS, b=st.integers(1, 3), nx=st.integers(8, 3000), ny=st.integers(1, 400), k=st.integers(1, 40),
       seed=st.integers(0, 999))
def test_knn_bit_exact_random_shapes(kind, b, nx, ny, k, seed):
    k = min(k, nx)
    x = _cloud(kind, b, nx, seed).reshape(-1, 3)
    y = _cloud(kind, b, ny, seed + 1).reshape(-1, 3)
    bx, by = torch.arange(b).repeat_interleave(nx), torch.arange(b).repeat_interleave(ny)
    want = oracle.knn(x, y, k, bx, by)
    got = ops.knn(x.to(DEV), y.to(DEV), k, bx.to(DEV), by.to(DEV)).cpu()
    assert torch.equal(got, want), (kind, b, nx, ny, k)


@settings(**_SETTINGS)
@given(kind=_KINDS, n=st.integers(64, 6000), m=st.integers(1, 300), radius=st.floats(0.05, 3.0), nsample=st.integers(1, 64),
       seed=st.integers(0, 999))
def test_ball_query_bit_exact_random_shapes(kind, n, m, radius, nsample, seed):
    xyz = _cloud(kind, 2, n, seed)
    fps = oracle.furthest_point_sample(xyz, min(m, n))
    new_xyz = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    want = oracle.ball_query(radius, nsample, xyz, new_xyz)
    got = ops.ball_query(radius, nsample, xyz.to(DEV), new_xyz.to(DEV)).cpu()
    assert torch.equal(got, want), (kind, n, m, radius, nsample)


# ------------------------------------------------------------------------------------------------
# backward of gather / group (reference wrappers pointnet2.patch:144-158, 290-304)
# ------------------------------------------------------------------------------------------------
def _group_ref(features, idx):
    """CPU oracle of the index semantics: out[b, c, j, s] = features[b, c, idx[b, j, s]] in plain torch (autograd)."""
    b, c, n = features.shape
    flat = idx.reshape(b, 1, -1).expand(-1, c, -1).long()
    return torch.gather(features, 2, flat).reshape(b, c, *idx.shape[1:])


@pytest.mark.parametrize('kind,b,c,n,npoint,nsample,radius', [
    ('dup', 2, 3, 1500, 64, 32, 0.25),        # duplicates: tie-heavy rows, long padding runs
    ('grid', 2, 4, 2048, 100, 64, 1.1),       # lattice
    ('kitti', 2, 1, 4096, 128, 512, 1.0),     # the model's own shapes: 1 feature channel, nsample 512 (mostly padding)
    ('normal', 3, 7, 333, 50, 16, 0.6),
    ('normal', 1, 2, 5000, 17, 1000, 3.0),    # nsample far beyond a wave: runs span many waves
])
def test_group_points_grad_matches_autograd_over_the_index_semantics(kind, b, c, n, npoint, nsample, radius):
    xyz = _cloud(kind, b, n, seed=n + nsample)
    fps = oracle.furthest_point_sample(xyz, npoint)
    new_xyz = torch.gather(xyz, 1, fps.long()[:, :, None].expand(-1, -1, 3))
    idx = oracle.ball_query(radius, nsample, xyz, new_xyz)                       # (b, npoint, nsample) with padding repeats
    assert int((idx[:, :, 1:] == idx[:, :, :1]).sum()) > 0                         # the padding the kernel's run folding is for
    rng = np.random.default_rng(7)
    feats = torch.from_numpy(rng.normal(size=(b, c, n)).astype(np.float32))
    grad_out = torch.from_numpy(rng.normal(size=(b, c, npoint, nsample)).astype(np.float32))
    # float64 reference gradient (exact up to rounding of the final sum) and the f32 autograd result on the CPU
    f64 = feats.double().requires_grad_(True)
    _group_ref(f64, idx).backward(grad_out.double())
    f32 = feats.clone().requires_grad_(True)
    _group_ref(f32, idx).backward(grad_out)
    # the HIP operator through its autograd Function
    from deepclr_amd.pointnet2 import grouping_operation
    f_dev = feats.to(DEV).requires_grad_(True)
    out = grouping_operation(f_dev, idx.to(DEV))
    assert torch.equal(out.detach().cpu(), _group_ref(feats, idx))               # forward: a pure copy
    out.backward(grad_out.to(DEV))
    got = f_dev.grad.cpu()
    scale = float(f64.grad.abs().max())
    assert float((got.double() - f64.grad).abs().max()) <= 2e-6 * max(1.0, scale) * np.sqrt(nsample)
    # comparable to the CPU f32 scatter-add's own error against float64 (the run sums are tree sums: usually closer)
    err_cpu = float((f32.grad.double() - f64.grad).abs().max())
    err_hip = float((got.double() - f64.grad).abs().max())
    assert err_hip <= 4 * err_cpu + 1e-5 * max(1.0, scale), (err_hip, err_cpu)
    # points no row refers to receive exactly zero
    untouched = torch.ones(b, n, dtype=torch.bool)
    untouched.scatter_(1, idx.reshape(b, -1).long(), False)
    assert float(got.abs().sum(dim=1)[untouched].max() if untouched.any() else 0.0) == 0.0
    # level 1 through the C symbol accumulates INTO the caller's buffer (the reference zero-fills it first)
    from deepclr_amd import lib
    pre = torch.full((b, c, n), 0.5, device=DEV)
    go_dev, idx_dev = grad_out.to(DEV), idx.to(DEV)                           # kept alive across the asynchronous call
    lib.check(lib.load().dclr_group_points_grad(b, c, n, npoint, nsample, go_dev.data_ptr(), idx_dev.data_ptr(),
                                                pre.data_ptr(), lib.stream_ptr()), 'group_points_grad')
    torch.cuda.synchronize()
    torch.testing.assert_close(pre.cpu() - 0.5, got, rtol=1e-4, atol=1e-4 * max(1.0, scale))


@pytest.mark.parametrize('kind,b,c,n,npoint', [('kitti', 2, 3, 4096, 1024), ('dup', 3, 67, 300, 64), ('grid', 1, 5, 40, 100)])
def test_gather_points_grad_matches_autograd_over_the_index_semantics(kind, b, c, n, npoint):
    xyz = _cloud(kind, b, n, seed=n + npoint)
    idx = oracle.furthest_point_sample(xyz, npoint)                              # npoint > n: index 0 repeats
    rng = np.random.default_rng(3)
    feats = torch.from_numpy(rng.normal(size=(b, c, n)).astype(np.float32))
    grad_out = torch.from_numpy(rng.normal(size=(b, c, npoint)).astype(np.float32))
    f64 = feats.double().requires_grad_(True)
    _group_ref(f64, idx).backward(grad_out.double())
    from deepclr_amd.pointnet2 import gather_operation
    f_dev = feats.to(DEV).requires_grad_(True)
    out = gather_operation(f_dev, idx.to(DEV))
    assert torch.equal(out.detach().cpu(), _group_ref(feats, idx))
    out.backward(grad_out.to(DEV))
    torch.testing.assert_close(f_dev.grad.cpu().double(), f64.grad, rtol=1e-5, atol=1e-5 * max(1.0, float(f64.grad.abs().max())))


def test_composed_set_abstraction_differentiates_through_the_hip_operators():
    """The training step differentiates through the module (/root/reference/deepclr/engine/engines.py:57-84): with gradients
    enabled the composed path gathers / groups through the HIP operators (and their HIP backward) and leaves the shared MLP
    to torch. Gradients of a scalar loss w.r.t. the input features and the MLP weights against the same module evaluated
    with plain torch indexing on the CPU (float64)."""
    from deepclr_amd.pointnet2 import PointnetSAModuleMSG
    rng = np.random.default_rng(11)
    n, npoint, feat = 600, 48, 5
    pts = rng.normal(size=(2, n, 3)); pts /= np.linalg.norm(pts, axis=2, keepdims=True); pts *= rng.uniform(0.3, 1.0, size=(2, n, 1))
    xyz = torch.from_numpy(pts.astype(np.float32))
    feats = torch.from_numpy(rng.normal(size=(2, feat, n)).astype(np.float32))
    torch.manual_seed(4)
    sam = PointnetSAModuleMSG(npoint=npoint, radii=[0.3, 0.6], nsamples=[8, 24], mlps=[[feat, 12, 20], [feat, 16, 8]],
                              bn=False, use_xyz=True)
    # CPU float64 twin: same indices (oracle), torch indexing, same weights
    fps = oracle.furthest_point_sample(xyz, npoint)
    new_xyz = torch.gather(xyz, 1, fps.long()[:, :, None].expand(-1, -1, 3))
    f64 = feats.double().requires_grad_(True)
    outs = []
    twin = [[(u.conv.weight.detach().double().clone().requires_grad_(True), u.conv.bias.detach().double().clone().requires_grad_(True))
             for u in stack] for stack in sam.mlps]
    for (radius, nsample), layers in zip(zip(sam.radii, sam.nsamples), twin):
        bq = oracle.ball_query(radius, nsample, xyz, new_xyz)
        g = torch.cat((_group_ref(xyz.transpose(1, 2).double(), bq) - new_xyz.transpose(1, 2).double().unsqueeze(-1),
                       _group_ref(f64, bq)), dim=1)
        for w, bias in layers:
            g = torch.relu(torch.einsum('oc,bcjs->bojs', w.reshape(w.shape[0], -1), g) + bias.view(1, -1, 1, 1))
        outs.append(g.max(dim=3).values)
    want = torch.cat(outs, dim=1)
    probe = torch.from_numpy(rng.normal(size=tuple(want.shape))).double()
    (want * probe).sum().backward()
    # the module on the GPU, gradients enabled
    sam = sam.to(DEV)
    f_dev = feats.to(DEV).requires_grad_(True)
    _, out = sam(xyz.to(DEV), f_dev)
    torch.testing.assert_close(out.detach().cpu().double(), want.detach(), rtol=1e-4, atol=1e-5)
    (out * probe.float().to(DEV)).sum().backward()
    torch.testing.assert_close(f_dev.grad.cpu().double(), f64.grad, rtol=1e-3, atol=1e-4 * float(f64.grad.abs().max()))
    for stack, layers in zip(sam.mlps, twin):
        for u, (w, bias) in zip(stack, layers):
            torch.testing.assert_close(u.conv.weight.grad.cpu().double(), w.grad, rtol=1e-3, atol=1e-4 * float(w.grad.abs().max()))
            torch.testing.assert_close(u.conv.bias.grad.cpu().double(), bias.grad, rtol=1e-3, atol=1e-4 * float(bias.grad.abs().max()))
