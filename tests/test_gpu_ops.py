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
    return torch.from_numpy(np.ascontiguousarray(x))


@pytest.mark.parametrize('kind,b,n,m', [
    ('normal', 2, 1, 4), ('normal', 3, 37, 20), ('dup', 2, 96, 200), ('grid', 2, 300, 128),
    ('normal', 2, 1000, 256), ('kitti', 2, 1024, 512), ('dup', 2, 1500, 512), ('grid', 2, 2048, 300),
    ('kitti', 2, 4096, 1024), ('normal', 2, 5000, 700), ('kitti', 3, 16384, 1024), ('dup', 2, 12000, 512),
    ('normal', 1, 20000, 200), ('kitti', 1, 65536, 150), ('grid', 1, 40000, 100),
    ('normal', 2, 1025, 300), ('dup', 2, 2049, 400), ('grid', 3, 16383, 600), ('kitti', 2, 8193, 1100),
    ('grid', 2, 9000, 9000),
    # a last group of one or two points that is not the first group of its wave (its box lane holds no point)
    ('kitti', 2, 1089, 64), ('kitti', 2, 2113, 64), ('kitti', 2, 4225, 64), ('kitti', 2, 8449, 64), ('kitti', 2, 8192 + 512 + 2, 64),
    ('normal', 2, 12288 + 768 + 1, 64),
])
def test_fps_bit_exact(kind, b, n, m):
    xyz = _cloud(kind, b, n, seed=n + m)
    want = oracle.furthest_point_sample(xyz, m)
    got = ops.furthest_point_sample(xyz.to(DEV), m).cpu()
    assert torch.equal(got, want), 'first mismatch at {}'.format((got != want).nonzero()[:3].tolist())


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
])
def test_knn_bit_exact(kind, b, nx, ny, k):
    x = _cloud(kind, b, nx, seed=1).reshape(-1, 3)
    y = _cloud(kind, b, ny, seed=2).reshape(-1, 3)
    bx, by = torch.arange(b).repeat_interleave(nx), torch.arange(b).repeat_interleave(ny)
    want = oracle.knn(x, y, k, bx, by)
    got = ops.knn(x.to(DEV), y.to(DEV), k, bx.to(DEV), by.to(DEV)).cpu()
    assert torch.equal(got, want)


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
