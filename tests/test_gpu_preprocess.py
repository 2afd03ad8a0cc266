"""Scan preparation kernel against the numpy restatement of the reference transforms: identical rows, identical order."""
import numpy as np
import pytest
import torch

from oracle import preprocess as opre
from deepclr_amd import preprocess

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _scan(n, c, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 25, size=(n, c)).astype(np.float32)
    if c > 2:
        x[:, 2] = rng.normal(-1, 0.5, size=n)
    return x


@pytest.mark.parametrize('n,c,kw', [
    (120000, 4, dict()),
    (120000, 4, dict(nth=3, start=2)),
    (120000, 4, dict(min_range=2.0, max_range=60.0)),
    (123457, 4, dict(nth=2, start=1, min_range=3.0, max_range=40.0, input_dim=3)),
    (70001, 5, dict(nth=7, start=0, min_range=0.0, max_range=30.0, input_dim=4)),
    (1, 3, dict()), (1023, 3, dict(min_range=10.0)), (1025, 3, dict(nth=2, start=1, max_range=20.0)),
    (5000, 3, dict(min_range=1e6)),                                  # everything cropped
    (4, 4, dict(nth=5, start=4)),                                    # start beyond the scan: empty
])
def test_prepare_cloud_matches_reference_transforms(n, c, kw):
    raw = _scan(n, c, seed=n)
    if n > 100:
        raw[17, 0] = np.nan                                          # a NaN coordinate is never inside a range
        raw[18, :2] = [60.0, -60.0] if 'max_range' in kw else raw[18, :2]   # on the boundary: kept (<=)
    want = opre.prepare_cloud(raw, **kw)
    got = preprocess.prepare_cloud(torch.from_numpy(raw).to(DEV), **kw).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)


def _golden_cases():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'preprocess.npz'))
    for name in sorted({k.split('/')[0] for k in g.files}):
        nth, start, lo, hi, dim = g[name + '/params']
        kw = dict(nth=int(nth), start=int(start), min_range=float(lo), max_range=float(hi))
        if dim >= 0:
            kw['input_dim'] = int(dim)
        yield name, g[name + '/raw'], g[name + '/want'], kw


def test_prepare_cloud_matches_vectors_written_by_the_reference_transforms():
    """tests/golden/preprocess.npz: outputs of the reference's own TruncateDimension / SystematicErasing / RangeSelection
    (tests/golden/make_preprocess_golden.py ran /root/reference/deepclr/data/transforms/transforms.py) -- NaN, +inf,
    boundary values, empty and single-point results included. Bit-exact, same row order."""
    seen = 0
    for name, raw, want, kw in _golden_cases():
        got = preprocess.prepare_cloud(torch.from_numpy(raw).to(DEV), **kw).cpu().numpy()
        assert got.shape == want.shape, (name, got.shape, want.shape)
        assert np.array_equal(got, want, equal_nan=True), name
        seen += 1
    assert seen == 11


def test_subsample_is_a_subset_without_repeats_and_rejects_bad_input():
    raw = torch.from_numpy(_scan(30000, 4, 1)).to(DEV)
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    sub = preprocess.subsample(raw, 16384, generator=g)
    assert sub.shape == (16384, 4)
    keys = {tuple(r) for r in raw.cpu().numpy().tolist()}
    rows = [tuple(r) for r in sub.cpu().numpy().tolist()]
    assert len(set(rows)) == 16384 and set(rows) <= keys
    assert preprocess.subsample(raw, 40000) is raw
    with pytest.raises(RuntimeError):
        preprocess.prepare_cloud(raw.cpu())
    with pytest.raises(RuntimeError):
        preprocess.prepare_cloud(raw, nth=2, start=2)
    with pytest.raises(RuntimeError):
        preprocess.prepare_cloud(raw, input_dim=5)
