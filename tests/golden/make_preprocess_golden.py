#!/usr/bin/env python3
"""Generate tests/golden/preprocess.npz by running the REFERENCE's own scan transforms.

Runs only in the build container (needs /root/reference). Loads
deepclr/data/transforms/{utils,transforms}.py by path (their package __init__ chain pulls in
torchvision / dataflow; `transforms3d`, which transforms.py imports for its random-transform classes only,
is an empty placeholder -- none of the three classes used here touches it) and applies, per case, the
evaluation-time composition of /root/reference/deepclr/data/transforms/build.py:36-42 restricted to its
deterministic members: TruncateDimension -> SystematicErasing(start fixed) -> RangeSelection, on the
'clouds' list of a sample. Inputs and outputs are stored; tests/test_oracle.py replays them against
oracle/preprocess.py (CPU) and tests/test_gpu_preprocess.py against csrc/prep.hip (GPU).

Usage:  python tests/golden/make_preprocess_golden.py  [--reference /root/reference]
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import preprocess as opre            # noqa: E402

# name -> (points, columns, seed, parameters); mirrors the cases of tests/test_gpu_preprocess.py at fixture size
CASES = {
    'identity':        (3000, 4, 11, dict()),
    'nth3_start2':     (3000, 4, 12, dict(nth=3, start=2)),
    'range_2_60':      (3000, 4, 13, dict(min_range=2.0, max_range=60.0)),
    'all_three':       (3457, 4, 14, dict(nth=2, start=1, min_range=3.0, max_range=40.0, input_dim=3)),
    'five_columns':    (2001, 5, 15, dict(nth=7, start=0, min_range=0.0, max_range=30.0, input_dim=4)),
    'single_point':    (1, 3, 16, dict()),
    'min_only':        (1023, 3, 17, dict(min_range=10.0)),
    'max_only_nth2':   (1025, 3, 18, dict(nth=2, start=1, max_range=20.0)),
    'all_cropped':     (500, 3, 19, dict(min_range=1e6)),
    'start_past_end':  (4, 4, 20, dict(nth=5, start=4)),
    'kitti_converter': (4000, 4, 21, dict(nth=2, start=0)),      # scripts/converter/kitti_odometry.py:14,22
}


def scan(n, c, seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 25, size=(n, c)).astype(np.float32)
    if c > 2:
        x[:, 2] = rng.normal(-1, 0.5, size=n)
    if n > 100:
        x[17, 0] = np.nan                          # a NaN coordinate is never inside a range
        x[18, :2] = [60.0, -60.0]                  # exactly on a boundary used below: kept (<=)
        x[19, :2] = [-2.0, 1.0]                    # exactly on the lower boundary: kept (>=)
        x[20, :2] = [np.inf, 0.0]
    return x


def load_reference_transforms(ref_root):
    for name in ('deepclr', 'deepclr.data', 'deepclr.data.transforms'):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules['transforms3d'] = types.ModuleType('transforms3d')       # imported, never called by the classes used

    def leaf(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(ref_root, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    leaf('deepclr.data.transforms.utils', 'deepclr/data/transforms/utils.py')
    return leaf('deepclr.data.transforms.transforms', 'deepclr/data/transforms/transforms.py')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference')
    args = ap.parse_args()
    ref = load_reference_transforms(args.reference)
    out = {}
    for name, (n, c, seed, kw) in CASES.items():
        raw = scan(n, c, seed)
        sample = {'clouds': [raw.copy(), raw[::-1].copy()]}
        steps = [ref.TruncateDimension(kw.get('input_dim', c)),
                 ref.SystematicErasing(kw.get('nth', 1), start=kw.get('start', 0)),
                 ref.RangeSelection(kw.get('min_range', 0.0), kw.get('max_range', np.inf), dim=3)]
        for t in steps:
            sample = t(sample)
        got = sample['clouds'][0]
        # the restatement must reproduce the reference on both clouds of the sample
        assert np.array_equal(opre.prepare_cloud(raw, **kw), got, equal_nan=True), name
        assert np.array_equal(opre.prepare_cloud(raw[::-1].copy(), **kw), sample['clouds'][1], equal_nan=True), name
        out[name + '/raw'] = raw
        out[name + '/want'] = np.ascontiguousarray(got)
        out[name + '/params'] = np.array([kw.get('nth', 1), kw.get('start', 0), kw.get('min_range', 0.0),
                                          kw.get('max_range', np.inf), kw.get('input_dim', -1)], dtype=np.float64)
        print('{:16s} raw {} -> {}'.format(name, raw.shape, got.shape))
    path = os.path.join(HERE, 'preprocess.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
