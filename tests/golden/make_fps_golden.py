#!/usr/bin/env python3
"""Generate tests/golden/fps_reference.npz by running the REFERENCE's own farthest point sampling.

The only sampling arithmetic the reference tree holds is the numpy transform
/root/reference/deepclr/data/transforms/transforms.py:31-59 (`FarthestPointSampling._fps`: float64
`pdist`, start at row 0, `argmax` of the running minimum). This script loads that file by path, as
make_preprocess_golden.py does, and runs the class itself -- nothing of it is restated here. `_fps`
returns the selected ROWS, not their indices; every input cloud therefore carries its row number in a
fourth column (the class looks at the first `dim` = 3 columns only), and the indices are read back from
the rows it returns.

The CUDA sampler the model uses (pointnet2, absent) works on float32 SQUARED distances; the two rules
pick the same rows whenever no two candidates are closer than float32 rounding. Each stored case is
checked for that on the reference's side alone: along the reference's own selection the best and the
second-best running minimum (float64, from the same `pdist`) differ by more than `MARGIN` relative at
every step. Cases that fail the margin are dropped and printed, never adjusted.

Runs only in the build container (needs /root/reference and scipy).
Usage:  python tests/golden/make_fps_golden.py [--reference /root/reference]
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from deepclr_amd import synthetic                               # noqa: E402
from make_preprocess_golden import load_reference_transforms    # noqa: E402

MARGIN = 1e-6          # relative gap between best and runner-up distance; float32 squared distances resolve ~1e-7 of it

# name -> (cloud kind, points, samples, seed)
CASES = {
    'normal_n700_m128':    ('normal', 700, 128, 3),
    'normal_n2048_m512':   ('normal', 2048, 512, 4),
    'kitti_n2048_m128':    ('kitti', 2048, 128, 5),
    'kitti_n4096_m512':    ('kitti', 4096, 512, 6),
    'modelnet_n2048_m512': ('modelnet', 2048, 512, 7),
    'modelnet_n4096_m128': ('modelnet', 4096, 128, 8),
    'sheet_n4096_m512':    ('sheet', 4096, 512, 9),
    'few_points_n96_m200': ('normal', 96, 200, 10),     # n >= points: the transform returns the cloud as it is
    # round 5: the sizes the headline kernels run at. fps_pruned_kernel<1024,16,4,3> takes N = 16384 (the c2 sampler),
    # fps_paged_kernel every 16384 < N <= 65536 (the reference's float64 distance matrix of N = 20000 is 3.2 GB)
    'kitti_n16384_m1024':  ('kitti', 16384, 1024, 11),
    'kitti_n20000_m256':   ('kitti', 20000, 256, 12),
}


def cloud(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == 'normal':
        return rng.normal(size=(n, 3)).astype(np.float32)
    if kind == 'sheet':
        return (rng.uniform(-1, 1, size=(n, 3)) * np.array([80.0, 60.0, 0.05])).astype(np.float32)
    return np.ascontiguousarray(synthetic.make_batch(kind, 1, n, first_pair=seed)[0, :, :3])


def margin_along(points64, picks):
    """Smallest relative gap between the largest and second largest running minimum over the reference's picks."""
    d = np.full(points64.shape[0], np.inf)
    worst = np.inf
    for j in range(len(picks) - 1):
        d = np.minimum(d, np.sqrt(((points64 - points64[picks[j]]) ** 2).sum(axis=1)))
        top = np.partition(d, -2)[-2:]
        worst = min(worst, (top[1] - top[0]) / top[1])
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reference', default='/root/reference')
    args = ap.parse_args()
    ref = load_reference_transforms(args.reference)
    out = {}
    for name, (kind, n, m, seed) in CASES.items():
        pts = cloud(kind, n, seed)
        tagged = np.concatenate([pts.astype(np.float64), np.arange(n, dtype=np.float64)[:, None]], axis=1)
        sample = ref.FarthestPointSampling(m)({'clouds': [tagged.copy()]})
        rows = sample['clouds'][0]
        picks = rows[:, 3].astype(np.int64)
        assert np.array_equal(rows[:, :3], pts[picks].astype(np.float64))
        if m < n:
            assert picks[0] == 0 and len(set(picks.tolist())) == m
            gap = margin_along(pts.astype(np.float64), picks)
            if not gap > MARGIN:
                print('{:22s} DROPPED: best / runner-up gap {:.2e} <= {:.0e}'.format(name, gap, MARGIN))
                continue
        else:
            gap = np.inf
            assert np.array_equal(picks, np.arange(n))
        out[name + '/points'] = pts
        out[name + '/picks'] = picks.astype(np.int32)
        out[name + '/m'] = np.int64(m)
        print('{:22s} {} points -> {} rows, smallest gap {:.2e}'.format(name, n, len(picks), gap))
    path = os.path.join(HERE, 'fps_reference.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
