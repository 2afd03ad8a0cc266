"""CPU restatement (numpy) of the reference's per-sample scan preparation -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/deepclr/data/transforms/transforms.py: SystematicErasing._systematic_erasing (257-268),
RangeSelection._range_selection (102-110), TruncateDimension (276-282), composed in the order the shipped data
configs list them (erase -> range -> truncate). PINNED: tests/golden/preprocess.npz holds outputs of the reference's
own classes on seeded scans (tests/golden/make_preprocess_golden.py, 11 cases with NaN / inf / boundary / empty
inputs) and tests/test_oracle.py replays them bit for bit. The reference's RandomErasing (113-134) draws from numpy's
global generator and has no device counterpart with equal draws.
"""
import numpy as np


def prepare_cloud(cloud: np.ndarray, nth: int = 1, start: int = 0, min_range: float = 0.0,
                  max_range: float = float('inf'), input_dim=None) -> np.ndarray:
    if nth != 1:
        cloud = cloud[start::nth, :]
    if not (min_range == 0.0 and np.isinf(max_range)):
        cloud_max = np.max(np.abs(cloud[:, :2]), axis=1)
        cloud = cloud[(cloud_max >= np.float32(min_range)) & (cloud_max <= np.float32(max_range)), :]
    if input_dim is not None:
        cloud = cloud[:, :input_dim]
    return cloud
