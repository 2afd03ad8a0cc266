"""Scan preparation on the GPU: the reference's per-sample numpy transforms
SystematicErasing -> RangeSelection -> TruncateDimension (/root/reference/deepclr/data/transforms/transforms.py:
244-268, 90-110, 271-282) as one order-preserving HIP pass, plus the equal-size random subsample of
ModelInferenceHelper.stack (/root/reference/deepclr/models/base.py:130-135)."""
from typing import Optional

import torch

from . import lib, ops


def prepare_cloud(raw: torch.Tensor, nth: int = 1, start: int = 0, min_range: float = 0.0,
                  max_range: float = float('inf'), input_dim: Optional[int] = None) -> torch.Tensor:
    """raw (n_raw, c_raw) on the GPU -> (kept, input_dim): rows start::nth whose max(|x|,|y|) lies in
    [min_range, max_range], original order. One host sync (the row count sizes the result)."""
    if raw.dim() != 2 or raw.shape[1] < 2:
        raise RuntimeError("prepare_cloud expects a (points, channels >= 2) tensor")
    if nth < 1 or not 0 <= start < nth:
        raise RuntimeError("need nth >= 1 and 0 <= start < nth")
    raw = lib.dev_f32(raw, 'raw')
    n_raw, c_raw = raw.shape
    c_out = c_raw if input_dim is None else int(input_dim)
    if not 1 <= c_out <= c_raw:
        raise RuntimeError("Wrong point dimension in cloud.")
    cap = max(0, (n_raw - start + nth - 1) // nth)
    out = torch.empty(max(cap, 1), c_out, device=raw.device)
    count = torch.zeros(1, dtype=torch.int32, device=raw.device)
    blocks = lib.load().dclr_prepare_cloud_blocks(n_raw, nth, start)
    scratch = torch.empty(max(blocks, 1), dtype=torch.int32, device=raw.device)
    ops._call('dclr_prepare_cloud', 'prepare_cloud', n_raw, c_raw, lib.ptr(raw), nth, start, float(min_range),
              float(max_range), c_out, lib.ptr(out), lib.ptr(count), lib.ptr(scratch), lib.stream_ptr())
    return out[:int(count.item())]


def subsample(cloud: torch.Tensor, n: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """At most n rows of `cloud`, a uniform random subset in random order (base.py:131: cloud[randperm(N)[:n]])."""
    if cloud.shape[0] <= n:
        return cloud
    perm = torch.randperm(cloud.shape[0], device=cloud.device, generator=generator)[:n]
    return cloud.index_select(0, perm)
