"""Calibration workload for profiles/collect.py: streaming copies of a known size (256 MiB read + 256 MiB written
per launch, more than the 256 MiB Infinity Cache holds together with the destination), so that FETCH_SIZE /
WRITE_SIZE can be compared with a known byte count in the 16-byte-per-lane access pattern
(MI355X_MICROARCH.md, HBM section)."""
import torch

n = 256 * 1024 * 1024 // 4
src = torch.empty(n, dtype=torch.float32, device='cuda:0').normal_()
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(6):
    torch.add(src, 1.0, out=dst)          # at::native::vectorized_elementwise_kernel<4, ...>: float4 per lane
torch.cuda.synchronize()
