"""prepare_tensor as scripts/timing.py calls it (reference: deepclr/utils/tensor.py:7-10, via ignite.convert_tensor)."""
from typing import Optional, Union

import torch


def prepare_tensor(x: torch.Tensor, device: Optional[Union[str, torch.device]] = None, non_blocking: bool = False)\
        -> torch.Tensor:
    return x.to(device=device, non_blocking=non_blocking) if device is not None else x
