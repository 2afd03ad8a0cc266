"""create_logger as the entry scripts call it (reference: deepclr/utils/logging.py:10-43)."""
import logging
import os
import sys
import time
from typing import Optional


def create_logger(name: Optional[str] = None, save_dir: Optional[str] = None, distributed_rank: int = 0)\
        -> logging.Logger:
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    if distributed_rank > 0 or logger.hasHandlers():          # only rank 0 prints; never attach handlers twice
        return logger
    fmt = logging.Formatter("%(asctime)s %(levelname)s: %(message)s" if name is None
                            else "%(asctime)s %(name)s %(levelname)s: %(message)s")
    targets = [logging.StreamHandler(stream=sys.stdout)]
    if save_dir:
        targets.append(logging.FileHandler(os.path.join(save_dir, time.strftime('log_%Y%m%d_%H%M%S.txt')), mode='w'))
    for handler in targets:
        handler.setLevel(logging.DEBUG)
        handler.setFormatter(fmt)
        logger.addHandler(handler)
    return logger
