from enum import Enum

from deepclr_amd.config import Config, ConfigEnum, load_model_config


class Mode(Enum):
    """Run modes of the reference's full configuration (config/utils.py); only named here."""
    NEW = 0
    CONTINUE = 1
    TEST = 2


def load_config(*_args, **_kwargs):
    raise RuntimeError("load_config reads the reference's training/data configuration tree, which is outside the "
                       "MI355X forward hot path; use load_model_config(model_config.yaml, weights) instead")


__all__ = ['Config', 'ConfigEnum', 'Mode', 'load_config', 'load_model_config']
