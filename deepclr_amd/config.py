"""Minimal model-config reader for the hot path.

Mirrors what ``load_model_config(cfg_filename, weights_filename)`` hands to
``build_model`` in the reference (/root/reference/deepclr/config/utils.py:250-256,
/root/reference/deepclr/models/build.py:24-29): an object with ``input_dim``,
``point_dim``, ``label_type``, ``model_type``, ``weights`` and a nested
``params`` mapping whose sub-configs expose ``.name`` / ``.params``. The
reference's full config system (defaults tree, ``extends:``, freezing, git
hash) is control plane and out of scope.
"""
from collections import OrderedDict
from enum import Enum
from typing import Any, Dict, Optional, Union

import yaml


class ConfigEnum(Enum):
    """Enum creatable from a case-insensitive string (reference: config/config.py:12-22)."""
    @classmethod
    def create(cls, x: Union[str, 'ConfigEnum']) -> Any:
        if isinstance(x, str):
            return cls[x.upper()]
        if isinstance(x, ConfigEnum):
            return x
        raise KeyError(f"Invalid config enum member '{x}'")


class Config(OrderedDict):
    """Ordered mapping with attribute access; usable as ``**cfg.params``."""

    def __getattr__(self, key: str) -> Any:
        try:
            return self[key]
        except KeyError:
            raise AttributeError("Attribute '{}' does not exist".format(key))

    def __setattr__(self, key: str, value: Any) -> None:
        self[key] = value

    @staticmethod
    def from_dict(data: Dict) -> 'Config':
        cfg = Config()
        for k, v in data.items():
            cfg[k] = Config.from_dict(v) if isinstance(v, dict) else v
        return cfg

    def dict(self) -> Dict:
        return {k: (v.dict() if isinstance(v, Config) else v) for k, v in self.items()}

    def dump(self) -> str:
        return yaml.safe_dump(_plain(self), sort_keys=False)

    def copy(self) -> 'Config':                      # deep, like the reference's Config.copy (scripts/inference.py:64)
        return Config.from_dict(self.dict())

    def write_file(self, filename: str, **_flags: Any) -> None:
        """YAML dump (reference: Config.write_file; its invalid=/internal= filters have nothing to act on here)."""
        with open(filename, 'w') as stream:
            stream.write(self.dump())


def _plain(x: Any) -> Any:
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, Enum):
        return x.name
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def model_config_from_dict(data: Dict, weights: Optional[str] = None) -> Config:
    """Finalize a parsed ``model_config.yaml`` mapping (enum coercion + dimension check)."""
    from .labels import LabelType
    from .models.build import ModelType

    cfg = Config.from_dict(data)
    cfg.weights = weights if weights is not None else cfg.get('weights')
    cfg.label_type = LabelType.create(cfg.label_type)
    cfg.model_type = ModelType.create(cfg.model_type)
    # reference: config/utils.py:224-226
    if cfg.point_dim > cfg.input_dim:
        raise RuntimeError("Model input dimension must be equal or smaller than point dimension.")
    return cfg


def load_model_config(cfg_filename: str, weights_filename: Optional[str]) -> Config:
    """Load the configuration of a model only (reference: config/utils.py:250-256)."""
    with open(cfg_filename, 'r') as stream:
        data = yaml.safe_load(stream)
    return model_config_from_dict(data, weights_filename)
