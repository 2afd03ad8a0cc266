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


class Mode(ConfigEnum):
    """Configuration modes (reference: config/utils.py:18-23)."""
    NEW = 1
    CONTINUE = 2
    INFERENCE = 3
    TEST = 4


def _merge(base: Dict, over: Dict) -> Dict:
    """Nested mapping update: what the reference's Config.read_dict does with a file that `extends:` another."""
    out = dict(base)
    for k, v in over.items():
        out[k] = _merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def _read_with_extends(filename: str) -> Dict:
    import os.path as osp
    with open(filename, 'r') as stream:
        data = yaml.safe_load(stream) or {}
    parent = data.pop('extends', None)
    if parent is not None:
        parent = osp.realpath(osp.join(osp.dirname(filename), parent))
        if osp.realpath(filename) != parent:                 # reference: config/utils.py:138-141
            data = _merge(_read_with_extends(parent), data)
    return data


def _expand(path: Any) -> Any:
    import os
    if isinstance(path, (list, tuple)):
        return [_expand(p) for p in path]
    if isinstance(path, str):
        return os.path.expandvars(os.path.expanduser(path))
    return path


def load_config(cfg_filename: str, mode: Any, ckpt_filename: Optional[str] = None) -> Config:
    """Read a run configuration (`*.yaml`, `extends:` chains followed) the way scripts/timing.py needs it
    (reference: config/utils.py:232-248; caller scripts/timing.py:57): `cfg.device`, `cfg.model` (finalized as by
    load_model_config: enums created, dimensions checked) and the `data` / `data_loader` / `transforms` sections as
    plain nested mappings for make_data_loader. What the reference's finish_config adds for TRAINING runs -- output
    directories, the `git rev-parse` of the package checkout (which fails outside a git checkout),
    optimizer / scheduler / metric validation, freezing -- is control plane of the training engine and not done here."""
    mode = Mode.create(mode.name) if isinstance(mode, Enum) else Mode.create(mode)
    data = _read_with_extends(cfg_filename)
    if 'model' not in data:
        raise RuntimeError("Configuration is not valid, missing required parameters.")      # reference wording
    cfg = Config.from_dict(data)
    cfg.mode = mode
    cfg.extends = None
    cfg.checkpoint = _expand(ckpt_filename if ckpt_filename is not None else cfg.get('checkpoint'))
    if mode == Mode.CONTINUE and cfg.checkpoint is None:
        raise RuntimeError("Please specify the checkpoint for continue")
    cfg.device = cfg.get('device', 'cuda')
    cfg.base_dir = _expand(cfg.get('base_dir'))
    cfg.model = model_config_from_dict(data['model'], _expand(data['model'].get('weights')))
    if mode == Mode.INFERENCE and cfg.model.weights is None:
        raise RuntimeError("Please specify the model weights for inference")
    section = cfg.get('data')
    if isinstance(section, Config):
        for key in ('training', 'validation'):
            if key in section:
                section[key] = _expand(section[key])
    return cfg
