"""Model construction (mirror of /root/reference/deepclr/models/build.py:14-49)."""
import os
import os.path as osp
import shutil
from enum import auto
from typing import Type

import torch

from ..config import Config, ConfigEnum
from .base import BaseModel


class ModelType(ConfigEnum):
    DEEPCLR = auto()

    def get_class(self) -> Type[BaseModel]:
        if self == ModelType.DEEPCLR:
            from .deepclr import DeepCLR
            return DeepCLR
        raise NotImplementedError("ModelType not implemented")


def build_model(model_cfg: Config) -> BaseModel:
    """Instantiate the configured model class from ``input_dim``, ``point_dim``, ``label_type`` and ``params``."""
    cls = model_cfg.model_type.get_class()
    return cls(input_dim=model_cfg.input_dim, point_dim=model_cfg.point_dim, label_type=model_cfg.label_type,
               **model_cfg.params)


def load_model_state(filename: str):
    """A bare ``state_dict`` as written by the reference's Checkpointer
    (/root/reference/deepclr/utils/checkpoint.py:40,97-99). ``weights_only=True``: nothing in the
    file is executed."""
    return torch.load(filename, map_location='cpu', weights_only=True)


def load_trained_model(model_cfg: Config) -> BaseModel:
    model = build_model(model_cfg)
    model.load_state_dict(load_model_state(model_cfg.weights))
    return model


def store_models_code(directory: str) -> None:
    """Copy this package's model sources next to an experiment (reference: build.py:32-41)."""
    here = osp.dirname(osp.realpath(__file__))
    os.mkdir(directory)
    for name in os.listdir(here):
        src = osp.join(here, name)
        if osp.isfile(src):
            shutil.copyfile(src, osp.join(directory, name))
