"""Model interface and the inference helper the drop-in scripts drive.

Mirrors ``BaseModel`` (/root/reference/deepclr/models/base.py:9-53) and
``ModelInferenceHelper`` (56-136): same method names, argument meaning, return
values and error behaviour (RuntimeError on too few point columns, a warning on
truncation, RuntimeError on a template in sequential mode or a missing template
otherwise). ``predict_batch`` is an addition: the reference only ever feeds one
pair per call (base.py:118-120), which leaves an MI355X idle.
"""
import abc
import warnings
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn


class BaseModel(nn.Module, metaclass=abc.ABCMeta):
    """Interface every registration model exposes to the scripts and engines."""

    @abc.abstractmethod
    def get_input_dim(self) -> int:
        """Number of columns expected per input point."""

    @abc.abstractmethod
    def has_loss(self) -> bool:
        """Whether forward() can return a loss."""

    @abc.abstractmethod
    def get_loss_weights(self) -> Dict:
        """Current loss weights by name."""

    @abc.abstractmethod
    def forward(self, x: torch.Tensor, is_feat: bool = False, m: Optional[torch.Tensor] = None,
                y: Optional[torch.Tensor] = None, debug: bool = False)\
            -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[Dict]]:
        """x: batch [T0..TB-1, S0..SB-1] of clouds (or of cloud features when is_feat);
        m: augmentation matrices; y: ground truth. Returns (prediction, loss, debug)."""

    @abc.abstractmethod
    def cloud_features(self, x: torch.Tensor, m: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Per-cloud feature extraction only."""


class ModelInferenceHelper:
    """Pairwise or sequential (cached previous cloud) prediction with a model in eval mode."""

    def __init__(self, model: BaseModel, is_sequential: bool = False):
        self._model = model
        self._model.eval()
        self._input_dim = model.get_input_dim()
        self._is_sequential = is_sequential
        self._state: Optional[torch.Tensor] = None

    def has_state(self) -> bool:
        return self._state is not None

    def reset_state(self) -> None:
        self._state = None

    def finish(self) -> None:
        """Wait for the model's work on the current stream and raise if any of it left the range of the split-f16 matrix
        path (deepclr_amd.models.DeepCLR.check_range). A clamped forward is never silent even without this call: the fused
        kernels write its poses as NaN (and every later one's, until the host has acknowledged the flag), and the next
        entry into the model raises. `predict_batch` and `predict_sequence` end with finish() (one wait per batch);
        `predict` -- one pair per call, timed per call by the reference's scripts (scripts/timing.py:36-44) -- does not:
        a host-side wait behind every pair costs ~20 us of its 0.82 ms, so a caller of `predict` gets NaN poses from a
        clamped pair, the exception at the next call, or the exception here when it calls finish() after its last pair.
        Models without such a check (the reference interface has none): a no-op."""
        check = getattr(self._model, 'check_range', None)
        if check is not None:
            check(synchronize='stream')

    def _fit_columns(self, cloud: torch.Tensor, which: str) -> torch.Tensor:
        cols = cloud.shape[1]
        if cols < self._input_dim:
            raise RuntimeError("Wrong point dimension in {}.".format(which))
        if cols > self._input_dim:
            warnings.warn(f"Truncate {which} point cloud from dimension {cols} "
                          f"to required dimension {self._input_dim}.")
            cloud = cloud[:, :self._input_dim]
        return cloud

    def predict(self, source: torch.Tensor, template: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """source / template: single clouds (N, C). Returns the (label_dim,) prediction, or None for the
        first cloud of a sequence."""
        source = self._fit_columns(source, 'source')
        if template is not None:
            template = self._fit_columns(template, 'template')

        with torch.no_grad():
            if self._is_sequential:
                if template is not None:
                    raise RuntimeError("Only the source cloud is required for sequential prediction.")
                feat = self._model.cloud_features(source.unsqueeze(0))[0]
                previous, self._state = self._state, feat
                if previous is None:
                    return None
                y, _, _ = self._model.forward(self.stack(previous, feat), is_feat=True)
                return y[0, :]

            if template is None:
                raise RuntimeError("Source and template clouds are required for non-sequential prediction.")
            y, _, _ = self._model.forward(self.stack(template, source), is_feat=False)
            return y[0, :]

    def predict_batch(self, sources: torch.Tensor, templates: torch.Tensor) -> torch.Tensor:
        """sources / templates: (B, N, C) equally sized clouds -> (B, label_dim)."""
        if sources.shape != templates.shape:
            raise RuntimeError("Batched prediction needs equally shaped source and template batches.")
        if sources.shape[2] < self._input_dim:
            raise RuntimeError("Wrong point dimension in source.")
        with torch.no_grad():
            x = torch.cat((templates[:, :, :self._input_dim], sources[:, :, :self._input_dim]), dim=0)
            y, _, _ = self._model.forward(x.contiguous(), is_feat=False)
        self.finish()
        return y

    def predict_sequence(self, frames: torch.Tensor) -> torch.Tensor:
        """Sequential mode over a chunk of consecutive frames (T, N, C): the same poses T calls of
        predict(frame) return (the first being None on a fresh state), with every frame's features computed
        once and in one batch. Returns (T, label_dim), or (T-1, label_dim) when no frame was cached."""
        if not self._is_sequential:
            raise RuntimeError("predict_sequence needs a sequential helper.")
        if frames.shape[2] < self._input_dim:
            raise RuntimeError("Wrong point dimension in source.")
        with torch.no_grad():
            feats = self._model.cloud_features(frames[:, :, :self._input_dim].contiguous())
            if self._state is not None:
                feats = torch.cat((self._state.unsqueeze(0), feats))
            self._state = feats[-1]
            if feats.shape[0] < 2:
                return feats.new_empty(0, getattr(self._model, 'label_dim', 0))
            y, _, _ = self._model.forward(torch.cat((feats[:-1], feats[1:])), is_feat=True)
        self.finish()
        return y

    @staticmethod
    def stack(template: torch.Tensor, source: torch.Tensor) -> torch.Tensor:
        """Two clouds -> batch (2, N, C); the larger one is randomly subsampled to the smaller size."""
        n_t, n_s = template.shape[0], source.shape[0]
        if n_s > n_t:
            source = source[torch.randperm(n_s)[:n_t], :]
        elif n_t > n_s:
            template = template[torch.randperm(n_t)[:n_s], :]
        return torch.stack((template, source), 0)
