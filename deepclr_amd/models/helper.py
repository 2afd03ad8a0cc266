"""Affine + ReLU stacks with the reference's parameter layout.

Mirrors ``Conv1d`` / ``Linear`` / ``Conv1dMultiLayer`` / ``LinearMultiLayer``
(/root/reference/deepclr/models/helper.py:11-123): xavier-uniform weights, zero
bias, ReLU after EVERY layer including the last, and the nesting
``_sequential.{i}._sequential.0.{weight,bias}`` that the shipped checkpoints use.
Here the modules are parameter holders plus a forward that calls the MFMA GEMM
(``dclr_linear``) or the small-row FC kernel (``dclr_fc``); they never run on CPU.
"""
from typing import List, Optional, Tuple

import torch
from torch import nn

from .. import ops


def _no_batch_norm(batch_norm: bool) -> None:
    if batch_norm:
        raise NotImplementedError("batch_norm=True is outside the MI355X hot path (every shipped "
                                  "model_config.yaml sets batch_norm: false)")


class Conv1d(nn.Module):
    """1x1 convolution + ReLU over (B, C, N)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 1, bias: bool = True,
                 batch_norm: bool = False):
        super().__init__()
        _no_batch_norm(batch_norm)
        if kernel_size not in (1, (1,)):
            raise NotImplementedError("only kernel_size 1 occurs on the hot path")
        conv = nn.Conv1d(in_channels, out_channels, 1, bias=bias)
        nn.init.xavier_uniform_(conv.weight)
        if conv.bias is not None:
            conv.bias.data.fill_(0.0)
        self._sequential = nn.Sequential(conv)
        self._output_dim = out_channels

    def output_dim(self) -> int:
        return self._output_dim

    @property
    def affine(self) -> nn.Conv1d:
        return self._sequential[0]


class Linear(nn.Module):
    """Fully connected layer + ReLU over (B, C)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, batch_norm: bool = False):
        super().__init__()
        _no_batch_norm(batch_norm)
        lin = nn.Linear(in_features, out_features, bias=bias)
        nn.init.xavier_uniform_(lin.weight)
        if bias:
            lin.bias.data.fill_(0.0)
        self._sequential = nn.Sequential(lin)
        self._output_dim = out_features

    def output_dim(self) -> int:
        return self._output_dim

    @property
    def affine(self) -> nn.Linear:
        return self._sequential[0]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.fc(x.contiguous(), self.affine.weight, self.affine.bias, act=1)


class _Stack(nn.Module):
    """Common layout of the two multi-layer containers. Dropout modules are kept as index
    placeholders (state_dict keys depend on their positions); at inference they are the identity."""

    def __init__(self, layer_cls, layer_sizes: List[int], batch_norm: bool, dropout_keep: float,
                 dropout_last: bool, **layer_kwargs):
        super().__init__()
        mods: List[nn.Module] = []
        pairs = list(zip(layer_sizes[:-1], layer_sizes[1:]))
        for i, (c_in, c_out) in enumerate(pairs):
            mods.append(layer_cls(c_in, c_out, bias=True, batch_norm=batch_norm, **layer_kwargs))
            last = i == len(pairs) - 1
            if dropout_keep < 1.0 and (not last or dropout_last):
                mods.append(nn.Dropout(1.0 - dropout_keep))
        self._sequential = nn.Sequential(*mods)
        self._output_dim = layer_sizes[-1]

    def output_dim(self) -> int:
        return self._output_dim

    def layers(self) -> List[nn.Module]:
        return [m for m in self._sequential if not isinstance(m, nn.Dropout)]

    def affine_params(self) -> List[Tuple[torch.Tensor, Optional[torch.Tensor]]]:
        return [(m.affine.weight, m.affine.bias) for m in self.layers()]


class Conv1dMultiLayer(_Stack):
    def __init__(self, layer_sizes: List[int], batch_norm: bool = False, dropout_keep: float = 1.0,
                 dropout_last: bool = False):
        super().__init__(Conv1d, layer_sizes, batch_norm, dropout_keep, dropout_last, kernel_size=1)


class LinearMultiLayer(_Stack):
    def __init__(self, layer_sizes: List[int], batch_norm: bool = False, dropout_keep: float = 1.0,
                 dropout_last: bool = False):
        super().__init__(Linear, layer_sizes, batch_norm, dropout_keep, dropout_last)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and any(isinstance(m, nn.Dropout) for m in self._sequential):
            raise NotImplementedError("dropout in training mode is outside the forward-only hot path")
        for layer in self.layers():
            x = layer(x)
        return x


def _structure_stamp(mods) -> int:
    """Cheap fingerprint of a module tree's SHAPE: registering a parameter or a submodule anywhere in it changes the
    count it sums (len() of the per-module dicts: no tensor is touched)."""
    return sum(len(m._parameters) + len(m._modules) for m in mods)


def flat_parameters(module: nn.Module):
    """The module's parameters in `parameters()` order without walking the module tree on every call: the (owner dict,
    name) slots are collected once per module and the CURRENT tensor of each slot is fetched from them, so that replaced
    parameters (`m.weight = nn.Parameter(...)`), in-place loads and device moves are all seen. The hot paths ask for the
    weights' versions several times per launch; `parameters()` cost ~0.1 ms each time (23 tensors behind a recursive
    generator) -- 0.4 ms of host time per dense call + sampling chain, exposed whenever the GPU waits for the host
    (the first launches of a timed window).
    The slot list belongs to ONE module object and one tree shape: it records the module it was built for and the
    tree's structure stamp, and is rebuilt when either differs -- a parameter or submodule registered later is picked
    up, and a replica made by copying `__dict__` (nn.Module._replicate_for_data_parallel, copy.copy) does not key its
    caches on the original's parameters."""
    cached = module.__dict__.get('_dclr_param_slots')
    if cached is None or cached[0] is not module or cached[2] != _structure_stamp(cached[1]):
        mods = list(module.modules())
        # every slot, also the ones that hold None now (`register_parameter('bias', None)` filled in later)
        slots = [(m._parameters, name) for m in mods for name in m._parameters]
        cached = (module, mods, _structure_stamp(mods), slots)
        module.__dict__['_dclr_param_slots'] = cached
    out, seen = [], set()
    for d, n in cached[3]:
        prm = d.get(n)
        if prm is not None and id(prm) not in seen:                # tied weights are listed once, as parameters() does
            seen.add(id(prm))
            out.append(prm)
    return out


class PackedCache:
    """Re-derive kernel-side weight buffers only when a parameter changed (load_state_dict, .to()).

    The pack kernels run on the stream of whoever asks first; any OTHER stream that later asks for the same
    buffers first waits for the event recorded behind them (the pipelined runner reads them from several side
    streams at once, and a fresh model's first use may well be on one of those)."""

    def __init__(self):
        self._key = None
        self._value = None
        self._event = None
        self._stream = None
        self._users = set()                 # streams (other than the builder's) that have been handed the value

    @staticmethod
    def _tensors(obj):
        if torch.is_tensor(obj):
            yield obj
        elif isinstance(obj, dict):
            for o in obj.values():
                yield from PackedCache._tensors(o)
        elif isinstance(obj, (tuple, list)):
            for o in obj:
                yield from PackedCache._tensors(o)

    def get(self, params, build):
        key = tuple((p.device, p.data_ptr(), p._version) for p in params)
        cuda = bool(params) and params[0].is_cuda
        if key != self._key:
            # the buffers being replaced may still be read by launches other streams have enqueued: their memory goes
            # back to the allocator only behind those streams' work (record_stream was called when they were handed out)
            self._value = build()
            self._key = key
            self._event, self._stream = None, None
            self._users = set()
            if cuda:
                self._stream = torch.cuda.current_stream(params[0].device).cuda_stream
                self._event = torch.cuda.Event()
                self._event.record()
        elif cuda:
            cur = torch.cuda.current_stream()
            if cur.cuda_stream != self._stream:
                if cur.cuda_stream not in self._users:          # first hand-out to this stream
                    self._users.add(cur.cuda_stream)
                    for t in self._tensors(self._value):
                        if t.is_cuda:
                            t.record_stream(cur)
                if self._event is not None:
                    if self._event.query():
                        self._event = None                      # packing finished: visible to every stream
                    else:
                        cur.wait_event(self._event)
        return self._value
