"""Affine + ReLU stacks with the reference's parameter layout.

Mirrors ``Conv1d`` / ``Linear`` / ``Conv1dMultiLayer`` / ``LinearMultiLayer``
(/root/reference/deepclr/models/helper.py:11-123): xavier-uniform weights, zero
bias, ReLU after EVERY layer including the last, and the nesting
``_sequential.{i}._sequential.0.{weight,bias}`` that the shipped checkpoints use.
Here the modules are parameter holders plus a forward that calls the MFMA GEMM
(``dclr_linear``) or the small-row FC kernel (``dclr_fc``); they never run on CPU.

``batch_norm=True`` (helper.py:27-30,57-60: a BatchNorm1d behind the affine layer, inside the same
``_sequential``) and ``dropout_keep < 1`` (helper.py:77-85,107-113) build the same module trees as the
reference, hence the same state_dict keys. In eval mode -- what ``ModelInferenceHelper`` sets and what
every inference script runs -- batch norm is the affine map of its running statistics and is FOLDED into
the layer's weight and bias before they are packed for the kernels (``folded()``); dropout is the
identity. In training mode both need batch statistics / random masks: the differentiable training step
runs the torch modules themselves (``forward_torch``), the fused inference kernels refuse.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops


def fold_batch_norm(weight: torch.Tensor, bias: Optional[torch.Tensor], bn: Optional[nn.modules.batchnorm._BatchNorm])\
        -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """(W, b) of `bn(W x + b)` in eval mode: W' = s W, b' = s (b - mean) + beta with s = gamma / sqrt(var + eps).
    Without a norm layer the parameters themselves come back (no copy). Float64 inside: the fold is done once per
    checkpoint and must not cost accuracy against torch's own eval-mode kernel."""
    if bn is None:
        return weight, bias
    if bn.training:
        raise RuntimeError("batch norm in training mode normalises with batch statistics: the inference kernels fold the "
                           "RUNNING statistics into the layer (call model.eval(), as ModelInferenceHelper does)")
    if bn.running_mean is None or bn.running_var is None:
        raise RuntimeError("batch norm without running statistics (track_running_stats=False) cannot be folded")
    s = (bn.running_var.detach().double() + bn.eps).rsqrt()
    if bn.weight is not None:
        s = s * bn.weight.detach().double()
    w = weight.detach().double() * s.view(-1, *([1] * (weight.dim() - 1)))
    b = (0.0 if bias is None else bias.detach().double()) - bn.running_mean.detach().double()
    b = b * s + (0.0 if bn.bias is None else bn.bias.detach().double())
    return w.float(), b.float()


class _Affine(nn.Module):
    """What Conv1d and Linear share: `_sequential` = [affine layer, optional BatchNorm1d] and ReLU behind it."""
    _sequential: nn.Sequential
    _output_dim: int

    def output_dim(self) -> int:
        return self._output_dim

    @property
    def affine(self):
        return self._sequential[0]

    @property
    def norm(self) -> Optional[nn.BatchNorm1d]:
        return self._sequential[1] if len(self._sequential) > 1 else None

    def folded(self) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """Weight and bias the kernels pack: the layer's own, or with the eval-mode batch norm folded in."""
        return fold_batch_norm(self.affine.weight, self.affine.bias, self.norm)

    def forward_torch(self, x: torch.Tensor) -> torch.Tensor:
        """The reference's forward, literally (helper.py:37-38,64-65): differentiable, batch statistics in training mode."""
        return F.relu(self._sequential(x))


class Conv1d(_Affine):
    """1x1 convolution (+ batch norm) + ReLU over (B, C, N)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 1, bias: bool = True,
                 batch_norm: bool = False):
        super().__init__()
        if kernel_size not in (1, (1,)):
            raise NotImplementedError("only kernel_size 1 occurs on the hot path")
        conv = nn.Conv1d(in_channels, out_channels, 1, bias=bias)
        nn.init.xavier_uniform_(conv.weight)
        if conv.bias is not None:
            conv.bias.data.fill_(0.0)
        self._sequential = nn.Sequential(conv, nn.BatchNorm1d(out_channels)) if batch_norm else nn.Sequential(conv)
        self._output_dim = out_channels


class Linear(_Affine):
    """Fully connected layer (+ batch norm) + ReLU over (B, C)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, batch_norm: bool = False):
        super().__init__()
        lin = nn.Linear(in_features, out_features, bias=bias)
        nn.init.xavier_uniform_(lin.weight)
        if bias:
            lin.bias.data.fill_(0.0)
        self._sequential = nn.Sequential(lin, nn.BatchNorm1d(out_features)) if batch_norm else nn.Sequential(lin)
        self._output_dim = out_features

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        w, b = self.folded()
        return ops.fc(x.contiguous(), w, b, act=1)


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
        """(weight, bias) per layer as the kernels need them: eval-mode batch norm folded in (the parameters themselves
        where there is none)."""
        return [m.folded() for m in self.layers()]

    def has_dropout(self) -> bool:
        return any(isinstance(m, nn.Dropout) for m in self._sequential)

    def forward_torch(self, x: torch.Tensor) -> torch.Tensor:
        """The reference's forward (helper.py:92-93,122-123): every module of `_sequential`, Dropout included, in torch."""
        for m in self._sequential:
            x = m(x) if isinstance(m, nn.Dropout) else m.forward_torch(x)
        return x


class Conv1dMultiLayer(_Stack):
    def __init__(self, layer_sizes: List[int], batch_norm: bool = False, dropout_keep: float = 1.0,
                 dropout_last: bool = False):
        super().__init__(Conv1d, layer_sizes, batch_norm, dropout_keep, dropout_last, kernel_size=1)


class LinearMultiLayer(_Stack):
    def __init__(self, layer_sizes: List[int], batch_norm: bool = False, dropout_keep: float = 1.0,
                 dropout_last: bool = False):
        super().__init__(Linear, layer_sizes, batch_norm, dropout_keep, dropout_last)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and self.has_dropout():
            raise RuntimeError("dropout in training mode draws random masks: the inference kernels treat it as the identity "
                               "(call model.eval(); a training step runs forward_torch)")
        for layer in self.layers():
            x = layer(x)
        return x


# Any registration of a submodule, parameter or buffer ANYWHERE in the process bumps this counter (torch's global
# registration hooks: `m.child = ...`, `seq[i] = ...`, `m.weight = nn.Parameter(...)`, register_buffer, add_module all pass
# through them). flat_parameters() rebuilds its cached slots when the counter has moved since they were collected --
# a replaced submodule of the same arity is seen for the price of one integer comparison per call; walking
# the tree for an identity stamp instead cost 14-19 us per call, ten calls per launch group (0.15 ms of host time at every
# group boundary of the pipelined runner). Deleting a parameter leaves its slot behind, which then reads None and is
# skipped; deleting a SUBMODULE is caught by a child count (flat_parameters); writing a module's private `_parameters`
# dict directly is not seen (nothing in torch does).
_TREE_EPOCH = [0]


def _bump_tree_epoch(*_args):
    _TREE_EPOCH[0] += 1
    return None                                  # keep the value being registered


for _reg in ('register_module_module_registration_hook', 'register_module_parameter_registration_hook',
             'register_module_buffer_registration_hook'):
    getattr(torch.nn.modules.module, _reg)(_bump_tree_epoch)


def flat_parameters(module: nn.Module):
    """The module's parameters in `parameters()` order, followed by its buffers (batch-norm running statistics: part of
    what the packed weights are derived from), without walking the module tree through torch's recursive generators on
    every call: the (owner dict, name) slots are collected once per tree shape and the CURRENT tensor of each slot is
    fetched from them, so that replaced parameters (`m.weight = nn.Parameter(...)`), in-place loads and device moves are
    all seen. The hot paths ask for the weights' versions several times per launch; `parameters()` cost ~0.1 ms each time
    (23 tensors behind a recursive generator) -- 0.4 ms of host time per dense call + sampling chain, exposed whenever the
    GPU waits for the host (the first launches of a timed window).
    The slot list belongs to ONE module object and one registration epoch (above): it is rebuilt when a submodule,
    parameter or buffer has been registered or replaced anywhere since, and a replica made by copying `__dict__`
    (nn.Module._replicate_for_data_parallel, copy.copy) does not key its caches on the original's parameters."""
    cached = module.__dict__.get('_dclr_param_slots')
    # REMOVING a submodule (del seq[i], Sequential.pop, delattr) fires no registration hook: the child count over the
    # cached module list is the structural term that catches it (one len() per module, ~2 us)
    if cached is None or cached[0] is not module or cached[1] != _TREE_EPOCH[0] \
            or cached[4] != sum(len(m._modules) for m in cached[3]):
        mods = list(module.modules())
        # every slot, also the ones that hold None now (`register_parameter('bias', None)` filled in later)
        slots = [(m._parameters, name) for m in mods for name in m._parameters]
        slots += [(m._buffers, name) for m in mods for name in m._buffers]
        cached = (module, _TREE_EPOCH[0], slots, mods, sum(len(m._modules) for m in mods))
        module.__dict__['_dclr_param_slots'] = cached
    out, seen = [], set()
    for d, n in cached[2]:
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
