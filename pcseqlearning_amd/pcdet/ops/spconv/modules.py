"""SparseModule / SparseSequential (pcdet/ops/spconv/modules.py:46-187): sparse modules take and
return a SparseConvTensor, plain nn.Modules are applied to `.features`."""
from collections import OrderedDict

import torch
from torch import nn

from .structure import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base class: modules that consume a SparseConvTensor."""


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError(f"index {idx} is out of range")
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        name = str(len(self._modules)) if name is None else name
        if name in self._modules:
            raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                # eval mode: conv -> BatchNorm1d [-> ReLU] runs as ONE kernel, the normalisation folded
                # into the conv epilogue (fused_spconv_ops.h:29-132 is the reference's own fused form)
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if (hasattr(module, "weight") and hasattr(module, "inverse")
                        and isinstance(nxt, nn.BatchNorm1d) and not nxt.training and nxt.track_running_stats
                        and nxt.running_mean is not None
                        and input.features.is_cuda and not torch.is_grad_enabled()):
                    scale, shift = fold_batchnorm(nxt)
                    relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                    input = module(input, scale=scale, shift=shift, relu=relu)
                    i += 3 if relu else 2
                    continue
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    input.features = module(input.features)
            else:
                input = module(input)
            i += 1
        return input


def fold_batchnorm(bn):
    """Eval-mode BatchNorm1d as y = x * scale + shift. Recomputed on every call (rsqrt + one small
    launch on the device, four [C]-sized tensor ops on the host):
    a cache keyed on tensor versions would miss in-place edits through `.data` and re-assigned
    buffers, and diverge from the module-by-module path without an error."""
    with torch.no_grad():
        if bn.running_var.is_cuda:  # three launches instead of five to seven (same arithmetic)
            from .... import ops as _ops
            return _ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        inv = torch.rsqrt(bn.running_var.float() + bn.eps)
        scale = inv * bn.weight.float() if bn.weight is not None else inv
        shift = -bn.running_mean.float() * scale
        if bn.bias is not None:
            shift = shift + bn.bias.float()
        return scale.contiguous(), shift.contiguous()
