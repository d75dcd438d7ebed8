"""Autograd functions of the sparse conv (pcdet/ops/spconv/functional.py:20-72):
forward on the output-major table kernel, backward on `pcs_indice_conv_backward_f32` (input gradient
= the same conv over the transposed rulebook, filter gradient = deterministic two-stage sum)."""
from torch.autograd import Function

from .... import ops as _ops
from . import ops


class _IndiceConv(Function):
    inverse, subm = False, False

    @classmethod
    def _forward(cls, ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out, table=None):
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters)
        if table is not None:
            return _ops.indice_conv_table(features.detach(), filters.detach(), table, num_activate_out,
                                          cls.subm and not cls.inverse)
        return ops.indice_conv(features.detach(), filters.detach(), indice_pairs, indice_pair_num,
                               num_activate_out, cls.inverse, cls.subm)

    @classmethod
    def _backward(cls, ctx, grad_output):
        indice_pairs, indice_pair_num, features, filters = ctx.saved_tensors
        input_bp, filters_bp = ops.indice_conv_backward(features, filters, grad_output, indice_pairs,
                                                        indice_pair_num, cls.inverse, cls.subm)
        return input_bp, filters_bp, None, None, None, None


class SparseConvFunction(_IndiceConv):
    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out, table=None):
        return SparseConvFunction._forward(ctx, features, filters, indice_pairs, indice_pair_num,
                                           num_activate_out, table)

    @staticmethod
    def backward(ctx, grad_output):
        return SparseConvFunction._backward(ctx, grad_output)


class SparseInverseConvFunction(_IndiceConv):
    inverse = True

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out, table=None):
        return SparseInverseConvFunction._forward(ctx, features, filters, indice_pairs, indice_pair_num,
                                                  num_activate_out, table)

    @staticmethod
    def backward(ctx, grad_output):
        return SparseInverseConvFunction._backward(ctx, grad_output)


class SubMConvFunction(_IndiceConv):
    subm = True

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out, table=None):
        return SubMConvFunction._forward(ctx, features, filters, indice_pairs, indice_pair_num,
                                         num_activate_out, table)

    @staticmethod
    def backward(ctx, grad_output):
        return SubMConvFunction._backward(ctx, grad_output)


class SparseMaxPoolFunction(Function):
    """functional.py:75-92 of the reference."""

    @staticmethod
    def forward(ctx, features, indice_pairs, indice_pair_num, num_activate_out):
        out = ops.indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out)
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, out)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, out = ctx.saved_tensors
        input_bp = ops.indice_maxpool_backward(features, out, grad_output, indice_pairs, indice_pair_num)
        return input_bp, None, None, None


indice_conv = SparseConvFunction.apply
indice_inverse_conv = SparseInverseConvFunction.apply
indice_subm_conv = SubMConvFunction.apply
indice_maxpool = SparseMaxPoolFunction.apply
