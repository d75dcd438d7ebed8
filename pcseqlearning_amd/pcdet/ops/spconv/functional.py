"""Autograd entry points of the sparse conv / max-pool (reference: pcdet/ops/spconv/functional.py:20-98;
SURVEY 8f row f4). Forward on the output-major table kernel, backward on `pcs_indice_conv_backward_f32`
(input gradient = the same conv over the transposed rulebook, filter gradient = deterministic
two-stage sum) and `pcs_indice_maxpool_backward_f32`. The three conv functions differ only in the
(inverse, subm) flags they hand to the kernels and are stamped out by one factory."""
from torch.autograd import Function

from .... import ops as _ops
from . import ops


def _conv_function(name, inverse, subm):
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out, table=None):
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters)
        f, w = features.detach(), filters.detach()
        if table is None:
            return ops.indice_conv(f, w, indice_pairs, indice_pair_num, num_activate_out, inverse, subm)
        return _ops.indice_conv_table(f, w, table, num_activate_out, subm and not inverse)

    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, filters = ctx.saved_tensors
        grads = ops.indice_conv_backward(features, filters, grad_output, indice_pairs, indice_pair_num, inverse, subm)
        return (*grads, None, None, None, None)

    return type(name, (Function,), {"forward": staticmethod(forward), "backward": staticmethod(backward),
                                    "__doc__": f"indice conv, inverse={inverse}, subm={subm}"})


SparseConvFunction = _conv_function("SparseConvFunction", False, False)
SparseInverseConvFunction = _conv_function("SparseInverseConvFunction", True, False)
SubMConvFunction = _conv_function("SubMConvFunction", False, True)


class SparseMaxPoolFunction(Function):
    @staticmethod
    def forward(ctx, features, indice_pairs, indice_pair_num, num_activate_out):
        pooled = ops.indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out)
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, pooled)
        return pooled

    @staticmethod
    def backward(ctx, grad_output):
        pairs, pair_num, features, pooled = ctx.saved_tensors
        return ops.indice_maxpool_backward(features, pooled, grad_output, pairs, pair_num), None, None, None


indice_conv = SparseConvFunction.apply
indice_inverse_conv = SparseInverseConvFunction.apply
indice_subm_conv = SubMConvFunction.apply
indice_maxpool = SparseMaxPoolFunction.apply
