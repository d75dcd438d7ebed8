"""SparseConvolution and its 3-D front-ends (pcdet/ops/spconv/conv.py:48-445): weight
[k,k,k,Cin,Cout], rulebook cached under `indice_key` in the tensor's `indice_dict` as
(outids, indices, indice_pairs, indice_pair_num, spatial_shape). The cached entry additionally
keeps the output-major neighbour table the HIP conv kernel consumes."""
import math

import numpy as np
import torch
from torch import nn
from torch.nn import init
from torch.nn.parameter import Parameter

from .... import ops as _ops
from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor

# Take the output rows of a cached rulebook in the order of pcs_neighbor_table_order (same results; see
# ops.neighbor_table_order). OFF: measured in round 5 on VoxelBackBone8x at B = 20 -- the matrix cores multiply a third
# less padding, but the reordered rows' gathers stop meeting in the caches and the order costs 92 us per rulebook:
# f32 forward 13.1 -> 13.9 ms, fp16 7.7 -> 9.1 ms (a sort over all rows instead of blocks of 4096: 15.2 / 10.3 ms).
ORDER_ROWS_BY_OFFSET_SET = False


def _fan_in_hwio(weight):
    receptive = weight[..., 0, 0].numel() if weight.dim() > 2 else 1
    return weight.size(-2) * receptive


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True, subm=False, output_padding=0, transposed=False,
                 inverse=False, indice_key=None, fused_bn=False):
        super().__init__()
        assert groups == 1
        as_list = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim  # noqa: E731
        kernel_size, stride, padding = as_list(kernel_size), as_list(stride), as_list(padding)
        dilation, output_padding = as_list(dilation), as_list(output_padding)
        for d, s in zip(dilation, stride):
            assert any([s == 1, d == 1]), "don't support this."
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.dilation, self.output_padding = dilation, output_padding
        self.conv1x1 = np.prod(kernel_size) == 1
        self.transposed, self.inverse = transposed, inverse
        self.groups, self.subm, self.indice_key, self.fused_bn = groups, subm, indice_key, fused_bn
        self.weight = Parameter(torch.Tensor(*kernel_size, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(_fan_in_hwio(self.weight))
            init.uniform_(self.bias, -bound, bound)

    def forward(self, input, scale=None, shift=None, relu=False, residual=None):
        """`scale` / `shift` [Cout] and `relu`: an eval-mode BatchNorm (+ ReLU) folded into the conv
        kernel's epilogue (out = relu((conv + bias) * scale + shift)), see SparseSequential. `residual`
        [n_out, Cout]: added after the normalisation and before the ReLU (SparseBasicBlock's `+ identity`): in the
        conv kernels' epilogue (fp32 and fp16), as a tensor op on the differentiable and 1x1 paths."""
        assert isinstance(input, SparseConvTensor)
        features, indices = input.features, input.indices
        spatial_shape, batch_size = input.spatial_shape, input.batch_size
        if self.subm:
            out_spatial_shape = spatial_shape
        elif self.transposed:
            out_spatial_shape = ops.get_deconv_output_size(spatial_shape, self.kernel_size, self.stride,
                                                           self.padding, self.dilation, self.output_padding)
        else:
            out_spatial_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride,
                                                         self.padding, self.dilation)
        if self.conv1x1:
            feats = torch.mm(features, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                feats = feats + self.bias
            if scale is not None:
                feats = feats * scale + shift
            if residual is not None:
                feats = feats + residual
            if relu:
                feats = torch.relu(feats)
            out = SparseConvTensor(feats, indices, spatial_shape, batch_size)
            out.indice_dict, out.grid = input.indice_dict, input.grid
            return out
        datas = input.find_indice_pair(self.indice_key)
        table = None
        extras = None  # per-rulebook cache beside the table (the row order of the matrix-core kernels)
        if self.inverse:
            assert datas is not None and self.indice_key is not None
            _, outids, indice_pairs, indice_pair_num, out_spatial_shape = datas[:5]
            assert indice_pairs.shape[0] == np.prod(self.kernel_size), \
                "inverse conv must have same kernel size as its couple conv"
        elif self.indice_key is not None and datas is not None:
            outids, _, indice_pairs, indice_pair_num, _ = datas[:5]
            table = datas[5] if len(datas) > 5 else None
            extras = datas[6] if len(datas) > 6 else None
        else:
            # (a submanifold rulebook leaves the conv kernels' neighbour table itself; the others' is built
            # from the pairs)
            outids, indice_pairs, indice_pair_num, table = ops.get_indice_pairs(
                indices, batch_size, spatial_shape, self.kernel_size, self.stride, self.padding,
                self.dilation, self.output_padding, self.subm, self.transposed, grid=input.grid, with_table=True)
            if table is None:
                table = _ops.build_neighbor_table(indice_pairs, indice_pair_num, outids.shape[0])
            extras = {}
            input.indice_dict[self.indice_key] = (outids, indices, indice_pairs, indice_pair_num,
                                                  spatial_shape, table, extras)
        if table is None:
            table = _ops.build_neighbor_table(indice_pairs, indice_pair_num, outids.shape[0], self.inverse)

        def row_order():
            # the rows sorted by their set of occupied offsets, cached beside the rulebook's table (every conv of the
            # key takes its rows in this order: the matrix cores multiply fragments of rows that want the same offsets)
            if not ORDER_ROWS_BY_OFFSET_SET or extras is None or table.shape[1] > 32 or self.inverse:
                return None
            if extras.get("order") is None:
                extras["order"] = _ops.neighbor_table_order(table, outids.shape[0])
            return extras["order"]

        if torch.is_grad_enabled() and (features.requires_grad or self.weight.requires_grad):
            # differentiable path (functional.py): the conv through its autograd function, bias and
            # any epilogue as ordinary tensor ops
            fn = (Fsp.indice_inverse_conv if self.inverse else
                  Fsp.indice_subm_conv if self.subm else Fsp.indice_conv)
            out_features = fn(features, self.weight, indice_pairs, indice_pair_num, outids.shape[0], table)
            if self.bias is not None:
                out_features = out_features + self.bias
            if scale is not None:
                out_features = out_features * scale + shift
            if residual is not None:
                out_features = out_features + residual
            if relu:
                out_features = torch.relu(out_features)
        elif features.dtype == torch.float16:
            # half-precision features: the fp16 matrix-core kernel (fp16 rows gathered as they are, fp32
            # accumulation, one rounding). A folded BatchNorm goes into the filters' columns and the bias.
            w, b = self.weight.float(), (self.bias.float() if self.bias is not None else None)
            if scale is not None:
                w = w * scale
                b = shift if b is None else b * scale + shift
            # (a residual: the fp16 sum of the conv's fp16 rows and the identity's, as the reference adds them)
            out_features = _ops.indice_conv_table_half(features, w, table, outids.shape[0],
                                                       self.subm and not self.inverse, bias=b, relu=relu,
                                                       order=row_order() if residual is None else None, residual=residual)
        else:
            out_features = _ops.indice_conv_table(features, self.weight, table, outids.shape[0],
                                                  self.subm and not self.inverse, bias=self.bias, scale=scale,
                                                  shift=shift, relu=relu, order=row_order(), residual=residual)
        out = SparseConvTensor(out_features, outids, out_spatial_shape, batch_size)
        out.indice_dict, out.grid = input.indice_dict, input.grid
        return out


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, indice_key=indice_key)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, True, indice_key=indice_key)


class SparseConvTranspose3d(SparseConvolution):
    """Transposed sparse conv (conv.py:308-331): every input emits its whole kernel footprint
    (getIndicePairsDeConv, geometry.h:197-245); output shape = get_deconv_output_size."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, transposed=True, indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                         indice_key=indice_key)


# ---- the 4-D forms (conv.py:257-279, 422-445 of the reference; get_indice_pairs_4d, src/all.cc:28-29): the rulebook is
# ops._indice_pairs_nd (tensor operations: nothing on the hot path is 4-D), the convolution over its 81-offset pairs
# the HIP kernels' any-volume form
class SparseConv4d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(4, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, indice_key=indice_key)


class SubMConv4d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(4, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, True, indice_key=indice_key)


# ---- the 2-D forms (conv.py:207-229, 282-305, 334-349, 370-393 of the reference): the same module on planes; the
# rulebook of a plane is a one-cell-deep volume's (ops.get_indice_pairs)
class SparseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, indice_key=indice_key)


class SubMConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, True, indice_key=indice_key)


class SparseConvTranspose2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                         bias, transposed=True, indice_key=indice_key)


class SparseInverseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(2, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                         indice_key=indice_key)
