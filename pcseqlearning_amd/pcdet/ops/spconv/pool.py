"""Sparse max-pooling modules (reference: pcdet/ops/spconv/pool.py:21-85). The pooling window's rulebook
comes from `get_indice_pairs` like a conv's; the values from `indice_maxpool` (zero floor, ties all
receive the gradient -- the reference's semantics, see include/pcs.h)."""
from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor


def _per_axis(value, ndim):
    return list(value) if isinstance(value, (list, tuple)) else [value] * ndim


class SparseMaxPool(SparseModule):
    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False):
        super().__init__()
        self.ndim, self.subm = ndim, subm
        self.kernel_size, self.stride, self.padding, self.dilation = (
            _per_axis(v, ndim) for v in (kernel_size, stride, padding, dilation))

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        window = (self.kernel_size, self.stride, self.padding, self.dilation)
        shape_out = list(input.spatial_shape) if self.subm else ops.get_conv_output_size(input.spatial_shape, *window)
        outids, pairs, pair_num = ops.get_indice_pairs(input.indices, input.batch_size, input.spatial_shape,
                                                       *window, 0, self.subm)
        pooled = Fsp.indice_maxpool(input.features, pairs, pair_num, outids.shape[0])
        out = SparseConvTensor(pooled, outids, shape_out, input.batch_size)
        out.indice_dict, out.grid = input.indice_dict, input.grid
        return out


class SparseMaxPool3d(SparseMaxPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__(3, kernel_size, stride, padding, dilation)


class SparseMaxPool2d(SparseMaxPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__(2, kernel_size, stride, padding, dilation)
