"""SparseMaxPool / SparseMaxPool3d (pcdet/ops/spconv/pool.py:21-85): the pooling window's rulebook
from `get_indice_pairs`, values by `indice_maxpool` (zero floor, as the reference)."""
from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor


class SparseMaxPool(SparseModule):
    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False):
        super().__init__()
        as_list = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim  # noqa: E731
        self.ndim = ndim
        self.kernel_size, self.stride = as_list(kernel_size), as_list(stride)
        self.padding, self.dilation = as_list(padding), as_list(dilation)
        self.subm = subm

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        spatial_shape = input.spatial_shape
        if not self.subm:
            out_spatial_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride,
                                                         self.padding, self.dilation)
        else:
            out_spatial_shape = spatial_shape
        outids, indice_pairs, indice_pairs_num = ops.get_indice_pairs(
            input.indices, input.batch_size, spatial_shape, self.kernel_size, self.stride, self.padding,
            self.dilation, 0, self.subm)
        out_features = Fsp.indice_maxpool(input.features, indice_pairs, indice_pairs_num, outids.shape[0])
        out = SparseConvTensor(out_features, outids, out_spatial_shape, input.batch_size)
        out.indice_dict, out.grid = input.indice_dict, input.grid
        return out


class SparseMaxPool3d(SparseMaxPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__(3, kernel_size, stride, padding, dilation)
