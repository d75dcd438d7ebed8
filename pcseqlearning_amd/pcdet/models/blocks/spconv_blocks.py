"""The two building blocks of the sparse backbones (reference: pcdet/models/blocks/spconv_blocks.py:8-64).

`post_act_block(cin, cout, k, ...)`   conv -> norm -> ReLU as a SparseSequential (children "0", "1", "2");
                                      conv_type picks subM / strided / inverse conv, all without bias
`SparseBasicBlock(inplanes, planes)`  residual pair of 3x3x3 subM convs on one rulebook (children conv1,
                                      bn1, relu, conv2, bn2, downsample): relu(bn2(conv2(relu(bn1(conv1 x)))) + x)

In eval mode under no_grad both normalisations (and the inner ReLU) are folded into the conv kernels'
epilogues (`fold_batchnorm`); otherwise the modules are applied one by one.
"""
import torch
from torch import nn

from ...ops.spconv.modules import fold_batchnorm
from ...utils.spconv_utils import replace_feature, spconv


FUSE_RESIDUAL_ADD = True  # (False: the add and the ReLU as tensor ops after the folded convs; same bits)


def _make_conv(conv_type, cin, cout, ksize, stride, padding, key):
    makers = {
        "subm": lambda: spconv.SubMConv3d(cin, cout, ksize, bias=False, indice_key=key),
        "spconv": lambda: spconv.SparseConv3d(cin, cout, ksize, stride=stride, padding=padding, bias=False,
                                              indice_key=key),
        "inverseconv": lambda: spconv.SparseInverseConv3d(cin, cout, ksize, indice_key=key, bias=False),
    }
    if conv_type not in makers:
        raise NotImplementedError(conv_type)
    return makers[conv_type]()


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0,
                   conv_type="subm", norm_fn=None):
    layers = (_make_conv(conv_type, in_channels, out_channels, kernel_size, stride, padding, indice_key),
              norm_fn(out_channels), nn.ReLU())
    return spconv.SparseSequential(*layers)


class SparseBasicBlock(spconv.SparseModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, indice_key=None, norm_fn=None):
        super().__init__()
        widths = ((inplanes, planes, stride), (planes, planes, 1))
        for n, (cin, cout, st) in enumerate(widths, start=1):
            setattr(self, f"conv{n}", spconv.SubMConv3d(cin, cout, kernel_size=3, stride=st, padding=1, bias=False,
                                                        indice_key=indice_key))
            setattr(self, f"bn{n}", norm_fn(cout))
            if n == 1:
                self.relu = nn.ReLU()
        self.downsample, self.stride = downsample, stride

    def _folded(self, x):
        def ok(bn):
            return (isinstance(bn, nn.BatchNorm1d) and not bn.training and bn.track_running_stats
                    and bn.running_mean is not None)
        return x.features.is_cuda and not torch.is_grad_enabled() and ok(self.bn1) and ok(self.bn2)

    def forward(self, x):
        skip = x.features
        if self._folded(x) and self.downsample is None and FUSE_RESIDUAL_ADD:
            # relu(bn2(conv2(.)) + x) in the second conv's epilogue (pcs_indice_conv_table_residual_f32): the same
            # operations in the same order as below, two passes over the block's output less
            return self.conv2(self.conv1(x, *fold_batchnorm(self.bn1), relu=True), *fold_batchnorm(self.bn2), relu=True,
                              residual=skip)
        if self._folded(x):
            y = self.conv2(self.conv1(x, *fold_batchnorm(self.bn1), relu=True), *fold_batchnorm(self.bn2), relu=False)
        else:
            y = self.conv1(x)
            y = self.conv2(replace_feature(y, self.relu(self.bn1(y.features))))
            y = replace_feature(y, self.bn2(y.features))
        if self.downsample is not None:
            skip = self.downsample(x)
        return replace_feature(y, self.relu(y.features + skip))
