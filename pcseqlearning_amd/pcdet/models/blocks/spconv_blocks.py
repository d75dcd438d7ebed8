"""post_act_block / SparseBasicBlock (pcdet/models/blocks/spconv_blocks.py:8-64)."""
import torch
import torch.nn as nn

from ...ops.spconv.modules import fold_batchnorm
from ...utils.spconv_utils import replace_feature, spconv


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0,
                   conv_type="subm", norm_fn=None):
    if conv_type == "subm":
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == "spconv":
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                   bias=False, indice_key=indice_key)
    elif conv_type == "inverseconv":
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key,
                                          bias=False)
    else:
        raise NotImplementedError
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class SparseBasicBlock(spconv.SparseModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, indice_key=None, norm_fn=None):
        super().__init__()
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1,
                                       bias=False, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False,
                                       indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x.features
        if (not self.training and x.features.is_cuda and not torch.is_grad_enabled()
                and self.bn1.track_running_stats and self.bn2.track_running_stats):
            # eval mode: both normalisations (and the first ReLU) ride in the conv epilogues
            out = self.conv1(x, *fold_batchnorm(self.bn1), relu=True)
            out = self.conv2(out, *fold_batchnorm(self.bn2), relu=False)
        else:
            out = self.conv1(x)
            out = replace_feature(out, self.relu(self.bn1(out.features)))
            out = self.conv2(out)
            out = replace_feature(out, self.bn2(out.features))
        if self.downsample is not None:
            identity = self.downsample(x)
        return replace_feature(out, self.relu(out.features + identity))
