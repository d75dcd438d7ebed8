from .spconv_backbone import VoxelBackBone8x, VoxelResBackBone8x
from .vfe import DynamicMeanVFE, MeanVFE  # noqa: F401

__all__ = {
    "VoxelBackBone8x": VoxelBackBone8x,
    "VoxelResBackBone8x": VoxelResBackBone8x,
}
