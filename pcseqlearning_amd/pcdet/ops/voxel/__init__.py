from .voxel_modules import VoxelAggregation

__all__ = ["VoxelAggregation"]
