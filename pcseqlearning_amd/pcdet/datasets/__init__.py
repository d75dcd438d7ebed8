"""Sequence I/O for the registration path: the reference's dataset interface
(`pcdet/datasets/__init__.py:18-33`) reduced to the one dataset the path reads."""
from .dataset import DatasetTemplate
from .waymo.waymo_dataset import WaymoDataset

__all__ = {
    "DatasetTemplate": DatasetTemplate,
    "WaymoDataset": WaymoDataset,
}
