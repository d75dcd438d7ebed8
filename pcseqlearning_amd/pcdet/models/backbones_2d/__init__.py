from .map_to_bev import HeightCompression

__all__ = {"HeightCompression": HeightCompression}
