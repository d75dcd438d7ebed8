from .dynamic_mean_vfe import DynamicMeanVFE
from .mean_vfe import MeanVFE

__all__ = {
    "MeanVFE": MeanVFE,
    "DynMeanVFE": DynamicMeanVFE,
}
