from .cluster_proposal import ClusterProposal
from .cluster_tracking import ClusterTracking
from .ground_plane_remover import GroundPlaneRemover

__all__ = {
    "GroundPlaneRemover": GroundPlaneRemover,
    "ClusterProposal": ClusterProposal,
    "ClusterTracking": ClusterTracking,
}
