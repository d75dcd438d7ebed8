"""Registry of the sequence preprocessors (reference: pcdet/models/registration/preprocessors/__init__.py):
`__all__[cfg.NAME]` is the class `SimpleReg` instantiates for every entry of MODEL.PREPROCESSORS."""
from . import cluster_proposal, cluster_tracking, ground_plane_remover

GroundPlaneRemover = ground_plane_remover.GroundPlaneRemover
ClusterProposal = cluster_proposal.ClusterProposal
ClusterTracking = cluster_tracking.ClusterTracking

__all__ = {cls.__name__: cls for cls in (GroundPlaneRemover, ClusterProposal, ClusterTracking)}
