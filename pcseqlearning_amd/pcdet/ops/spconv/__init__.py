"""Drop-in for the spconv-1.x module surface the backbone uses (pcdet/ops/spconv/__init__.py):
SparseConvTensor, SubMConv3d, SparseConv3d, SparseInverseConv3d, SparseSequential, SparseModule,
and `ops.get_indice_pairs` / `ops.indice_conv`. Inference path (no autograd: SURVEY.md 8f.4)."""
from . import ops  # noqa: F401
from .conv import SparseConv3d, SparseConvolution, SparseInverseConv3d, SubMConv3d  # noqa: F401
from .modules import SparseModule, SparseSequential  # noqa: F401
from .structure import SparseConvTensor  # noqa: F401
