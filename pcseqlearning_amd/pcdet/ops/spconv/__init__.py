"""Drop-in for the spconv-1.x module surface the backbone uses (pcdet/ops/spconv/__init__.py):
SparseConvTensor, SubMConv3d, SparseConv3d, SparseConvTranspose3d, SparseInverseConv3d, SparseSequential, SparseModule,
`ops.get_indice_pairs` / `ops.indice_conv` / `ops.indice_conv_backward` and the autograd functions of
`functional.py`, `SparseMaxPool3d`."""
from . import functional, ops  # noqa: F401
from .conv import (SparseConv3d, SparseConvolution, SparseConvTranspose3d, SparseInverseConv3d,  # noqa: F401
                   SubMConv3d)
from .modules import SparseModule, SparseSequential  # noqa: F401
from .pool import SparseMaxPool, SparseMaxPool3d  # noqa: F401
from .structure import SparseConvTensor  # noqa: F401
