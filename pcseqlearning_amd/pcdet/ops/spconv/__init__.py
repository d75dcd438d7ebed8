"""Drop-in for the spconv-1.x module surface the backbone uses (pcdet/ops/spconv/__init__.py):
SparseConvTensor, SubMConv3d, SparseConv3d, SparseConvTranspose3d, SparseInverseConv3d, SparseSequential, SparseModule,
`ops.get_indice_pairs` / `ops.indice_conv` / `ops.indice_conv_backward` and the autograd functions of
`functional.py`, `SparseMaxPool3d`; the 2-D forms (`SparseConv2d`, `SubMConv2d`, `SparseConvTranspose2d`,
`SparseInverseConv2d`, `SparseMaxPool2d`) on the same kernels (a plane = a volume one cell deep)."""
from . import functional, ops  # noqa: F401
from .conv import (SparseConv2d, SparseConv3d, SparseConv4d, SparseConvolution, SparseConvTranspose2d,  # noqa: F401
                   SparseConvTranspose3d, SparseInverseConv2d, SparseInverseConv3d, SubMConv2d, SubMConv3d, SubMConv4d)
from .modules import SparseModule, SparseSequential  # noqa: F401
from .pool import SparseMaxPool, SparseMaxPool2d, SparseMaxPool3d  # noqa: F401
from .structure import SparseConvTensor  # noqa: F401
