"""The C-ABI library loads on a machine without a GPU and exports every symbol include/pcs.h
declares (no compute calls here)."""
import ctypes
import os

from pcseqlearning_amd import lib as L


def test_library_exports_every_declared_symbol():
    names = L.declared_symbols()
    assert len(names) >= 15
    assert os.path.exists(L.LIB_PATH), "run `python -m pcseqlearning_amd.build` first"
    handle = ctypes.CDLL(L.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, f"declared in include/pcs.h but not exported: {missing}"


def test_no_cpu_fallback():
    """The product path refuses host tensors instead of silently computing on the CPU."""
    import pytest
    import torch
    from pcseqlearning_amd import ops
    with pytest.raises(L.PcsError):
        ops.minmax4(torch.zeros(4, 4))
