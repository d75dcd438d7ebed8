"""ctypes binding of libpcs_hip.so (the C ABI declared in include/pcs.h).

The product path has no CPU fallback: if the library is missing or a call returns an error code,
`PcsError` is raised. Tensors are passed as raw device pointers (`tensor.data_ptr()`), the stream
is torch's current HIP stream.
"""
import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libpcs_hip.so")
HEADER_PATH = os.path.join(ROOT, "include", "pcs.h")

_ERRORS = {-1: "PCS_ERR_ARG", -2: "PCS_ERR_WORKSPACE", -3: "PCS_ERR_HIP", -4: "PCS_ERR_RANGE"}


class PcsError(RuntimeError):
    pass


_lib = None


def declared_symbols():
    """Names of every function include/pcs.h declares."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcs_\w+)\s*\(", text)))


def load():
    """Load libpcs_hip.so; raises PcsError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PcsError(
                f"{LIB_PATH} not found: build it with `python -m pcseqlearning_amd.build` "
                "(there is no CPU fallback for the HIP path)")
        _lib = ctypes.CDLL(LIB_PATH)
        for name in declared_symbols():
            fn = getattr(_lib, name)  # AttributeError if the library lacks a declared symbol
            if name.endswith("_bytes"):
                fn.restype = ctypes.c_size_t
            elif name.endswith("_capacity"):
                fn.restype = ctypes.c_int64
            else:
                fn.restype = ctypes.c_int
    return _lib


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_contiguous(), "pcs kernels need contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def i64(x):
    return ctypes.c_int64(int(x))


def i32(x):
    return ctypes.c_int(int(x))


def f32(x):
    return ctypes.c_float(float(x))


def host_ints(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def host_floats(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def check(rc, what):
    if rc != 0:
        raise PcsError(f"{what} failed: {_ERRORS.get(rc, rc)}")


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def require_gpu(t, what):
    if not t.is_cuda:
        raise PcsError(f"{what}: tensor is on {t.device}; the HIP path needs device memory "
                       "(no CPU fallback)")
