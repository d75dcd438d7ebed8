"""The two helpers of pcdet/utils/common_utils.py that the hot path uses."""
import numpy as np
import torch


class EasyDict(dict):
    """Attribute-style dict (the reference uses the `easydict` package, absent on the target)."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        for k, v in dict(d or {}, **kwargs).items():
            self[k] = v

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value


def filter_dict(data_dict, mask, ignore_keys=()):
    """Index every value of a dict with the same mask / index tensor
    (pcdet/utils/common_utils.py:67-78), checking lengths for boolean masks."""
    out = {}
    index = mask
    is_bool = isinstance(mask, (torch.Tensor, np.ndarray)) and mask.dtype in (torch.bool, np.bool_)
    if is_bool and isinstance(mask, torch.Tensor) and mask.dim() == 1 and mask.is_cuda:
        # one compaction of the mask shared by every entry (a boolean index would redo it per entry)
        index = mask.nonzero(as_tuple=True)[0]
    for key, val in data_dict.items():
        if key in ignore_keys:
            out[key] = val
            continue
        if is_bool:
            assert mask.shape[0] == len(val), \
                f"MisMatch for key={key}, mask.shape={mask.shape}, data.shape={len(val)}"
        out[key] = val[index] if isinstance(val, torch.Tensor) else val[mask]
    return out


def indexing_list_elements(data_dict, idx):
    """Pick element idx of every list-valued entry (graph configs with per-level lists)."""
    return {k: (v[idx] if isinstance(v, list) else v) for k, v in data_dict.items()}


def run_deferred(data_dict):
    """Run the side computations a module parked under `_deferred` (callables returning a dict of
    new entries) and merge their results. A module defers work whose results no later stage
    consumes (the GT-evaluation side channel of ClusterProposal), so that the main thread can do it
    while the tracking units run in native code on their own threads and streams."""
    for fn in data_dict.pop("_deferred", []):
        data_dict.update(fn())
    return data_dict


# ---- side streams that survive the call that used them. torch's caching allocator keeps its free
# blocks per stream: a fresh torch.cuda.Stream() per call lands on another stream of torch's pool of
# 32 every time, so nothing cached is ever reused and every call goes back to hipMalloc (measured: a
# batch of sequences 10-40x slower once the pools had drifted apart).
import contextlib
import threading

_STREAMS = {}
_STREAMS_LOCK = threading.Lock()


@contextlib.contextmanager
def pooled_stream(device):
    """A side stream of `device` for the duration of the block; returned to the pool afterwards."""
    import torch
    index = torch.device(device).index
    if index is None:
        index = torch.cuda.current_device()
    with _STREAMS_LOCK:
        free = _STREAMS.setdefault(index, [])
        stream = free.pop() if free else None
    if stream is None:
        stream = torch.cuda.Stream(device=index)
    try:
        yield stream
    finally:
        with _STREAMS_LOCK:
            _STREAMS[index].append(stream)



_THREAD_STREAMS = threading.local()


def thread_stream(device, which=0):
    """The calling thread's own side stream number `which` of `device`, created once and kept for the
    thread's life. torch's caching allocator keeps its free blocks per stream, so a worker that takes the
    SAME stream every time finds the blocks it freed last time; with streams handed out from a shared pool
    a worker met a different pool of blocks at every call and kept going back to hipMalloc (~60-100 device
    allocations per batch, the reserved size growing by 10 GiB per batch)."""
    import torch
    index = torch.device(device).index
    if index is None:
        index = torch.cuda.current_device()
    slot = getattr(_THREAD_STREAMS, "slot", None)
    if slot is not None:
        # a worker SLOT's streams (stream_slot): the pool's threads take the slots in any order, and the allocator's
        # pools belong to streams -- keyed by slot, slot i meets the same sizes at every step whichever thread runs it
        with _SLOT_LOCK:
            stream = _SLOT_STREAMS.get((slot, index, which))
            if stream is None:
                stream = _SLOT_STREAMS[(slot, index, which)] = torch.cuda.Stream(device=index)
        return stream
    table = getattr(_THREAD_STREAMS, "table", None)
    if table is None:
        table = _THREAD_STREAMS.table = {}
    stream = table.get((index, which))
    if stream is None:
        stream = table[(index, which)] = torch.cuda.Stream(device=index)
    return stream


_SLOT_STREAMS = {}
_SLOT_LOCK = threading.Lock()


class stream_slot:
    """Within the block, thread_stream hands out the streams of worker slot `slot` instead of the calling thread's."""

    def __init__(self, slot):
        self.slot = slot

    def __enter__(self):
        self.prev = getattr(_THREAD_STREAMS, "slot", None)
        _THREAD_STREAMS.slot = self.slot
        return self

    def __exit__(self, *exc):
        _THREAD_STREAMS.slot = self.prev
        return False
