"""`Timer` with the reference's name and use (pcdet/utils/timer.py:3-13) but device-correct:
the span is measured with HIP events on the current stream when a GPU tensor path is active,
and printing is optional so the hot path does not have to synchronise."""
import threading
import time

import torch

VERBOSE = False
SYNC = False  # synchronise the device around every span (stage-time tooling)
RECORDS = {}
STATS = {}    # sizes the stages log for the roofline report: name -> accumulated count
_LOCK = threading.Lock()  # the sequences of a batch log from their own host threads


def count(name, value):
    """Accumulate a size (points, voxels, edges ...) under `name`; host integers only (no device sync)."""
    with _LOCK:
        STATS[name] = STATS.get(name, 0) + int(value)


class Timer:
    def __init__(self, name, sync=False):
        self.name = name
        self.sync = sync or VERBOSE or SYNC

    def __enter__(self):
        if self.sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        dt = time.perf_counter() - self.t0
        with _LOCK:
            RECORDS[self.name] = RECORDS.get(self.name, 0.0) + dt
        if VERBOSE:
            print(f"{self.name}: {dt:.4f}s")
        return False
