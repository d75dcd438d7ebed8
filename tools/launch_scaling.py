"""Host launch-rate probe: how do kernel launches from T host threads (one HIP stream each) scale?

  torch : each thread issues small in-place torch ops (interpreter + ATen dispatch per launch)
  native: each thread calls pcs_profile_launch_loop once (a C loop of launches, GIL released)

Prints launches/s per mode and thread count. Used to decide where the tracking control flow has
to live.
"""
import argparse
import ctypes
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pcseqlearning_amd import lib as L  # noqa: E402


def run(mode, T, n):
    lib = L.load()
    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream(dev) for _ in range(T)]
    xs = [torch.zeros(64, device=dev) for _ in range(T)]
    barrier = threading.Barrier(T + 1)

    def work(i):
        with torch.cuda.stream(streams[i]):
            barrier.wait()
            if mode == "torch":
                x = xs[i]
                for _ in range(n):
                    x.add_(1.0)
            elif mode.startswith("busy"):
                ticks, blocks = (int(x) for x in mode[4:].split("x"))
                rc = lib.pcs_profile_busy_loop(ctypes.c_int64(n), ctypes.c_int(ticks), ctypes.c_int(blocks),
                                               ctypes.c_void_p(streams[i].cuda_stream))
                assert rc == 0
            elif mode.startswith("graph"):
                # graph<nodes>[:<ticks>x<blocks>]: the chain captured once, launched n / nodes times
                spec = mode[5:].split(":")
                ticks, blocks = (int(x) for x in spec[1].split("x")) if len(spec) > 1 else (0, 1)
                rc = lib.pcs_profile_graph_loop(ctypes.c_int64(n), ctypes.c_int(int(spec[0])), ctypes.c_int(ticks),
                                                ctypes.c_int(blocks), ctypes.c_void_p(streams[i].cuda_stream))
                assert rc == 0
            elif mode.startswith("sync"):
                rc = lib.pcs_profile_sync_loop(ctypes.c_int64(n), ctypes.c_int(int(mode[4:])),
                                               ctypes.c_void_p(streams[i].cuda_stream))
                assert rc == 0
            else:
                rc = lib.pcs_profile_launch_loop(ctypes.c_int64(n), ctypes.c_void_p(streams[i].cuda_stream))
                assert rc == 0
            streams[i].synchronize()
            barrier.wait()

    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    for t in th:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    barrier.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    return T * n / dt


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--touch", action="store_true", help="only: begin-to-end time of launches that merely touch memory")
    ap.add_argument("--graphs", action="store_true", help="only: dependent chains as stream launches vs captured graphs")
    args = ap.parse_args()
    run("native", 1, 1000)
    if args.touch:
        # begin-to-end time of a launch that only touches memory, in a dependent chain on one stream
        lib = L.load()
        lib.pcs_profile_touch_loop.restype = ctypes.c_double
        s = torch.cuda.Stream()
        names = {0: "load one row per thread", 1: "store one row per thread", 2: "alternate store / load",
                 3: "two dependent loads"}
        for rows in (4096, 131072, 1048576):
            for mode in (0, 1, 2, 3):
                us = lib.pcs_profile_touch_loop(ctypes.c_int64(1024), ctypes.c_int(mode), ctypes.c_int(rows),
                                                ctypes.c_void_p(s.cuda_stream))
                print(f"touch rows={rows:8d} ({rows * 16 / 1e6:6.2f} MB) {names[mode]:26s} {us:6.2f} us per launch", flush=True)
        sys.exit(0)
    if args.graphs:
        # 500 ticks = 5 us kernels, 64 workgroups: the size of an ICP iteration kernel
        # graphN:-1x1 = chains of kernels that load one row per thread (N = 1: plain stream launches)
        for mode in ("native", "graph24", "busy500x64", "graph24:500x64", "graph96:500x64", "graph1:-1x1",
                     "graph24:-1x1", "graph96:-1x1"):
            for T in (1, 4):
                r = run(mode, T, 9600)
                print(f"{mode:16s} threads={T}  {1e6 / r * T:7.2f} us per kernel per chain  (aggregate {r / 1e3:7.1f} k kernels/s)",
                      flush=True)
        sys.exit(0)
    for mode in ("torch", "native"):
        for T in (1, 2, 3, 5, 9):
            r = run(mode, T, args.n)
            print(f"{mode:7s} threads={T}  {r / 1e3:8.1f} k launches/s  ({1e6 / r * T:6.2f} us per launch per thread)", flush=True)
    for mode in ("busy500x1", "busy500x64", "busy1000x512"):
        for T in (1, 2, 3, 4, 6, 9):
            r = run(mode, T, args.n // 4)
            print(f"{mode:12s} threads={T}  {1e6 / r * T:7.2f} us per kernel per chain  (aggregate {r / 1e3:7.1f} k kernels/s)", flush=True)
    for mode in ("sync0", "sync1", "sync2"):
        for T in (1, 3, 9):
            r = run(mode, T, args.n // 10)
            print(f"{mode:7s} threads={T}  {1e6 / r * T:7.1f} us per (8 launches + read + sync) per thread", flush=True)
