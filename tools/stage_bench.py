#!/usr/bin/env python3
"""The pre-tracking stages of ONE full-size synthetic sequence, alone on the device, each timed over a few
device-synchronised repetitions: ground removal and the proposal stage (the 32-NN graph at the largest radius + the
components of its three cuts), with the cell-sorted search's own counters (runs, fallback queries).

    python tools/stage_bench.py [--reps 5] [--seed 0]      -> one JSON line
"""
import argparse
import copy
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--points", type=int, default=160000)
    args = ap.parse_args()
    import bench
    host = bench.load_sequences([args.seed], args.frames, args.points)[0]
    import torch
    from pcseqlearning_amd import lib as L
    from pcseqlearning_amd import ops
    from pcseqlearning_amd.pcdet import models
    from tools.run_sequence import build_model
    dev = torch.device("cuda", 0)
    batch = models.load_data_to_gpu(host, dev)
    model = build_model(save=False, device=dev)
    lib = L.load()
    out = {}
    with torch.no_grad():
        seq = model.build_sequence(batch, 0)
        ground, proposal = model.preprocessors[0], model.preprocessors[1]

        def timed(fn, reps):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            return r, ts

        after, ts = timed(lambda: ground(copy.copy(seq)), args.reps)
        out["ground_ms"] = [round(t, 3) for t in ts]
        out["points_after_ground_removal"] = int(after["point_fxyz"].shape[0])
        _, ts = timed(lambda: proposal.propose_cluster(copy.copy(after)), args.reps)
        out["propose_cluster_ms"] = [round(t, 3) for t in ts]
        # the search of the first 10-frame chunk alone, and its counters
        fr = after["point_sweep"].reshape(-1)
        pts = after["point_fxyz"][(fr < 10).nonzero(as_tuple=True)[0]].float().contiguous()
        n = pts.shape[0]
        d_num = torch.empty(1, dtype=torch.int64, device=dev)
        ws = L.workspace(lib.pcs_radius_graph_workspace_bytes(L.i64(n), L.i64(n), L.i32(32)), dev)

        def search():
            L.check(lib.pcs_radius_graph_search(L.ptr(pts), L.i64(n), L.ptr(pts), L.i64(n), L.f32(1.25), L.f32(-0.625),
                                                L.i32(0), L.i32(0), L.i32(32), L.i32(1), L.ptr(d_num), L.ptr(ws),
                                                L.i64(ws.numel()), L.stream()), "search")
        _, ts = timed(search, args.reps)
        cnt = (ctypes.c_int32 * 8)()
        L.check(lib.pcs_radius_graph_counters(L.i64(n), L.i64(n), L.i32(32), L.ptr(ws), L.i64(ws.numel()), cnt,
                                              L.stream()), "counters")
        out["chunk0"] = {"points": n, "edges": int(d_num.item()), "search_ms": [round(t, 3) for t in ts],
                         "counters": list(cnt)}
        _, ts = timed(lambda: ops.knn_components(pts, [1.25, 0.75, 0.25], 32), args.reps)
        out["chunk0"]["knn_components_ms"] = [round(t, 3) for t in ts]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
