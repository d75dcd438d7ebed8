"""Sequence loader, device pass: `pcs_assemble_sweeps` on a configs[2]-sized
sample (20 sweeps x 160 k rows, with seg labels) timed with HIP events on the launch stream;
algorithmic bytes = 40 B read (32 raw + 8 labels) + 40 B written (16 bxyz + 12 feat + 4 sweep + 8
labels) per row. Beside it: the pinned-host -> HBM copy of the same rows (PCIe-inclusive rate) and
the numpy oracle of the same arithmetic on one host core. Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pcseqlearning_amd import ops  # noqa: E402

HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweeps", type=int, default=20)
    ap.add_argument("--points", type=int, default=160000)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    F, n = args.sweeps, args.points
    N = F * n
    raw = np.empty((N, 8), dtype=np.float32)
    raw[:, :3] = rng.uniform(-75, 75, (N, 3)) * [1, 1, 0.05]
    raw[:, 3] = rng.gamma(1.0, 0.8, N)
    raw[:, 4:] = rng.uniform(0, 1, (N, 4))
    raw[:, 5] *= 75
    seg = rng.integers(0, 1000, (N, 2)).astype(np.int32)
    yaw = 0.03 * np.arange(F)
    Ts = np.zeros((F, 3, 4))
    Ts[:, 0, 0], Ts[:, 0, 1], Ts[:, 1, 0], Ts[:, 1, 1], Ts[:, 2, 2] = np.cos(yaw), -np.sin(yaw), np.sin(yaw), np.cos(yaw), 1
    Ts[:, :, 3] = rng.uniform(-20, 20, (F, 3))
    off = torch.arange(F + 1, dtype=torch.int64) * n
    h_raw, h_seg = torch.from_numpy(raw).pin_memory(), torch.from_numpy(seg).pin_memory()
    d_raw, d_seg = h_raw.to(dev, non_blocking=True), h_seg.to(dev, non_blocking=True)  # first touch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d_raw.copy_(h_raw, non_blocking=True)
    d_seg.copy_(h_seg, non_blocking=True)
    torch.cuda.synchronize()
    t_h2d = time.perf_counter() - t0
    d_off, d_T = off.to(dev), torch.from_numpy(Ts).to(dev)
    d_ids = torch.arange(F, dtype=torch.int32, device=dev)
    for _ in range(5):
        out = ops.assemble_sweeps(d_raw, d_seg, d_off, d_T, d_ids, 0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.reps)]
    torch.cuda.synchronize()
    # ops allocates its outputs per call (caching allocator, no device malloc after warm-up); the
    # events bracket each launch on the launch stream
    from pcseqlearning_amd import lib as L
    lib = L.load()
    bxyz, feat, sweep, seg_l, inst_l = out
    for r in range(args.reps):
        ev[2 * r].record()
        L.check(lib.pcs_assemble_sweeps(L.ptr(d_raw), L.ptr(d_seg), L.ptr(d_off), L.ptr(d_T), L.ptr(d_ids), L.i32(F),
                                        L.i64(N), L.i32(0), L.ptr(bxyz), L.ptr(feat), L.ptr(sweep), L.ptr(seg_l),
                                        L.ptr(inst_l), L.stream()), "pcs_assemble_sweeps")
        ev[2 * r + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[2 * r].elapsed_time(ev[2 * r + 1]) for r in range(args.reps))
    avg_ms = float(np.mean(ms))
    nbytes = 80.0 * N
    line = {
        "metric": "sequence loader device pass (pcs_assemble_sweeps)", "unit": "rows/s",
        "value": round(N / (avg_ms * 1e-3), 1), "rows": N, "sweeps": F, "launches": args.reps,
        "avg_us_per_launch": round(avg_ms * 1e3, 2), "median_us_per_launch": round(ms[len(ms) // 2] * 1e3, 2),
        "roofline": {"bound": "hbm", "kernel": "assemble_sweeps_kernel", "achieved": round(nbytes / (avg_ms * 1e-3) / 1e9, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "algorithmic_bytes_per_launch": nbytes, "traffic": None},
        "h2d_pinned": {"seconds": round(t_h2d, 5), "GB/s": round((raw.nbytes + seg.nbytes) / t_h2d / 1e9, 2),
                       "rows_per_s_pcie_inclusive": round(N / (t_h2d + avg_ms * 1e-3), 1)},
    }
    pmc_path = os.path.join(ROOT, "profiles", "r01_loader_hbm_pmc.json")
    if os.path.exists(pmc_path) and N == 3200000:
        with open(pmc_path) as fh:
            pmc = json.load(fh)
        line["roofline"]["traffic"] = pmc["corrected_fetch_bytes_per_dispatch"] + pmc["write_bytes_per_dispatch"]
        line["roofline"]["traffic_source"] = "profiles/r01_loader_hbm_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE)"
    if not args.no_cpu:
        from oracle import loader_oracle as LO
        m = N
        t0 = time.perf_counter()
        for s in range(m // n):
            p = LO.get_lidar(raw[s * n:(s + 1) * n]).astype(np.float32)
            T = np.vstack([Ts[s], [0, 0, 0, 1]])
            xyz = p[:, :3].copy()
            xyz[:, :3] = xyz[:, :3] @ T[:3, :3].T + T[:3, 3]
            np.pad(xyz, ((0, 0), (1, 0)), mode="constant", constant_values=0)
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(m / dt, 1), "unit": "rows/s", "cores": 1, "kind": "port",
                                "sample": f"numpy oracle (get_lidar + pose product + batch column) on all {F} sweeps, {dt:.2f}s"}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
