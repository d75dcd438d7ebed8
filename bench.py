"""bench.py -- end-to-end cluster-tracking throughput on synthetic Waymo-shaped sequences.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Workload = BASELINE.json configs[2]: a 20-frame x 160k-pt sequence through the reference's
cluster_tracking_TLS_multiradius_every8 pipeline (8 cm subsample -> GroundPlaneRemover ->
ClusterProposal at 3 radii -> ClusterTracking: 3 keys x anchors {0, 8, 16} x +-8 frames x 3 ICP
levels), one sequence per GPU per step, inputs resident in HBM before the timed region, nothing
written to disk. N GPUs process N different sequences (weak scaling); the only collective is the
final all-reduce of {max wall time, sequences done} over RCCL.

Rank 0 prints ONE JSON line: whole-job sequences/s plus
  roofline      dominant kernel of the tracking stage (icp_nn_kernel: two-direction 1-NN
                search of the fused ICP), timed with HIP events on its launch
                stream inside the timed steps; achieved = algorithmic bytes / kernel time.
  cpu_baseline  the CPU oracle (serial numpy/C++ restatement, kind "port") timed on a bounded
                sample of the same sequence and scaled to sequences/s (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pcseqlearning_amd import lib as L  # noqa: E402
from pcseqlearning_amd import synthetic  # noqa: E402
from pcseqlearning_amd.pcdet import models  # noqa: E402
from pcseqlearning_amd.pcdet.utils import timer  # noqa: E402
from tools.run_sequence import build_model  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def load_sequence(seed, frames, points):
    """Synthetic sequence, cached under /tmp (generation is ~2 s of numpy ray casting per frame)."""
    import pickle
    path = f"/tmp/pcs_seq_s{seed}_f{frames}_p{points}.pkl"
    if os.path.exists(path):
        with open(path, "rb") as f:
            return pickle.load(f)
    batch = synthetic.gen_sequence(seed, frames, points)
    try:
        with open(path + f".{os.getpid()}", "wb") as f:
            pickle.dump(batch, f)
        os.replace(path + f".{os.getpid()}", path)
    except OSError:
        pass
    return batch


def baseline_metric():
    """BASELINE.json's metric string (the file sits next to this script)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "Waymo sequences/s (20-frame, 160k-pt) cluster-tracking, 1\u21928 MI355X + HBM-roofline %"


def cpu_baseline(batch, out, n_reg_calls):
    """Oracle timed on: 8 cm subsample of the whole sequence, one 10-frame proposal chunk at
    r = 0.25, and the 3 ICP levels of ONE anchor->next-frame pair; scaled by the number of such
    units the GPU run executed. Ground removal has no CPU oracle (its third-party leaves are
    unpinned) and is left out of the CPU estimate, which therefore flatters the CPU."""
    from oracle import oracle as O
    from oracle import reg_oracle as R
    from oracle import track_oracle as T
    t_all = time.perf_counter()
    fxyz = np.concatenate([batch["point_sweep"].astype(np.float32), batch["point_bxyz"][:, 1:]], 1)
    t0 = time.perf_counter()
    O.grid_sample(fxyz, [0.08, 0.08, 0.08])
    t_sub = time.perf_counter() - t0
    pts = out["point_fxyz"].cpu().numpy()
    frame = pts[:, 0].astype(np.int64)
    comp = out["point_component_rad0x25"].cpu().numpy()
    # one proposal chunk (10 frames) at the finest radius, 32-NN graph + components
    m = frame < 10
    t0 = time.perf_counter()
    e0, e1 = O.radius_graph(pts[m][:60000], pts[m][:60000], 0.25, 32, True)
    O.connected_components(e0, e1, min(int(m.sum()), 60000))
    t_prop = (time.perf_counter() - t0) * (m.sum() / min(int(m.sum()), 60000))
    # one frame pair, 3 levels
    diam = T.component_diameter(pts, comp, int(comp.max()) + 1)
    stat = diam[comp] > 12.5
    f0 = dict(fxyz=pts[frame == 0].copy(), component=comp[frame == 0] - comp[frame == 0].min(),
              stationary=stat[frame == 0], frame=frame[frame == 0])
    f1 = dict(fxyz=pts[frame == 1], component=comp[frame == 1], stationary=stat[frame == 1], frame=frame[frame == 1])
    C = int(f0["component"].max()) + 1
    t0 = time.perf_counter()
    for radius, vs in ((2.5, (0.4, 0.4, 0.6)), (1.25, (0.2, 0.2, 0.3)), (1.0, (0.1, 0.1, 0.15))):
        a, b = T.sample_frame(vs, f0), T.sample_frame(vs, f1)
        R.register_to_next_frame(R.Graph(radius), a, b, C, 10.0, 80, 0.05)
    t_pair = time.perf_counter() - t0
    n_pairs = n_reg_calls / 3.0
    est = t_sub + t_prop * 2 * 3 + t_pair * n_pairs  # 2 chunks x 3 radii (coarser radii cost more)
    return {"value": round(1.0 / est, 6), "unit": "sequences/s", "cores": 1, "kind": "port",
            "sample": (f"oracle timed {time.perf_counter() - t_all:.1f}s on: 8cm subsample ({t_sub:.1f}s), one "
                       f"10-frame r=0.25 proposal chunk ({t_prop:.1f}s, x6), one frame pair x 3 ICP levels "
                       f"({t_pair:.1f}s, x{n_pairs:.0f} pairs); ground removal excluded; scaled to one sequence")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--points", type=int, default=160000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")  # RCCL over xGMI on ROCm
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    host_batch = load_sequence(rank, args.frames, args.points)  # sequence s -> GPU s mod world
    batch = models.load_data_to_gpu(host_batch, device)
    model = build_model(save=False, device=device)
    lib = L.load()

    def step():
        with torch.no_grad():
            model(batch)

    for _ in range(args.warmup):
        step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    lib.pcs_profile_icp(ctypes.c_int(1))
    timer.RECORDS.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    prof = (ctypes.c_double * 9)()
    lib.pcs_profile_icp_read(prof)
    lib.pcs_profile_icp(ctypes.c_int(0))

    units_in_flight = int(getattr(model.preprocessors[2], "last_num_workers", 0))
    total_sequences = float(args.steps)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        n = torch.tensor([float(args.steps)], dtype=torch.float64, device=device)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        total_sequences = float(n.item())

    if rank == 0:
        out = model.last_outputs[0]
        # HIP events attached to the dispatch itself (hipExtLaunchKernelGGL): begin-to-end time of the
        # launch, the quantity rocprofv3 reports as the kernel duration
        ms, launches, nbytes = prof[0], prof[1], prof[2]
        achieved = (nbytes / (ms * 1e-3) / 1e9) if ms > 0 else 0.0
        # fabric-side bytes per launch from the committed PMC passes of this workload (counters cannot
        # be read live; see the note in the file for what FETCH_SIZE counts on gfx950)
        traffic, traffic_src = None, None
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_icp_nn_hbm_pmc_v10.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)
            traffic = round((pmc["FETCH_SIZE_KB_per_dispatch"] + pmc["WRITE_SIZE_KB_per_dispatch"]) * 1024.0, 1)
            traffic_src = "profiles/r01_icp_nn_hbm_pmc_v10.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, serial schedule)"
        line = {
            "metric": baseline_metric(),
            "value": round(total_sequences / dt, 5),
            "unit": "sequences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 points / f64 solve",
            "data": "synthetic",
            "config": {
                "workload": ("configs[2]: 20-frame x 160k-pt synthetic sequence, "
                             "cluster_tracking_TLS_multiradius_every8 end to end "
                             "(subsample + ground removal + 3-radius proposal + tracking)"),
                "frames": args.frames, "points_per_frame": args.points,
                "points_after_subsample": int(out["full_point_fxyz"].shape[0]),
                "points_after_ground_removal": int(out["point_fxyz"].shape[0]),
                "parallelism": f"one sequence per GPU x{world}",
                "tracking_units_in_flight": units_in_flight,
                # host wall spans (not device-synchronised; concurrent units overlap, so they do not add up)
                "stage_seconds_per_step": {k: round(v / args.steps, 3) for k, v in timer.RECORDS.items()
                                           if not k.startswith(("unit ", "Tracking Frame", "ground:"))},
            },
            "roofline": {
                "bound": "hbm", "kernel": "icp_nn_kernel (fused ICP: two-direction 1-NN search, 32 lanes per query)",
                "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                "launches": int(launches), "avg_us_per_launch": round(ms * 1e3 / max(launches, 1), 2),
                "algorithmic_bytes_per_launch": round(nbytes / max(launches, 1), 1),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            n_reg = model.preprocessors[2].num_registration_calls  # of the last step
            line["config"]["icp_level_calls_per_sequence"] = n_reg
            line["cpu_baseline"] = cpu_baseline(host_batch, out, n_reg)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
