"""bench.py -- end-to-end cluster-tracking throughput on synthetic Waymo-shaped sequences.

  python bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[2]: a 20-frame x 160k-pt sequence through the reference's
cluster_tracking_TLS_multiradius_every8 pipeline (8 cm subsample -> GroundPlaneRemover ->
ClusterProposal at 3 radii -> ClusterTracking: 3 keys x anchors {0, 8, 16} x +-8 frames x 3 ICP
levels); a step = one batch of `--batch` (default 12) independent sequences per GPU, all in flight at once,
inputs resident in HBM before the timed region, nothing written to disk. N GPUs process different sequences
(sequence s -> rank s mod N, weak scaling); the only collectives are the final reductions of the metric vector and of
every rank's step times over RCCL.

With N > 1 and no launcher in the environment, this script starts the N ranks itself
(`python -m torch.distributed.run --nproc-per-node N ...` as a CHILD process, before anything
touches the GPU); started by a launcher it checks WORLD_SIZE against --gpus. `--backend gloo
--dry-run` runs the same rank / sequence assignment and the same reduction on CPU tensors without
a GPU (tests/test_distributed.py).

Rank 0 prints ONE JSON line: whole-job sequences/s plus
  roofline      the dominant kernel(s) of the path, the fused ICP's iteration, timed per CHUNK: what a schedule
                launches for the iterations of all registrations an engine has in flight (wide schedule: 8
                iterations x {budget test + moments, searches, solve, apply}; resident schedules: one kernel).
                HIP events on the chunk's first and last dispatch inside the timed steps; achieved = the chunk's
                algorithmic bytes / its span; traffic from the PMC passes of the same command
                (profiles/r05_icp_hbm_pmc.json). `stages` lists the other stages (one extra, untimed,
                device-synchronised step over one sequence).
  cpu_baseline  the CPU oracle (C++ restatement, kind "port"), serial and OpenMP over the host's
                cores, timed on a bounded sample of the same sequence (subsample, ground removal, a proposal
                chunk, 18 registrations with their sampling) and scaled to sequences/s (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


# ---------------------------------------------------------------------------------- distribution
def sequences_of_rank(rank, world, steps):
    """Synthetic-sequence seeds a rank processes: sequence s -> rank s mod world (the reference's
    DistributedSampler striding, pcdet/datasets/__init__.py:61-64); one sequence per step."""
    return [rank + world * k for k in range(steps)]


def reduce_metrics(dist, device, seconds, values):
    """The job's metric vector: wall time = MAX over ranks, everything else = SUM over ranks.
    `values`: list of floats (sequences done, points in, ...). Returns (seconds, values)."""
    import torch
    if dist is None:
        return float(seconds), [float(v) for v in values]
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    v = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in v.tolist()]


def gather_rank_steps(dist, device, step_ms):
    """Every rank's step times on rank 0 (one all-gather of `steps` doubles): the reduced line then shows a
    straggler rank instead of hiding it behind the MAX of the wall time. Returns a list (per rank) of
    step_stats dictionaries; with one rank: that rank's."""
    import torch
    if dist is None:
        return [step_stats(step_ms)]
    t = torch.tensor(list(step_ms), dtype=torch.float64, device=device)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [step_stats([float(x) for x in p.tolist()]) for p in parts]


def spawn_ranks(args):
    """Start the N ranks as a child job and pass its output / exit code through. Nothing in this
    process has touched the GPU yet (no torch.cuda call), and the child is a fresh process: a
    process that initialised the GPU is never re-exec'ed."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------- workload
def load_sequence(seed, frames, points):
    """Synthetic sequence, cached under /tmp (generation is ~2 s of numpy ray casting per frame)."""
    import pickle
    from pcseqlearning_amd import synthetic
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


def load_sequences(seeds, frames, points, world=1):
    """The rank's sequences; the ones not cached yet are generated side by side in child processes
    (~40 s of numpy ray casting each; the /tmp cache is keyed by seed, so ranks never generate the same
    sequence twice and a later run finds them all). Called before this process touches the GPU. With
    `world` ranks on the host each rank takes its share of the cores (all ranks generate at once)."""
    missing = [s for s in seeds if not os.path.exists(f"/tmp/pcs_seq_s{s}_f{frames}_p{points}.pkl")]
    if len(missing) > 1:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        workers = max(1, min(len(missing), (os.cpu_count() or 2) // (2 * max(1, world))))
        with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as pool:
            list(pool.map(_generate_sequence, [(s, frames, points) for s in missing]))
    return [load_sequence(s, frames, points) for s in seeds]


def _generate_sequence(job):
    load_sequence(*job)
    return job[0]


def step_stats(ms):
    """min / median / p95 / max of the timed steps (rank 0's clock), so that a stalled step shows."""
    v = sorted(ms)
    if not v:
        return None
    pick = lambda q: v[min(len(v) - 1, int(round(q * (len(v) - 1))))]
    return {"min": round(v[0], 2), "median": round(pick(0.5), 2), "p95": round(pick(0.95), 2), "max": round(v[-1], 2)}


def baseline_metric():
    """BASELINE.json's metric string (the file sits next to this script)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "Waymo sequences/s (20-frame, 160k-pt) cluster-tracking, 1→8 MI355X + HBM-roofline %"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(batch, out, n_reg_calls):
    """The CPU oracle timed on a bounded sample of this sequence, once on ONE core and once with
    OpenMP over all cores: the 8 cm subsample of the whole sequence, ground removal on its first two
    frames (oracle/pcs_oracle_ground.cpp behind ground_plane_removal_canonical; scaled by points), one
    10-frame proposal chunk (32-NN radius graph + components at r = 0.25), and -- the bulk of the job --
    the registrations of anchor frame 8 against frames 7 and 9 for all three component keys and all
    three levels (18 of the sequence's ~315 register_to_next_frame calls) with the voxel sampling of
    both frames in front of each (sample_frame). Scaled by the call counts of the GPU run."""
    import numpy as np
    from oracle import ground_oracle as GO
    from oracle import oracle as O
    from oracle import reg_oracle as R
    from oracle import track_oracle as T
    from tools.run_sequence import ground_cfg
    t_all = time.perf_counter()
    fxyz = np.concatenate([batch["point_sweep"].astype(np.float32), batch["point_bxyz"][:, 1:]], 1)
    t0 = time.perf_counter()
    O.grid_sample(fxyz, [0.08, 0.08, 0.08])
    t_sub = time.perf_counter() - t0
    full = out["full_point_fxyz"].cpu().numpy().astype(np.float32)
    g_sel = full[:, 0] < 2  # the ground stage of two frames, scaled to the sequence by points
    t0 = time.perf_counter()
    GO.ground_plane_removal_canonical(np.ascontiguousarray(full[g_sel]), ground_cfg())
    t_ground = (time.perf_counter() - t0) * (full.shape[0] / max(int(g_sel.sum()), 1))
    pts = out["point_fxyz"].cpu().numpy()
    frame = pts[:, 0].astype(np.int64)
    m = frame < 10
    n_prop = min(int(m.sum()), 60000)
    t0 = time.perf_counter()
    e0, e1 = O.radius_graph(pts[m][:n_prop], pts[m][:n_prop], 0.25, 32, True)
    O.connected_components(e0, e1, n_prop)
    t_prop = (time.perf_counter() - t0) * (m.sum() / n_prop)
    levels = ((2.5, (0.4, 0.4, 0.6)), (1.25, (0.2, 0.2, 0.3)), (1.0, (0.1, 0.1, 0.15)))
    threads_all = R.cpu_threads()
    t_reg = {1: 0.0, threads_all: 0.0}
    t_smp = 0.0
    n_timed = 0
    for key in ("point_component_rad1x25", "point_component_rad0x75", "point_component_rad0x25"):
        comp = out[key].cpu().numpy()
        diam = T.component_diameter(pts, comp, int(comp.max()) + 1)
        stat = diam[comp] > 12.5
        fa = 8 if (frame == 8).any() else int(frame.min())
        base = comp[frame == fa].min()
        f0 = dict(fxyz=pts[frame == fa].copy(), component=comp[frame == fa] - base, stationary=stat[frame == fa],
                  frame=frame[frame == fa])
        C = int(f0["component"].max()) + 1
        for fb in (fa - 1, fa + 1):
            if not (frame == fb).any():
                continue
            f1 = dict(fxyz=pts[frame == fb], component=comp[frame == fb], stationary=stat[frame == fb],
                      frame=frame[frame == fb])
            for radius, vs in levels:
                t0 = time.perf_counter()
                a, b = T.sample_frame(vs, f0), T.sample_frame(vs, f1)
                t_smp += time.perf_counter() - t0
                for th in t_reg:
                    t0 = time.perf_counter()
                    R.register_to_next_frame_cpp(radius, a, b, C, 10.0, 80, 0.05, threads=th)
                    t_reg[th] += time.perf_counter() - t0
                n_timed += 1
    res = {}
    for th, t in t_reg.items():
        est = t_sub + t_ground + t_prop * 2 * 3 + (t + t_smp) / max(n_timed, 1) * n_reg_calls  # 2 chunks x 3 radii
        res[th] = 1.0 / est
    return {"value": round(res[threads_all], 6), "unit": "sequences/s", "cores": threads_all, "kind": "port",
            "value_1_core": round(res[1], 6), "host_cores": os.cpu_count(), "cpu_model": cpu_model(),
            "sample": (f"CPU oracle timed {time.perf_counter() - t_all:.1f}s: 8cm subsample ({t_sub:.2f}s), ground "
                       f"removal of frames 0-1 (scaled by points: {t_ground:.1f}s per sequence), one "
                       f"10-frame r=0.25 proposal chunk ({t_prop:.2f}s, x6), {n_timed} of the sequence's "
                       f"{n_reg_calls} registrations (anchor 8 <-> frames 7 / 9, 3 keys x 3 levels: "
                       f"{t_reg[1]:.2f}s on 1 core, {t_reg[threads_all]:.2f}s on {threads_all}; the voxel sampling "
                       f"of both frames in front of each: {t_smp:.2f}s); scaled to one sequence")}


def stage_rooflines(records, stats, prof, out):
    """Achieved algorithmic GB/s per stage from device-synchronised spans of ONE extra step and the
    sizes the stages logged (SURVEY.md 8d formulas). ICP kernels: HIP-event kernel time."""
    stages = []

    def add(name, seconds, nbytes, note):
        if seconds and seconds > 0 and nbytes > 0:
            gbs = nbytes / seconds / 1e9
            stages.append({"stage": name, "ms": round(seconds * 1e3, 3), "algorithmic_MB": round(nbytes / 1e6, 2),
                           "achieved_GBs": round(gbs, 2), "frac": round(gbs / HBM_PEAK_GBS, 5), "bytes": note})

    n, v = stats.get("subsample points in", 0), stats.get("subsample cells out", 0)
    add("subsample (8 cm grid pick)", records.get("Subsample (8 cm grid pick)"), 16.0 * n + 8.0 * n + 16.0 * v,
        "16 n + 8 n + 16 v")
    n_full = int(out["full_point_fxyz"].shape[0])
    add("ground removal", records.get("Ground Removal"), 50.0 * 16.0 * n_full, "50 x 16 n (IRLS rounds)")
    e, p = stats.get("proposal graph edges", 0), stats.get("proposal graph points", 0)
    ecc, nn = stats.get("proposal cc edges", 0), stats.get("proposal cc nodes", 0)
    t_graph = sum(val for key, val in records.items() if key.startswith("Propose Cluster: graph"))
    t_cc = sum(val for key, val in records.items() if key.startswith("Propose Cluster for"))
    if t_cc > 0:  # (SEPARATE_GRAPHS: one graph and one labelling per key)
        add("proposal radius graph (32-NN)", t_graph, 16.0 * p + 20.0 * e, "16 N + 16 M + 16 E + 4 E")
        add("proposal components", t_cc, 16.0 * ecc + 8.0 * nn, "16 E + 8 N")
    else:
        # one search at the largest radius + the components of its three cuts from the neighbour table, one span
        add("proposal radius graph (32-NN) + components of the three cuts", t_graph,
            16.0 * p + 20.0 * e + 16.0 * ecc + 8.0 * nn, "16 N + 16 M + 16 E + 4 E (graph) + 16 E + 8 N per cut")
    # the ICP chunks: HIP-event time of their launches; the phase rows (resident schedules only) are CU time (the
    # workgroups' own s_memrealtime stamps summed over the units), not wall time
    ms, launches, nbytes, unit_iters = prof[0], prof[1], prof[2], prof[3]
    if ms > 0:
        gbs = nbytes / (ms * 1e-3) / 1e9
        stages.append({"stage": "icp iterations (budget test + moments, searches, solve, apply), per chunk",
                       "ms": round(ms, 3), "launches": int(launches),
                       "avg_us_per_launch": round(ms * 1e3 / max(launches, 1), 2),
                       "unit_iterations": int(unit_iters), "algorithmic_MB": round(nbytes / 1e6, 2),
                       "achieved_GBs": round(gbs, 2), "frac": round(gbs / HBM_PEAK_GBS, 5),
                       "bytes": "76 (Mn + Rn) + 68 Mn + 688 C per unit and iteration"})
        phase_us = [prof[4 + q] for q in range(4)]
        if sum(phase_us) > 0 and unit_iters > 0:
            stages.append({"stage": "icp loop phases, CU-microseconds per unit-iteration",
                           "budget_test": round(phase_us[0] / unit_iters, 2), "searches": round(phase_us[1] / unit_iters, 2),
                           "solve": round(phase_us[2] / unit_iters, 2), "apply": round(phase_us[3] / unit_iters, 2)})
    return stages


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--points", type=int, default=160000)
    ap.add_argument("--batch", type=int, default=12,
                    help="sequences per step (one batch, as the reference's collate_batch hands them to "
                         "SimpleReg.forward, which tracks the sequences of a batch concurrently)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="sequences of a batch processed at once (SimpleReg's SEQUENCES_IN_FLIGHT; 0 = its default)")
    ap.add_argument("--stagger-ms", type=float, default=-1.0,
                    help="SimpleReg's SEQUENCE_START_STAGGER_MS (delay between the starts of the sequences of a batch); "
                         "< 0 = its default")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--dry-run", action="store_true", help="no GPU: rank / sequence assignment and reduction only")
    ap.add_argument("--schedule", default="wide", choices=("wide", "resident_tiles", "resident_cache"),
                    help="how the ICP iterations run (ops.icp_set_schedule): wide launches over all registrations in "
                         "flight (default), or one resident workgroup per registration with LDS-staged / hash-grid searches")
    ap.add_argument("--worker-streams", default="", choices=("", "own", "caller"),
                    help="SimpleReg's WORKER_STREAMS: every sequence worker on a stream of its own (default) or all on the caller's")
    ap.add_argument("--engines", type=int, default=-1, help="tracker engines per device (ops.track_configure; default 3)")
    ap.add_argument("--chunk-us", type=int, default=-1, help="chunk budget of the resident schedules (default 3000)")
    ap.add_argument("--pre-streams", type=int, default=0,
                    help="create this many HIP streams before the library creates any (robustness check: the engines' "
                         "hardware queues must not depend on what the embedding application created first)")
    ap.add_argument("--pre-slots", type=int, default=-1,
                    help="SimpleReg's PRE_TRACKING_CONCURRENCY: sequences of a batch in ground removal / proposal at once "
                         "(0 = no limit, the default; measured in round 5: 2 / 3 / 4 / 6 slots are 1-4 % slower)")
    ap.add_argument("--icp-widening", type=float, default=0.0,
                    help="pcs_icp_set_widening: searches look radius / this further (0 = the library's default, 16)")
    ap.add_argument("--extract-unit-by-unit", type=int, default=0,
                    help="1: ClusterTracking's EXTRACT_UNITS_TOGETHER off (the final filter and the extraction of traces unit "
                         "by unit: round 4's form; same results)")
    ap.add_argument("--chunk-iterations", type=int, default=0,
                    help="pcs_icp_set_chunk_iterations: iterations per chunk of wide launches (0 = the library's default, 10)")
    ap.add_argument("--trace", action="store_true", help="per-step memory lines and stack dumps of steps that hang")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true", help="skip the extra device-synchronised step")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)

    seeds = sequences_of_rank(rank, world, max(1, args.batch))  # every step re-tracks the rank's batch
    # (before this process initialises the GPU: the generator runs in child processes)
    host_batches = None if args.dry_run else load_sequences(seeds, args.frames, args.points, world)

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not args.dry_run and args.backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(args.backend)  # "nccl" is RCCL over xGMI on ROCm

    if args.dry_run:
        # the N > 1 logic without a GPU: a rank "tracks" for 10 ms x (1 + rank) per step
        device = torch.device("cpu")
        t0 = time.perf_counter()
        dry_steps = []
        for _ in range(args.steps):
            ts = time.perf_counter()
            time.sleep(0.01 * (1 + rank))
            dry_steps.append((time.perf_counter() - ts) * 1e3)
        dt = time.perf_counter() - t0
        dt, (nseq, seed_sum) = reduce_metrics(dist, device, dt, [float(args.steps), float(seeds[0])])
        per_rank = gather_rank_steps(dist, device, dry_steps)
        if rank == 0:
            print(json.dumps({"metric": baseline_metric(), "value": round(nseq / dt, 5), "unit": "sequences/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "data": "none (dry run)",
                              "per_rank_step_ms": per_rank,
                              "config": {"workload": "dry run: rank / sequence assignment and metric reduction",
                                         "sequence_seed_sum": seed_sum, "backend": args.backend}}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from pcseqlearning_amd import lib as L
    from pcseqlearning_amd.pcdet import models
    from pcseqlearning_amd.pcdet.utils import timer
    from tools.run_sequence import build_model
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    from pcseqlearning_amd import ops, synthetic
    # streams an embedding application might have created before the library sees the device
    pre_streams = [torch.cuda.Stream(device) for _ in range(args.pre_streams)]
    ops.track_configure(args.engines, args.chunk_us)
    if args.chunk_iterations > 0:
        L.check(L.load().pcs_icp_set_chunk_iterations(ctypes.c_int(args.chunk_iterations)), "pcs_icp_set_chunk_iterations")
    if args.icp_widening > 0:
        L.check(L.load().pcs_icp_set_widening(ctypes.c_double(args.icp_widening)), "pcs_icp_set_widening")
    ops.icp_set_schedule(*{"wide": (-1, True), "resident_tiles": (1, True), "resident_cache": (1, False)}[args.schedule])
    host_batch = host_batches[0]
    batch1 = models.load_data_to_gpu(host_batch, device)  # the first sequence alone (latency / stage report)
    batch = batch1 if len(host_batches) == 1 else \
        models.load_data_to_gpu(synthetic.collate_sequences(host_batches), device)
    n_points_in = int(batch["point_bxyz"].shape[0])
    model = build_model(save=False, device=device)
    if args.inflight > 0:
        model.model_cfg["SEQUENCES_IN_FLIGHT"] = args.inflight
    if args.worker_streams:
        model.model_cfg["WORKER_STREAMS"] = args.worker_streams
    if args.stagger_ms >= 0:
        model.model_cfg["SEQUENCE_START_STAGGER_MS"] = args.stagger_ms
    if args.pre_slots >= 0:
        model.model_cfg["PRE_TRACKING_CONCURRENCY"] = args.pre_slots
    if args.extract_unit_by_unit:
        model.preprocessors[2].model_cfg["EXTRACT_UNITS_TOGETHER"] = False
    lib = L.load()

    def step():
        with torch.no_grad():
            model(batch)

    if args.warmup > 0 and batch is not batch1:
        # the first sequence alone, on one host thread, before the first batch: every code object the path
        # uses is loaded by one thread (twelve threads taking first launches of the same torch / rocPRIM
        # kernels at once have crashed rocprofv3's tool library twice; a plain run does not need this)
        with torch.no_grad():
            model(batch1)
    for _ in range(args.warmup):
        step()
    # one sequence alone through the same model (its 9 units in the engines, nothing beside it): its latency, and
    # the outputs / tracker counters the report below quotes (the batch's worker copies do not survive a step).
    # BEFORE the timed steps: a kernel trace of this command then ends with the timed steps and nothing else
    # (tools/trace_summary.py looks at the second half of a trace)
    with torch.no_grad():
        model(batch1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    with torch.no_grad():
        model(batch1)
    torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t1) * 1e3
    out = model.last_outputs[0]
    tracker = model.preprocessors[2]
    n_reg_calls = int(tracker.num_registration_calls)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    lib.pcs_profile_icp(ctypes.c_int(1))
    timer.RECORDS.clear()
    barrier()
    t0 = time.perf_counter()
    trace = args.trace
    step_ms = []
    alloc0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    from pcseqlearning_amd.pcdet.models.registration.preprocessors import cluster_tracking as CT
    fallbacks0 = CT.HOST_FALLBACKS
    for i in range(args.steps):
        spans0 = dict(timer.RECORDS)
        ts = time.perf_counter()
        if trace:  # a step that hangs shows where: every thread's Python stack after 5 s, then every 5 s
            import faulthandler
            faulthandler.dump_traceback_later(5, repeat=True, file=sys.stderr)
        step()
        torch.cuda.current_stream().synchronize()  # (a step's outputs are complete when forward returns; this only orders the clock)
        step_ms.append((time.perf_counter() - ts) * 1e3)
        if len(step_ms) > 3 and step_ms[-1] > 1.8 * sorted(step_ms)[len(step_ms) // 2]:
            # a step far above the median: which stage's host spans (summed over the batch's sequences) it went into
            d = {k: round(v - spans0.get(k, 0.0), 3) for k, v in timer.RECORDS.items()
                 if v - spans0.get(k, 0.0) > 0.05 and not k.startswith(("Tracking Frame", "ground:"))}
            print(f"slow step {i}: {step_ms[-1]:.0f} ms (median {sorted(step_ms)[len(step_ms) // 2]:.0f}); host spans of its stages, "
                  f"summed over the batch (s): {d}", file=sys.stderr, flush=True)
        if trace:
            faulthandler.cancel_dump_traceback_later()
            ms_ = torch.cuda.memory_stats()
            print(f"step {i}: {(time.perf_counter() - ts) * 1e3:.1f} ms reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB"
                  f" alloc_retries {ms_.get('num_alloc_retries', 0)} device_alloc {ms_.get('num_device_alloc', 0)}"
                  f" device_free {ms_.get('num_device_free', 0)}", file=sys.stderr, flush=True)
    barrier()
    dt = time.perf_counter() - t0
    host_fallbacks = CT.HOST_FALLBACKS - fallbacks0
    device_allocs = torch.cuda.memory_stats().get("num_device_alloc", 0) - alloc0
    prof = (ctypes.c_double * 12)()
    lib.pcs_profile_icp_read_ex(prof, ctypes.c_int(12))
    lib.pcs_profile_icp(ctypes.c_int(0))
    B = len(host_batches)
    host_spans = {k: round(v / (args.steps * B), 4) for k, v in timer.RECORDS.items()
                  if not k.startswith(("Tracking Frame", "ground:"))}
    dt, (total_sequences, points_in) = reduce_metrics(
        dist, device, dt, [float(args.steps * B), float(args.steps) * float(n_points_in)])
    per_rank_steps = gather_rank_steps(dist, device, step_ms)

    if rank == 0:
        # one extra, untimed step with device-synchronised stage spans for the per-stage report
        stages = []
        if not args.no_stages:
            timer.RECORDS.clear()
            timer.STATS.clear()
            timer.SYNC = True
            with torch.no_grad():
                model(batch1)
            torch.cuda.synchronize()
            timer.SYNC = False
            stages = stage_rooflines(dict(timer.RECORDS), dict(timer.STATS), prof, out)
        # the dominant kernel(s): the fused ICP's iterations, chunk by chunk (the launches of the schedule in use).
        # HIP events attached to the dispatches themselves (hipExtLaunchKernelGGL): start of a chunk's first
        # launch to the stop of its last, the span a kernel trace shows for the same dispatches.
        ms, launches, nbytes, unit_iters = prof[0], prof[1], prof[2], prof[3]
        # `achieved` / `frac` price the chunk at SURVEY 8(d)'s algorithmic bytes (112 (Mn + Rn) + 144 C_live per unit and
        # iteration); `frac_impl` at this implementation's own traffic model (76 (Mn + Rn) + 68 Mn + 688 C: 20-word
        # fixed-point moment records and double-buffered f64 transforms per component ID, live or not)
        bytes_8d, many_iters, many_8d = prof[9], prof[10], prof[11]
        achieved = (bytes_8d / (ms * 1e-3) / 1e9) if ms > 0 else 0.0
        achieved_impl = (nbytes / (ms * 1e-3) / 1e9) if ms > 0 else 0.0
        # HBM traffic per launch from the PMC passes of THIS schedule (tools/pmc_batch.sh ->
        # profiles/r05_icp_hbm_pmc.json, written by tools/derive_icp_traffic.py); null when no such file exists
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "r05_icp_hbm_pmc.json")
        on_8d = os.path.exists(pmc_path)  # (round 5's file: the ratio is taken against SURVEY 8(d)'s bytes)
        if not on_8d:
            pmc_path = os.path.join(ROOT, "profiles", "r04_icp_hbm_pmc.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)
            ratio = pmc.get("icp_chunk_traffic_over_algorithmic")
            if ratio and launches > 0:
                traffic = round(ratio * (bytes_8d if on_8d else nbytes) / launches, 1)
            traffic_src = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (same schedule), "
                           "(fetch + write) / %s = %s (raw FETCH_SIZE; %s with the guide's x2 correction for gfx950), times "
                           "this run's %s" % (os.path.basename(pmc_path),
                                              "the algorithmic bytes (SURVEY 8(d))" if on_8d else "the implementation's bytes",
                                              ratio, pmc.get("icp_chunk_traffic_over_algorithmic_fetch_x2"),
                                              "algorithmic_bytes_per_launch" if on_8d else "impl_bytes_per_launch"))
        line = {
            "metric": baseline_metric(),
            "value": round(total_sequences / dt, 5),
            "unit": "sequences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "step_ms": step_stats(step_ms),
            "per_rank_step_ms": per_rank_steps,
            "host_fallbacks": int(host_fallbacks),
            "device_allocations_in_timed_steps": int(device_allocs),
            "reserved_GiB": round(torch.cuda.memory_reserved() / 2**30, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 points / f64 solve",
            "data": "synthetic",
            "config": {
                "workload": ("configs[2]: 20-frame x 160k-pt synthetic sequences, "
                             "cluster_tracking_TLS_multiradius_every8 end to end "
                             "(subsample + ground removal + 3-radius proposal + tracking); a step is one "
                             f"batch of {B} independent sequences through SimpleReg.forward, which processes "
                             "the sequences of a batch concurrently (one host thread and stream per sequence "
                             "for the ground / proposal / extraction stages; the tracking units of all "
                             "sequences in flight share the launches of the device's three lock-step engines)" if B > 1 else
                             "configs[2]: one 20-frame x 160k-pt synthetic sequence per step, "
                             "cluster_tracking_TLS_multiradius_every8 end to end"),
                "batch_size": B,
                "sequences_in_flight": min(B, int(model.model_cfg.get("SEQUENCES_IN_FLIGHT", 12))),
                "pre_created_streams": len(pre_streams),
                "one_sequence_alone_ms": round(single_ms, 2),
                "one_sequence_alone_sequences_per_s": round(1e3 / single_ms, 4),
                "frames": args.frames, "points_per_frame": args.points,
                "points_in_per_second": round(points_in / dt, 1),
                "points_after_subsample": int(out["full_point_fxyz"].shape[0]),
                "points_after_ground_removal": int(out["point_fxyz"].shape[0]),
                "parallelism": (f"sequences sharded over {world} GPU(s), sequence s -> rank s mod {world}; per GPU a "
                                f"batch of {B} per step"),
                "tracking_units_per_sequence": int(getattr(tracker, "last_num_workers", 0)),
                "tracker_host_ticks": int(getattr(tracker, "last_ticks", 0)),
                "icp_level_calls_per_sequence": n_reg_calls,
                # host wall spans of the timed steps per sequence (not device-synchronised; the sequences of a
                # batch overlap, so the spans add up to more than the step)
                "stage_seconds_per_sequence": host_spans,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("the iteration launches of the fused ICP, chunk by chunk: icp_nn_kernel (verified-cache budget "
                           "test, moments) + icp_nn_search_kernel + icp_solve_kernel + icp_apply_kernel, "
                           f"{args.chunk_iterations or 10} iterations of every registration in flight per chunk"
                           if args.schedule == "wide" else
                           "icp_tile_kernel / icp_loop_kernel (fused ICP, one 1024-thread workgroup per tracking unit: "
                           "budget test, searches, moments, solve, stop rule, apply -- all iterations of a chunk)"),
                "schedule": args.schedule, "iterations_per_chunk": args.chunk_iterations or 10,
                "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                "bytes": "SURVEY 8(d): 112 (Mn + Rn) + 144 C_live per unit and iteration",
                "frac_impl": round(achieved_impl / HBM_PEAK_GBS, 6),
                "bytes_impl": "76 (Mn + Rn) + 68 Mn + 688 C per unit and iteration (this implementation's own records)",
                # the same bytes over the whole timed window: what the ICP iterations of ALL engines move per second of
                # the step (the chunks of the three engines overlap, so this is not the per-chunk figure times three)
                "chip_level": {"achieved_GBs": round(bytes_8d / dt / 1e9, 2) if dt > 0 else 0.0,
                               "frac": round(bytes_8d / dt / 1e9 / HBM_PEAK_GBS, 6) if dt > 0 else 0.0,
                               "note": "8(d) bytes of every chunk of the timed steps / the steps' wall time"},
                # by component count (chunks mix the keys, so only bytes and iterations split, not time): the 0.25 m
                # key numbers ~13 k components of one or two points (C_live counts those with a moving point); the others ~1 k
                "by_key": {"more_than_5000_component_ids": {"unit_iterations": int(many_iters),
                                                         "MB_per_unit_iteration": round(many_8d / max(many_iters, 1) / 1e6, 3)},
                           "others": {"unit_iterations": int(unit_iters - many_iters),
                                      "MB_per_unit_iteration": round((bytes_8d - many_8d) / max(unit_iters - many_iters, 1) / 1e6, 3)}},
                "launches_timed": int(launches), "avg_us_per_launch": round(ms * 1e3 / max(launches, 1), 2),
                "timing": ("HIP events on the dispatches themselves, inside the timed steps: start of a chunk's first "
                           "iteration launch to the stop of its last, every chunk in which at least one unit iterated; "
                           "a 'launch' below is one chunk"),
                "algorithmic_bytes_per_launch": round(bytes_8d / max(launches, 1), 1),
                "impl_bytes_per_launch": round(nbytes / max(launches, 1), 1),
                "unit_iterations_timed": int(unit_iters),
                "unit_iterations_per_launch": round(unit_iters / max(launches, 1), 1),
                "stages": stages,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host_batch, out, n_reg_calls)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
