"""The N>1 logic of bench.py (sequence s -> rank s mod world, no data-path collective, final
all-reduce of {max wall time, sequences done}) on a world-size-2 gloo group on CPU."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json, time
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    # every rank owns the sequences s with s % world == rank (DistributedSampler striding)
    mine = [s for s in range(5) if s % world == rank]
    t0 = time.perf_counter(); time.sleep(0.05 * (rank + 1)); dt = time.perf_counter() - t0
    dist.barrier()
    t = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([float(len(mine))], dtype=torch.float64); dist.all_reduce(n, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"max_dt": float(t), "own_dt": dt, "total": float(n), "mine": mine}))
    dist.destroy_process_group()
""")


def test_two_rank_gloo_reduction(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    import json
    res = json.loads(outs[0].strip().splitlines()[-1])
    assert res["total"] == 5.0 and res["mine"] == [0, 2, 4]
    assert res["max_dt"] >= res["own_dt"]  # rank 1 sleeps longer: MAX over ranks is reported


GRAD_WORKER = textwrap.dedent("""
    import os, sys, json
    import torch, torch.distributed as dist
    sys.path.insert(0, os.environ["PCS_ROOT"])
    from pcseqlearning_amd.pcdet.utils.dist_utils import allreduce_gradients
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.manual_seed(0)                      # same parameters on every rank
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    for p in net[3].parameters():
        p.requires_grad_(True)
    g = torch.Generator().manual_seed(100 + rank)   # different data per rank
    x = torch.randn(8, 6, generator=g)
    net[2](net[1](net[0](x))).pow(2).sum().backward()   # the last layer gets no gradient on any rank
    own = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    nb = allreduce_gradients(list(net.parameters()), bucket_bytes=64)   # tiny buckets: several of them
    # reference: gather every rank's own gradients and average
    ok = True
    for p, o in zip(net.parameters(), own):
        o = torch.zeros_like(p) if o is None else o
        parts = [torch.zeros_like(o) for _ in range(world)]
        dist.all_gather(parts, o)
        ok = ok and torch.allclose(p.grad, sum(parts) / world, atol=1e-6)
    if rank == 0:
        print(json.dumps({"ok": bool(ok), "buckets": nb}))
    dist.destroy_process_group()
""")


def test_two_rank_gradient_allreduce(tmp_path):
    """pcdet/utils/dist_utils.allreduce_gradients (bucketed, async) on a world-size-2 gloo group."""
    script = tmp_path / "grad_worker.py"
    script.write_text(GRAD_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PCS_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    import json
    res = json.loads(outs[0].strip().splitlines()[-1])
    assert res["ok"] and res["buckets"] >= 3
