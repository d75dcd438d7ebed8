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
