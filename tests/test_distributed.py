"""The N>1 logic of bench.py (it launches the ranks itself; sequence s -> rank s mod world, no
data-path collective, final all-reduce of the metric vector) driven through bench.py on a
world-size-2 gloo group on CPU, and the gradient all-reduce of the backbone's training path."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import json


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError(f"no JSON line in: {text!r}")


def test_bench_launches_two_ranks_and_reduces():
    """`python bench.py --gpus 2 --backend gloo --dry-run`: bench.py itself starts the two ranks
    (torch.distributed.run as a child process), assigns sequence s to rank s mod 2 and reduces the
    metric vector (MAX of the wall time, SUM of the rest) -- the code the GPU run uses, on CPU."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--dry-run", "--steps", "3", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    res = _last_json(p.stdout)
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["scaling"] == "weak"
    assert res["config"]["sequence_seed_sum"] == 1.0       # rank 0 tracks sequence 0, rank 1 sequence 1
    # 6 sequences in the time of the slower rank (rank 1 sleeps 20 ms per step)
    assert res["ms_per_step"] >= 19.0
    assert abs(res["value"] - 6.0 / (res["ms_per_step"] * 3e-3)) / res["value"] < 1e-3
    # every rank's own step times travel with the reduced line: a straggler shows (rank 1 sleeps twice as long)
    pr = res["per_rank_step_ms"]
    assert len(pr) == 2 and pr[0]["median"] >= 9.0 and pr[1]["median"] >= 19.0 and pr[1]["median"] > pr[0]["median"] * 1.5


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """Started by a launcher with WORLD_SIZE != --gpus the script fails fast (it used to run one rank
    and print n_gpus: 1)."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_sequence_assignment_and_single_rank_reduction():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.sequences_of_rank(1, 4, 3) == [1, 5, 9]
    owned = sorted(s for r in range(4) for s in bench.sequences_of_rank(r, 4, 2))
    assert owned == list(range(8))                         # every sequence exactly once
    assert bench.reduce_metrics(None, None, 0.5, [3, 7]) == (0.5, [3.0, 7.0])


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
