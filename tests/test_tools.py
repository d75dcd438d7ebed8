"""The measurement tools run (CPU only): the trace / counter summarisers on tiny synthetic inputs, the
shell recipes parse, bench.py's defaults are the documented ones."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **kw):
    return subprocess.run([sys.executable, *args], cwd=ROOT, capture_output=True, text=True, timeout=120, **kw)


def test_trace_concurrency_on_a_synthetic_trace(tmp_path):
    rows = ['"Kind","Agent_Id","Queue_Id","Kernel_Name","Start_Timestamp","End_Timestamp","Grid_Size","Workgroup_Size"']
    t = 1000
    for i in range(400):  # two queues, kernels of 10 us, half overlapping
        for q, off in ((1, 0), (2, 5000)):
            name = ("pcs::icp_nn_kernel(a)", "pcs::icp_nn_search_kernel(a)", "pcs::icp_apply_kernel(a)")[i % 3]
            rows.append(f'"KERNEL_DISPATCH","Agent 2",{q},"{name}",{t + off},{t + off + 10000},256,256')
        t += 12000
    path = tmp_path / "trace.csv"
    path.write_text("\n".join(rows) + "\n")
    r = _run(["tools/trace_concurrency.py", str(path)])
    assert r.returncode == 0, r.stderr
    assert "mean kernels running" in r.stdout and "queue 1" in r.stdout
    mean = float(r.stdout.split("mean kernels running:")[1].split()[0])
    assert 1.0 < mean < 2.0


def test_summarize_pmc_on_a_synthetic_csv(tmp_path):
    rows = ['"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"']
    for d in range(4):
        rows.append(f'{d},"pcs::icp_nn_kernel(x)","FETCH_SIZE",{1000 + d},{d * 100},{d * 100 + 50}')
    path = tmp_path / "pmc.csv"
    path.write_text("\n".join(rows) + "\n")
    r = _run(["tools/summarize_pmc.py", str(path)])
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["pcs::icp_nn_kernel"]["dispatches"] == 4
    assert abs(out["pcs::icp_nn_kernel"]["counters_per_dispatch"]["FETCH_SIZE"] - 1001.5) < 1e-6


def test_shell_recipes_parse():
    for name in ("profile_sequence.sh", "profile_batch.sh", "pmc_pass.sh", "pmc_batch.sh", "ab_bench.sh"):
        r = subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", name)], capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stderr)


def test_bench_defaults():
    r = _run(["bench.py", "--help"])
    assert r.returncode == 0, r.stderr
    text = " ".join(r.stdout.split())
    assert "--batch" in text and "--inflight" in text and "--gpus" in text
    assert "--schedule" in text and "--pre-streams" in text and "--engines" in text
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'add_argument("--batch", type=int, default=12' in src
    assert 'add_argument("--steps", type=int, default=3' in src
