"""The measurement tools run (CPU only): the trace / counter summarisers on tiny synthetic inputs, the
shell recipes parse, bench.py's defaults are the documented ones."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **kw):
    return subprocess.run([sys.executable, *args], cwd=ROOT, capture_output=True, text=True, timeout=120, **kw)


def test_trace_summary_on_a_synthetic_trace(tmp_path):
    """tools/trace_summary.py: per-kernel table, queue busy time, concurrency, and the ICP chunk spans (what bench.py's
    roofline row times): two queues, each running chunks of 2 iterations x 4 kernels closed by icp_decide_kernel."""
    rows = ['"Kind","Agent_Id","Queue_Id","Kernel_Name","Start_Timestamp","End_Timestamp","Grid_Size","Workgroup_Size",'
            '"Grid_Size_Y","Workgroup_Size_Y"']
    names = ("pcs::icp_nn_kernel(a)", "pcs::icp_nn_search_kernel(a)", "pcs::icp_solve_kernel(a)", "pcs::icp_apply_kernel(a)")
    for q, off in ((1, 0), (2, 5000)):
        t = 1000 + off
        for chunk in range(200):
            for it in range(2):
                for n in names:
                    rows.append(f'"KERNEL_DISPATCH","Agent 2",{q},"{n}",{t},{t + 10000},2560,256,10,1')
                    t += 12000
            rows.append(f'"KERNEL_DISPATCH","Agent 2",{q},"pcs::icp_decide_kernel(a)",{t},{t + 2000},640,64,1,1')
            t += 5000
    path = tmp_path / "trace.csv"
    path.write_text("\n".join(rows) + "\n")
    r = _run(["tools/trace_summary.py", str(path)])
    assert r.returncode == 0, r.stderr
    assert "mean kernels running" in r.stdout and "queue 1" in r.stdout
    mean = float(r.stdout.split("mean kernels running:")[1].split()[0])
    assert 1.0 < mean < 2.0
    # a chunk = 8 kernels of 10 us, 12 us apart: first start to last end = 7 * 12 + 10 = 94 us, 10 entries
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("ICP chunks")][0]
    span = float(line.split("mean span")[1].split()[0])  # (the summary looks at the second half of the trace: its first chunk is cut)
    assert 93.0 <= span <= 94.0 and "mean entries 10.0" in line, line
    # gaps between dependent launches: 2 us (kernels 12 us apart, 10 us long) and 3 us behind the decide kernel
    gaps = [ln for ln in r.stdout.splitlines() if ln.startswith("engine queues, idle time")][0]
    assert "0-5:" in gaps and "1000-inf: 0 gaps" in gaps, gaps
    assert "by kernel: " in r.stdout
    # a window of the last 100 ms instead of the second half (what profile_batch.sh passes: the timed steps)
    r2 = _run(["tools/trace_summary.py", str(path), "10", "1.0"])
    assert r2.returncode == 0 and r2.stdout.startswith("window 1.0 ms"), r2.stdout[:200] + r2.stderr


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
    for name in ("profile_sequence.sh", "profile_batch.sh", "pmc_pass.sh", "pmc_batch.sh", "ab_bench.sh", "ab_multi.sh",
                 "ab_trees.sh"):
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
