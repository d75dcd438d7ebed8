"""Per-kernel and per-queue summary of a rocprofv3 --kernel-trace run (the second half of the trace, or its last
`last_ms` milliseconds: bench.py's timed steps are the last thing a `--no-stages --no-cpu-baseline` run does, so
last_ms = steps x ms_per_step of the line it printed selects exactly what the bench line timed):
  python tools/trace_summary.py <..._kernel_trace.csv> [top] [last_ms]
Per kernel: launches, total ms, mean / median us, mean grid size (workgroups), and the share of the
window; per queue: launches, busy ms; overall: time-weighted mean number of kernels running. For the
three ICP iteration kernels: mean duration by number of batch entries (gridDim.y where it is the
entry count) and by how many other kernels ran beside them; the spans of the ICP chunks (what bench.py's
roofline row times, so that its frac can be recomputed from the trace alone)."""
import bisect
import collections
import csv
import sys


def main():
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        wg = int(r.get("Workgroup_Size", 0) or 0)
        gs = int(r.get("Grid_Size", 0) or 0)
        gy = int(r.get("Grid_Size_Y", 0) or 0)
        wy = int(r.get("Workgroup_Size_Y", 1) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0],
                     r.get("Queue_Id", "?"), gs // max(wg, 1), gy // max(wy, 1)))
    rows.sort()
    t0 = rows[0][0]
    t_end = max(r[1] for r in rows)
    tmid = t0 + (t_end - t0) // 2
    if len(sys.argv) > 3 and float(sys.argv[3]) > 0:
        tmid = max(t0, t_end - int(float(sys.argv[3]) * 1e6))
    rows = [r for r in rows if r[0] >= tmid]
    span = max(r[1] for r in rows) - rows[0][0]
    print(f"window {span / 1e6:.1f} ms, {len(rows)} dispatches")
    per_k = collections.defaultdict(list)
    per_q = collections.defaultdict(lambda: [0, 0])
    grid = collections.defaultdict(int)
    for s, e, n, q, g, gy in rows:
        per_k[n].append(e - s)
        grid[n] += g
        per_q[q][0] += 1
        per_q[q][1] += e - s
    tot = sum(sum(v) for v in per_k.values())
    print(f"sum of kernel durations {tot / 1e6:.1f} ms = {tot / span:.2f} x the window")
    print(f"{'kernel':58s} {'calls':>8s} {'total ms':>9s} {'mean us':>8s} {'med us':>7s} {'mean WGs':>9s} {'% sum':>6s}")
    for n, v in sorted(per_k.items(), key=lambda kv: -sum(kv[1]))[:top]:
        v2 = sorted(v)
        print(f"{n[-58:]:58s} {len(v):8d} {sum(v) / 1e6:9.2f} {sum(v) / len(v) / 1e3:8.1f} {v2[len(v2) // 2] / 1e3:7.1f} "
              f"{grid[n] / len(v):9.0f} {100 * sum(v) / tot:6.1f}")
    icp_q = collections.Counter(q for s_, e_, n, q, g, gy in rows if "icp_apply_kernel" in n)
    for q, (c, b) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
        print(f"  queue {q}: {c} dispatches, busy {b / 1e6:.1f} ms ({100 * b / span:.1f} % of the window), "
              f"{icp_q.get(q, 0)} ICP iterations (the engines' chains)")
    by_qk = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
    for s_, e_, n, q, g, gy in rows:
        by_qk[q][n][0] += 1
        by_qk[q][n][1] += e_ - s_
    for q in sorted(by_qk):  # whose kernels a queue runs (another stream's long kernel on an engine's queue delays its chain)
        topk = sorted(by_qk[q].items(), key=lambda kv: -kv[1][1])[:8]
        print(f"  queue {q} by kernel: " + "; ".join(f"{n.split('::')[-1][:28]} {c} x {t / c / 1e3:.0f} us" for n, (c, t) in topk))
    ev = []
    for s, e, *_ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    level, last, hist = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        hist[level] += t - last
        last = t
        level += d
    tt = sum(hist.values())
    print("time at concurrency level (%):", {k: round(100 * v / tt, 1) for k, v in sorted(hist.items())})
    print("mean kernels running:", round(sum(k * v for k, v in hist.items()) / tt, 2))
    starts = sorted((s, e) for s, e, *_ in rows)
    ss = [s for s, _ in starts]
    for name in ("pcs::icp_nn_kernel", "pcs::icp_nn_search_kernel", "pcs::icp_apply_kernel", "pcs::smp_sort_kernel"):
        by_n, by_c = collections.defaultdict(list), collections.defaultdict(list)
        for s, e, n, q, g, gy in rows:
            if not (n == name or n.endswith(name) or (name + "<") in n):  # (template instances: "void pcs::k<false>")
                continue
            by_n[min(gy // 16 * 16, 128)].append(e - s)
            mid = (s + e) // 2
            hi = bisect.bisect_right(ss, mid)
            lo = max(0, hi - 64)
            c = sum(1 for a, b in starts[lo:hi] if b > mid) - 1
            by_c[min(c, 8)].append(e - s)
        print(name, "by entries (gridDim.y, bins of 16):", {k: (len(v), round(sum(v) / len(v) / 1e3, 1)) for k, v in sorted(by_n.items())})
        print(name, "by kernels beside it:", {k: (len(v), round(sum(v) / len(v) / 1e3, 1)) for k, v in sorted(by_c.items())})
    chunk_spans(rows)
    first_iteration_searches(rows)
    queue_gaps(rows)
    tick_composition(rows)


def first_iteration_searches(rows):
    """The search launches split by what they serve: the launch that follows an icp_first_kernel on its queue is the
    full search of the registrations that begin in that chunk (every query of theirs), the others serve the queries
    whose cached neighbour failed the budget test."""
    by_q = collections.defaultdict(list)
    for s, e, n, q, g, gy in rows:
        if "icp_first_kernel" in n or "icp_nn_search_kernel" in n:
            by_q[q].append((s, e, n))
    cls = {"after icp_first": [], "others": []}
    for q, v in by_q.items():
        armed = False
        for s, e, n in v:
            if "icp_first_kernel" in n:
                armed = True
            else:
                cls["after icp_first" if armed else "others"].append(e - s)
                armed = False
    for k, v in cls.items():
        if v:
            v2 = sorted(v)
            print(f"icp_nn_search_kernel, {k}: {len(v)} launches, total {sum(v) / 1e6:.1f} ms, mean {sum(v) / len(v) / 1e3:.1f} us, "
                  f"median {v2[len(v2) // 2] / 1e3:.1f} us")


def queue_gaps(rows):
    """Idle time of the engines' queues by the length of the gap between two dispatches: short gaps are the cost of
    one dependent launch after another, long ones the host between two ticks."""
    icp_q = set(q for s, e, n, q, g, gy in rows if "icp_apply_kernel" in n)
    by_q = collections.defaultdict(list)
    for s, e, n, q, g, gy in rows:
        if q in icp_q:
            by_q[q].append((s, e, n))
    bins = [(0, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1000), (1000, 10 ** 9)]
    tot = collections.Counter()
    cnt = collections.Counter()
    around = collections.Counter()   # gaps of 20-1000 us: (kernel before, kernel after) -> idle ns
    around_n = collections.Counter()
    span = 0
    short = lambda n: n.split("(")[0].split("::")[-1][:34]  # noqa: E731
    for q, v in by_q.items():
        v.sort()
        span += v[-1][1] - v[0][0]
        end, last = v[0][1], v[0][2]
        for s, e, n in v[1:]:
            gap = s - end
            if gap > 0:
                for lo, hi in bins:
                    if lo * 1000 <= gap < hi * 1000:
                        tot[(lo, hi)] += gap
                        cnt[(lo, hi)] += 1
                        break
                if 20000 <= gap < 1000000:
                    around[(short(last), short(n))] += gap
                    around_n[(short(last), short(n))] += 1
            if e >= end:
                end, last = e, n
    if span:
        print("engine queues, idle time by gap length (us): " + ", ".join(
            f"{lo}-{hi if hi < 10 ** 9 else 'inf'}: {cnt[(lo, hi)]} gaps = {100 * tot[(lo, hi)] / span:.1f} %" for lo, hi in bins))
        print("engine queues, gaps of 20-1000 us by the kernels around them (before -> after: gaps, mean us, % of the queues' time):")
        for (a, b), t in around.most_common(14):
            print(f"    {a:34s} -> {b:34s} {around_n[(a, b)]:6d} x {t / around_n[(a, b)] / 1e3:7.1f} us = {100.0 * t / span:5.2f} %")


def tick_composition(rows):
    """What an engine's tick is made of: a tick = the dispatches of an engine's queue from one icp_decide_kernel to the
    next (other streams' kernels that the runtime put on the same hardware queue included). Per kernel: launches per
    tick, their time per tick; then the tick's span, busy time and idle time."""
    icp_q = set(q for s, e, n, q, g, gy in rows if "icp_decide_kernel" in n)
    per_k = collections.defaultdict(lambda: [0, 0])
    ticks = spans = busy = disp = 0
    for q in icp_q:
        v = sorted((s, e, n) for s, e, n, qq, g, gy in rows if qq == q)
        start = None
        for s, e, n in v:
            if start is not None:
                per_k[n][0] += 1
                per_k[n][1] += e - s
                busy += e - s
                disp += 1
            if "icp_decide_kernel" in n:
                if start is not None:
                    ticks += 1
                    spans += e - start
                start = e
    if not ticks:
        return
    print(f"engine ticks: {ticks}, mean span {spans / ticks / 1e3:.0f} us, busy {busy / ticks / 1e3:.0f} us, "
          f"{disp / ticks:.1f} dispatches per tick; per tick and kernel (launches x mean us = us per tick):")
    for n, (c, t) in sorted(per_k.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"    {n.split('::')[-1][:44]:44s} {c / ticks:6.2f} x {t / c / 1e3:7.1f} = {t / ticks / 1e3:7.1f}")


def chunk_spans(rows):
    """The quantity bench.py's roofline row times with HIP events, recomputed from the trace: a CHUNK of the fused
    ICP = the iteration launches between two icp_decide_kernel dispatches of one queue; its span = start of its
    first icp_nn_kernel to the end of its last icp_apply_kernel (resident schedules: start of icp_tile_kernel /
    icp_loop_kernel to the end of icp_loop_kernel). Entries = gridDim.y of the budget-test launches."""
    by_q = collections.defaultdict(list)
    for s, e, n, q, g, gy in rows:
        if "icp_" in n:
            by_q[q].append((s, e, n, gy))
    spans, entries, inner = [], [], []
    for q, v in by_q.items():
        first, last, ent, ksum = None, None, 0, 0
        for s, e, n, gy in v:
            if "icp_decide_kernel" in n:
                if first is not None and last is not None:
                    spans.append(last - first)
                    entries.append(ent)
                    inner.append(ksum)
                first, last, ent, ksum = None, None, 0, 0
                continue
            it = any(t in n for t in ("icp_nn_kernel", "icp_nn_search_kernel", "icp_solve_kernel", "icp_apply_kernel"))
            if not it:
                continue
            if "icp_nn_kernel" in n and "search" not in n:
                if first is None:
                    first = s
                ent = max(ent, gy)
            if first is not None:
                ksum += e - s
                if "icp_apply_kernel" in n:
                    last = e
    if spans:
        m = sum(spans) / len(spans)
        print(f"ICP chunks (wide schedule): {len(spans)} chunks, mean span {m / 1e3:.1f} us (first budget-test start to last "
              f"apply end; bench.py's roofline 'avg_us_per_launch'), sum of their kernels' durations {sum(inner) / len(inner) / 1e3:.1f} us, "
              f"mean entries {sum(entries) / len(entries):.1f}")
        by_e = collections.defaultdict(list)
        for sp, en in zip(spans, entries):
            by_e[min(en // 8 * 8, 64)].append(sp)
        print("  chunk span by entries (bins of 8):", {k: (len(v), round(sum(v) / len(v) / 1e3, 1)) for k, v in sorted(by_e.items())})


if __name__ == "__main__":
    main()
