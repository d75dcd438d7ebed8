"""How much do the kernels of a rocprofv3 --kernel-trace run overlap?
  python tools/trace_concurrency.py <..._kernel_trace.csv>
Prints per queue: dispatches and busy time; overall: wall time covered, time-weighted mean number of
kernels running, histogram of the concurrency level, and for the three ICP kernels the mean duration
split by how many other kernels ran beside them."""
import csv
import collections
import sys


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0],
                     r.get("Queue_Id", "?"), int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0)))
    rows.sort()
    t0 = rows[0][0]
    # keep the second half (steady state of the timed steps)
    tmid = t0 + (max(r[1] for r in rows) - t0) // 2
    rows = [r for r in rows if r[0] >= tmid]
    per_q = collections.defaultdict(lambda: [0, 0])
    for s, e, n, q, g, w in rows:
        per_q[q][0] += 1
        per_q[q][1] += e - s
    span = max(r[1] for r in rows) - rows[0][0]
    print("window ms", span / 1e6, "dispatches", len(rows))
    for q, (c, b) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
        print(f"  queue {q}: {c} dispatches, busy {b / 1e6:.1f} ms ({100 * b / span:.1f} % of the window)")
    # gaps between consecutive kernels of a queue (end -> next start)
    by_q = collections.defaultdict(list)
    for s_, e_, n_, q_, g_, w_ in rows:
        by_q[q_].append((s_, e_, n_))
    for q, lst in sorted(by_q.items(), key=lambda kv: -len(kv[1]))[:5]:
        lst.sort()
        gaps = sorted(max(0, b[0] - a[1]) for a, b in zip(lst, lst[1:]))
        small = [g for g in gaps if g < 200000]  # (longer ones: the host is between ticks or in another phase)
        if not gaps:
            continue
        print(f"  queue {q} gaps: median {gaps[len(gaps) // 2] / 1e3:.2f} us, p90 {gaps[int(len(gaps) * 0.9)] / 1e3:.2f} us, "
              f"sum of gaps < 200 us: {sum(small) / 1e6:.1f} ms, all: {sum(gaps) / 1e6:.1f} ms")
        hist = collections.Counter(min(int(g / 5000), 12) for g in gaps)
        print("    gap histogram (5 us bins, last = >= 60 us):", [hist.get(k, 0) for k in range(13)])
    ev = []
    for i, (s, e, *_) in enumerate(rows):
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    level, last, hist = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        hist[level] += t - last
        last = t
        level += d
    tot = sum(hist.values())
    print("time at concurrency level:", {k: round(100 * v / tot, 1) for k, v in sorted(hist.items())})
    print("mean kernels running:", sum(k * v for k, v in hist.items()) / tot)
    # ICP kernels: duration vs number of kernels overlapping their interval (sampled at the midpoint)
    starts = sorted((s, e) for s, e, *_ in rows)
    import bisect
    ss = [s for s, _ in starts]
    for name in ("pcs::icp_nn_kernel", "pcs::icp_nn_search_kernel", "pcs::icp_apply_kernel"):
        by = collections.defaultdict(list)
        for s, e, n, q, g, w in rows:
            if n != name:
                continue
            mid = (s + e) // 2
            hi = bisect.bisect_right(ss, mid)
            lo = max(0, hi - 64)
            c = sum(1 for a, b in starts[lo:hi] if b > mid) - 1
            by[min(c, 6)].append(e - s)
        print(name, {k: (len(v), round(sum(v) / len(v) / 1e3, 1)) for k, v in sorted(by.items())})


if __name__ == "__main__":
    main()
