"""Summarise a rocprofv3 --kernel-trace CSV: span, union of busy time, sum of kernel durations,
mean concurrency, per-queue load and the kernels with the largest total duration, restricted to
the last `--tail-frac` of the trace (the timed step after warm-up)."""
import argparse
import csv
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--tail-frac", type=float, default=0.5)
    ap.add_argument("--top", type=int, default=25)
    args = ap.parse_args()
    rows = []
    with open(args.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - (t1 - t0) * args.tail_frac
    rows = [r for r in rows if r[0] >= cut]
    span = (max(r[1] for r in rows) - rows[0][0]) / 1e6
    total = sum(r[1] - r[0] for r in rows) / 1e6
    # union of intervals
    busy, cur_s, cur_e = 0, None, None
    for s, e, *_ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    busy /= 1e6
    print(f"launches {len(rows)}  span {span:.1f} ms  union-busy {busy:.1f} ms  sum-of-durations {total:.1f} ms  "
          f"mean concurrency while busy {total / busy:.2f}")
    perq = collections.Counter()
    perq_t = collections.Counter()
    for s, e, n, q, st in rows:
        perq[(q, st)] += 1
        perq_t[(q, st)] += e - s
    print("queue/stream: launches, busy ms")
    for k in sorted(perq, key=lambda k: -perq_t[k])[:16]:
        print(f"   {k}: {perq[k]:7d} {perq_t[k] / 1e6:9.1f}")
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n, q, st in rows:
        a = agg[n.split("(")[0][:70]]
        a[0] += 1
        a[1] += e - s
    print("kernel: calls, total ms, avg us")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f"   {c:7d} {t / 1e6:9.1f} {t / c / 1e3:8.1f}  {n}")


if __name__ == "__main__":
    main()
