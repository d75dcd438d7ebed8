"""Summarise a rocprofv3 --pmc counter_collection.csv (+ kernel durations from the same file's
timestamps) per kernel name: dispatches, mean duration, counter sums per dispatch, and for MFMA
kernels MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (duration x clock x 1024 SIMDs).

  python tools/summarize_pmc.py <counter_collection.csv> [filter] > profiles/<name>.json
"""
import collections
import csv
import json
import sys

CLOCK_GHZ = 2.4   # MI355X max shader clock (MI355X_MICROARCH.md); DVFS lowers it under load
SIMDS = 256 * 4


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    per = collections.defaultdict(lambda: {"counters": collections.defaultdict(float), "dur_ns": {}, })
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if flt and flt not in name:
            continue
        e = per[name]
        e["counters"][r["Counter_Name"]] += float(r["Counter_Value"])
        e["dur_ns"][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for name, e in per.items():
        n = len(e["dur_ns"])
        dur = sum(e["dur_ns"].values()) / n
        row = {"dispatches": n, "mean_duration_us": round(dur / 1e3, 2),
               "counters_per_dispatch": {k: round(v / n, 1) for k, v in e["counters"].items()}}
        busy = e["counters"].get("SQ_VALU_MFMA_BUSY_CYCLES")
        if busy:
            row["mfma_util_pct_at_2.4GHz"] = round(100 * (busy / n) / (dur * CLOCK_GHZ * SIMDS), 2)
        out[name] = row
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
