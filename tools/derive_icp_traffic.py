"""(fetch + write) / algorithmic bytes of the ICP iteration kernels from the two PMC passes of
tools/pmc_batch.sh and the bench line one of them printed (algorithmic bytes per chunk -- a chunk's iterations
of the four iteration kernels -- in that same schedule).
  python tools/derive_icp_traffic.py <FETCH.json> <WRITE.json> <bench line .json>  > profiles/<tag>_icp_hbm_pmc.json
FETCH_SIZE / WRITE_SIZE are KB per dispatch as rocprofv3 reports them (TCC_EA request counters x 64 B at the
fabric side of the L2, Infinity-Cache hits included). MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half
of the bytes of a wide coalesced streaming read; these kernels read 16-byte probes, 64-320-byte cell runs and
16-byte rows, a pattern the guide calls uncalibrated, so both the raw figure and the x2-corrected upper bound
are given."""
import json
import sys


def main():
    fetch, write = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
    line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])

    def kb(table, name, counter):
        tot, n, dur = 0.0, 0, 0.0  # (template instances of a kernel are separate rows: dispatch-weighted mean)
        for k, v in table.items():
            if ("::" + name) in k:
                tot += v["counters_per_dispatch"].get(counter, 0.0) * v["dispatches"]
                dur += v["mean_duration_us"] * v["dispatches"]
                n += v["dispatches"]
        return (tot / n, n, dur / n) if n else (0.0, 0, 0.0)

    out = {"command": "tools/pmc_batch.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py "
                      "--steps 1 --warmup 1 --no-cpu-baseline --no-stages (the default schedule: batches of 12)",
           "per_dispatch_KB": {}}
    tot_f = tot_w = 0.0
    for name in ("icp_nn_kernel", "icp_nn_search_kernel", "icp_solve_kernel", "icp_apply_kernel"):
        f, nf, du = kb(fetch, name, "FETCH_SIZE")
        w, nw, _ = kb(write, name, "WRITE_SIZE")
        out["per_dispatch_KB"][name] = {"FETCH_SIZE": f, "WRITE_SIZE": w, "dispatches": nf,
                                        "mean_duration_us_under_pmc": du}
        tot_f += f * 1024.0
        tot_w += w * 1024.0
    # bench.py's roofline row is per CHUNK (a "launch" there): a chunk's iterations = that many dispatches of each
    # kernel (pcs_icp_set_chunk_iterations; the bench line says how many: 8 until the end of round 5, then 10)
    import re
    m = re.search(r"(\d+) iterations of every registration", line["roofline"].get("kernel", ""))
    per_chunk = float(line["roofline"].get("iterations_per_chunk") or (m.group(1) if m else 8))
    alg = float(line["roofline"].get("algorithmic_bytes_per_launch") or 0.0)
    out["icp_chunk_fetch_bytes"] = round(tot_f * per_chunk, 1)
    out["icp_chunk_write_bytes"] = round(tot_w * per_chunk, 1)
    out["icp_chunk_algorithmic_bytes"] = alg
    if alg > 0:
        out["icp_chunk_traffic_over_algorithmic"] = round((tot_f + tot_w) * per_chunk / alg, 3)
        out["icp_chunk_traffic_over_algorithmic_fetch_x2"] = round((2 * tot_f + tot_w) * per_chunk / alg, 3)
    out["note"] = ("every dispatch of the four iteration kernels is counted (live and gated alike; " + str(int(per_chunk)) + " of each per chunk), "
                   "the algorithmic bytes are those of the event-timed chunks of the same run (live unit-iterations only): "
                   "the ratio describes the schedule, not one launch")
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
