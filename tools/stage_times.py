"""Per-stage wall time (device-synchronised) of the cluster-tracking pipeline on one synthetic
sequence: subsample, ground removal, cluster proposal, tracking (track_frame / extract traces)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from pcseqlearning_amd.pcdet import models  # noqa: E402
from pcseqlearning_amd.pcdet.utils import timer  # noqa: E402
from tools.run_sequence import build_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--units-per-call", type=int, default=None)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--units", action="store_true", help="print the per-unit track / extract spans")
    ap.add_argument("--detail", action="store_true", help="device-synchronised spans inside the stages")
    args = ap.parse_args()
    timer.SYNC = args.detail
    batch = bench.load_sequence(0, 20, 160000)
    model = build_model(units_per_call=args.units_per_call)
    batch = models.load_data_to_gpu(batch)
    spans = {}

    def wrap(mod, name):
        orig = mod.forward

        def fwd(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = orig(*a, **k)
            torch.cuda.synchronize()
            spans[name] = spans.get(name, 0.0) + time.perf_counter() - t0
            return out
        mod.forward = fwd

    for i, m in enumerate(model.preprocessors):
        wrap(m, f"{i}:{type(m).__name__}")
    for r in range(args.repeat):
        spans.clear()
        timer.RECORDS.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            model(batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"run {r}: total {dt * 1e3:.1f} ms")
        for k, v in spans.items():
            print(f"   {k:36s} {v * 1e3:8.1f} ms")
        print(f"   {'(outside the preprocessors)':36s} {(dt - sum(spans.values())) * 1e3:8.1f} ms")
        trk = sum(v for k, v in timer.RECORDS.items() if k.startswith("Tracking Frame"))
        ext = timer.RECORDS.get("Extract Traces from all points", 0.0)
        print(f"   sum over units: track_frame {trk * 1e3:.1f} ms, extract traces {ext * 1e3:.1f} ms")
        print(f"   native tracker: {getattr(model.preprocessors[2], 'last_ticks', 0)} host ticks, "
              f"{model.preprocessors[2].num_registration_calls} registrations")
        if args.detail:
            for k in sorted(timer.RECORDS):
                if not k.startswith(("unit ", "Tracking Frame", "REGISTRATION")):
                    print(f"      {k:44s} {timer.RECORDS[k] * 1e3:8.1f} ms")
        if args.units:
            for k in sorted(timer.RECORDS):
                if k.startswith("unit "):
                    print(f"      {k:44s} {timer.RECORDS[k] * 1e3:8.1f} ms")


if __name__ == "__main__":
    main()
