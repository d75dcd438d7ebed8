#!/usr/bin/env python3
"""One sequence per step with and without SAVE (the reference writes `<frame:03d>_<key>.pth` per unit and `all.pth`,
cluster_tracking.py:884,913): what the drop-in's step costs with the result files written. One JSON line.
    python tools/time_save.py [--steps 5]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import bench
    host = bench.load_sequences([0], 20, 160000)[0]
    import torch
    from pcseqlearning_amd.pcdet import models
    from tools.run_sequence import build_model
    dev = torch.device("cuda", 0)
    batch = models.load_data_to_gpu(host, dev)
    out = {}
    tmp = tempfile.mkdtemp(prefix="pcs_save_")
    try:
        for save in (False, True):
            model = build_model(save=save, device=dev)
            def fresh_dirs(step):
                # (a sequence whose result files exist is skipped, as in the reference: every step writes into new folders)
                if not save:
                    return
                model.model_cfg["SAVE_DIR"] = None
                for m in model.preprocessors:
                    if hasattr(m, "model_cfg") and "DIR" in m.model_cfg:
                        m.model_cfg["DIR"] = os.path.join(tmp, f"step{step}", type(m).__name__)
            ts = []
            with torch.no_grad():
                fresh_dirs(-1)
                model(batch)
                torch.cuda.synchronize()
                for i in range(args.steps):
                    fresh_dirs(i)
                    t0 = time.perf_counter()
                    model(batch)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
            out["save" if save else "no_save"] = {"ms_per_sequence": round(sorted(ts)[len(ts) // 2], 2), "all": [round(t, 1) for t in ts]}
        files = sum(len(f) for _, _, f in os.walk(tmp))
        size = sum(os.path.getsize(os.path.join(d, x)) for d, _, f in os.walk(tmp) for x in f)
        out["files_written_per_step"] = files // (args.steps + 1)
        out["MB_written_per_step"] = round(size / 1e6 / (args.steps + 1), 1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
