#!/usr/bin/env python3
"""Counters of the fused ICP's verified cache over ONE sequence (library built with -DPCS_ICP_STATS:
PCS_EXTRA_HIP_FLAGS=-DPCS_ICP_STATS python -m pcseqlearning_amd.build --force):  python tools/cache_stats.py"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcseqlearning_amd import lib as L  # noqa: E402
from pcseqlearning_amd.pcdet import models  # noqa: E402
from tools.run_sequence import build_model  # noqa: E402


def main():
    batch = models.load_data_to_gpu(bench.load_sequence(0, 20, 160000))
    model = build_model(save=False)
    lib = L.load()
    out = (ctypes.c_ulonglong * 16)()
    with torch.no_grad():
        model(batch)
        L.check(lib.pcs_icp_cache_stats(out), "stats")
        model(batch)
        L.check(lib.pcs_icp_cache_stats(out), "stats")
    v = list(out)
    names = ["tested", "budget_ran_out", "kept_by_second_level", "searched"]
    print(json.dumps({"all": dict(zip(names, v[:4])), "backward": dict(zip(names, v[4:8])),
                      "forward": dict(zip(names, [a - b for a, b in zip(v[:4], v[4:8])])),
                      "cooperative_searches": v[8], "seeded_searches": v[9],
                      "coop_later_without_partner": v[10], "coop_later_with_partner": v[11], "coop_later_found_none": v[12],
                      "coop_first": v[13], "coop_first_found_none": v[14]}))


if __name__ == "__main__":
    main()
