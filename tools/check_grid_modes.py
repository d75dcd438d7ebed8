"""Cross-check of the moving grid's two modes on whole synthetic sequences: in place (default) against rebuilt at every
ICP iteration (ops.icp_set_grid_mode), every tracking output bit for bit, for several generator seeds.
    python tools/check_grid_modes.py [seed ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcseqlearning_amd import ops  # noqa: E402
from pcseqlearning_amd.pcdet import models  # noqa: E402
from tools.run_sequence import build_model  # noqa: E402

FIELDS = ("fxyz", "component", "original_indices", "frame_indices", "moving", "component_hit")


def run(batch, rebuild_always):
    ops.icp_set_grid_mode(rebuild_always)
    try:
        model = build_model(save=False)
        with torch.no_grad():
            model(batch)
        return model.last_outputs[0], int(model.preprocessors[2].num_registration_calls)
    finally:
        ops.icp_set_grid_mode(False)


def main():
    seeds = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
    hosts = bench.load_sequences(seeds, 20, 160000)
    bad = 0
    for seed, host in zip(seeds, hosts):
        batch = models.load_data_to_gpu(host)
        a, na = run(batch, False)
        b, nb = run(batch, True)
        ok = na == nb and a["cluster_tracking_results"].keys() == b["cluster_tracking_results"].keys()
        rows = 0
        for key, ra in a["cluster_tracking_results"].items():
            rb = b["cluster_tracking_results"][key]
            for f in FIELDS:
                ok = ok and torch.equal(ra[f], rb[f])
            rows += int(ra["fxyz"].shape[0])
        ok = ok and torch.equal(a["gt_box_best_iou_tracking"], b["gt_box_best_iou_tracking"])
        print(f"seed {seed}: {len(a['cluster_tracking_results'])} units, {na} registrations, {rows} extracted rows: "
              f"{'identical' if ok else 'DIFFERENT'}", flush=True)
        bad += 0 if ok else 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
