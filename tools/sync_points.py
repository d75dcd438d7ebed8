"""Where one sequence's host thread WAITS for the device: every synchronising torch call of one pipeline pass
(torch.cuda.set_sync_debug_mode), grouped by the source line of this package that made it.
    python tools/sync_points.py"""
import collections
import os
import sys
import traceback
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcseqlearning_amd.pcdet import models  # noqa: E402
from tools.run_sequence import build_model  # noqa: E402


def main():
    batch = models.load_data_to_gpu(bench.load_sequence(0, 20, 160000))
    model = build_model(save=False)
    with torch.no_grad():
        model(batch)
        torch.cuda.synchronize()
        sites = collections.Counter()
        orig = warnings.showwarning

        def show(message, category, filename, lineno, file=None, line=None):
            if "synchroniz" not in str(message):
                return orig(message, category, filename, lineno, file, line)
            for fr in reversed(traceback.extract_stack()):
                if "pcseqlearning_amd" in fr.filename and "sync_points" not in fr.filename:
                    sites[f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.line}"] += 1
                    break
            else:
                sites[f"{filename}:{lineno}"] += 1

        warnings.showwarning = show
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode(1)
        model(batch)
        torch.cuda.set_sync_debug_mode(0)
        warnings.showwarning = orig
    print(f"{sum(sites.values())} synchronising calls in one sequence")
    for k, v in sites.most_common():
        print(f"{v:4d}  {k}")


if __name__ == "__main__":
    main()
