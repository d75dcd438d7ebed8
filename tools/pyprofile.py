"""cProfile of one pipeline pass (after a warm-up pass), cumulative time per host function."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcseqlearning_amd.pcdet import models  # noqa: E402
from tools.run_sequence import build_model  # noqa: E402

batch = bench.load_sequence(0, 20, 160000)
model = build_model(streams=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
batch = models.load_data_to_gpu(batch)
with torch.no_grad():
    model(batch)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    model(batch)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
