"""VoxelResBackBone8x forward (pcdet/models/backbones_3d/spconv_backbone.py:161-271 of the reference) on
configs[1]-shaped frames, with the residual blocks' `+ identity` and ReLU in the second conv's epilogue
(pcs_indice_conv_table_residual_f32 / _f16, the default) and as tensor ops, alternating; fp32 (and fp16 with --half):
  python tools/bench_res_backbone.py --batch 20 [--half] [--reps 20]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pcseqlearning_amd import synthetic  # noqa: E402
from pcseqlearning_amd.pcdet.models.backbones_3d import VoxelResBackBone8x  # noqa: E402
from pcseqlearning_amd.pcdet.models.backbones_3d.vfe import DynamicMeanVFE  # noqa: E402
from pcseqlearning_amd.pcdet.models.blocks import spconv_blocks as SB  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=160000)
    ap.add_argument("--batch", type=int, default=20)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--half", action="store_true")
    args = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    frames = []
    for b in range(args.batch):
        f = synthetic.gen_voxel_frame(b, args.points)
        f[:, 0] = b
        frames.append(f)
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    vfe = DynamicMeanVFE({}, 5, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE, synthetic.PC_RANGE)
    net = VoxelResBackBone8x({}, 5, synthetic.GRID_SIZE.tolist()).to(dev).eval()

    def forward(half):
        batch = vfe({"points": pts, "batch_size": args.batch})
        if half:
            batch["voxel_features"] = batch["voxel_features"].half()
        with torch.no_grad():
            return net(batch)["encoded_spconv_tensor"].features

    def timed(half, fuse):
        SB.FUSE_RESIDUAL_ADD = fuse
        try:
            out = forward(half)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.reps):
                out = forward(half)
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / args.reps, out
        finally:
            SB.FUSE_RESIDUAL_ADD = True

    res = {"model": "VoxelResBackBone8x", "batch": args.batch, "points_per_frame": args.points, "reps": args.reps}
    for half in ([False, True] if args.half else [False]):
        ms = {True: [], False: []}
        outs = {}
        for _ in range(args.rounds):
            for fuse in (True, False):
                t, outs[fuse] = timed(half, fuse)
                ms[fuse].append(round(t, 3))
        res["fp16" if half else "fp32"] = {"ms_per_forward_fused_add": ms[True], "ms_per_forward_tensor_ops": ms[False],
                                           "identical_bits": bool(torch.equal(outs[True], outs[False]))}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
