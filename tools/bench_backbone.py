"""BASELINE config 5: full VoxelBackBone8x forward on one 160k-pt synthetic frame (dynamic VFE ->
12 sparse convs over 8 rulebooks), eval-mode BN. Times the forward with the MFMA tiles and with
the FMA tiles (bit-identical outputs are asserted) and prints one JSON line.

  python tools/bench_backbone.py [--points 160000] [--batch 1] [--reps 20] [--mfma 0|1|both]

--batch B puts B frames (seeds 0..B-1) into one batch, as the reference's API does (B = 1 is a
launch-latency benchmark). The line carries a `roofline` block: conv FLOP/s against the f32 MFMA peak
and the algorithmic HBM bytes of the 12 convs (4 V Cin + 4 V' Cout + 8 sum P_k + 4 KV Cin
Cout per layer) against the HBM peak; MFMA busy % comes from a separate rocprofv3 --pmc pass
(tools/summarize_pmc.py -> profiles/*_backbone_mfma_pmc*.json).
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pcseqlearning_amd import lib as L  # noqa: E402
from pcseqlearning_amd import synthetic  # noqa: E402
from pcseqlearning_amd.pcdet.models.backbones_3d import VoxelBackBone8x  # noqa: E402
from pcseqlearning_amd.pcdet.models.backbones_3d.vfe import DynamicMeanVFE  # noqa: E402


def conv_flops(out):
    """2 * sum(P_k) * Cin * Cout over the 12 convs, from the cached rulebooks."""
    t = out["encoded_spconv_tensor"]
    return t.indice_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=160000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--mfma", default="both")
    ap.add_argument("--train", action="store_true", help="also time one forward + backward step")
    ap.add_argument("--half", action="store_true",
                    help="fp16 voxel features through the backbone (pcs_indice_conv_table_f16: fp16 rows gathered as "
                         "they are, v_mfma_f32_16x16x32_f16, fp32 accumulation); reported beside the fp32 forward")
    ap.add_argument("--f16-tiles", type=int, default=0,
                    help="pcs_indice_conv_f16_short_tiles: 0 = direct gathers, 256 threads (default), 3 = the same on 512 "
                         "threads, 2 = 128-row staged tiles, 1 = round 4's 64-row tiles")
    args = ap.parse_args()
    dev = "cuda:0"
    lib = L.load()
    L.check(lib.pcs_indice_conv_f16_short_tiles(ctypes.c_int(args.f16_tiles)), "pcs_indice_conv_f16_short_tiles")
    torch.manual_seed(0)
    frames = []
    for b in range(args.batch):
        f = synthetic.gen_voxel_frame(b, args.points)
        f[:, 0] = b
        frames.append(f)
    import numpy as np
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    vfe = DynamicMeanVFE({}, 5, synthetic.VOXEL_SIZE, synthetic.GRID_SIZE, synthetic.PC_RANGE)
    net = VoxelBackBone8x({}, 5, synthetic.GRID_SIZE.tolist()).to(dev).eval()

    def forward():
        batch = vfe({"points": pts, "batch_size": args.batch})
        with torch.no_grad():
            return net(batch)

    modes = {"both": (1, 0), "1": (1,), "0": (0,)}[args.mfma]
    results, outs = {}, {}
    for mode in modes:
        lib.pcs_conv_use_mfma(ctypes.c_int(mode))
        out = forward()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.reps):
            out = forward()
        e.record()
        torch.cuda.synchronize()
        results["mfma" if mode else "fma"] = s.elapsed_time(e) / args.reps
        outs[mode] = out["encoded_spconv_tensor"].features.clone()
    lib.pcs_conv_use_mfma(ctypes.c_int(1))
    if len(modes) == 2:
        assert torch.equal(outs[1], outs[0]), "MFMA and FMA tiles must agree bit for bit"
    half = None
    if args.half:
        def forward_half():
            batch = vfe({"points": pts, "batch_size": args.batch})
            batch["voxel_features"] = batch["voxel_features"].half()
            with torch.no_grad():
                return net(batch)

        o16 = forward_half()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.reps):
            o16 = forward_half()
        e.record()
        torch.cuda.synchronize()
        f16 = o16["encoded_spconv_tensor"].features
        ref = outs[modes[0]]
        assert f16.dtype == torch.float16 and f16.shape == ref.shape
        err = (f16.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
        half = {"ms_per_forward": round(s.elapsed_time(e) / args.reps, 3),
                "max_abs_error_over_fp32_max": round(err, 5)}
    # per-layer time of the conv kernels alone (one extra forward per precision, HIP events around every conv call on
    # the cached table; the rulebooks, the VFE and the BatchNorm folds are what is left of a forward)
    from pcseqlearning_amd import ops as pops

    def layer_times(fwd):
        spans = []
        orig32, orig16 = pops.indice_conv_table, pops.indice_conv_table_half

        def timed(fn):
            def call(features, filters, table, num_out, *a, **k):
                s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                r = fn(features, filters, table, num_out, *a, **k)
                e0.record()
                spans.append((int(filters.shape[-2]), int(filters.shape[-1]), int(num_out), s0, e0))
                return r
            return call
        pops.indice_conv_table, pops.indice_conv_table_half = timed(orig32), timed(orig16)
        try:
            fwd()
            torch.cuda.synchronize()
        finally:
            pops.indice_conv_table, pops.indice_conv_table_half = orig32, orig16
        return [(ci, co, n, s0.elapsed_time(e0)) for ci, co, n, s0, e0 in spans]

    conv_ms = layer_times(forward)
    conv_ms16 = layer_times(forward_half) if args.half else None
    # flops of the conv layers from the rulebooks of the last forward
    d = out["encoded_spconv_tensor"].indice_dict
    chans = {"subm1": [(5, 16), (16, 16)], "spconv2": [(16, 32)], "subm2": [(32, 32)] * 2,
             "spconv3": [(32, 64)], "subm3": [(64, 64)] * 2, "spconv4": [(64, 64)],
             "subm4": [(64, 64)] * 2, "spconv_down2": [(64, 128)]}
    flops = 0
    hbm_bytes = 0
    voxels = {}
    layers_out = []
    v_in = int(out["multi_scale_3d_features"]["x_conv1"].indices.shape[0])
    for key, layers in chans.items():
        num = int(d[key][3].sum())
        kv = int(d[key][3].numel())
        v_out = int(d[key][0].shape[0])
        voxels[key] = v_out
        for cin, cout in layers:
            fl = 2 * num * cin * cout
            by = 4 * v_in * cin + 4 * v_out * cout + 8 * num + 4 * kv * cin * cout
            flops += fl
            hbm_bytes += by
            row = {"rulebook": key, "cin": cin, "cout": cout, "pairs": num, "mflop": round(fl / 1e6, 1),
                   "algorithmic_MB": round(by / 1e6, 2)}
            li = len(layers_out)
            if li < len(conv_ms) and conv_ms[li][:2] == (cin, cout):
                ms = conv_ms[li][3]
                row["conv_kernel_ms"] = round(ms, 4)
                row["tflops"] = round(fl / (ms * 1e-3) / 1e12, 2)
                row["frac_of_f32_mfma_peak"] = round(fl / (ms * 1e-3) / 1e12 / 157.3, 4)
                row["algorithmic_GBs"] = round(by / (ms * 1e-3) / 1e9, 1)
                row["frac_of_8TBs"] = round(by / (ms * 1e-3) / 1e9 / 8000.0, 4)
            if conv_ms16 is not None and li < len(conv_ms16) and conv_ms16[li][:2] == (cin, cout):
                ms = conv_ms16[li][3]
                by16 = by - (4 * v_in * cin + 4 * v_out * cout + 4 * kv * cin * cout) / 2.0
                row["fp16"] = {"conv_kernel_ms": round(ms, 4), "tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                               "algorithmic_GBs": round(by16 / (ms * 1e-3) / 1e9, 1),
                               "frac_of_8TBs": round(by16 / (ms * 1e-3) / 1e9 / 8000.0, 4)}
            layers_out.append(row)
            v_in = v_out
    t_best = results.get("mfma", results.get("fma")) * 1e-3
    line = {"workload": f"config5: VoxelBackBone8x forward, {args.batch} synthetic frame(s) per batch",
            "points": args.points, "batch": args.batch,
            "ms_per_frame": round(t_best * 1e3 / args.batch, 3),
            "roofline": {"bound": "mfma", "achieved": round(flops / t_best / 1e12, 3), "peak": 157.3,
                         "unit": "TFLOP/s", "frac": round(flops / t_best / 1e12 / 157.3, 4),
                         "hbm": {"algorithmic_MB": round(hbm_bytes / 1e6, 1),
                                 "achieved_GBs": round(hbm_bytes / t_best / 1e9, 1),
                                 "frac_of_8TBs": round(hbm_bytes / t_best / 1e9 / 8000.0, 4)},
                         "note": ("whole forward (VFE + 8 rulebooks + 12 convs + BN/ReLU epilogues) over the "
                                  "convs' flops / bytes; f32 in / f32 accumulate (v_mfma_f32_16x16x4_f32)"),
                         "layers": layers_out},
            "voxels": int(out["multi_scale_3d_features"]["x_conv1"].indices.shape[0]),
            "ms_per_forward": {k: round(v, 3) for k, v in results.items()},
            "conv_gflop": round(flops / 1e9, 3),
            "conv_tflops_mfma_path": round(flops / (results.get("mfma", results.get("fma")) * 1e-3) / 1e12, 3),
            "output_voxels_per_rulebook": voxels, "bit_identical_mfma_vs_fma": len(modes) == 2,
            "conv_kernels_ms_per_forward": {"fp32": round(sum(x[3] for x in conv_ms), 3),
                                            "fp16": round(sum(x[3] for x in conv_ms16), 3) if conv_ms16 else None}}
    if half is not None:
        t16 = half["ms_per_forward"] * 1e-3
        hb16 = hbm_bytes / 2.0  # fp16 rows: half the feature and filter bytes (the pair lists stay 8 bytes)
        half["roofline"] = {"bound": "mfma", "achieved": round(flops / t16 / 1e12, 3), "peak": 2500.0, "unit": "TFLOP/s",
                            "frac": round(flops / t16 / 1e12 / 2500.0, 5),
                            "hbm": {"algorithmic_MB": round(hb16 / 1e6, 1), "achieved_GBs": round(hb16 / t16 / 1e9, 1),
                                    "frac_of_8TBs": round(hb16 / t16 / 1e9 / 8000.0, 4)},
                            "note": "fp16 in, fp32 accumulate (v_mfma_f32_16x16x32_f16); the dense fp16 MFMA peak"}
        line["half"] = half
    if args.train:
        # one training-mode forward + backward through the same network (batch-stat
        # BatchNorm in torch, conv forward on the table kernel, conv backward on
        # pcs_indice_conv_backward_f32); loss = sum of the encoded features
        net.train()
        batch0 = vfe({"points": pts, "batch_size": 1})

        def step():
            net.zero_grad(set_to_none=True)
            out_t = net(dict(batch0))
            out_t["encoded_spconv_tensor"].features.sum().backward()

        step()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.reps):
            step()
        e.record()
        torch.cuda.synchronize()
        line["ms_per_train_step_fwd_bwd"] = round(s.elapsed_time(e) / args.reps, 3)
        line["conv_backward_gflop"] = round(2 * flops / 1e9, 3)
        g0 = net.conv_input[0].weight.grad.clone()
        step()
        line["backward_reproducible"] = bool(torch.equal(g0, net.conv_input[0].weight.grad))
    print(json.dumps(line))


if __name__ == "__main__":
    main()
