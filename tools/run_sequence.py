"""Run the cluster-tracking pipeline (SimpleReg: GroundPlaneRemover -> ClusterProposal ->
ClusterTracking) on one synthetic Waymo-shaped sequence and print per-stage timings.

  python tools/run_sequence.py --frames 20 --points 160000 [--seed 0] [--save]
  python tools/run_sequence.py --data-root /tmp/waymo_synth --frames 20     (sequence on disk, through
      the WaymoDataset loader; tools/make_waymo_sequence.py writes one)
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pcseqlearning_amd import synthetic  # noqa: E402
from pcseqlearning_amd.pcdet import config, models  # noqa: E402
from pcseqlearning_amd.pcdet.utils import timer  # noqa: E402


def build_model(save=False, verbose=False, device="cuda", units_per_call=None):
    cfg = config.cfg_from_yaml_file(config.CLUSTER_TRACKING_TLS_MULTIRADIUS_EVERY8)
    if not save:
        cfg.MODEL.SAVE_DIR = None
        for p in cfg.MODEL.PREPROCESSORS:
            p["SAVE"] = False
    if units_per_call is not None:
        cfg.MODEL.PREPROCESSORS[2]["UNITS_PER_CALL"] = int(units_per_call)
    model = models.build_network(cfg.MODEL)
    for m in model.modules():
        if hasattr(m, "verbose"):
            m.verbose = verbose
    return model.to(device).eval()


def ground_cfg():
    """The GroundPlaneRemover entry of cluster_tracking_TLS_multiradius_every8.yaml as the plain dictionary the
    ground stage (and its CPU oracle) takes."""
    cfg = config.cfg_from_yaml_file(config.CLUSTER_TRACKING_TLS_MULTIRADIUS_EVERY8)
    g = dict(cfg.MODEL.PREPROCESSORS[0])
    return {k: g[k] for k in ("PILLAR_SIZE", "RANSAC", "SIGMA2", "K", "JointOpt", "LR", "DECAY_STEPS", "RIGID_WEIGHT",
                              "MAX_NUM_ITERS", "TRUNCATE_HEIGHT")}


def loader_cfg(root, num_sweeps):
    """tools/cfgs/dataset_configs/waymo/PCsequence/registration/all_sequence.yaml of the reference."""
    from pcseqlearning_amd.pcdet.utils.common_utils import EasyDict
    return EasyDict({
        "DATASET": "WaymoDataset", "DATA_PATH": str(root), "PROCESSED_DATA_TAG": "waymo_processed_data_v0_5_0",
        "BOX_CLASSES": ["Vehicle", "Pedestrian", "Cyclist"], "POINT_CLOUD_RANGE": [-75.2, -75.2, -2, 75.2, 75.2, 4],
        "DATA_SPLIT": {"train": "train", "test": "val"}, "NUM_SWEEPS": int(num_sweeps), "SWEEP_DIR": 1,
        "SAMPLED_INTERVAL": {"train": 1, "test": 1}, "SAMPLED_SEQUENCE_INTERVAL": {"train": 1, "test": 1},
        "SAMPLE_IDX_RANGE": {"train": [0, 1], "test": [0, 1]}, "SYNC_MOVING_POINTS": False,
        "SEGMENTATION_CFG": {"NUM_SEG_CLASSES": 23, "USE_ONLY_SAMPLES_WITH_SEG_LABELS": False, "LOAD_SEG": True},
        "DROP_POINTS_BY_LIDAR_INDEX": [1, 2, 3, 4],
        "POINT_FEATURE_ENCODING": {"encoding_type": "absolute_coordinates_encoding",
                                   "used_feature_list": ["intensity", "elongation", "range"],
                                   "src_feature_list": ["intensity", "elongation", "range", "rimage_w", "rimage_h"]},
        "MAX_NUM_POINTS": 40000000,
        "DATA_PROCESSOR": [{"NAME": "transform_points_to_voxels", "VOXEL_SIZE": [0.1, 0.1, 0.15],
                            "POINT_CLOUD_RANGE": [-75.2, -75.2, -2, 75.2, 75.2, 4], "MAX_POINTS_PER_VOXEL": 5,
                            "MAX_NUMBER_OF_VOXELS": {"train": 150000, "test": 150000}, "DRY": True}],
    })


def load_from_disk(root, num_sweeps, device="cuda"):
    from pcseqlearning_amd.pcdet.datasets import WaymoDataset
    ds = WaymoDataset(loader_cfg(root, num_sweeps), training=False, root_path=root)
    return ds.load_batch_to_gpu([0], device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-root", default=None, help="sequence on disk in the reference's layout")
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--points", type=int, default=160000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--save", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--units-per-call", type=int, default=None,
                    help="tracking units per native call (default: all units of the sequence in lock-step)")
    args = ap.parse_args()
    t0 = time.time()
    if args.data_root:
        batch = load_from_disk(args.data_root, args.frames)
        print(f"loaded {batch['point_bxyz'].shape[0]} points from {args.data_root} in {time.time() - t0:.2f}s", flush=True)
    else:
        batch = synthetic.gen_sequence(args.seed, args.frames, args.points)
        print(f"generated {batch['point_bxyz'].shape[0]} points in {time.time() - t0:.1f}s", flush=True)
    model = build_model(args.save, args.verbose, units_per_call=args.units_per_call)
    batch = models.load_data_to_gpu(batch)
    for r in range(args.repeat):
        timer.RECORDS.clear()
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.no_grad():
            model(batch)
        torch.cuda.synchronize()
        dt = time.time() - t0
        out = model.last_outputs[0]
        print(f"run {r}: {dt:.3f}s  ({1.0 / dt:.3f} sequences/s)")
        for k, v in timer.RECORDS.items():
            print(f"   {k:40s} {v:.3f}s")
    n_sub = out['full_point_fxyz'].shape[0]
    print(f"subsampled {n_sub} pts, non-ground {out['point_fxyz'].shape[0]}")
    for key in model.preprocessors[1].component_keys:
        print(f"   {key}: {int(out['point_' + key].max()) + 1} components")
    res = out.get("cluster_tracking_results", {})
    for (fid, key), ex in res.items():
        print(f"   anchor {fid:3d} {key}: extracted {ex['fxyz'].shape[0]} pts, "
              f"{int(ex['component'].unique().numel()) if ex['fxyz'].shape[0] else 0} tracked components")
    if "gt_box_best_iou_tracking" in out:
        print(f"   box best IoU mean {float(out['gt_box_best_iou_tracking'].mean()):.4f}")


if __name__ == "__main__":
    main()
