"""Write one synthetic sequence to disk in the layout the reference's WaymoDataset reads
(`<root>/<tag>/<sequence>/%04d.npy` raw rows, `%04d_seg.npy`, `<sequence>.pkl` infos with poses and
boxes, `<root>/ImageSets/val.txt`), every frame in its OWN vehicle frame under a smooth ego motion,
so that the sequence loader (pose alignment, feature normalisation) has real work to do.

  python tools/make_waymo_sequence.py --root /tmp/waymo_synth --frames 20 --points 160000 [--seed 0]
"""
import argparse
import os
import pickle
import sys
from pathlib import Path

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pcseqlearning_amd import synthetic  # noqa: E402

TAG = "waymo_processed_data_v0_5_0"
NAMES = {1: "Vehicle", 2: "Pedestrian", 3: "Cyclist"}


def ego_pose(f, last):
    """Vehicle-to-world pose of frame f: gentle left turn, ~1 m per frame; identity at the last frame
    (the loader's reference frame), so the aligned cloud is the generator's scene."""
    d = f - last
    yaw = 0.01 * d
    T = np.eye(4)
    T[:3, :3] = [[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]]
    T[:3, 3] = [1.0 * d, 0.02 * d * d, 0.0]
    return T


def write(root, seed, frames, points, **scene_kw):
    root = Path(root)
    seq = f"segment-synthetic{seed:04d}_with_camera_labels"
    seq_dir = root / TAG / seq
    seq_dir.mkdir(parents=True, exist_ok=True)
    (root / "ImageSets").mkdir(exist_ok=True)
    (root / "ImageSets" / "val.txt").write_text(seq + ".tfrecord\n")
    scene = synthetic.Scene(seed, **scene_kw)
    rng = np.random.default_rng(seed + 77)
    infos = []
    for f in range(frames):
        fr = synthetic.gen_frame(scene, f, points)
        P = ego_pose(f, frames - 1)
        Pinv = np.linalg.inv(P)
        n = fr["points"].shape[0]
        raw = np.empty((n, 8), dtype=np.float32)
        raw[:, :3] = fr["points"].astype(np.float64) @ Pinv[:3, :3].T + Pinv[:3, 3]
        raw[:, 3] = np.arctanh(np.clip(fr["feat"][:, 0].astype(np.float64), -0.999999, 0.999999))
        raw[:, 4] = fr["feat"][:, 1]
        raw[:, 5] = fr["feat"][:, 2] * 75.0
        raw[:, 6:] = rng.uniform(0, 1, (n, 2))
        boxes = fr["boxes"].astype(np.float32).copy()
        boxes[:, :3] = boxes[:, :3].astype(np.float64) @ Pinv[:3, :3].T + Pinv[:3, 3]
        boxes[:, 6] -= np.arctan2(P[1, 0], P[0, 0])
        np.save(seq_dir / ("%04d.npy" % f), raw)
        np.save(seq_dir / ("%04d_seg.npy" % f), np.stack([fr["inst"], fr["seg"]], axis=1).astype(np.int32))
        infos.append({
            "point_cloud": {"num_features": 5, "lidar_sequence": seq, "sample_idx": f},
            "frame_id": f"{seq}_{f:03d}", "metadata": {"context_name": seq, "timestamp_micros": 100000 * f},
            "pose": P, "num_points_of_each_lidar": [n, 0, 0, 0, 0],
            "annos": {"name": np.array([NAMES[int(c)] for c in fr["box_cls"]]), "gt_boxes_lidar": boxes,
                      "obj_ids": np.array([f"obj{seed}_{i}" for i in fr["obj_ids"]]),
                      "num_points_in_gt": np.zeros(boxes.shape[0], dtype=np.int64)},
        })
    with open(seq_dir / f"{seq}.pkl", "wb") as fh:
        pickle.dump(infos, fh)
    return seq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--points", type=int, default=160000)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    seq = write(args.root, args.seed, args.frames, args.points)
    print(f"wrote {args.frames} frames of {seq} under {args.root}")


if __name__ == "__main__":
    main()
