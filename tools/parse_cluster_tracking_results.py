"""Coverage / mIoU report over saved cluster-tracking results: same inputs, arithmetic and text
output as the reference's tools/parse_cluster_tracking_results.py (:19-96).

Reads `<result_dir>/<algorithm>/<sequence>/all.pth` (the dict ClusterTracking.forward saves: attr
[B,7], frame, trace_id, best_iou, ...; cluster_tracking.py:913), re-estimates the per-trace box
velocity (mean corner displacement between consecutive frames of a trace, `moving` = velo > 5e-2),
and per sequence reports coverage (share of boxes with best IoU > --min_iou) and mean IoU, for all
boxes and for the moving ones; `<output_dir>/<algorithm>.txt` lists the sequences in ascending
moving-coverage order, `<output_dir>/compare.txt` contrasts algorithms per sequence.

  python tools/parse_cluster_tracking_results.py --result_dir output/cluster_tracking --output_dir stats
"""
import argparse
import glob
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ALGORITHMS = ("TLS_multiradius_every8",)


def boxes_to_corners_3d(boxes3d):
    """pcdet/utils/box_utils.py:28-53: 8 corners of (x, y, z, dx, dy, dz, heading) boxes."""
    template = boxes3d.new_tensor((
        [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
        [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1],
    )) / 2
    corners = boxes3d[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    angle = boxes3d[:, 6]
    cosa, sina = torch.cos(angle), torch.sin(angle)
    zeros, ones = angle.new_zeros(angle.shape[0]), angle.new_ones(angle.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    corners = torch.matmul(corners.view(-1, 8, 3)[:, :, 0:3], rot)   # common_utils.rotate_points_along_z
    return corners + boxes3d[:, None, 0:3]


def estimate_velo(seq_boxes):
    """parse_cluster_tracking_results.py:19-36."""
    velo = torch.zeros_like(seq_boxes["attr"][:, 0])
    for trace_id in seq_boxes["trace_id"].unique().tolist():
        trace_mask = (seq_boxes["trace_id"] == trace_id).reshape(-1)
        sorted_idx = torch.argsort(seq_boxes["frame"][trace_mask])
        corners = boxes_to_corners_3d(seq_boxes["attr"][trace_mask][sorted_idx])
        trace_velo = torch.zeros_like(corners[:, 0, 0])
        if trace_velo.numel() > 1:
            trace_velo[1:] = (corners[1:] - corners[:-1]).norm(p=2, dim=-1).mean(dim=-1)
            trace_velo[0] = trace_velo[1]
        velo[trace_mask.nonzero()[:, 0][sorted_idx]] = trace_velo
    seq_boxes["velo"] = velo
    seq_boxes["moving"] = velo > 5e-2
    return seq_boxes


def sequence_stats(data, iou_threshold):
    """(coverage, num_boxes, mIoU, moving_coverage, num_moving_boxes, moving_mIoU) or None when the
    sequence has no boxes / no moving box (parse_cluster_tracking_results.py:66-79)."""
    mask = data["best_iou"] > iou_threshold
    num_boxes = mask.shape[0]
    if num_boxes == 0 or not bool(data["moving"].any()):
        return None
    return (mask.float().mean(), num_boxes, data["best_iou"].mean(), mask[data["moving"]].float().mean(),
            data["moving"].long().sum(), data["best_iou"][data["moving"]].mean())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--result_dir", type=str, default="output/cluster_tracking/")
    ap.add_argument("--min_iou", dest="iou_threshold", type=float, default=0.7)
    ap.add_argument("--output_dir", type=str, default="cluster_tracking_stats")
    args = ap.parse_args(argv)
    os.makedirs(args.output_dir, exist_ok=True)
    seq_dict = defaultdict(list)
    for path in glob.glob(f"{args.result_dir}/*"):
        algorithm = path.split("/")[-1]
        if algorithm not in ALGORITHMS:
            continue
        with open(f"{args.output_dir}/{algorithm}.txt", "w") as fout:
            fout.write(f"algorithm={algorithm}\n")
            rows = []
            for pth_file in glob.glob(f"{args.result_dir}/{algorithm}/*/all.pth"):
                sequence_id = pth_file.split("/")[-2]
                data = estimate_velo(dict(torch.load(pth_file, map_location="cpu")))
                stats = sequence_stats(data, args.iou_threshold)
                if stats is None:
                    continue
                rows.append([sequence_id, *stats])
                seq_dict[sequence_id].append([algorithm, stats[0], stats[3]])
            for sequence_id, cov, nb, miou, mcov, nmb, mmiou in sorted(rows, key=lambda r: r[4]):
                fout.write(f"{sequence_id} num_boxes={nb} coverage(all)={cov:.4f} mIoU(all)={miou:.4f} "
                           f"coverage(moving)={mcov:.4f} num_moving_boxes={nmb} mIoU(moving)={mmiou}\n")
    with open(f"{args.output_dir}/compare.txt", "w") as fout:
        for key in sorted(seq_dict, key=lambda k: seq_dict[k][0][2] - seq_dict[k][-1][2]):
            val = seq_dict[key]
            if len(val) > 1 and min(v[2] for v in val) != max(v[2] for v in val):
                fout.write(f"{key}\n")
                for v in val:
                    fout.write(f"\t{v[0]}: all={v[1]:.4f}, moving={v[2]:.4f}\n")


if __name__ == "__main__":
    main()
