"""Golden vectors for the sequence loader, produced by IMPORTING the reference's
own `WaymoDataset` (pcdet/datasets/waymo/waymo_dataset.py) and `DatasetTemplate.collate_batch`
(pcdet/datasets/dataset.py) in the authoring container and running them on a small synthetic
sequence written to a temporary directory in the reference's on-disk layout
(`<tag>/<sequence>/%04d.npy`, `%04d_seg.npy`, `<sequence>.pkl`, `ImageSets/<split>.txt`).

  loader_fwd.npz   NUM_SWEEPS 5, SWEEP_DIR +1 from sample 0 (the registration configs' direction)
  loader_bwd.npz   NUM_SWEEPS 4, SWEEP_DIR -1 from the last sample
  loader_sync.npz  (round 3) NUM_SWEEPS 5 forward with SYNC_MOVING_POINTS + WITH_TIME_FEAT, on a sequence whose
                   boxes hold points (waymo_dataset.py:513-569, 587-618, 625-630)
  loader_filter.npz (round 3) NUM_SWEEPS 4 forward with FILTER_FOREGROUND + REMOVE_GROUND (waymo_dataset.py:479-492);
                   the per-frame prediction / height files are part of the inputs (`pred_<f>`, `height_<f>`)

Each file holds the inputs (raw frames, seg labels, poses, boxes, names, ids, per-lidar counts) and
every array of the collated batch the reference returns. Third-party modules absent from this image
are provided in-memory for the import only and are never called on this path (skimage,
torch_cluster, torch_scatter, SharedArray), except `roiaware_pool3d_utils.points_in_boxes_cpu`
(compiled extension) which is answered by the oracle's restatement and only feeds `is_foreground`.
The reference never travels: only inputs and outputs are stored.
Run: python tests/golden/make_golden_loader.py
"""
import logging
import os
import pickle
import shutil
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

TAG = "waymo_processed_data_v0_5_0"
SEQ = "segment-1005081002024129653_5313_150_5333_150_with_camera_labels"


def _install_shims():
    for name in ("skimage", "skimage.transform", "torch_cluster", "torch_scatter", "SharedArray"):
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    sys.modules["torch_cluster"].fps = None
    sys.modules["torch_cluster"].grid_cluster = None
    sys.modules["torch_cluster"].knn = None
    sys.modules["torch_scatter"].scatter = None

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = EasyDict(v) if isinstance(v, dict) and not isinstance(v, EasyDict) else v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    # namespace packages: the real files of pcdet.utils / pcdet.datasets resolve, the package
    # __init__ files (setup-generated version.py, every dataset of the zoo) do not run
    for name, path in (("pcdet", f"{REF}/pcdet"), ("pcdet.utils", f"{REF}/pcdet/utils"),
                       ("pcdet.datasets", f"{REF}/pcdet/datasets"),
                       ("pcdet.datasets.waymo", f"{REF}/pcdet/datasets/waymo"),
                       ("pcdet.datasets.processor", f"{REF}/pcdet/datasets/processor"),
                       ("pcdet.datasets.augmentor", f"{REF}/pcdet/datasets/augmentor"),
                       ("pcdet.ops", f"{REF}/pcdet/ops")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    roi_pkg = types.ModuleType("pcdet.ops.roiaware_pool3d")
    roi_pkg.__path__ = []
    roi = types.ModuleType("pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils")

    def points_in_boxes_cpu(points, boxes):
        points = np.asarray(points, dtype=np.float32)[:, :3]
        boxes = np.asarray(boxes, dtype=np.float32)[:, :7]
        if boxes.shape[0] == 0:
            return np.zeros((0, points.shape[0]), dtype=np.int32)
        return O.points_in_boxes(np.ascontiguousarray(points), np.ascontiguousarray(boxes))

    roi.points_in_boxes_cpu = points_in_boxes_cpu
    roi_pkg.roiaware_pool3d_utils = roi
    sys.modules["pcdet.ops.roiaware_pool3d"] = roi_pkg
    sys.modules["pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils"] = roi
    # the augmentor is only built for training; its module imports the whole augmentation zoo
    aug = types.ModuleType("pcdet.datasets.augmentor.data_augmentor")
    aug.DataAugmentor = None
    sys.modules["pcdet.datasets.augmentor.data_augmentor"] = aug
    return EasyDict


def pose_of(rng, f):
    """Smooth ego motion: yaw drift + forward translation, small roll / pitch, in f64."""
    yaw = 0.03 * f + 0.4
    pitch = 0.004 * np.sin(0.7 * f)
    roll = 0.003 * np.cos(0.5 * f)
    cz, sz, cy, sy, cx, sx = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
    Ry = np.array([[cy, 0, sy], [0, 1.0, 0], [-sy, 0, cy]])
    Rx = np.array([[1.0, 0, 0], [0, cx, -sx], [0, sx, cx]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [1000.0 + 1.1 * f * np.cos(yaw), -2500.0 + 1.1 * f * np.sin(yaw), 30.0 + 0.01 * f]
    return T


def write_sequence(root, num_frames, seed, points_in_boxes=0, side_files=None):
    """points_in_boxes > 0: that many points of the first lidar are placed inside every present box (so that
    SYNC_MOVING_POINTS has something to move); side_files = (pred_dir, height_dir, filter_classes, truncate):
    per-frame prediction / height files for FILTER_FOREGROUND / REMOVE_GROUND."""
    rng = np.random.default_rng(seed)
    seq_dir = Path(root) / TAG / SEQ
    seq_dir.mkdir(parents=True)
    (Path(root) / "ImageSets").mkdir()
    (Path(root) / "ImageSets" / "val.txt").write_text(SEQ + ".tfrecord\n")
    names_pool = np.array(["Vehicle", "Pedestrian", "Cyclist", "Sign", "unknown"])
    n_obj = 9
    obj_names = names_pool[rng.integers(0, 5, n_obj)]
    obj_names[:3] = ["Vehicle", "Pedestrian", "Cyclist"]
    obj_ids = np.array([f"obj_{k:02d}_{rng.integers(1 << 30):x}" for k in range(n_obj)])
    obj_pos = rng.uniform(-30, 30, (n_obj, 3)) * [1, 1, 0.03]
    obj_vel = rng.uniform(-1, 1, (n_obj, 3)) * [1, 1, 0]
    obj_dim = rng.uniform(0.6, 5.0, (n_obj, 3))
    obj_yaw = rng.uniform(-np.pi, np.pi, n_obj)
    infos, inputs = [], {}
    for f in range(num_frames):
        counts = [int(rng.integers(220, 330)), int(rng.integers(10, 30)), int(rng.integers(0, 20)),
                  int(rng.integers(5, 25)), int(rng.integers(5, 25))]
        n = sum(counts)
        pts = np.zeros((n, 8), dtype=np.float32)
        pts[:, :3] = rng.uniform(-60, 60, (n, 3)) * [1, 1, 0.05]
        pts[:, 3] = rng.gamma(1.0, 0.8, n)          # raw intensity (tanh'd by get_lidar)
        pts[:, 4] = rng.uniform(0, 1.5, n)          # elongation
        pts[:, 5] = np.linalg.norm(pts[:, :3], axis=1)  # range in metres
        pts[:, 6] = rng.uniform(0, 1, n)            # range-image column / 2650
        pts[:, 7] = rng.integers(0, 64, n) / 64.0   # range-image row / 64
        seg = np.stack([rng.integers(0, 40, n), rng.integers(0, 23, n)], axis=1).astype(np.int32)
        present = rng.uniform(size=n_obj) < 0.8
        present[0] = True
        k = int(present.sum())
        boxes = np.concatenate([obj_pos[present] + obj_vel[present] * f, obj_dim[present],
                                (obj_yaw[present] + 0.02 * f)[:, None]], axis=1).astype(np.float32)
        if points_in_boxes > 0:
            at = 0
            for b in boxes:
                m = points_in_boxes
                loc = rng.uniform(-0.45, 0.45, (m, 3)) * b[3:6]
                c, s_ = np.cos(b[6]), np.sin(b[6])
                rot = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])
                pts[at:at + m, :3] = (loc @ rot.T + b[:3]).astype(np.float32)
                at += m
            pts[:, 5] = np.linalg.norm(pts[:, :3], axis=1)
        pose = pose_of(rng, f)
        if side_files is not None:
            pred_dir, height_dir, filter_classes, truncate = side_files
            n_kept = counts[0]  # DROP_POINTS_BY_LIDAR_INDEX [1, 2, 3, 4] keeps the first lidar's rows
            pred = rng.integers(0, 6, n_kept).astype(np.int32)
            n_fg = int((~np.isin(pred, filter_classes)).sum())
            height = rng.uniform(-0.5, 2.0, n_fg).astype(np.float32)
            (Path(pred_dir) / SEQ).mkdir(parents=True, exist_ok=True)
            (Path(height_dir) / SEQ).mkdir(parents=True, exist_ok=True)
            np.save(Path(pred_dir) / SEQ / f"{f:03d}_pred.npy", pred)
            np.save(Path(height_dir) / SEQ / ("%04d_point_height.npy" % f), height)
            inputs[f"pred_{f}"] = pred
            inputs[f"height_{f}"] = height
        np.save(seq_dir / ("%04d.npy" % f), pts)
        np.save(seq_dir / ("%04d_seg.npy" % f), seg)
        infos.append({
            "point_cloud": {"num_features": 5, "lidar_sequence": SEQ, "sample_idx": f},
            "frame_id": f"{SEQ}_{f:03d}",
            "metadata": {"context_name": SEQ, "timestamp_micros": 1000000 * f},
            "pose": pose,
            "num_points_of_each_lidar": counts,
            "annos": {"name": obj_names[present].copy(), "gt_boxes_lidar": boxes,
                      "obj_ids": obj_ids[present].copy(),
                      "num_points_in_gt": rng.integers(0, 50, k).astype(np.int64),
                      "difficulty": rng.integers(0, 3, k).astype(np.int64)},
        })
        inputs[f"raw_{f}"] = pts
        inputs[f"seg_{f}"] = seg
        inputs[f"pose_{f}"] = pose
        inputs[f"counts_{f}"] = np.array(counts, dtype=np.int64)
        inputs[f"boxes_{f}"] = boxes
        inputs[f"names_{f}"] = obj_names[present].astype(str)
        inputs[f"ids_{f}"] = obj_ids[present].astype(str)
        inputs[f"npts_{f}"] = infos[-1]["annos"]["num_points_in_gt"]
    with open(seq_dir / f"{SEQ}.pkl", "wb") as fh:
        pickle.dump(infos, fh)
    inputs["num_frames"] = np.int64(num_frames)
    return inputs


def dataset_cfg(EasyDict, root, num_sweeps, sweep_dir, sample_range):
    return EasyDict({
        "DATASET": "WaymoDataset", "DATA_PATH": str(root), "PROCESSED_DATA_TAG": TAG,
        "BOX_CLASSES": ["Vehicle", "Pedestrian", "Cyclist"],
        "POINT_CLOUD_RANGE": [-75.2, -75.2, -2, 75.2, 75.2, 4],
        "DATA_SPLIT": {"train": "train", "test": "val"},
        "NUM_SWEEPS": num_sweeps, "SWEEP_DIR": sweep_dir,
        "SAMPLED_INTERVAL": {"train": 1, "test": 1},
        "SAMPLED_SEQUENCE_INTERVAL": {"train": 1, "test": 1},
        "SAMPLE_IDX_RANGE": {"train": sample_range, "test": sample_range},
        "GT_VELOCITY": True, "SYNC_MOVING_POINTS": False, "EVALUATION_LIST": ["seg"],
        "SEGMENTATION_CFG": {"NUM_SEG_CLASSES": 23, "USE_ONLY_SAMPLES_WITH_SEG_LABELS": False, "LOAD_SEG": True},
        "FILTER_EMPTY_BOXES_FOR_TRAIN": True, "DISABLE_NLZ_FLAG_ON_POINTS": True,
        "USE_SHARED_MEMORY": False,
        "DATA_AUGMENTOR": {"DISABLE_AUG_LIST": ["placeholder"], "AUG_CONFIG_LIST": []},
        "DROP_POINTS_BY_LIDAR_INDEX": [1, 2, 3, 4],
        "POINT_FEATURE_ENCODING": {"encoding_type": "absolute_coordinates_encoding",
                                   "used_feature_list": ["intensity", "elongation", "range"],
                                   "src_feature_list": ["intensity", "elongation", "range", "rimage_w", "rimage_h"]},
        "MAX_NUM_POINTS": 40000000,
        "DATA_PROCESSOR": [EasyDict({"NAME": "transform_points_to_voxels", "VOXEL_SIZE": [0.1, 0.1, 0.15],
                                     "POINT_CLOUD_RANGE": [-75.2, -75.2, -2, 75.2, 75.2, 4],
                                     "MAX_POINTS_PER_VOXEL": 5,
                                     "MAX_NUMBER_OF_VOXELS": {"train": 150000, "test": 150000}, "DRY": True})],
    })


def run_reference(EasyDict, WaymoDataset, root, num_sweeps, sweep_dir, sample_range, overrides=None):
    log = logging.getLogger("golden")
    cfg = dataset_cfg(EasyDict, root, num_sweeps, sweep_dir, sample_range)
    for k, v in (overrides or {}).items():
        cfg[k] = EasyDict(v) if isinstance(v, dict) else v
    ds = WaymoDataset(cfg, training=False, root_path=Path(root), logger=log)
    assert len(ds) == 1
    sample = ds[0]
    batch = ds.collate_batch([sample])
    out = {}
    for k, v in batch.items():
        if k == "obj_ids":
            out["out_obj_ids"] = np.asarray(v[0]).astype(str)
        elif isinstance(v, np.ndarray):
            out["out_" + k] = v.astype(str) if v.dtype.kind in "OU" else v
        else:
            out["out_" + k] = np.asarray(v)
    return out


FILTER_CLASSES, TRUNCATE = [1, 4], 0.3


def main():
    EasyDict = _install_shims()
    import importlib
    wd = importlib.import_module("pcdet.datasets.waymo.waymo_dataset")
    only = sys.argv[1:]  # optional: names of the files to (re)generate
    cases = (("loader_fwd", 6, 5, 1, [0, 1], 11, {}), ("loader_bwd", 5, 4, -1, [4, 5], 12, {}),
             ("loader_sync", 6, 5, 1, [0, 1], 13, {"SYNC_MOVING_POINTS": True, "WITH_TIME_FEAT": True}),
             ("loader_filter", 5, 4, 1, [0, 1], 14, {"REMOVE_GROUND": True, "FILTER_FOREGROUND": True}))
    for name, frames, sweeps, direction, rng_, seed, opts in cases:
        if only and name not in only:
            continue
        root = tempfile.mkdtemp(prefix="pcs_golden_loader_")
        try:
            side = None
            overrides = dict(opts)
            if opts.get("REMOVE_GROUND"):
                side = (os.path.join(root, "pred"), os.path.join(root, "height"), FILTER_CLASSES, TRUNCATE)
                overrides["FILTER_FOREGROUND"] = {"PRED_LABEL_PATH": side[0], "FILTER_CLASS": FILTER_CLASSES}
                overrides["POINT_HEIGHT_PATH"] = Path(side[1])
                overrides["TRUNCATE_GROUND_HEIGHT"] = TRUNCATE
            inputs = write_sequence(root, frames, seed=seed, points_in_boxes=12 if opts.get("SYNC_MOVING_POINTS") else 0,
                                    side_files=side)
            out = run_reference(EasyDict, wd.WaymoDataset, root, sweeps, direction, rng_, overrides)
        finally:
            shutil.rmtree(root, ignore_errors=True)
        for k in ("SYNC_MOVING_POINTS", "WITH_TIME_FEAT", "REMOVE_GROUND", "FILTER_FOREGROUND"):
            inputs["opt_" + k] = np.bool_(bool(opts.get(k, False)))
        inputs["num_sweeps"] = np.int64(sweeps)
        inputs["sweep_dir"] = np.int64(direction)
        inputs["start_index"] = np.int64(rng_[0])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **inputs, **out)
        print(name, {k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
