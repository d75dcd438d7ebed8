"""`DatasetTemplate` with the reference's name and contract (pcdet/datasets/dataset.py:15-298) for
the test-mode registration path: `prepare_data` (class labels, `gt_boxes`, feature selection) and
`collate_batch`. Augmentation, voxelisation in the loader and image keys are out of scope and
refused loudly."""
from collections import defaultdict
from pathlib import Path

import numpy as np
import torch

# collate_batch key classes (dataset.py:209-292), restricted to the keys this path produces
_CONCAT_KEYS = {"point_sweep", "point_feat", "segmentation_label", "is_foreground", "instance_label",
                "point_ground_mask", "point_rimage_w", "point_rimage_h"}
_BOX_KEYS = {"gt_boxes", "gt_box_attr", "gt_box_cls_label", "difficulty", "num_points_in_gt", "augmented",
             "obj_sweep", "gt_box_corners_3d", "gt_box_velo"}
_INT_BOX_KEYS = {"gt_box_cls_label", "difficulty", "num_points_in_gt", "obj_sweep"}


def _cat(vals):
    return torch.cat(vals, dim=0) if torch.is_tensor(vals[0]) else np.concatenate(vals, axis=0)


class DatasetTemplate(torch.utils.data.Dataset):
    def __init__(self, dataset_cfg=None, training=True, root_path=None, logger=None):
        super().__init__()
        self.dataset_cfg = dataset_cfg
        if dataset_cfg is None:
            return
        self.training = training
        self.logger = logger
        self.box_classes = dataset_cfg.get("BOX_CLASSES", None)
        seg_cfg = dataset_cfg.get("SEGMENTATION_CFG", {})
        self.segmentation_cfg = seg_cfg
        self.num_seg_classes = seg_cfg.get("NUM_SEG_CLASSES", None)
        self.use_only_samples_with_seg_labels = seg_cfg.get("USE_ONLY_SAMPLES_WITH_SEG_LABELS", False)
        self.load_seg = seg_cfg.get("LOAD_SEG", False)
        self.max_num_points = dataset_cfg.get("MAX_NUM_POINTS", 200000)
        self.root_path = Path(root_path) if root_path is not None else Path(dataset_cfg["DATA_PATH"])
        self.drop_points_by_lidar_index = dataset_cfg.get("DROP_POINTS_BY_LIDAR_INDEX", None)
        enc = dataset_cfg.get("POINT_FEATURE_ENCODING", None)
        if enc is not None:
            if enc["encoding_type"] != "absolute_coordinates_encoding":
                raise NotImplementedError(f"POINT_FEATURE_ENCODING.encoding_type={enc['encoding_type']}")
            self.used_feature_list = list(enc["used_feature_list"])
            self.src_feature_list = list(enc["src_feature_list"])
            self.num_point_features = len(self.used_feature_list)
        else:
            self.used_feature_list = self.src_feature_list = None
            self.num_point_features = dataset_cfg.get("NUM_POINT_FEATURES", 0)
        pcr = dataset_cfg.get("POINT_CLOUD_RANGE", None)
        self.point_cloud_range = np.array(pcr, dtype=np.float32) if isinstance(pcr, list) else pcr
        self.grid_size = self.voxel_size = None
        for proc in dataset_cfg.get("DATA_PROCESSOR", []):
            # data_processor.py:163-171: the voxel grid is declared here even when the loader does
            # not voxelise (DRY); anything that would change the sample is refused
            if proc["NAME"] == "transform_points_to_voxels" and proc.get("DRY", False):
                rng = np.array(proc["POINT_CLOUD_RANGE"])
                self.grid_size = np.round((rng[3:6] - rng[0:3]) / np.array(proc["VOXEL_SIZE"])).astype(np.int64)
                self.voxel_size = proc["VOXEL_SIZE"]
            else:
                raise NotImplementedError(f"DATA_PROCESSOR {proc['NAME']} (only a DRY transform_points_to_voxels)")
        self.total_epochs = 0
        self._merge_all_iters_to_one_epoch = False
        self.runtime_cfg = dict(point_cloud_range=self.point_cloud_range, num_point_features=self.num_point_features,
                                grid_size=self.grid_size, voxel_size=self.voxel_size, box_classes=self.box_classes,
                                num_seg_classes=self.num_seg_classes)

    @property
    def mode(self):
        return "train" if self.training else "test"

    def encode_point_feat(self, point_feat):
        """PointFeatureEncoder.absolute_coordinates_encoding (point_feature_encoder.py:41-55)."""
        if self.used_feature_list is None:
            return point_feat
        cols = [self.src_feature_list.index(x) for x in self.used_feature_list]
        if cols == list(range(point_feat.shape[1])):
            return point_feat
        return point_feat[:, cols]

    def prepare_data(self, data_dict):
        """dataset.py:143-191, test mode (augmentation is a training-side subsystem, out of scope)."""
        if self.training:
            raise NotImplementedError("training-mode prepare_data (data augmentor) is out of scope")
        ow = data_dict["object_wise"]
        if ow.get("gt_box_attr", None) is not None:
            names = ow["gt_box_cls_label"]
            if names.dtype != np.int32:
                names = np.array([self.box_classes.index(n) + 1 if len(n) > 0 else 0 for n in names], dtype=np.int32)
            ow["gt_box_cls_label"] = names
            ow["gt_boxes"] = np.concatenate((ow["gt_box_attr"], names.reshape(-1, 1).astype(np.float32)), axis=1)
        pw = data_dict["point_wise"]
        if pw.get("point_feat", None) is not None:
            pw["point_feat"] = self.encode_point_feat(pw["point_feat"])
        return data_dict

    @staticmethod
    def collate_batch(batch_list, _unused=False, num_mix3d_samples=1):
        """dataset.py:193-298 for the keys of this path. Point-wise values may be numpy arrays (the
        reference's contract) or device tensors (WaymoDataset.load_sample_to_gpu): concatenation and
        the batch-index column keep them where they are."""
        data = defaultdict(lambda: defaultdict(list))
        for sample in batch_list:
            for key0, val0 in sample.items():
                for key, val in val0.items():
                    data[key0][key].append(val)
        B = len(batch_list)
        ret = {}
        for key0, val0 in data.items():
            for key, val in val0.items():
                if key in _CONCAT_KEYS:
                    ret[key] = _cat(val)
                elif key == "point_xyz":
                    coors = []
                    for i, c in enumerate(val):
                        if torch.is_tensor(c):
                            coors.append(torch.nn.functional.pad(c, (1, 0), mode="constant", value=float(i)))
                        else:
                            coors.append(np.pad(c, ((0, 0), (1, 0)), mode="constant", constant_values=i))
                    ret["point_bxyz"] = _cat(coors)
                elif key == "point_bxyz":  # already carries its batch index (device path)
                    ret[key] = _cat(val)
                elif key in _BOX_KEYS:
                    if key in _INT_BOX_KEYS:
                        val, dtype = [v.reshape(-1, 1) for v in val], np.int32
                    elif key == "augmented":
                        val, dtype = [v.reshape(-1, 1) for v in val], bool
                    else:
                        dtype = np.float32
                    max_gt = max(len(x) for x in val)
                    out = np.zeros((B, max_gt, val[0].shape[-1]), dtype=dtype)
                    for k in range(B):
                        out[k, :len(val[k]), :] = val[k]
                    ret[key] = out
                elif key == "obj_ids":
                    ret["obj_ids"] = val
                elif key in ("num_points_in_box", "voxel_instance_label"):
                    continue
                else:
                    ret[key] = np.stack(val, axis=0)
        ret["batch_size"] = B
        return ret
