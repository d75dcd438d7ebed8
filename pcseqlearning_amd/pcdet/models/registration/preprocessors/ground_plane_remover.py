"""GroundPlaneRemover (pcdet/models/registration/preprocessors/ground_plane_remover.py:185-249):
estimates the ground height per 2 m pillar, keeps `full_*` copies of the per-point arrays and
drops the points lower than TRUNCATE_HEIGHT above ground. Caches the pillar grids in
`<DIR>/<seq>/pillar_height.pth` like the reference (its path-level resume)."""
import os

import torch
import torch.nn as nn

from ....utils.common_utils import EasyDict
from ....utils.timer import Timer
from .preprocessor_utils import ground_plane_removal


class GroundPlaneRemover(nn.Module):
    def __init__(self, model_cfg, runtime_cfg):
        super().__init__()
        self.model_cfg = model_cfg
        self.forward_dict = EasyDict()
        self.save = model_cfg.get("SAVE", True)

    def forward(self, seq_dict):
        sequence_id = seq_dict["frame_id"][0][:-4]
        point_fxyz = seq_dict["point_fxyz"]
        path = f"{self.model_cfg['DIR']}/{sequence_id}"
        cache = f"{path}/pillar_height.pth"
        if self.save and os.path.exists(cache):
            saved = torch.load(cache, map_location=point_fxyz.device)
            out = ground_plane_removal(point_fxyz, self.model_cfg, warmup=saved)
        else:
            with Timer("Ground Removal"):
                out = ground_plane_removal(point_fxyz, self.model_cfg)
            if self.save:
                os.makedirs(path, exist_ok=True)
                torch.save(dict(pillar_height=out[3], pillar_min_z=out[4]), cache)
        point_height, point_horizon, point_error, pillar_height, pillar_min_z = out
        seq_dict["point_horizon"] = point_horizon
        seq_dict["point_error"] = point_error
        ground_mask = None
        for height in self.model_cfg["TRUNCATE_HEIGHT"]:
            ground_mask = point_height < height
        seq_dict["point_height"] = point_height
        seq_dict["pillar_height"] = pillar_height
        seq_dict["pillar_min_z"] = pillar_min_z
        kept = (~ground_mask).nonzero(as_tuple=True)[0]  # (one compaction of the mask for all six arrays)
        for key in ("point_fxyz", "segmentation_label", "point_sweep", "point_height",
                    "instance_label", "point_horizon"):
            if key in seq_dict:
                seq_dict[f"full_{key}"] = seq_dict[key].clone()
                seq_dict[key] = seq_dict[key][kept]
        return seq_dict

    def get_output_feature_dim(self):
        return 0
