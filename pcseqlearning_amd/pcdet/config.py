"""YAML -> attribute dict with `_BASE_CONFIG_` merging (pcdet/config.py:16-85), enough to load the
reference's model YAMLs, e.g. tools/cfgs/waymo_models/PCsequence/registration/
cluster_tracking_TLS_multiradius_every8.yaml."""
import os

import yaml

from .utils.common_utils import EasyDict


def _to_easydict(obj):
    if isinstance(obj, dict):
        return EasyDict({k: _to_easydict(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_to_easydict(v) for v in obj]
    return obj


def merge_new_config(config, new_config):
    if "_BASE_CONFIG_" in new_config:
        with open(new_config["_BASE_CONFIG_"]) as f:
            config.update(_to_easydict(yaml.safe_load(f)))
    for key, val in new_config.items():
        if isinstance(val, dict) and isinstance(config.get(key), dict):
            merge_new_config(config[key], val)
        else:
            config[key] = _to_easydict(val)
    return config


def cfg_from_yaml_file(cfg_file, config=None):
    config = EasyDict() if config is None else config
    with open(cfg_file) as f:
        merge_new_config(config, yaml.safe_load(f))
    return config


# the shipped model config of the hot path, so the GPU box needs no copy of the reference tree
CLUSTER_TRACKING_TLS_MULTIRADIUS_EVERY8 = os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "cfgs", "cluster_tracking_TLS_multiradius_every8.yaml")
