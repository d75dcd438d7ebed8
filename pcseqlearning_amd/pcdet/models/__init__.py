"""build_network / load_data_to_gpu of pcdet/models/__init__.py:16-63 for the registration path."""
import numpy as np
import torch

from .registration import build_registration


def build_network(model_cfg, cfg=None, dataset=None):
    runtime_cfg = {}
    if model_cfg["NAME"] in ("SimpleReg",):
        return build_registration(model_cfg=model_cfg, runtime_cfg=runtime_cfg, dataset=dataset)
    raise NotImplementedError(f"{model_cfg['NAME']}: only the registration path is provided")


def load_data_to_gpu(batch_dict, device="cuda"):
    """The host->device boundary: numeric numpy arrays become device tensors, the rest stays."""
    out = {}
    for key, val in batch_dict.items():
        if isinstance(val, np.ndarray) and val.dtype.kind in "fiub":
            out[key] = torch.from_numpy(val).to(device)
        elif isinstance(val, list) and val and isinstance(val[0], np.ndarray) and val[0].dtype.kind in "fiub":
            out[key] = [torch.from_numpy(v).to(device) for v in val]
        else:
            out[key] = val
    return out
