"""RegistrationTemplate (pcdet/models/registration/registration_module_template.py:11-76): module
topology `preprocessors -> registration -> visualizer`; only the preprocessor stage exists on the
hot path (MODEL.PREPROCESSORS is an ordered list of {NAME: ...} dicts resolved through
`preprocessors.__all__`)."""
import torch.nn as nn

from . import preprocessors


class RegistrationTemplate(nn.Module):
    def __init__(self, model_cfg, runtime_cfg, dataset):
        super().__init__()
        self.model_cfg = model_cfg
        self.runtime_cfg = runtime_cfg or {}
        self.dataset = dataset
        self.module_topology = ["preprocessors"]
        self.preprocessors = None

    def build_networks(self):
        info = {"module_list": []}
        for name in self.module_topology:
            module, info = getattr(self, f"build_{name}")(model_info_dict=info)
        return info["module_list"]

    def build_preprocessors(self, model_info_dict):
        cfgs = self.model_cfg.get("PREPROCESSORS", None)
        if cfgs is None:
            return None, model_info_dict
        mods = []
        for cfg in cfgs:
            mod = preprocessors.__all__[cfg["NAME"]](model_cfg=cfg, runtime_cfg=self.runtime_cfg)
            model_info_dict["module_list"].append(mod)
            mods.append(mod)
        self.preprocessors = nn.ModuleList(mods)
        return self.preprocessors, model_info_dict
