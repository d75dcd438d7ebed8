from .simple_reg import SimpleReg

__all__ = {
    "SimpleReg": SimpleReg,
}


def build_registration(model_cfg, runtime_cfg, dataset):
    return __all__[model_cfg["NAME"]](model_cfg=model_cfg, runtime_cfg=runtime_cfg, dataset=dataset)
