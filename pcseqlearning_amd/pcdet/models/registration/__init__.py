"""Registration models by name. The reference keeps a module-level dict called `__all__`
(pcdet/models/registration/__init__.py) that `build_network` indexes with MODEL.NAME; the same
name and call signature are kept here so that configs and callers carry over unchanged."""
from . import simple_reg

REGISTRY = {cls.__name__: cls for cls in (simple_reg.SimpleReg,)}
SimpleReg = REGISTRY["SimpleReg"]
__all__ = REGISTRY


def build_registration(model_cfg, runtime_cfg, dataset):
    name = model_cfg["NAME"]
    if name not in REGISTRY:
        raise KeyError(f"unknown registration model {name!r}; available: {sorted(REGISTRY)}")
    return REGISTRY[name](model_cfg=model_cfg, runtime_cfg=runtime_cfg, dataset=dataset)
