"""Golden vectors for the registration maths, produced by IMPORTING the reference's own Python
(pcdet/models/registration/preprocessors/registration_utils.py) in the authoring container and
running it on CPU tensors:

  reg_utils.npz   robust_mean / efficient_robust_mean / truncated_robust_mean / robust_median
  reg_icp_*.npz   register_to_next_frame on small synthetic clouds (inputs + T, l1, ratio, moved)

Third-party modules absent from this image are provided in-memory for the import only:
torch_scatter.scatter / segment_coo (-> torch index_add_/scatter_reduce, same documented
semantics), easydict.EasyDict (attribute dict). The graph object handed to
register_to_next_frame is the oracle's radius graph (the reference's own RadiusGraph needs its
CUDA extension). The reference never travels: only inputs and outputs are stored.
Run: python tests/golden/make_golden_reg.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def _install_shims():
    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
        assert dim == 0
        n = int(dim_size) if dim_size is not None else int(index.max()) + 1
        out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
        idx = index.reshape((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
        if reduce == "sum":
            return out.scatter_add_(0, idx, src)
        if reduce == "mean":
            out = out.scatter_add_(0, idx, src)
            cnt = torch.zeros(n, dtype=src.dtype).scatter_add_(0, index, torch.ones_like(index, dtype=src.dtype))
            return out / cnt.clamp(min=1).reshape((-1,) + (1,) * (src.dim() - 1))
        red = {"max": "amax", "min": "amin"}[reduce]
        return out.scatter_reduce_(0, idx, src, red, include_self=False)

    def segment_coo(src, index, dim_size=None, reduce="sum"):
        return scatter(src, index, 0, dim_size, reduce)

    ts.scatter, ts.segment_coo = scatter, segment_coo
    sys.modules["torch_scatter"] = ts

    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed

    # namespace packages so that `from pcdet.utils import common_utils` resolves to the real files
    for name, path in (("pcdet", f"{REF}/pcdet"), ("pcdet.utils", f"{REF}/pcdet/utils")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    sa = types.ModuleType("SharedArray")
    sys.modules["SharedArray"] = sa
    return EasyDict


def load_reference():
    EasyDict = _install_shims()
    spec = importlib.util.spec_from_file_location(
        "ref_registration_utils", f"{REF}/pcdet/models/registration/preprocessors/registration_utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, EasyDict


class OracleGraph:
    """Duck-type of graph_utils.RadiusGraph for register_to_next_frame (radius/qmin/qmax mutable)."""

    def __init__(self, radius):
        self.radius = radius
        self.qmin = torch.tensor([0, -1, -1, -1]).int()
        self.qmax = torch.tensor([0, 1, 1, 1]).int()

    def __call__(self, ref, query):
        r = float((query.fxyz.new_zeros(1) + self.radius)[0])  # fp32 rounding as build_graph does
        e_ref, e_q = O.radius_graph(ref.fxyz.numpy(), query.fxyz.numpy(), r, 1, True,
                                    self.qmin.tolist(), self.qmax.tolist())
        return torch.from_numpy(e_ref), torch.from_numpy(e_q), None


def make_pair(rng, n_obj=6, pts_per_obj=120, frame_a=3, frame_b=4, shift=0.25, rot=0.03, extra_static=True):
    """Two frames of a few rigid blobs; frame_b = blobs moved by a small rigid motion + noise."""
    mov, comp, ref = [], [], []
    for c in range(n_obj):
        centre = rng.uniform(-20, 20, 3) * np.array([1, 1, 0.05])
        size = rng.uniform(0.4, 2.0, 3)
        pts = centre + (rng.uniform(-0.5, 0.5, (pts_per_obj, 3)) * size)
        a = rng.normal(0, rot)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        t = rng.normal(0, shift, 3) * np.array([1, 1, 0.1])
        moved = (pts - centre) @ R.T + centre + t + rng.normal(0, 0.01, pts.shape)
        keep = rng.uniform(size=pts_per_obj) < 0.85
        mov.append(pts); comp.append(np.full(pts_per_obj, c)); ref.append(moved[keep])
    mov = np.concatenate(mov); comp = np.concatenate(comp); ref = np.concatenate(ref)
    moving = dict(fxyz=np.concatenate([np.full((len(mov), 1), frame_a), mov], 1).astype(np.float32),
                  component=comp.astype(np.int64), frame=np.full(len(mov), frame_a, np.int64),
                  stationary=np.zeros(len(mov), bool))
    refd = dict(fxyz=np.concatenate([np.full((len(ref), 1), frame_b), ref], 1).astype(np.float32),
                frame=np.full(len(ref), frame_b, np.int64), stationary=np.zeros(len(ref), bool))
    if extra_static:  # a stationary component and stationary reference points (excluded from ICP)
        moving["stationary"][comp == n_obj - 1] = True
        refd["stationary"][rng.uniform(size=len(ref)) < 0.1] = True
    return moving, refd


def main():
    mod, EasyDict = load_reference()
    rng = np.random.default_rng(2024)
    # --- small helpers
    C = 7
    idx = rng.integers(0, C - 1, 500)          # segment C-1 stays empty
    data = rng.normal(0, 1, (500, 3))
    ints = rng.integers(-5, 40, 500)
    t = torch.from_numpy
    out = dict(index=idx, data=data, ints=ints, C=np.array(C),
               robust_mean=mod.robust_mean(t(data), t(idx), C).numpy(),
               eff_mean=mod.efficient_robust_mean(t(data), t(idx), C).numpy(),
               eff_sum=mod.efficient_robust_sum(t(data), t(idx), C).numpy(),
               trunc_mean=mod.truncated_robust_mean(t(np.abs(data[:, 0])), t(idx), C).numpy(),
               median=mod.robust_median(t(ints), t(idx), C).numpy())
    np.savez_compressed(os.path.join(HERE, "reg_utils.npz"), **out)
    # --- ICP
    cases = [("near", dict(frame_a=3, frame_b=4, shift=0.25), 2.5),
             ("far", dict(frame_a=8, frame_b=3, shift=0.15), 1.25),
             ("fine", dict(frame_a=0, frame_b=1, shift=0.05, rot=0.01), 1.0)]
    for name, kw, radius in cases:
        moving, refd = make_pair(rng, **kw)
        Cn = int(moving["component"].max()) + 1
        m = EasyDict({k: t(v.copy()) for k, v in moving.items()})
        r = EasyDict({k: t(v.copy()) for k, v in refd.items()})
        g = OracleGraph(radius)
        moved, T, l1, ratio = mod.register_to_next_frame(g, m, r, Cn, 10, max_iter=80, stopping_delta=0.05)
        np.savez_compressed(os.path.join(HERE, f"reg_icp_{name}.npz"),
                            mov_fxyz=moving["fxyz"], mov_comp=moving["component"],
                            mov_frame=moving["frame"], mov_stationary=moving["stationary"],
                            ref_fxyz=refd["fxyz"], ref_frame=refd["frame"],
                            ref_stationary=refd["stationary"], radius=np.array(radius),
                            C=np.array(Cn), T=T.numpy(), l1=l1.numpy(), ratio=ratio.numpy(),
                            moved=moved.fxyz.numpy())
        print(name, "C", Cn, "T00", T[0, :3, 3].numpy(), "l1", l1[:3].numpy(), "ratio", ratio[:3].numpy())


if __name__ == "__main__":
    main()
