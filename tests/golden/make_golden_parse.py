"""Golden text outputs of the REFERENCE's tools/parse_cluster_tracking_results.py, run in the
authoring container on small synthetic `all.pth` result files (the generator below writes them, and
their tensors are committed alongside the expected text so the test can rebuild the files).
Modules absent from this image are provided in memory for the import only: `_init_path`,
`easydict`, `SharedArray` (imported, never called, by the reference's common_utils) and `pcdet.ops.roiaware_pool3d` (imported, never called, by the
reference's box_utils); box_utils / common_utils are the reference's own files.

  parse_results.npz   per sequence: attr, frame, trace_id, best_iou (+ the file order glob returned)
                      and the reference's `TLS_multiradius_every8.txt` / `compare.txt`
Run: python tests/golden/make_golden_parse.py
"""
import importlib.util
import os
import runpy
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def synth_sequences(seed=0, n_seq=5):
    rng = np.random.default_rng(seed)
    out = {}
    for s in range(n_seq):
        n_tr, n_fr = int(rng.integers(3, 9)), int(rng.integers(4, 12))
        attr, frame, trace = [], [], []
        for t in range(n_tr):
            p0 = rng.uniform(-40, 40, 3) * [1, 1, 0.02]
            v = rng.uniform(-1.2, 1.2, 2) if t % 3 else np.zeros(2)     # every third trace stands still
            size = rng.uniform([1.5, 0.6, 1.2], [5.0, 2.2, 2.0])
            frames = sorted(rng.choice(n_fr, size=int(rng.integers(1, n_fr + 1)), replace=False).tolist())
            for f in frames:
                yaw = float(np.arctan2(v[1], v[0]) + rng.normal(0, 0.02)) if t % 3 else 0.3 * t
                attr.append([p0[0] + v[0] * f, p0[1] + v[1] * f, p0[2], *size, yaw])
                frame.append(f)
                trace.append(t)
        order = rng.permutation(len(attr))                                # boxes in arbitrary order
        out[f"seq_{s:02d}"] = dict(
            attr=torch.tensor(np.asarray(attr, np.float32)[order]),
            frame=torch.tensor(np.asarray(frame, np.int64)[order]),
            trace_id=torch.tensor(np.asarray(trace, np.int64)[order]),
            best_iou=torch.tensor(rng.uniform(0, 1, len(attr)).astype(np.float32)[order]),
        )
    # one sequence without a moving box (skipped by the parser) and one with a single box
    out["seq_still"] = dict(attr=torch.tensor([[0, 0, 0, 4, 2, 1.5, 0.1]] * 3, dtype=torch.float32),
                            frame=torch.tensor([0, 1, 2]), trace_id=torch.tensor([0, 0, 0]),
                            best_iou=torch.tensor([0.9, 0.8, 0.75]))
    return out


def write_results(root, seqs):
    for name, d in seqs.items():
        os.makedirs(f"{root}/TLS_multiradius_every8/{name}", exist_ok=True)
        torch.save(d, f"{root}/TLS_multiradius_every8/{name}/all.pth")
    os.makedirs(f"{root}/some_other_algorithm/x", exist_ok=True)             # ignored by the parser


def main():
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in {**(d or {}), **kw}.items():
                self[k] = v
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    for name in ("_init_path", "SharedArray"):
        sys.modules[name] = types.ModuleType(name)
    m = types.ModuleType("easydict"); m.EasyDict = EasyDict; sys.modules["easydict"] = m
    pk = types.ModuleType("pcdet"); pk.__path__ = [f"{REF}/pcdet"]; sys.modules["pcdet"] = pk
    ops = types.ModuleType("pcdet.ops"); ops.__path__ = []; sys.modules["pcdet.ops"] = ops
    roi = types.ModuleType("pcdet.ops.roiaware_pool3d"); roi.roiaware_pool3d_utils = types.ModuleType("x")
    sys.modules["pcdet.ops.roiaware_pool3d"] = roi
    ut = types.ModuleType("pcdet.utils"); ut.__path__ = [f"{REF}/pcdet/utils"]; sys.modules["pcdet.utils"] = ut
    for name in ("common_utils", "box_utils"):
        spec = importlib.util.spec_from_file_location(f"pcdet.utils.{name}", f"{REF}/pcdet/utils/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"pcdet.utils.{name}"] = mod
        try:
            spec.loader.exec_module(mod)
        except ImportError as err:      # common_utils pulls optional packages at import time
            raise SystemExit(f"cannot import reference {name}: {err}")
        setattr(ut, name, mod)
    seqs = synth_sequences()
    with tempfile.TemporaryDirectory() as tmp:
        write_results(f"{tmp}/res", seqs)
        argv = sys.argv
        sys.argv = ["parse", "--result_dir", f"{tmp}/res", "--output_dir", f"{tmp}/out"]
        try:
            runpy.run_path(f"{REF}/tools/parse_cluster_tracking_results.py", run_name="__main__")
        finally:
            sys.argv = argv
        txt = open(f"{tmp}/out/TLS_multiradius_every8.txt").read()
        cmp_txt = open(f"{tmp}/out/compare.txt").read()
    arrays = {}
    for name, d in seqs.items():
        for k, v in d.items():
            arrays[f"{name}/{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "parse_results.npz"), algorithm_txt=np.array(txt),
                        compare_txt=np.array(cmp_txt), names=np.array(sorted(seqs)), **arrays)
    print(txt)


if __name__ == "__main__":
    main()
