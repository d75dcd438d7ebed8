"""Golden vectors for the sparse max-pool (forward + backward), produced by the REFERENCE's own CPU
functors compiled in place (oracle/_ref/libref_maxpool.so <- pcdet/ops/spconv/src/maxpool.cc) driven
offset by offset as pool_ops.h:26-98 does, on the rulebook of a k3 s2 p1 pooling window. Features
are quantised so that ties occur (the backward hands the gradient to every tied input).
Run: python tests/golden/make_golden_maxpool.py
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def reference_maxpool(lib, feat, pairs, num, num_out, out_grad):
    C = feat.shape[1]
    out = np.zeros((num_out, C), np.float32)
    for k in range(pairs.shape[0]):
        if num[k] > 0:
            blk = np.ascontiguousarray(pairs[k])
            lib.ref_maxpool_fwd(_p(out), _p(feat), _p(blk), num_out, feat.shape[0], C, blk.shape[1], int(num[k]))
    din = np.zeros_like(feat)
    for k in range(pairs.shape[0]):
        if num[k] > 0:
            blk = np.ascontiguousarray(pairs[k])
            lib.ref_maxpool_bwd(_p(out), _p(feat), _p(out_grad), _p(din), _p(blk), num_out, feat.shape[0], C,
                                blk.shape[1], int(num[k]))
    return out, din


def make_case(seed, quant):
    rng = np.random.default_rng(seed)
    shape = [7, 20, 22]
    lin = rng.choice(int(np.prod(shape)), 700, replace=False)
    idx = np.concatenate([np.zeros((700, 1), np.int64), np.stack(np.unravel_index(lin, shape), 1)], 1).astype(np.int32)
    ks, st, pd = [3, 3, 3], [2, 2, 2], [1, 1, 1]
    oshape = O.conv_out_shape(shape, ks, st, pd, [1, 1, 1])
    out_idx, pairs, num = O.get_indice_pairs(idx, oshape, ks, st, pd, [1, 1, 1], False)
    feat = rng.normal(size=(700, 6)).astype(np.float32)
    if quant:
        feat = (np.round(feat * 2) / 2).astype(np.float32)  # few distinct values: ties, and negatives below the 0 floor
    og = rng.normal(size=(out_idx.shape[0], 6)).astype(np.float32)
    return dict(indices=idx, out_ids=out_idx, pairs=pairs, num=num, features=feat, out_grad=og)


def main():
    lib = O.ref_maxpool()
    assert lib is not None, "build oracle/_ref first (make -C oracle)"
    for name, seed, quant in (("maxpool_ties", 21, True), ("maxpool_float", 22, False)):
        c = make_case(seed, quant)
        out, din = reference_maxpool(lib, c["features"], c["pairs"], c["num"], c["out_ids"].shape[0], c["out_grad"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), out=out, in_grad=din, **c)
        print(name, out.shape, float((out == 0).mean()), din.shape)


if __name__ == "__main__":
    main()
