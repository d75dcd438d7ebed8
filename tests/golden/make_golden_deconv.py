"""Golden vectors of the transposed-conv rulebook from the REFERENCE's own getIndicePairsDeConv
(pcdet/ops/spconv/include/spconv/geometry.h:197-245, compiled in place by oracle/Makefile into
oracle/_ref/libref_spconv.so): deconvbook_*.npz = inputs + outputs. Data only.
Run: python tests/golden/make_golden_deconv.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests.golden.make_golden import random_indices  # noqa: E402

# (name, spatial_shape zyx, ksize, stride, padding, dilation, output_padding)
DECONV_CASES = (("deconv_k3_s2_p1", [6, 9, 8], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], [1, 1, 1]),
                ("deconv_k2_s2_p0", [5, 7, 6], [2, 2, 2], [2, 2, 2], [0, 0, 0], [1, 1, 1], [0, 0, 0]),
                ("deconv_k311_s211_p100", [7, 6, 5], [3, 1, 1], [2, 1, 1], [1, 0, 0], [1, 1, 1], [0, 0, 0]),
                ("deconv_k3_s1_p0_d2", [6, 6, 6], [3, 3, 3], [1, 1, 1], [0, 0, 0], [2, 2, 2], [0, 0, 0]))


def main():
    assert O.ref_available(), "oracle/_ref not built (needs /root/reference)"
    rng = np.random.default_rng(777)
    for name, shape, ks, st, pd, dl, opad in DECONV_CASES:
        idx = random_indices(rng, shape, 150, batch=2)
        out_shape = O.deconv_out_shape(shape, ks, st, pd, dl, opad)
        out_ids, pairs, num = O.ref_get_indice_pairs_deconv(idx, 2, out_shape, ks, st, pd, dl)
        np.savez_compressed(os.path.join(HERE, f"deconvbook_{name}.npz"), indices=idx, spatial_shape=np.array(shape),
                            out_shape=np.array(out_shape), ksize=np.array(ks), stride=np.array(st),
                            padding=np.array(pd), dilation=np.array(dl), out_ids=out_ids, pairs=pairs, num=num)
        print(name, "V", idx.shape[0], "out", out_ids.shape[0], "pairs", int(num.sum()))


if __name__ == "__main__":
    main()
