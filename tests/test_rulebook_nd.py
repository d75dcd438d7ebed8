"""The N-D rulebook (get_indice_pairs_4d of the reference, src/all.cc:28-29): the loop-by-loop restatement of
include/spconv/geometry.h in oracle/oracle.py is pinned at three dimensions by the compiled, reference-pinned oracle;
the product's tensor-operation form (pcdet/ops/spconv/ops._indice_pairs_nd) is then held to it at four dimensions --
output ids, counts and the whole pair arrays, bit for bit. CPU only (the form is device-agnostic tensor code)."""
import numpy as np
import torch

from oracle import oracle as O
from pcseqlearning_amd.pcdet.ops.spconv import ops as S
from tests.golden.make_golden import random_indices

CASES3 = [(True, False, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1]),
          (False, False, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]),
          (False, False, [3, 1, 2], [1, 2, 1], [0, 0, 1], [2, 1, 1]),
          (False, False, [2, 3, 2], [2, 1, 2], [0, 1, 0], [1, 2, 1]),
          (False, True, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1])]
CASES4 = [(True, False, [3] * 4, [1] * 4, [1] * 4, [1] * 4),
          (False, False, [3] * 4, [2] * 4, [1] * 4, [1] * 4),
          (False, False, [2, 3, 1, 2], [1, 2, 1, 1], [0, 1, 0, 0], [2, 1, 1, 1]),
          (False, True, [3, 2, 3, 2], [2, 2, 2, 2], [1, 0, 1, 0], [1] * 4)]


def _out_shape(shape, ks, st, pd, dl, subm, tr):
    if subm:
        return list(shape)
    if tr:
        return S.get_deconv_output_size(shape, ks, st, pd, dl, [0] * len(shape))
    return S.get_conv_output_size(shape, ks, st, pd, dl)


def _same(a, b, subm):
    assert subm or np.array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[2], b[2])
    for k in range(len(a[2])):
        np.testing.assert_array_equal(a[1][k, :, :a[2][k]], b[1][k, :, :a[2][k]])


def test_loop_restatement_is_the_compiled_oracle_in_three_dimensions():
    rng = np.random.default_rng(4)
    shape = [7, 10, 9]
    idx = random_indices(rng, shape, 150, batch=2)
    for subm, tr, ks, st, pd, dl in CASES3:
        osh = _out_shape(shape, ks, st, pd, dl, subm, tr)
        _same(O.get_indice_pairs_nd_py(idx, osh, ks, st, pd, dl, subm, tr), O.get_indice_pairs(idx, osh, ks, st, pd, dl, subm, tr),
              subm)
        got = S._indice_pairs_nd(torch.from_numpy(idx), osh, ks, st, pd, dl, subm, tr)
        _same([g.numpy() for g in got], O.get_indice_pairs(idx, osh, ks, st, pd, dl, subm, tr), subm)


def random_indices_nd(rng, shape, n, batch):
    vol = int(np.prod(shape))
    cells = rng.choice(vol * batch, n, replace=False)
    cols = [cells // vol]
    rem = cells % vol
    for d in range(len(shape)):
        step = int(np.prod(shape[d + 1:]))
        cols.append(rem // step)
        rem = rem % step
    return np.stack(cols, 1).astype(np.int32)


def test_four_dimensional_rulebook_vs_loop_restatement():
    rng = np.random.default_rng(5)
    shape = [4, 6, 7, 5]
    idx = random_indices_nd(rng, shape, 300, 2)
    for subm, tr, ks, st, pd, dl in CASES4:
        osh = _out_shape(shape, ks, st, pd, dl, subm, tr)
        want = O.get_indice_pairs_nd_py(idx, osh, ks, st, pd, dl, subm, tr)
        got = [g.numpy() for g in S._indice_pairs_nd(torch.from_numpy(idx), osh, ks, st, pd, dl, subm, tr)]
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[2], want[2])
        np.testing.assert_array_equal(got[1], want[1])
        assert want[2].sum() > len(idx) // 2
    # duplicate input coordinates in a submanifold rulebook: the later row owns the cell (gridsOut[index] = j)
    dup = np.concatenate([idx[:50], idx[:10]])
    want = O.get_indice_pairs_nd_py(dup, shape, [3] * 4, [1] * 4, [1] * 4, [1] * 4, True)
    got = [g.numpy() for g in S._indice_pairs_nd(torch.from_numpy(dup), shape, [3] * 4, [1] * 4, [1] * 4, [1] * 4, True, False)]
    np.testing.assert_array_equal(got[1], want[1])
    # empty input
    e = S._indice_pairs_nd(torch.zeros(0, 5, dtype=torch.int32), [2] * 4, [3] * 4, [2] * 4, [1] * 4, [1] * 4, False, False)
    assert e[0].shape == (0, 5) and int(e[2].sum()) == 0
