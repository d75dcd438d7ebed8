"""The CPU twins of include/pcs_cpu.h beside the HIP entries they mirror: the SAME argument list goes to both (host
pointers to the twin, device pointers to libpcs_hip.so) and the outputs are compared -- exactly for index / integer
results and for the fp32 results whose order of operations the header fixes, within a stated tolerance where the device
sums in another order. The twins are test infrastructure (oracle/): this file is the only place both are called."""
import ctypes
import os

import numpy as np
import pytest
import torch

from pcseqlearning_amd import lib as L

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda:0")


class Buf:
    """one argument array: its host copy (the twin's) and its device copy (the entry's)"""

    def __init__(self, a):
        self.h = np.array(a, order="C", copy=True)  # (the twin writes into it)
        self.d = torch.from_numpy(self.h.copy()).to(DEV)

    def device(self):
        return self.d.cpu().numpy()


class Ws:
    def __init__(self, nbytes):
        self.n = int(nbytes)


def _cpu():
    from oracle import oracle as O
    O.lib()
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "libpcs_oracle.so"))


def both(name, *args):
    """pcs_<name>(args..., stream) on the device and pcs_cpu_<name>(args..., NULL) on the host"""
    g, c, keep = [], [], []
    for a in args:
        if isinstance(a, Buf):
            g.append(L.ptr(a.d))
            c.append(a.h.ctypes.data_as(ctypes.c_void_p))
        elif isinstance(a, Ws):
            ws = L.workspace(max(a.n, 16), DEV)
            keep.append(ws)
            g += [L.ptr(ws), ctypes.c_size_t(max(a.n, 16))]
            c += [None, ctypes.c_size_t(0)]
        else:
            g.append(a)
            c.append(a)
    L.check(getattr(L.load(), "pcs_" + name)(*g, L.stream()), "pcs_" + name)
    assert getattr(_cpu(), "pcs_cpu_" + name)(*c, None) == 0, "pcs_cpu_" + name
    torch.cuda.synchronize()


def same(*bufs):
    for b in bufs:
        np.testing.assert_array_equal(b.device(), b.h)


I64, I32, F32 = (lambda x: ctypes.c_int64(int(x))), (lambda x: ctypes.c_int(int(x))), (lambda x: ctypes.c_float(float(x)))


def test_segment_twins_beside_the_entries():
    rng = np.random.default_rng(61)
    n = 20000
    ids = rng.integers(0, 1500, n)
    order = np.argsort(ids, kind="stable").astype(np.int32)
    uniq, cnt = np.unique(ids, return_counts=True)
    v = len(uniq)
    seg = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    order, seg = Buf(order), Buf(seg)
    for last in (0, 1):
        out = Buf(np.zeros(v, np.int64))
        both("segment_pick", order, seg, I64(v), I32(last), out)
        same(out)
    data, med = Buf(rng.integers(-40, 40, n).astype(np.int64)), Buf(np.zeros(v, np.int64))
    both("segment_median_i64", data, order, seg, I64(v), med)
    same(med)
    f = Buf(rng.normal(0, 1, (n, 3)).astype(np.float32))
    for is_max in (0, 1):
        mm = Buf(np.zeros((v, 3), np.float32))
        both("segment_minmax_f32", f, I32(3), order, seg, I64(v), I32(is_max), mm)
        same(mm)


def test_hash_table_twins_beside_the_entries():
    """The torch_hash drop-ins on the reference's table layout: each side fills its own table (the device's slot
    assignment follows its atomicCAS race, the host inserts in order -- no query result depends on it), then degree
    count, sorted fill, correspondence and points-in-radius must agree exactly."""
    rng = np.random.default_rng(62)
    N, M, D, r = 20000, 6000, 4, 0.5
    ref = np.concatenate([rng.integers(0, 3, (N, 1)), rng.uniform(0, 12, (N, 3))], 1).astype(np.float32)
    qry = np.concatenate([rng.integers(0, 3, (M, 1)), rng.uniform(0, 12, (M, 3))], 1).astype(np.float32)
    vs = np.array([1 - 1e-3, r, r, r], np.float32)
    lo = np.minimum(ref.min(0), qry.min(0)) - vs * 2
    hi = np.maximum(ref.max(0), qry.max(0)) + vs * 2
    dims = Buf(np.rint((hi - lo) / vs).astype(np.int64) + 3)
    rk = Buf(np.rint((ref - lo) / vs).astype(np.int64) + 1)
    qk = Buf(np.rint((qry - lo) / vs).astype(np.int64) + 1)
    ref, qry = Buf(ref), Buf(qry)
    H = 2 * N
    keys, vals, rev = Buf(np.full(H, -1, np.int64)), Buf(np.zeros((H, D), np.float32)), Buf(np.zeros(H, np.int64))
    both("hash_insert", keys, vals, rev, I64(H), dims, I32(D), rk, ref, I64(N))
    assert sorted(keys.device().tolist()) == sorted(keys.h.tolist())  # (the same keys, possibly in other slots)
    qmin, qmax = Buf(np.array([0, -1, -1, -1], np.int32)), Buf(np.array([0, 1, 1, 1], np.int32))
    table = (keys, vals, rev, I64(H), dims, I32(D), qk, qry, I64(M), qmin, qmax)
    rad = Buf(np.full(M, r, np.float32))
    deg, off, E = Buf(np.zeros(M, np.int32)), Buf(np.zeros(M, np.int32)), Buf(np.zeros(1, np.int64))
    nws = L.load().pcs_radius_graph_table_workspace_bytes(I64(M))
    both("radius_graph_table_count", *table, rad, I32(16), deg, off, E, Ws(nws))
    same(deg, off, E)
    assert E.h[0] > M
    edges, dist = Buf(np.zeros((int(E.h[0]), 2), np.int64)), Buf(np.zeros(int(E.h[0]), np.float32))
    both("radius_graph_table_fill", *table, rad, I32(1), deg, off, edges, dist)
    same(edges, dist)
    cor = Buf(np.zeros(M, np.int64))
    both("correspondence", *table, cor)
    same(cor)
    assert (cor.h >= 0).mean() > 0.9
    vis = Buf(np.zeros(N, np.int64))
    both("points_in_radius", *table, F32(0.3), vis)
    same(vis)
    assert 0 < vis.h.sum() < N


def test_sparse_conv_twins_beside_the_entries(oracle):
    rng = np.random.default_rng(63)
    from tests.golden.make_golden import random_indices
    shape = [11, 40, 40]
    idx = random_indices(rng, shape, 6000, batch=2)
    V, kv, cin, cout = len(idx), 27, 8, 16
    feat = Buf(rng.normal(0, 1, (V, cin)).astype(np.float32))
    W = Buf(rng.normal(0, 0.2, (kv, cin, cout)).astype(np.float32))
    _, pairs, num = oracle.get_indice_pairs(idx, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True)
    centre = int(np.argmax(num))
    pairs_b, num_b = Buf(pairs), Buf(num)
    table = Buf(np.zeros((V, kv), np.int32))
    both("build_neighbor_table", pairs_b, I64(pairs.shape[2]), num_b, I32(kv), I32(0), I64(V), table)
    same(table)
    gamma, beta = Buf(rng.uniform(0.5, 2, cout).astype(np.float32)), Buf(rng.normal(0, 1, cout).astype(np.float32))
    mean = Buf(rng.normal(0, 1, cout).astype(np.float32))
    inv_std = Buf((1 / np.sqrt(rng.uniform(0.5, 2, cout) + 1e-3)).astype(np.float32))
    scale, shift = Buf(np.zeros(cout, np.float32)), Buf(np.zeros(cout, np.float32))
    both("bn_fold_f32", gamma, beta, mean, inv_std, I32(cout), scale, shift)
    same(scale, shift)
    bias = Buf(rng.normal(0, 1, cout).astype(np.float32))
    out = Buf(np.zeros((V, cout), np.float32))
    # (the same order of offsets and channels on both sides; the device keeps one running sum per output where the
    #  reference -- and the twin -- round every offset's product sum before adding it: 1e-5, as tests/test_gpu_voxel_conv.py)
    both("indice_conv_table_f32", feat, I32(cin), W, I32(kv), I32(cout), I32(centre), table, I64(V), bias, scale, shift,
         I32(1), out)
    np.testing.assert_allclose(out.device(), out.h, rtol=1e-5, atol=2e-5)
    skip, out_r = Buf(rng.normal(0, 1, (V, cout)).astype(np.float32)), Buf(np.zeros((V, cout), np.float32))
    both("indice_conv_table_residual_f32", feat, I32(cin), W, I32(kv), I32(cout), I32(centre), table, I64(V), bias, scale,
         shift, skip, I32(1), out_r)
    np.testing.assert_allclose(out_r.device(), out_r.h, rtol=1e-5, atol=2e-5)
    # strided window: pooling and its backward exactly, the conv's backward within 1e-4 (two-stage sums on the device)
    oshape = oracle.conv_out_shape(shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1])
    outids, pairs, num = oracle.get_indice_pairs(idx, oshape, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False)
    Vo, stride = len(outids), pairs.shape[2]
    pairs_b, num_b = Buf(pairs), Buf(num)
    nt = L.load().pcs_neighbor_table_bytes
    nt.restype = ctypes.c_size_t
    pooled = Buf(np.zeros((Vo, cin), np.float32))
    both("indice_maxpool_f32", feat, I32(cin), pairs_b, I64(stride), num_b, I32(kv), I64(Vo), pooled, Ws(nt(I64(Vo), I32(kv))))
    same(pooled)
    pg, din = Buf(rng.normal(0, 1, (Vo, cin)).astype(np.float32)), Buf(np.zeros((V, cin), np.float32))
    both("indice_maxpool_backward_f32", feat, pooled, pg, I32(cin), pairs_b, I64(stride), num_b, I32(kv), I64(V), din,
         Ws(nt(I64(V), I32(kv))))
    same(din)
    og = Buf(rng.normal(0, 1, (Vo, cout)).astype(np.float32))
    gi, gw = Buf(np.zeros((V, cin), np.float32)), Buf(np.zeros((kv, cin, cout), np.float32))
    bw = L.load().pcs_indice_conv_backward_workspace_bytes
    bw.restype = ctypes.c_size_t
    both("indice_conv_backward_f32", feat, I32(cin), W, I32(kv), I32(cout), og, pairs_b, I64(stride), num_b, I64(V), I64(Vo),
         I32(0), I32(0), gi, gw, Ws(bw(I64(V), I64(stride), I32(kv), I32(cin), I32(cout))))
    np.testing.assert_allclose(gi.device(), gi.h, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gw.device(), gw.h, rtol=1e-4, atol=2e-4)
    dense, status = Buf(np.zeros((2, cin) + tuple(shape), np.float32)), Buf(np.zeros(1, np.int32))
    both("sparse_to_dense", feat, Buf(idx), I64(V), I32(cin), I32(2), I32(shape[0]), I32(shape[1]), I32(shape[2]), dense, status)
    same(dense, status)


def test_box_and_loader_twins_beside_the_entries():
    rng = np.random.default_rng(64)
    F, N = 6, 40000
    nb = [9, 0, 14, 3, 30, 1]
    off = np.concatenate([[0], np.cumsum(nb)]).astype(np.int32)
    boxes = Buf(np.concatenate([rng.uniform(-20, 20, (off[-1], 3)), rng.uniform(1, 8, (off[-1], 3)),
                                rng.uniform(-3.2, 3.2, (off[-1], 1))], 1).astype(np.float32))
    pts, frame = Buf(rng.uniform(-24, 24, (N, 3)).astype(np.float32)), Buf(rng.integers(0, F, N).astype(np.int64))
    bmax, offb = max(nb), Buf(off)
    member, first = Buf(np.zeros((N, bmax), np.int32)), Buf(np.zeros(N, np.int32))
    both("points_in_frame_boxes", pts, frame, I64(N), boxes, offb, I32(F), I32(bmax), member, first)
    same(member, first)
    assert member.h.sum() > 100
    G = 50
    group, counts = Buf(rng.integers(0, G, N).astype(np.int64)), Buf(np.zeros((G, bmax), np.int32))
    both("frame_box_counts", pts, frame, I64(N), boxes, offb, I32(F), I32(bmax), group, I64(G), counts)
    same(counts)
    assert counts.h.sum() == member.h.sum()
    S, n = 5, 30000
    so = Buf(np.array([0, 7000, 7000, 15000, 29000, 30000], np.int64))
    raw, seg = Buf(rng.normal(0, 20, (n, 8)).astype(np.float32)), Buf(rng.integers(0, 20, (n, 2)).astype(np.int32))
    T, sid = Buf(rng.normal(0, 1, (S, 3, 4))), Buf(np.arange(S, 0, -1).astype(np.int32))
    bxyz, feat = Buf(np.zeros((n, 4), np.float32)), Buf(np.zeros((n, 3), np.float32))
    sweep, sl, il = Buf(np.zeros(n, np.int32)), Buf(np.zeros(n, np.int32)), Buf(np.zeros(n, np.int32))
    both("assemble_sweeps", raw, seg, so, T, sid, I32(S), I64(n), I32(1), bxyz, feat, sweep, sl, il)
    same(bxyz, sweep, sl, il)
    # (tanh: the device's and the host's libm within one unit in the last place of the fp32 result)
    np.testing.assert_allclose(feat.device(), feat.h, rtol=1.3e-7, atol=0)


def test_smooth_velo_twin_beside_the_entry():
    """AdamW steps in fp32 on both sides (the twin is oracle/track_oracle.py's arithmetic bit for bit); the device sums the
    loss and forms the step in its own order: 2e-5, the bound tests/test_gpu_tracking.py holds the entry to."""
    rng = np.random.default_rng(65)
    C, Fv = 40, 17
    d = rng.normal(0, 1, (C, Fv, 3)).astype(np.float32)
    v0 = (d + rng.normal(0, 0.3, d.shape)).astype(np.float32)
    sw = L.load().pcs_smooth_velo_workspace_bytes
    sw.restype = ctypes.c_size_t
    for lo, hi in ((2, 12), (0, 16)):
        v, db = Buf(v0), Buf(d)
        both("smooth_velo", v, db, I64(C), I32(Fv), I32(lo), I32(hi), F32(1), F32(10), I32(300), F32(1e-3), Ws(sw(I64(C), I32(Fv))))
        assert np.abs(v.device() - v.h).max() < 2e-5
        assert np.abs(v.h - v0).max() > 1e-2


def test_radius_graph_twins_beside_the_entries():
    """pcs_radius_graph_search / _emit / _components and their twins on one cloud: each side keeps the edges in its own
    workspace between the calls; the edge lists (grouped by query, nearest first, ties by the reference index), the
    squared distances and the component labels at three cut radii must be identical."""
    rng = np.random.default_rng(66)
    n, K = 30000, 32
    centres = rng.uniform(0, 60, (900, 3))
    pts = Buf(np.concatenate([rng.integers(0, 2, (n, 1)), centres[rng.integers(0, 900, n)] + rng.normal(0, 0.3, (n, 3))],
                             1).astype(np.float32))
    lib, cpu = L.load(), _cpu()
    cpu.pcs_cpu_radius_graph_workspace_bytes.restype = ctypes.c_size_t
    nb_g = lib.pcs_radius_graph_workspace_bytes(I64(n), I64(n), I32(K))
    nb_c = cpu.pcs_cpu_radius_graph_workspace_bytes(I64(n), I64(n), I32(K))
    ws_g, ws_c = L.workspace(nb_g, DEV), np.zeros(nb_c, np.uint8)
    E_g, E_c = torch.zeros(1, dtype=torch.int64, device=DEV), np.zeros(1, np.int64)
    L.check(lib.pcs_radius_graph_search(L.ptr(pts.d), I64(n), L.ptr(pts.d), I64(n), F32(1.25), F32(0), I32(0), I32(0), I32(K), I32(1),
                                        L.ptr(E_g), L.ptr(ws_g), ctypes.c_size_t(nb_g), L.stream()), "search")
    hp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert cpu.pcs_cpu_radius_graph_search(hp(pts.h), I64(n), hp(pts.h), I64(n), F32(1.25), F32(0), I32(0), I32(0), I32(K), I32(1),
                                           hp(E_c), hp(ws_c), ctypes.c_size_t(nb_c), None) == 0
    E = int(E_g.item())
    assert E == E_c[0] and E > n
    eg = [torch.zeros(E, dtype=torch.int64, device=DEV), torch.zeros(E, dtype=torch.int64, device=DEV),
          torch.zeros(E, dtype=torch.float32, device=DEV)]
    ec = [np.zeros(E, np.int64), np.zeros(E, np.int64), np.zeros(E, np.float32)]
    L.check(lib.pcs_radius_graph_emit(I64(n), I64(n), I32(K), *(L.ptr(t) for t in eg), L.ptr(ws_g), ctypes.c_size_t(nb_g),
                                      L.stream()), "emit")
    assert cpu.pcs_cpu_radius_graph_emit(I64(n), I64(n), I32(K), *(hp(a) for a in ec), hp(ws_c), ctypes.c_size_t(nb_c), None) == 0
    for t, a in zip(eg, ec):
        np.testing.assert_array_equal(t.cpu().numpy(), a)
    cuts = np.array([1.25, 0.75, 0.25], np.float32)
    nb_cc = lib.pcs_radius_graph_components_workspace_bytes(I64(n), I32(3))
    ws_cc = L.workspace(nb_cc, DEV)
    comp_g, num_g = torch.zeros(3, n, dtype=torch.int64, device=DEV), torch.zeros(3, dtype=torch.int64, device=DEV)
    comp_c, num_c = np.zeros((3, n), np.int64), np.zeros(3, np.int64)
    L.check(lib.pcs_radius_graph_components(I64(n), I64(n), I32(K), hp(cuts), I32(3), L.ptr(comp_g), L.ptr(num_g), L.ptr(ws_g),
                                            ctypes.c_size_t(nb_g), L.ptr(ws_cc), ctypes.c_size_t(nb_cc), L.stream()), "components")
    assert cpu.pcs_cpu_radius_graph_components(I64(n), I64(n), I32(K), hp(cuts), I32(3), hp(comp_c), hp(num_c), hp(ws_c),
                                               ctypes.c_size_t(nb_c), None, ctypes.c_size_t(0), None) == 0
    np.testing.assert_array_equal(num_g.cpu().numpy(), num_c)
    np.testing.assert_array_equal(comp_g.cpu().numpy(), comp_c)
    assert num_c[0] < num_c[1] < num_c[2]
