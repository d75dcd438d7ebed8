"""include/pcs_cpu.h: every declared CPU twin is exported by the oracle libraries with the argument
list of its pcs_* counterpart, and gives the oracle's results (CPU only)."""
import ctypes
import os
import re

import numpy as np

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pcs_cpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcs_cpu_\w+)\s*\(", text)))


def _libs():
    O.lib()
    from oracle import reg_oracle as R
    R._icp_lib()
    return (ctypes.CDLL(os.path.join(ROOT, "oracle", "libpcs_oracle.so")),
            ctypes.CDLL(os.path.join(ROOT, "oracle", "libpcs_oracle_icp.so")))


def test_every_twin_is_exported_and_mirrors_a_pcs_symbol():
    from pcseqlearning_amd import lib as L
    names = _declared()
    assert len(names) >= 10
    main, icp = _libs()
    gpu = set(L.declared_symbols())
    for n in names:
        assert hasattr(main, n) or hasattr(icp, n), n
        assert n.replace("pcs_cpu_", "pcs_") in gpu, f"{n} has no pcs_* counterpart in include/pcs.h"


def test_twins_give_the_oracle_results():
    main, icp = _libs()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(5)
    # voxel keys -> unique -> means (GridSampling3D)
    pts = np.concatenate([rng.integers(0, 3, (4000, 1)), rng.uniform(-5, 5, (4000, 3))], 1).astype(np.float32)
    size4 = np.array([1, 0.4, 0.4, 0.3], np.float32)
    b8 = np.zeros(8, np.float32)
    assert main.pcs_cpu_minmax4_f32(p(pts), ctypes.c_int64(len(pts)), p(b8), None) == 0
    ids = np.zeros(len(pts), np.int64)
    assert main.pcs_cpu_grid_cluster_ids(p(pts), ctypes.c_int64(len(pts)), p(b8), p(size4), ctypes.c_float(0.5), p(ids), None) == 0
    want_ids, _ = O.grid_cluster_ids(pts, size4.tolist(), 0.5)
    np.testing.assert_array_equal(ids, want_ids)
    n = len(ids)
    uniq, inv = np.zeros(n, np.int64), np.zeros(n, np.int64)
    order, seg = np.zeros(n, np.int32), np.zeros(n + 1, np.int32)
    d_num = np.zeros(1, np.int64)
    assert main.pcs_cpu_unique_i64(p(ids), ctypes.c_int64(n), ctypes.c_int(64), p(uniq), p(inv), p(order), p(seg), p(d_num),
                                   None, ctypes.c_size_t(0), None) == 0
    w_uniq, w_inv, w_order, w_seg = O.unique_i64(ids)
    v = int(d_num[0])
    assert v == len(w_uniq)
    np.testing.assert_array_equal(uniq[:v], w_uniq)
    np.testing.assert_array_equal(inv, w_inv)
    means = np.zeros((v, 4), np.float32)
    assert main.pcs_cpu_segment_reduce_f32(p(pts), ctypes.c_int(4), p(order), p(seg), ctypes.c_int64(v), ctypes.c_int(1),
                                           p(means), None) == 0
    np.testing.assert_array_equal(means, O.segment_reduce(pts, w_order, w_seg, mean=True))
    # rulebook + conv (subM 3x3x3)
    from tests.golden.make_golden import random_indices
    shape = [9, 20, 20]
    idx = random_indices(rng, shape, 500, batch=2)
    ints = lambda a: (ctypes.c_int * 3)(*a)
    V, kv = len(idx), 27
    pairs, num, dn = np.zeros((kv, 2, V), np.int32), np.zeros(kv, np.int32), np.zeros(1, np.int64)
    assert main.pcs_cpu_get_indice_pairs(p(idx), ctypes.c_int64(V), ctypes.c_int(2), ints(shape), ints(shape), ints([3, 3, 3]),
                                         ints([1, 1, 1]), ints([1, 1, 1]), ints([1, 1, 1]), ctypes.c_int(1), ctypes.c_int(0),
                                         None, ctypes.c_int64(0), p(pairs), p(num), p(dn), None, ctypes.c_size_t(0), None) == 0
    _, w_pairs, w_num = O.get_indice_pairs(idx, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True)
    np.testing.assert_array_equal(pairs, w_pairs)
    np.testing.assert_array_equal(num, w_num)
    feat = rng.normal(0, 1, (V, 8)).astype(np.float32)
    W = rng.normal(0, 0.1, (27, 8, 16)).astype(np.float32)
    out = np.zeros((V, 16), np.float32)
    assert main.pcs_cpu_indice_conv_f32(p(feat), ctypes.c_int(8), p(W), ctypes.c_int(27), ctypes.c_int(16), p(pairs),
                                        ctypes.c_int64(V), p(num), ctypes.c_int64(V), ctypes.c_int(0), ctypes.c_int(1), p(out),
                                        None, ctypes.c_size_t(0), None) == 0
    np.testing.assert_array_equal(out, O.indice_conv(feat, W, w_pairs, w_num, V, False, True))
    # registration
    from oracle import reg_oracle as R
    from tests.golden.make_golden_reg import make_pair
    moving, refd = make_pair(np.random.default_rng(7), n_obj=5, pts_per_obj=80, frame_a=2, frame_b=3, shift=0.2)
    C = int(moving["component"].max()) + 1
    _, w_T, w_l1, w_ratio, w_it = R.register_to_next_frame_cpp(1.25, moving, refd, C, 10.0, 80, 0.05)
    nsm, nsr = ~moving["stationary"], ~refd["stationary"]
    mov = np.ascontiguousarray(moving["fxyz"][nsm], np.float32).copy()
    comp = np.ascontiguousarray(moving["component"][nsm], np.int32)
    r_ns, r_all = np.ascontiguousarray(refd["fxyz"][nsr], np.float32), np.ascontiguousarray(refd["fxyz"], np.float32)
    deg = np.bincount(moving["component"], minlength=C).astype(np.int64)
    T, l1, ratio, info = np.zeros((C, 16)), np.zeros(C), np.zeros(C, np.float32), np.zeros(4, np.int32)
    assert icp.pcs_cpu_register_to_next_frame(
        p(mov), p(comp), ctypes.c_int64(len(mov)), p(r_ns), ctypes.c_int64(len(r_ns)), p(r_all), ctypes.c_int64(len(r_all)),
        p(deg), ctypes.c_int64(C), ctypes.c_double(1.25), ctypes.c_int(1), ctypes.c_double(10.0), ctypes.c_int(80),
        ctypes.c_double(0.05), ctypes.c_float(0), ctypes.c_int(8), p(T), p(l1), p(ratio), p(info), None, ctypes.c_size_t(0),
        None) == 0
    assert int(info[0]) == w_it
    np.testing.assert_array_equal(T.reshape(C, 4, 4), w_T)
    np.testing.assert_array_equal(ratio, w_ratio)


def test_ground_stage_twins():
    """The three ground-stage twins (round 5) behind their entries' argument lists: the plane hypotheses of a few
    synthetic pillars (a tilted plane + clutter: the fitted normal is the plane's), the pruning loop (an outlier
    hypothesis goes, the coplanar ones stay) and the L1 smoothing (a flat field stays flat, a spike is pulled in)."""
    main, _ = _libs()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(8)
    P, per = 6, 300
    xyz, seg = [], [0]
    for c in range(P):
        xy = rng.uniform(0, 8, (per, 2)) + np.array([8.0 * c, 0.0])
        z = 0.05 * xy[:, 0] - 0.02 * xy[:, 1] + rng.normal(0, 0.005, per)
        z[: per // 10] += rng.uniform(0.5, 2.0, per // 10)  # clutter above the plane
        xyz.append(np.concatenate([xy, z[:, None]], 1))
        seg.append(seg[-1] + per)
    xyz = np.ascontiguousarray(np.concatenate(xyz), np.float32)
    seg = np.asarray(seg, np.int32)
    zmin = np.array([xyz[seg[c]:seg[c + 1], 2].min() for c in range(P)], np.float32)
    zmax = np.array([xyz[seg[c]:seg[c + 1], 2].max() for c in range(P)], np.float32)
    origin = np.ascontiguousarray(np.stack([xyz[seg[c]:seg[c + 1]].mean(0) for c in range(P)]), np.float32)
    ratios = np.linspace(0.3, 1, 30).astype(np.float32)
    bc, bn, conf = np.zeros((P, 3), np.float32), np.zeros((P, 3), np.float32), np.zeros(P, np.float32)
    rc = main.pcs_cpu_ransac_plane_hypotheses(p(xyz), p(seg), ctypes.c_int64(len(xyz)), p(zmin), p(zmax), p(ratios),
                                              ctypes.c_int(30), p(origin), ctypes.c_int64(P), ctypes.c_float(0.0025),
                                              ctypes.c_float(1e-2), ctypes.c_int(50), ctypes.c_float(0.05), p(bc), p(bn),
                                              p(conf), None, ctypes.c_size_t(0), None)
    assert rc == 0
    want = np.array([-0.05, 0.02, 1.0]) / np.linalg.norm([-0.05, 0.02, 1.0])
    assert np.abs(bn - want).max() < 5e-3 and (conf > 0.8 * per).all()
    # pruning: hypothesis 2 gets a wild normal
    bn2 = bn.copy()
    bn2[2] = [0.6, 0.0, 0.8]
    keep = np.zeros(P, np.uint8)
    thr = np.logspace(np.log10(5), np.log10(0.01), 100).astype(np.float32)
    assert main.pcs_cpu_prune_plane_hypotheses(p(bc), p(bn2), ctypes.c_int64(P), ctypes.c_int(8), p(thr), ctypes.c_int(100),
                                               p(keep), None) == 0
    assert keep[2] == 0 and keep.sum() >= P - 2
    # smoothing
    X, Y = 12, 10
    mz = np.full((X, Y), 0.3, np.float32)
    mz[5, 5] = 2.0
    w = np.ones((X, Y), np.float32)
    h = np.zeros((X, Y), np.float32)
    iters = np.zeros(1, np.int32)
    dec = np.array([1600], np.int32)
    assert main.pcs_cpu_l1_minimization(p(mz), p(w), ctypes.c_int(X), ctypes.c_int(Y), ctypes.c_double(0.01), p(dec),
                                        ctypes.c_int(1), ctypes.c_float(0.5), ctypes.c_int(10000), p(h), p(iters), None,
                                        ctypes.c_size_t(0), None) == 0
    assert 10 < iters[0] <= 10000 and abs(float(np.median(h)) - 0.3) < 0.05 and h[5, 5] < 1.0


def test_radius_graph_twins():
    """pcs_cpu_radius_graph_search / _emit (the entries' own argument lists; the workspace carries the edges between the
    two phases): 16 nearest within the radius, grouped by query, nearest first, against brute force -- one frame offset,
    ties broken by the reference index."""
    main, _ = _libs()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(21)
    N, M, K, r = 1500, 400, 16, 0.8
    ref = np.concatenate([rng.integers(0, 2, (N, 1)), rng.uniform(0, 6, (N, 3))], 1).astype(np.float32)
    qry = np.concatenate([rng.integers(0, 2, (M, 1)), rng.uniform(0, 6, (M, 3))], 1).astype(np.float32)
    main.pcs_cpu_radius_graph_workspace_bytes.restype = ctypes.c_size_t
    nb = main.pcs_cpu_radius_graph_workspace_bytes(ctypes.c_int64(N), ctypes.c_int64(M), ctypes.c_int(K))
    ws = np.zeros(nb, np.uint8)
    E = np.zeros(1, np.int64)
    assert main.pcs_cpu_radius_graph_search(p(ref), ctypes.c_int64(N), p(qry), ctypes.c_int64(M), ctypes.c_float(r),
                                            ctypes.c_float(0.0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(K),
                                            ctypes.c_int(1), p(E), p(ws), ctypes.c_size_t(nb), None) == 0
    e_ref, e_q, d = np.zeros(E[0], np.int64), np.zeros(E[0], np.int64), np.zeros(E[0], np.float32)
    assert main.pcs_cpu_radius_graph_emit(ctypes.c_int64(N), ctypes.c_int64(M), ctypes.c_int(K), p(e_ref), p(e_q), p(d),
                                          p(ws), ctypes.c_size_t(nb), None) == 0
    d2 = ((ref[None, :, :] - qry[:, None, :]) ** 2).sum(-1)  # 4-D: the frame column counts
    want_q, want_r = [], []
    for q in range(M):
        cand = np.nonzero((d2[q] <= np.float32(r) * np.float32(r)) & (ref[:, 0] == qry[q, 0]))[0]
        cand = cand[np.lexsort((cand, d2[q, cand]))][:K]
        want_q += [q] * len(cand)
        want_r += cand.tolist()
    assert E[0] == len(want_q) and e_q.tolist() == want_q
    # (fp32 sums in the restatement's own order: compare the neighbour SETS per query and the sortedness)
    for q in range(0, M, 13):
        sel = e_q == q
        assert set(e_ref[sel].tolist()) == {b for a, b in zip(want_q, want_r) if a == q}
        assert (np.diff(d[sel]) >= 0).all()


def _i64(x):
    return ctypes.c_int64(int(x))


def test_segment_and_table_twins():
    """Second set (oracle/pcs_cpu_twins2.cpp): segment pick / median / min-max against numpy on the segments of
    pcs_cpu_unique_i64; the torch_hash table drop-ins (insert, degree count + fill, correspondence, points in radius) on
    a table the twin filled itself, against brute force over the walked cells."""
    main, _ = _libs()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(31)
    n = 5000
    ids = rng.integers(0, 700, n).astype(np.int64)
    uniq, inv = np.zeros(n, np.int64), np.zeros(n, np.int64)
    order, seg, d_num = np.zeros(n, np.int32), np.zeros(n + 1, np.int32), np.zeros(1, np.int64)
    assert main.pcs_cpu_unique_i64(p(ids), _i64(n), ctypes.c_int(64), p(uniq), p(inv), p(order), p(seg), p(d_num), None,
                                   ctypes.c_size_t(0), None) == 0
    v = int(d_num[0])
    for last in (0, 1):
        pick = np.zeros(v, np.int64)
        assert main.pcs_cpu_segment_pick(p(order), p(seg), _i64(v), ctypes.c_int(last), p(pick), None) == 0
        want = np.zeros(v, np.int64) if last else np.full(v, n, np.int64)
        (np.maximum if last else np.minimum).at(want, inv, np.arange(n))
        np.testing.assert_array_equal(pick, want)
    data = rng.integers(-50, 50, n).astype(np.int64)
    med = np.zeros(v, np.int64)
    assert main.pcs_cpu_segment_median_i64(p(data), p(order), p(seg), _i64(v), p(med), None) == 0
    np.testing.assert_array_equal(med, O.robust_median(data, inv, v))
    f = rng.normal(0, 1, (n, 3)).astype(np.float32)
    for is_max in (0, 1):
        mm = np.zeros((v, 3), np.float32)
        assert main.pcs_cpu_segment_minmax_f32(p(f), ctypes.c_int(3), p(order), p(seg), _i64(v), ctypes.c_int(is_max), p(mm),
                                               None) == 0
        want = np.full((v, 3), -np.inf if is_max else np.inf, np.float32)
        (np.maximum if is_max else np.minimum).at(want, inv, f)
        np.testing.assert_array_equal(mm, want)

    # ---- the reference's hash table: points (frame, x, y, z), cells of one radius, voxel coordinates as
    #      graph_utils.py:169-183 forms them
    N, M, D, r = 3000, 700, 4, 0.6
    ref = np.concatenate([rng.integers(0, 2, (N, 1)), rng.uniform(0, 8, (N, 3))], 1).astype(np.float32)
    qry = np.concatenate([rng.integers(0, 2, (M, 1)), rng.uniform(0, 8, (M, 3))], 1).astype(np.float32)
    vs = np.array([1 - 1e-3, r, r, r], np.float32)
    lo = np.minimum(ref.min(0), qry.min(0)) - vs * 2
    hi = np.maximum(ref.max(0), qry.max(0)) + vs * 2
    dims = (np.rint((hi - lo) / vs).astype(np.int64) + 3)
    rk = np.rint((ref - lo) / vs).astype(np.int64) + 1
    qk = np.rint((qry - lo) / vs).astype(np.int64) + 1
    H = 2 * N
    keys, vals, rev = np.full(H, -1, np.int64), np.zeros((H, D), np.float32), np.zeros(H, np.int64)
    assert main.pcs_cpu_hash_insert(p(keys), p(vals), p(rev), _i64(H), p(dims), ctypes.c_int(D), p(rk), p(ref), _i64(N),
                                    None) == 0
    assert (keys >= 0).sum() == N and sorted(rev[keys >= 0].tolist()) == list(range(N))
    np.testing.assert_array_equal(vals[keys >= 0], ref[rev[keys >= 0]])
    qmin, qmax = np.array([0, -1, -1, -1], np.int32), np.array([0, 1, 1, 1], np.int32)
    rad = np.full(M, r, np.float32)
    near = (np.abs(rk[None, :, 1:] - qk[:, None, 1:]).max(-1) <= 1) & (rk[None, :, 0] == qk[:, None, 0])  # walked cells
    d2 = np.zeros((M, N), np.float32)
    for d in range(D):  # (the kernels' own order of the four squared differences)
        d2 = d2 + (ref[None, :, d] - qry[:, None, d]) ** 2
    K = 8
    deg, off, E = np.zeros(M, np.int32), np.zeros(M, np.int32), np.zeros(1, np.int64)
    args = (p(keys), p(vals), p(rev), _i64(H), p(dims), ctypes.c_int(D), p(qk), p(qry), _i64(M), p(qmin), p(qmax))
    assert main.pcs_cpu_radius_graph_table_count(*args, p(rad), ctypes.c_int(K), p(deg), p(off), p(E), None,
                                                 ctypes.c_size_t(0), None) == 0
    inr = near & (d2 <= np.float32(r) * np.float32(r))
    np.testing.assert_array_equal(deg, np.minimum(inr.sum(1), K))
    assert E[0] == deg.sum() and (off == np.cumsum(deg) - deg).all()
    edges, dist = np.zeros((int(E[0]), 2), np.int64), np.zeros(int(E[0]), np.float32)
    assert main.pcs_cpu_radius_graph_table_fill(*args, p(rad), ctypes.c_int(1), p(deg), p(off), p(edges), p(dist), None) == 0
    for q in range(0, M, 11):
        cand = np.nonzero(inr[q])[0]
        cand = cand[np.lexsort((cand, d2[q, cand]))][:K]
        sl = slice(off[q], off[q] + deg[q])
        assert edges[sl, 0].tolist() == cand.tolist() and (edges[sl, 1] == q).all()
        np.testing.assert_array_equal(dist[sl], d2[q, cand])
    cor = np.zeros(M, np.int64)
    assert main.pcs_cpu_correspondence(*args, p(cor), None) == 0
    masked = np.where(near, d2, np.float32(np.inf))
    want = np.where(near.any(1), masked.argmin(1), -1)  # (argmin: the lowest index among equal distances)
    np.testing.assert_array_equal(cor, want)
    vis = np.zeros(N, np.int64)
    assert main.pcs_cpu_points_in_radius(*args, ctypes.c_float(0.4), p(vis), None) == 0
    np.testing.assert_array_equal(vis, (near & (d2 < np.float32(0.4) * np.float32(0.4))).any(0).astype(np.int64))


def test_sparse_conv_twins():
    """Second set: neighbour table, table conv with the fused epilogue, BatchNorm fold, conv backward, max pool +
    backward, dense() -- against the oracle's numpy / C restatements on a random strided rulebook."""
    main, _ = _libs()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(32)
    from tests.golden.make_golden import random_indices
    shape = [9, 16, 16]
    idx = random_indices(rng, shape, 600, batch=2)
    V, kv, cin, cout = len(idx), 27, 8, 16
    feat = rng.normal(0, 1, (V, cin)).astype(np.float32)
    W = rng.normal(0, 0.2, (kv, cin, cout)).astype(np.float32)
    # submanifold: table conv == the oracle's pair-list conv, bit for bit; then the epilogue
    _, pairs, num = O.get_indice_pairs(idx, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True)
    table = np.zeros((V, kv), np.int32)
    assert main.pcs_cpu_build_neighbor_table(p(pairs), _i64(pairs.shape[2]), p(num), ctypes.c_int(kv), ctypes.c_int(0),
                                             _i64(V), p(table), None) == 0
    for k in range(kv):
        hit = table[:, k] >= 0
        assert hit.sum() == num[k]
        assert set(zip(table[hit, k].tolist(), np.nonzero(hit)[0].tolist())) == \
            set(zip(pairs[k, 0, :num[k]].tolist(), pairs[k, 1, :num[k]].tolist()))
    centre = int(np.argmax(num))
    out = np.zeros((V, cout), np.float32)
    conv = lambda *a: main.pcs_cpu_indice_conv_table_f32(p(feat), ctypes.c_int(cin), p(W), ctypes.c_int(kv), ctypes.c_int(cout),
                                                         ctypes.c_int(centre), p(table), _i64(V), *a)
    assert conv(None, None, None, ctypes.c_int(0), p(out), None) == 0
    plain = O.indice_conv(feat, W, pairs, num, V, False, True)
    np.testing.assert_array_equal(out, plain)
    gamma, beta = rng.uniform(0.5, 2, cout).astype(np.float32), rng.normal(0, 1, cout).astype(np.float32)
    mean, var = rng.normal(0, 1, cout).astype(np.float32), rng.uniform(0.5, 2, cout).astype(np.float32)
    inv_std = (np.float32(1) / np.sqrt(var + np.float32(1e-3))).astype(np.float32)
    scale, shift = np.zeros(cout, np.float32), np.zeros(cout, np.float32)
    assert main.pcs_cpu_bn_fold_f32(p(gamma), p(beta), p(mean), p(inv_std), ctypes.c_int(cout), p(scale), p(shift), None) == 0
    np.testing.assert_array_equal(scale, inv_std * gamma)
    np.testing.assert_array_equal(shift, -mean * scale + beta)
    bias = rng.normal(0, 1, cout).astype(np.float32)
    fused = np.zeros((V, cout), np.float32)
    assert conv(p(bias), p(scale), p(shift), ctypes.c_int(1), p(fused), None) == 0
    np.testing.assert_array_equal(fused, np.maximum((plain + bias) * scale + shift, 0))
    skip = rng.normal(0, 1, (V, cout)).astype(np.float32)
    res_out = np.zeros((V, cout), np.float32)
    assert main.pcs_cpu_indice_conv_table_residual_f32(p(feat), ctypes.c_int(cin), p(W), ctypes.c_int(kv), ctypes.c_int(cout),
                                                       ctypes.c_int(centre), p(table), _i64(V), p(bias), p(scale), p(shift),
                                                       p(skip), ctypes.c_int(1), p(res_out), None) == 0
    np.testing.assert_array_equal(res_out, np.maximum(((plain + bias) * scale + shift) + skip, 0))
    # strided conv: backward against the numpy restatement (BLAS sums: tolerance), pooling exactly
    oshape = O.conv_out_shape(shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1])
    outids, pairs, num = O.get_indice_pairs(idx, oshape, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False)
    Vo = len(outids)
    og = rng.normal(0, 1, (Vo, cout)).astype(np.float32)
    gi, gw = np.zeros((V, cin), np.float32), np.zeros((kv, cin, cout), np.float32)
    assert main.pcs_cpu_indice_conv_backward_f32(p(feat), ctypes.c_int(cin), p(W), ctypes.c_int(kv), ctypes.c_int(cout), p(og),
                                                 p(pairs), _i64(pairs.shape[2]), p(num), _i64(V), _i64(Vo), ctypes.c_int(0),
                                                 ctypes.c_int(0), p(gi), p(gw), None, ctypes.c_size_t(0), None) == 0
    w_gi, w_gw = O.indice_conv_backward(feat, W, og, pairs, num, False, False)
    np.testing.assert_allclose(gi, w_gi, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gw, w_gw, rtol=1e-4, atol=1e-4)
    pooled = np.zeros((Vo, cin), np.float32)
    assert main.pcs_cpu_indice_maxpool_f32(p(feat), ctypes.c_int(cin), p(pairs), _i64(pairs.shape[2]), p(num), ctypes.c_int(kv),
                                           _i64(Vo), p(pooled), None, ctypes.c_size_t(0), None) == 0
    np.testing.assert_array_equal(pooled, O.indice_maxpool(feat, pairs, num, Vo))
    pg = rng.normal(0, 1, (Vo, cin)).astype(np.float32)
    din = np.zeros((V, cin), np.float32)
    assert main.pcs_cpu_indice_maxpool_backward_f32(p(feat), p(pooled), p(pg), ctypes.c_int(cin), p(pairs),
                                                    _i64(pairs.shape[2]), p(num), ctypes.c_int(kv), _i64(V), p(din), None,
                                                    ctypes.c_size_t(0), None) == 0
    np.testing.assert_array_equal(din, O.indice_maxpool_backward(feat, pooled, pg, pairs, num))
    dense = np.zeros((2, cin) + tuple(shape), np.float32)
    status = np.zeros(1, np.int32)
    assert main.pcs_cpu_sparse_to_dense(p(feat), p(idx), _i64(V), ctypes.c_int(cin), ctypes.c_int(2), *(ctypes.c_int(s) for s in shape),
                                        p(dense), p(status), None) == 0
    want = np.zeros_like(dense)
    want[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feat
    np.testing.assert_array_equal(dense, want)
    assert status[0] == 0


def test_box_loader_and_velocity_twins():
    """Second set: the per-frame box tests (membership, first box, counts per group) against the oracle's
    points_in_boxes frame by frame; the loader's sweep assembly against oracle/loader_oracle.py's arithmetic; smooth_velo
    against oracle/track_oracle.py (same stop step, same bits: both are the reference's AdamW written out serially)."""
    main, _ = _libs()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(33)
    F, N = 4, 3000
    nb = [5, 0, 7, 3]
    off = np.concatenate([[0], np.cumsum(nb)]).astype(np.int32)
    boxes = np.concatenate([rng.uniform(-10, 10, (off[-1], 3)), rng.uniform(1, 6, (off[-1], 3)),
                            rng.uniform(-3.2, 3.2, (off[-1], 1))], 1).astype(np.float32)
    pts = rng.uniform(-12, 12, (N, 3)).astype(np.float32)
    frame = rng.integers(0, F, N).astype(np.int64)
    bmax = max(nb)
    member, first = np.full((N, bmax), -7, np.int32), np.zeros(N, np.int32)
    assert main.pcs_cpu_points_in_frame_boxes(p(pts), p(frame), _i64(N), p(boxes), p(off), ctypes.c_int(F), ctypes.c_int(bmax),
                                              p(member), p(first), None) == 0
    want = np.zeros((N, bmax), np.int32)
    for f in range(F):
        sel = np.nonzero(frame == f)[0]
        if nb[f]:
            want[sel, :nb[f]] = O.points_in_boxes(pts[sel], boxes[off[f]:off[f + 1]]).T
    np.testing.assert_array_equal(member, want)
    np.testing.assert_array_equal(first, np.where(want.any(1), want.argmax(1), -1))
    G = 9
    group = rng.integers(0, G, N).astype(np.int64)
    counts = np.zeros((G, bmax), np.int32)
    assert main.pcs_cpu_frame_box_counts(p(pts), p(frame), _i64(N), p(boxes), p(off), ctypes.c_int(F), ctypes.c_int(bmax),
                                         p(group), _i64(G), p(counts), None) == 0
    wc = np.zeros((G, bmax), np.int32)
    np.add.at(wc, group, want)
    np.testing.assert_array_equal(counts, wc)
    # ---- sweeps
    S = 3
    so = np.array([0, 400, 400, 1000], np.int64)
    raw = rng.normal(0, 20, (1000, 8)).astype(np.float32)
    seg = rng.integers(0, 20, (1000, 2)).astype(np.int32)
    T = rng.normal(0, 1, (S, 3, 4))
    sid = np.array([5, 4, 3], np.int32)
    bxyz, feat = np.zeros((1000, 4), np.float32), np.zeros((1000, 3), np.float32)
    sweep, sl, il = np.zeros(1000, np.int32), np.zeros(1000, np.int32), np.zeros(1000, np.int32)
    assert main.pcs_cpu_assemble_sweeps(p(raw), p(seg), p(so), p(T), p(sid), ctypes.c_int(S), _i64(1000), ctypes.c_int(2), p(bxyz),
                                        p(feat), p(sweep), p(sl), p(il), None) == 0
    for s in range(S):
        a, b = so[s], so[s + 1]
        xyz = (raw[a:b, :3].astype(np.float64) @ T[s, :, :3].T + T[s, :, 3]).astype(np.float32)
        assert np.abs(bxyz[a:b, 1:] - xyz).max(initial=0) <= 4e-6 * max(1.0, np.abs(xyz).max(initial=0))
        assert (sweep[a:b] == sid[s]).all()
    assert (bxyz[:, 0] == 2).all()
    np.testing.assert_allclose(feat[:, 0], np.tanh(raw[:, 3]), rtol=2e-7, atol=1e-7)
    np.testing.assert_array_equal(feat[:, 1], raw[:, 4])
    np.testing.assert_array_equal(feat[:, 2], raw[:, 5] / np.float32(75))
    np.testing.assert_array_equal(il, seg[:, 0])
    np.testing.assert_array_equal(sl, seg[:, 1])
    # ---- smooth_velo
    from oracle import track_oracle as TO
    C, Fv = 7, 12
    d = rng.normal(0, 1, (C, Fv, 3)).astype(np.float32)
    v0 = (d + rng.normal(0, 0.3, d.shape)).astype(np.float32)
    for lo, hi in ((3, 9), (9, 3), (0, 11)):
        it = [0]
        want = TO.smooth_velo(v0.copy(), d, lo, hi, iters_out=it)
        got = v0.copy()
        assert main.pcs_cpu_smooth_velo(p(got), p(d), _i64(C), ctypes.c_int(Fv), ctypes.c_int(min(lo, hi)), ctypes.c_int(max(lo, hi)),
                                        ctypes.c_float(1), ctypes.c_float(10), ctypes.c_int(300), ctypes.c_float(1e-3), None,
                                        ctypes.c_size_t(0), None) == 0
        assert 3 < it[0] <= 300
        np.testing.assert_array_equal(got, want)


def test_radius_graph_components_twin():
    """pcs_cpu_radius_graph_components on the workspace of pcs_cpu_radius_graph_search (a cloud on itself, 32 nearest,
    nearest first): per cut radius the labels scipy gives the same edge set (rank of the component's smallest node)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    main, _ = _libs()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(41)
    n, K = 1200, 32
    centres = rng.uniform(0, 30, (60, 3))
    pts = np.concatenate([np.zeros((n, 1)), centres[rng.integers(0, 60, n)] + rng.normal(0, 0.35, (n, 3))], 1).astype(np.float32)
    main.pcs_cpu_radius_graph_workspace_bytes.restype = ctypes.c_size_t
    nb = main.pcs_cpu_radius_graph_workspace_bytes(_i64(n), _i64(n), ctypes.c_int(K))
    ws, E = np.zeros(nb, np.uint8), np.zeros(1, np.int64)
    assert main.pcs_cpu_radius_graph_search(p(pts), _i64(n), p(pts), _i64(n), ctypes.c_float(1.25), ctypes.c_float(0), ctypes.c_int(0),
                                            ctypes.c_int(0), ctypes.c_int(K), ctypes.c_int(1), p(E), p(ws), ctypes.c_size_t(nb), None) == 0
    cuts = np.array([1.25, 0.75, 0.25], np.float32)
    comp, num = np.zeros((3, n), np.int64), np.zeros(3, np.int64)
    assert main.pcs_cpu_radius_graph_components(_i64(n), _i64(n), ctypes.c_int(K), p(cuts), ctypes.c_int(3), p(comp), p(num), p(ws),
                                                ctypes.c_size_t(nb), None, ctypes.c_size_t(0), None) == 0
    e_ref, e_q, d = np.zeros(E[0], np.int64), np.zeros(E[0], np.int64), np.zeros(E[0], np.float32)
    assert main.pcs_cpu_radius_graph_emit(_i64(n), _i64(n), ctypes.c_int(K), p(e_ref), p(e_q), p(d), p(ws), ctypes.c_size_t(nb), None) == 0
    assert num[0] < num[1] < num[2] <= n
    for c, r in enumerate(cuts):
        keep = d <= r * r
        g = coo_matrix((np.ones(keep.sum()), (e_ref[keep], e_q[keep])), shape=(n, n))
        want_n, want = connected_components(g, directed=True, connection="weak")
        assert num[c] == want_n
        np.testing.assert_array_equal(comp[c], want)
