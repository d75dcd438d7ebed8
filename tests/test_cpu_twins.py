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
