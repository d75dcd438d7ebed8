"""Parity of the HIP radius graph / connected components / medians / fused ICP against the CPU
oracle and the reference-generated golden vectors. Edges, labels, medians bit-exact; transforms
within 1e-5 (BASELINE.json north_star)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pcseqlearning_amd import ops  # noqa: E402

DEV = "cuda:0"
G = os.path.join(os.path.dirname(__file__), "golden")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def cloud(rng, n, frames, extent):
    return np.concatenate([rng.integers(0, frames, (n, 1)).astype(np.float32),
                           rng.uniform(-extent, extent, (n, 3)).astype(np.float32)], axis=1)


def test_radius_graph_hand_case():
    pts = dev(np.array([[0, 0, 0, 0], [0, .1, .1, 0], [0, .2, .2, 0]], np.float32))
    e_ref, e_q, d = ops.radius_graph(pts, pts, 0.15, 1, True, want_dists=True)
    assert e_ref.tolist() == [0, 1, 2] and e_q.tolist() == [0, 1, 2] and d.tolist() == [0, 0, 0]


@pytest.mark.parametrize("n,m,r,k,cell", [(3000, 1000, 0.8, 4, 0.0), (5000, 5000, 1.25, 32, 0.0),
                                            (5000, 5000, 0.25, 32, 0.25), (4000, 2000, 2.5, 1, 0.4),
                                            (1, 1, 1.0, 1, 0.0), (50, 3, 100.0, 32, 25.0)])
def test_radius_graph_vs_oracle(oracle, n, m, r, k, cell):
    rng = np.random.default_rng(n + m + k)
    ref, qry = cloud(rng, n, 3, 6.0), cloud(rng, m, 3, 6.0)
    for sort in (True, False):
        o_ref, o_q, o_d = oracle.radius_graph(ref, qry, r, k, sort, want_dists=True)
        e_ref, e_q, d = ops.radius_graph(dev(ref), dev(qry), r, k, sort, cell_size=cell, want_dists=True)
        np.testing.assert_array_equal(e_q.cpu().numpy(), o_q)
        np.testing.assert_array_equal(e_ref.cpu().numpy(), o_ref)
        if sort:
            np.testing.assert_array_equal(d.cpu().numpy(), o_d)


def test_radius_graph_frame_offsets(oracle):
    """register_to_next_frame's use: radius' = sqrt(r^2 + df^2), qmin[0] = qmax[0] = df."""
    rng = np.random.default_rng(3)
    for df in (1, -1, 5, -8):
        a = cloud(rng, 3000, 1, 8.0); a[:, 0] = 8
        b = cloud(rng, 2500, 1, 8.0); b[:, 0] = 8 + df
        rad = np.float32((1.25 ** 2 + df ** 2) ** 0.5)
        o_ref, o_q = oracle.radius_graph(b, a, rad, 1, True, qmin=(df, -1, -1, -1), qmax=(df, 1, 1, 1))
        e_ref, e_q = ops.radius_graph(dev(b), dev(a), float(rad), 1, True, df, df, cell_size=0.3)
        np.testing.assert_array_equal(e_q.cpu().numpy(), o_q)
        np.testing.assert_array_equal(e_ref.cpu().numpy(), o_ref)
        assert len(o_q) > 100


def test_torch_hash_dropins_vs_oracle(oracle):
    """hash_insert_gpu + radius_graph_gpu on the reference's table layout, driven exactly as
    RadiusGraph.build_graph does (graph_utils.py:169-206)."""
    rng = np.random.default_rng(4)
    ref, qry = cloud(rng, 4000, 2, 5.0), cloud(rng, 1500, 2, 5.0)
    r = 0.9
    rt, qt = dev(ref), dev(qry)
    radius = qt.new_zeros(qt.shape[0]) + r
    vs = torch.tensor([1 - 1e-3, r, r, r], device=DEV)
    allp = torch.cat([rt, qt], 0)
    lo = allp.min(0)[0] - vs * 2
    hi = allp.max(0)[0] + vs * 2
    cr = torch.round((rt - lo) / vs).long() + 1
    cq = torch.round((qt - lo) / vs).long() + 1
    dims = torch.round((hi - lo) / vs).long() + 3
    H = int(rt.shape[0] / 0.5)
    keys = cr.new_zeros(H) - 1
    values = rt.new_empty(H, 4)
    rev = cr.new_zeros(H)
    ops.hash_insert_gpu(keys, values, rev, dims, cr, rt)
    qmin = torch.tensor([0, -1, -1, -1], dtype=torch.int32, device=DEV)
    qmax = torch.tensor([0, 1, 1, 1], dtype=torch.int32, device=DEV)
    for k, sort in ((32, True), (1, True), (3, False)):
        edges = ops.radius_graph_gpu(keys, values, rev, dims, cq, qt, qmin, qmax, radius, k, sort)
        o_ref, o_q = oracle.radius_graph(ref, qry, r, k, sort)
        np.testing.assert_array_equal(edges[:, 0].cpu().numpy(), o_ref)
        np.testing.assert_array_equal(edges[:, 1].cpu().numpy(), o_q)
    assert int((keys >= 0).sum()) == rt.shape[0]  # every point owns a slot (duplicates not merged)
    # the other two functions graph_utils.py / torch_hash_modules.py import: correspondence (nearest
    # hashed point among the walked cells) and points_in_radius_gpu (hashed points near any query)
    from pcseqlearning_amd.pcdet.ops.torch_hash import torch_hash_cuda
    corres = cq.new_zeros(cq.shape[0]) - 7
    torch_hash_cuda.correspondence(keys, values, rev, dims, cq, qt, qmin, qmax, corres)
    np.testing.assert_array_equal(corres.cpu().numpy(), oracle.correspondence(ref, qry, r))
    assert int((corres >= 0).sum()) > 0
    for rq in (0.3, 0.9):
        visited = cr.new_zeros(rt.shape[0])
        torch_hash_cuda.points_in_radius_gpu(keys, values, rev, dims, cq, qt, qmin, qmax, rq, visited)
        want = oracle.points_in_radius(ref, qry, r, rq)
        np.testing.assert_array_equal(visited.cpu().numpy(), want)
        assert 0 < int(want.sum()) < ref.shape[0]


def test_connected_components_vs_oracle(oracle):
    rng = np.random.default_rng(5)
    for n, E in ((1, 0), (10, 3), (20000, 15000), (50000, 200000)):
        e0 = rng.integers(0, n, E)
        e1 = rng.integers(0, n, E)
        num, comp = ops.connected_components(dev(e0), dev(e1), n)
        o_num, o_comp = oracle.connected_components(e0, e1, n)
        assert num == o_num
        np.testing.assert_array_equal(comp.cpu().numpy(), o_comp)


def test_proposal_graph_components(oracle):
    """ClusterProposal's unit of work: 32-NN radius graph of a cloud on itself -> components."""
    rng = np.random.default_rng(6)
    pts = cloud(rng, 20000, 2, 12.0)
    pts[:, 3] *= 0.1
    p = dev(pts)
    e0, e1 = ops.radius_graph(p, p, 0.75, 32, True)
    num, comp = ops.connected_components(e0, e1, pts.shape[0])
    o0, o1 = oracle.radius_graph(pts, pts, 0.75, 32, True)
    o_num, o_comp = oracle.connected_components(o0, o1, pts.shape[0])
    np.testing.assert_array_equal(e0.cpu().numpy(), o0)
    assert num == o_num
    np.testing.assert_array_equal(comp.cpu().numpy(), o_comp)


def test_segment_median_and_minmax(oracle):
    rng = np.random.default_rng(7)
    n = 30000
    ids = rng.integers(0, 4000, n)
    data = rng.integers(-3, 50, n)
    seg = ops.unique_i64(dev(ids))
    med = ops.segment_median(dev(data), seg).cpu().numpy()
    uq = np.unique(ids)
    want = oracle.robust_median(data, np.searchsorted(uq, ids), len(uq))
    np.testing.assert_array_equal(med, want)
    f = rng.normal(size=(n, 3)).astype(np.float32)
    mx = ops.segment_minmax(dev(f), seg, True).cpu().numpy()
    mn = ops.segment_minmax(dev(f), seg, False).cpu().numpy()
    inv = np.searchsorted(uq, ids)
    wmx = np.full((len(uq), 3), -np.inf, np.float32); np.maximum.at(wmx, inv, f)
    wmn = np.full((len(uq), 3), np.inf, np.float32); np.minimum.at(wmn, inv, f)
    np.testing.assert_array_equal(mx, wmx)
    np.testing.assert_array_equal(mn, wmn)


def test_nn1_dense_cells_spill(oracle):
    """Cells holding far more points than their fixed run (kCellCap = 16): the spill list of the
    two-level grid must give the same exact nearest neighbours (ties by index) as the oracle, for
    the sorted 1-NN radius graph and through the fused ICP."""
    rng = np.random.default_rng(11)
    blobs = rng.uniform(-5, 5, (12, 3)).astype(np.float32)
    dense = (blobs[rng.integers(0, 12, 1500)] + rng.normal(0, 0.004, (1500, 3))).astype(np.float32)
    ref = np.concatenate([np.zeros((1500, 1), np.float32), dense], 1)
    ref = np.concatenate([ref, cloud(rng, 2000, 1, 6.0)], 0)
    ref[:, 0] = 0
    dup = ref[rng.integers(0, len(ref), 200)].copy()          # exact duplicates: index ties
    ref = np.concatenate([ref, dup], 0)
    qry = np.concatenate([ref[rng.integers(0, len(ref), 800)] + rng.normal(0, 0.02, (800, 4)).astype(np.float32),
                          cloud(rng, 800, 1, 6.0)], 0).astype(np.float32)
    qry[:, 0] = 0
    o_ref, o_q, o_d = oracle.radius_graph(ref, qry, 0.5, 1, True, want_dists=True)
    e_ref, e_q, d = ops.radius_graph(dev(ref), dev(qry), 0.5, 1, True, cell_size=0.13, want_dists=True)
    np.testing.assert_array_equal(e_q.cpu().numpy(), o_q)
    np.testing.assert_array_equal(e_ref.cpu().numpy(), o_ref)
    np.testing.assert_array_equal(d.cpu().numpy(), o_d)
    assert len(o_q) > 900

    # the same clouds through the ICP (static grids AND the per-iteration moving grid spill)
    from oracle import reg_oracle as R
    comp = (np.arange(len(qry)) % 7).astype(np.int64)
    moving = dict(fxyz=qry.copy(), component=comp, frame=np.zeros(len(qry), np.int64),
                  stationary=np.zeros(len(qry), bool))
    moving["fxyz"][:, 1:] += np.float32(0.05)
    dense_q = (blobs[rng.integers(0, 12, 600)] + rng.normal(0, 0.004, (600, 3))).astype(np.float32)
    extra = np.concatenate([np.zeros((600, 1), np.float32), dense_q], 1)
    moving = dict(fxyz=np.concatenate([moving["fxyz"], extra]), component=np.concatenate([comp, np.full(600, 3)]),
                  frame=np.zeros(len(qry) + 600, np.int64), stationary=np.zeros(len(qry) + 600, bool))
    refd = dict(fxyz=ref.copy(), frame=np.ones(len(ref), np.int64), stationary=np.zeros(len(ref), bool))
    refd["fxyz"][:, 0] = 1
    graph = R.Graph(0.5)
    o_moved, o_T, o_l1, o_ratio, o_it = R.register_to_next_frame(graph, moving, refd, 7, 10.0, 40, 0.05)
    g = dict(mov_fxyz=moving["fxyz"], mov_comp=moving["component"], mov_frame=moving["frame"],
             mov_stationary=moving["stationary"], ref_fxyz=refd["fxyz"], ref_frame=refd["frame"],
             ref_stationary=refd["stationary"], C=np.array(7))
    nsm = ~g["mov_stationary"]
    mov = dev(g["mov_fxyz"][nsm].copy())
    T, l1, ratio, iters = ops.register_to_next_frame(mov, dev(g["mov_comp"]), dev(g["ref_fxyz"]), dev(g["ref_fxyz"]),
                                                     dev(np.bincount(g["mov_comp"], minlength=7)), 7, 0.5, 1,
                                                     10.0, 40, 0.05, 0.0, 8)
    assert iters == o_it
    np.testing.assert_allclose(T.cpu().numpy(), o_T, rtol=0, atol=1e-5)
    np.testing.assert_allclose(mov.cpu().numpy(), o_moved, rtol=0, atol=2e-5)


def test_points_in_frame_boxes_vs_oracle(oracle):
    """All frames at once against the oracle's points_in_boxes (roiaware_pool3d.cpp:121-168) frame by
    frame: membership matrix in within-frame box order, first containing box."""
    rng = np.random.default_rng(4)
    F = 5
    nbox = np.array([4, 0, 7, 2, 5])
    off = np.concatenate([[0], np.cumsum(nbox)]).astype(np.int32)
    boxes = np.concatenate([rng.uniform(-20, 20, (off[-1], 3)) * [1, 1, 0.05],
                            rng.uniform([2, 1, 1], [8, 4, 3], (off[-1], 3)),
                            rng.uniform(-3.2, 3.2, (off[-1], 1))], 1).astype(np.float32)
    n = 6000
    frame = rng.integers(0, F, n)
    centre = boxes[rng.integers(0, off[-1], n), :3]
    pts = (centre + rng.normal(0, 2.5, (n, 3)) * [1, 1, 0.4]).astype(np.float32)
    bmax = int(nbox.max())
    member, first = ops.points_in_frame_boxes(dev(pts), dev(frame.astype(np.int64)), dev(boxes),
                                              dev(off), bmax)
    member, first = member.cpu().numpy(), first.cpu().numpy()
    for f in range(F):
        sel = np.nonzero(frame == f)[0]
        want = oracle.points_in_boxes(pts[sel], boxes[off[f]:off[f + 1]]) if nbox[f] else np.zeros((0, len(sel)), np.int32)
        np.testing.assert_array_equal(member[sel, :nbox[f]].T, want)
        assert not member[sel, nbox[f]:].any()
        want_first = np.where(want.any(0), want.argmax(0), -1) if nbox[f] else np.full(len(sel), -1)
        np.testing.assert_array_equal(first[sel], want_first)
    assert member.sum() > 200
    # the same hits summed per group inside the test (ops.frame_box_counts: no membership matrix)
    group = rng.integers(0, 37, n)
    counts = ops.frame_box_counts(dev(pts), dev(frame.astype(np.int64)), dev(boxes), dev(off), bmax,
                                  dev(group.astype(np.int64)), 37).cpu().numpy()
    want_counts = np.zeros((37, bmax), np.int64)
    np.add.at(want_counts, group, member)
    np.testing.assert_array_equal(counts, want_counts)


def _run_icp(g, radius, cell=0.0, chunk=8):
    nsm = ~g["mov_stationary"]
    nsr = ~g["ref_stationary"]
    C = int(g["C"])
    mov = dev(g["mov_fxyz"][nsm].copy())
    comp = dev(g["mov_comp"][nsm])
    deg = dev(np.bincount(g["mov_comp"], minlength=C))
    fo = int(g["ref_frame"][0] - g["mov_frame"][0])
    T, l1, ratio, iters = ops.register_to_next_frame(mov, comp, dev(g["ref_fxyz"][nsr]), dev(g["ref_fxyz"]),
                                                     deg, C, radius, fo, 10.0, 80, 0.05, cell, chunk)
    moved = g["mov_fxyz"].copy()
    moved[nsm] = mov.cpu().numpy()
    return T.cpu().numpy(), l1.cpu().numpy(), ratio.cpu().numpy(), moved, iters


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "reg_icp_*.npz"))))
def test_icp_vs_reference_golden(path):
    g = np.load(path)
    T, l1, ratio, moved, iters = _run_icp(g, float(g["radius"]))
    np.testing.assert_allclose(T, g["T"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(l1, g["l1"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(ratio, g["ratio"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(moved, g["moved"], rtol=0, atol=2e-5)
    # chunking of the device-side stop rule must not change the result
    T2, _, _, _, iters2 = _run_icp(g, float(g["radius"]), cell=0.5, chunk=1)
    assert iters2 == iters
    np.testing.assert_allclose(T2, T, rtol=0, atol=1e-9)


# the three schedules of the iterations (pcs_icp_set_schedule): every one computes the same iterations
SCHEDULES = {"wide": (-1, True), "resident_tiles": (1, True), "resident_cache": (1, False)}


@pytest.fixture(params=sorted(SCHEDULES))
def icp_schedule(request):
    wide_iters, tiles = SCHEDULES[request.param]
    ops.icp_set_schedule(wide_iters, tiles)
    yield request.param
    ops.icp_set_schedule(-1, True)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "reg_icp_*.npz"))))
def test_icp_schedules_agree_bit_for_bit(path, icp_schedule):
    """Wide launches, the resident loop with LDS-staged searches and the resident loop on the hash grids run the
    same iterations: transforms, errors, ratios, moved points and iteration counts are IDENTICAL (integer
    moments, exact neighbours), also across chunk lengths."""
    g = np.load(path)
    ops.icp_set_schedule(-1, True)
    T0, l0, r0, m0, it0 = _run_icp(g, float(g["radius"]))
    ops.icp_set_schedule(*SCHEDULES[icp_schedule])
    for chunk in (8, 3, 80):
        T, l1, ratio, moved, iters = _run_icp(g, float(g["radius"]), chunk=chunk)
        assert iters == it0
        np.testing.assert_array_equal(T, T0)
        np.testing.assert_array_equal(l1, l0)
        np.testing.assert_array_equal(ratio, r0)
        np.testing.assert_array_equal(moved, m0)


def test_icp_larger_all_schedules(oracle, icp_schedule):
    """The larger pair (several bands of rows in the tile path, open queries, an empty component) on every
    schedule against the oracle."""
    test_icp_vs_oracle_larger(oracle)


@pytest.mark.parametrize("case", ["far_steps", "dense_cells"])
def test_icp_moving_grid_in_place(oracle, case):
    """The moving cloud's grid is updated in place between iterations (g2_move_point): far_steps -- most points change
    their cell in the first iterations (tombstones, appends, rebuilds decided by the solve phase); dense_cells -- cells
    with far more than kCellCap points (rows in the overflow list, moved and tombstoned there). Same result as the
    oracle's brute-force neighbours either way."""
    from oracle import reg_oracle as R
    from tests.golden.make_golden_reg import make_pair
    rng = np.random.default_rng(314)
    if case == "far_steps":
        moving, ref = make_pair(rng, n_obj=24, pts_per_obj=400, frame_a=5, frame_b=6, shift=0.7, rot=0.1)
        radius, cell = 1.25, 0.0
    else:
        moving, ref = make_pair(rng, n_obj=2, pts_per_obj=1500, frame_a=2, frame_b=3, shift=0.2, extra_static=False)
        radius, cell = 0.75, 0.6  # (hundreds of points per 0.6 m cell)
    C = int(moving["component"].max()) + 1
    o_moved, o_T, o_l1, o_ratio, o_it = R.register_to_next_frame(R.Graph(radius), moving, ref, C, 10.0, 80, 0.05)
    g = dict(mov_fxyz=moving["fxyz"], mov_comp=moving["component"], mov_frame=moving["frame"],
             mov_stationary=moving["stationary"], ref_fxyz=ref["fxyz"], ref_frame=ref["frame"],
             ref_stationary=ref["stationary"], C=np.array(C))
    T, l1, ratio, moved, iters = _run_icp(g, radius, cell=cell)
    assert iters == o_it and iters > 3
    np.testing.assert_allclose(T, o_T, rtol=0, atol=1e-5)
    np.testing.assert_allclose(l1, o_l1, rtol=0, atol=1e-5)
    np.testing.assert_allclose(ratio, o_ratio, rtol=0, atol=1e-6)
    np.testing.assert_allclose(moved, o_moved, rtol=0, atol=2e-5)


def test_icp_vs_oracle_larger(oracle):
    """Bigger synthetic pair (stationary parts, empty component, negative frame offset)."""
    from oracle import reg_oracle as R
    from tests.golden.make_golden_reg import make_pair
    rng = np.random.default_rng(99)
    moving, ref = make_pair(rng, n_obj=40, pts_per_obj=300, frame_a=12, frame_b=9, shift=0.3)
    C = int(moving["component"].max()) + 3  # two ids with no points at all
    graph = R.Graph(1.25)
    o_moved, o_T, o_l1, o_ratio, o_it = R.register_to_next_frame(graph, moving, ref, C, 10.0, 80, 0.05)
    g = dict(mov_fxyz=moving["fxyz"], mov_comp=moving["component"], mov_frame=moving["frame"],
             mov_stationary=moving["stationary"], ref_fxyz=ref["fxyz"], ref_frame=ref["frame"],
             ref_stationary=ref["stationary"], C=np.array(C))
    T, l1, ratio, moved, iters = _run_icp(g, 1.25, cell=0.4)
    assert iters == o_it
    np.testing.assert_allclose(T, o_T, rtol=0, atol=1e-5)
    np.testing.assert_allclose(l1, o_l1, rtol=0, atol=1e-5)
    np.testing.assert_allclose(ratio, o_ratio, rtol=0, atol=1e-6)
    np.testing.assert_allclose(moved, o_moved, rtol=0, atol=2e-5)



def _lidar_like(rng, n, frames=2):
    """Points on a few planes and poles (dense surfaces, as a voxel-sampled LiDAR frame has them) plus a thin
    uniform background (queries with fewer than k points inside the radius)."""
    n_s, n_p = int(n * 0.7), int(n * 0.2)
    uv = rng.uniform(-6, 6, (n_s, 2))
    which = rng.integers(0, 3, n_s)
    surf = np.stack([np.where(which == 0, uv[:, 0], np.where(which == 1, 3.0 + 0.01 * uv[:, 0], uv[:, 0])),
                     np.where(which == 0, uv[:, 1], np.where(which == 1, uv[:, 1], -4.0)),
                     np.where(which == 0, 0.02 * uv[:, 0], np.where(which == 1, 0.3 * uv[:, 0], 0.2 * uv[:, 1]))], 1)
    poles = np.concatenate([rng.normal(0, 0.03, (n_p, 2)) + rng.integers(-5, 6, (n_p, 2)),
                            rng.uniform(0, 2, (n_p, 1))], 1)
    back = rng.uniform(-12, 12, (n - n_s - n_p, 3))
    xyz = np.concatenate([surf, poles, back], 0).astype(np.float32)
    rng.shuffle(xyz)
    return np.concatenate([rng.integers(0, frames, (n, 1)).astype(np.float32), xyz], 1)


@pytest.mark.parametrize("case", ["lidar", "ties", "query_cloud"])
def test_knn_tiled_search_vs_engines(oracle, case):
    """The cell-sorted wave search (pcs_knn.hip) against the oracle, against the one-thread engine it replaces and
    against its own one-thread fallback: edges and distances bit for bit. `ties`: points on a lattice (dozens of
    equal distances at the k-th place: rows overflow, the queries take the fallback)."""
    from pcseqlearning_amd import lib as L
    lib = L.load()
    rng = np.random.default_rng(11)
    if case == "ties":
        g = np.stack(np.meshgrid(np.arange(24), np.arange(24), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
        ref = np.concatenate([np.zeros((g.shape[0], 1)), g * 0.125], 1).astype(np.float32)
        rng.shuffle(ref)
        qry = ref
        r, k = 0.5, 32
    elif case == "query_cloud":
        ref = _lidar_like(rng, 30000)
        qry = _lidar_like(rng, 7000)
        r, k = 1.25, 8
    else:
        ref = _lidar_like(rng, 40000)
        qry = ref
        r, k = 1.25, 32
    rt = dev(ref)
    qt = rt if qry is ref else dev(qry)
    got = {}
    try:
        for mode in ("tiled", "engine", "hard"):
            lib.pcs_radius_graph_engine_only(1 if mode == "engine" else 0)
            lib.pcs_knn_force_hard(1 if mode == "hard" else 0)
            got[mode] = [t.cpu().numpy() for t in ops.radius_graph(rt, qt, r, k, True, want_dists=True)]
    finally:
        lib.pcs_radius_graph_engine_only(0)
        lib.pcs_knn_force_hard(0)
    for mode in ("engine", "hard"):
        for a, b in zip(got["tiled"], got[mode]):
            np.testing.assert_array_equal(a, b)
    if case != "lidar":  # (the oracle's brute force is quadratic)
        o_ref, o_q, o_d = oracle.radius_graph(ref, qry, r, k, True, want_dists=True)
        np.testing.assert_array_equal(got["tiled"][0], o_ref)
        np.testing.assert_array_equal(got["tiled"][1], o_q)
        np.testing.assert_array_equal(got["tiled"][2], o_d)
    assert got["tiled"][0].shape[0] > 4 * qry.shape[0]


def test_knn_components_of_three_cuts(oracle):
    """ops.knn_components (one 32-NN search at the largest radius, components of its cuts from the neighbour table)
    = a graph and a labelling per radius, as ClusterProposal.propose_cluster asks for them."""
    rng = np.random.default_rng(12)
    pts = _lidar_like(rng, 30000)
    p = dev(pts)
    radii = [1.25, 0.75, 0.25]
    comp, num, E = ops.knn_components(p, radii, 32)
    for i, r in enumerate(radii):
        e0, e1 = ops.radius_graph(p, p, r, 32, True)
        o_num, o_comp = oracle.connected_components(e0.cpu().numpy(), e1.cpu().numpy(), pts.shape[0])
        assert int(num[i]) == o_num
        np.testing.assert_array_equal(comp[i].cpu().numpy(), o_comp)
        if i == 0:
            assert E == e0.shape[0]
    assert int(num[2]) > int(num[1]) > int(num[0]) > 1


def test_icp_component_beyond_fixed_point_range_is_refused():
    """the moments are 64-bit fixed point about each component's first moving point; a component that
    reaches further than 64 m from it must fail the call (PCS_ERR_RANGE), not return wrong moments (include/pcs.h)."""
    from pcseqlearning_amd.lib import PcsError
    rng = np.random.default_rng(21)
    n = 400
    xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    xyz[n // 2:, 0] += 70.0  # one component, two blobs 70 m apart
    mov = np.concatenate([np.full((n, 1), 3, np.float32), xyz], 1)
    ref = mov.copy()
    ref[:, 0] = 4
    ref[:, 1:] += 0.05
    comp = np.zeros(n, np.int32)
    with pytest.raises(PcsError):
        ops.register_to_next_frame(dev(mov), dev(comp), dev(ref), dev(ref), dev(np.array([n], np.int64)), 1, 1.25, 1,
                                   10.0, 20, 0.05)
    # the same clouds as two components (each 2 m across) register
    comp[n // 2:] = 1
    T, _, _, iters = ops.register_to_next_frame(dev(mov), dev(comp), dev(ref), dev(ref),
                                                dev(np.array([n // 2, n // 2], np.int64)), 2, 1.25, 1, 10.0, 20, 0.05)
    assert iters >= 1 and np.isfinite(T.cpu().numpy()).all()


@pytest.mark.parametrize("seed", [4, 9, 13, 15, 28])
def test_icp_second_level_of_the_cache_is_exact(seed):
    """Round 5: a query whose budget has run out re-ranks its four cached points instead of searching. The same
    registration with the second level switched off (every such query searched: the form of rounds 1-4) must give the
    same transforms, errors, moved points and iteration counts BIT FOR BIT -- on crowded clouds, where partners of
    neighbouring objects compete and the second level changes partners (five seeds that told a wrong pair apart)."""
    import ctypes
    from pcseqlearning_amd import lib as L
    from tests.golden.make_golden_reg import make_pair
    lib = L.load()
    rng = np.random.default_rng(seed)
    n_obj = int(rng.integers(20, 400))
    ppo = int(rng.integers(3, 120))
    radius = [2.5, 1.25, 1.0][seed % 3]
    moving, refd = make_pair(rng, n_obj=n_obj, pts_per_obj=ppo, shift=float(rng.uniform(0.05, 0.6)), rot=0.05)
    scale = 0.25 if seed % 2 else 0.5
    for d in (moving, refd):
        d["fxyz"][:, 1:3] *= scale
    nsm, nsr = ~moving["stationary"], ~refd["stationary"]
    got = {}
    try:
        for levels in (2, 1):
            L.check(lib.pcs_icp_set_cache_levels(ctypes.c_int(levels)), "pcs_icp_set_cache_levels")
            mov = dev(moving["fxyz"][nsm].copy())
            T, l1, ratio, iters = ops.register_to_next_frame(
                mov, dev(moving["component"][nsm]), dev(refd["fxyz"][nsr]), dev(refd["fxyz"]),
                dev(np.bincount(moving["component"], minlength=n_obj)), n_obj, radius, 1, 10.0, 80, 0.05)
            got[levels] = (T.cpu().numpy(), l1.cpu().numpy(), ratio.cpu().numpy(), iters, mov.cpu().numpy())
    finally:
        lib.pcs_icp_set_cache_levels(ctypes.c_int(2))
    assert got[2][3] == got[1][3] and got[2][3] >= 5
    for a, b in zip(got[2], got[1]):
        np.testing.assert_array_equal(a, b)


def test_torch_hash_module_fronts(oracle):
    """pcdet.ops.torch_hash.RadiusGraph / ChamferDistance (torch_hash_modules.py:10-129 of the reference): the module
    fronts on the drop-in table ops -- the reference's own hand case (:146-151), its edges against the oracle, and the
    Chamfer distance against brute force."""
    from pcseqlearning_amd.pcdet.ops.torch_hash import ChamferDistance, RadiusGraph
    rg = RadiusGraph(ndim=3).to(DEV)
    pts = dev(np.array([[0, 0, 0, 0], [0, .1, .1, 0], [0, .2, .2, 0]], np.float32))
    e = rg(pts, pts, 0.15, 1, sort_by_dist=True)
    assert e.tolist() == [[0, 1, 2], [0, 1, 2]]
    rng = np.random.default_rng(31)
    ref, qry = cloud(rng, 3000, 2, 5.0), cloud(rng, 1200, 2, 5.0)
    for k, sort in ((16, True), (1, True), (4, False)):
        e = rg(dev(ref), dev(qry), 0.9, k, sort_by_dist=sort)
        o_ref, o_q = oracle.radius_graph(ref, qry, 0.9, k, sort)
        np.testing.assert_array_equal(e[0].cpu().numpy(), o_ref)
        np.testing.assert_array_equal(e[1].cpu().numpy(), o_q)
    # per-query radii: a query with radius 0.3 keeps the neighbours within 0.3
    rad = torch.full((qry.shape[0],), 0.9, device=DEV)
    rad[::2] = 0.3
    e = rg(dev(ref), dev(qry), rad, 32, sort_by_dist=True)
    d = (dev(ref)[e[0]] - dev(qry)[e[1]]).square().sum(-1).sqrt()
    assert bool((d <= rad[e[1]] + 1e-6).all()) and e.shape[1] > 0
    cd = ChamferDistance(ndim=3).to(DEV)
    a, b = dev(ref[ref[:, 0] == 0]), dev(qry[qry[:, 0] == 0])
    got = float(cd(a, b, 2.0))
    dm = torch.cdist(a[:, 1:].double(), b[:, 1:].double())
    f, g = dm.min(1).values, dm.min(0).values
    want = float((f[f <= 2.0] ** 2).mean() + (g[g <= 2.0] ** 2).mean())
    assert abs(got - want) < 1e-5 * max(want, 1.0), (got, want)


def test_torch_hash_table_front():
    """pcdet.ops.torch_hash.HashTable (torch_hash_utils.py:9-363 of the reference): the stateful front on the drop-in
    table ops. Points [n, 4] = (x, y, z, frame), cells as wide as the radius, so everything within the radius of a
    query lies in the 3 x 3 x 3 cells the ops walk: correspondences, points in radius and the voxel graph against brute
    force; the one-call forms against insert + step 2."""
    from pcseqlearning_amd.pcdet.ops.torch_hash import HashTable
    rng = np.random.default_rng(51)
    n, m = 4000, 1500
    ref = np.concatenate([rng.uniform(0, 10, (n, 3)), rng.integers(0, 2, (n, 1))], 1).astype(np.float32)
    qry = np.concatenate([rng.uniform(0, 10, (m, 3)), rng.integers(0, 2, (m, 1))], 1).astype(np.float32)
    vs = torch.tensor([0.5, 0.5, 0.5, 1.0])
    pcr = torch.tensor([0.0, 0.0, 0.0, 0.0, 10.0, 10.0, 10.0, 2.0])
    ht = HashTable(size=2 ** 15, device=DEV)
    ht.hash_into_gpu(dev(ref), vs, pcr)
    cells_r = np.round(ref / vs.numpy()).astype(np.int64)
    cells_q = np.round(qry / vs.numpy()).astype(np.int64)
    for off in (0, 1):
        c = ht.find_corres_step2(dev(qry), off).cpu().numpy()
        # brute force: the nearest reference point among the cells within one step (frame cell + off exactly)
        found = dict(zip(c[0].tolist(), c[1].tolist()))
        for q in range(0, m, 7):
            dc = cells_r[:, :3] - cells_q[q, :3]
            near = (np.abs(dc).max(1) <= 1) & (cells_r[:, 3] == cells_q[q, 3] + off)
            if not near.any():
                assert q not in found
                continue
            cand = np.nonzero(near)[0]
            d = ((ref[cand] - qry[q]) ** 2).sum(1)
            assert q in found and abs(d.min() - ((ref[found[q]] - qry[q]) ** 2).sum()) < 1e-5
    # points within 0.5 of some query of the same frame
    got = set(ht.points_in_radius_step2(dev(qry), 0, radius=0.5).cpu().numpy().tolist())
    same = ref[:, None, 3] == qry[None, :, 3]
    d2 = ((ref[:, None, :3] - qry[None, :, :3]) ** 2).sum(-1)
    want = set(np.nonzero(((d2 < 0.25 - 1e-5) & same).any(1))[0].tolist())
    maybe = set(np.nonzero(((d2 < 0.25 + 1e-5) & same).any(1))[0].tolist())
    assert want <= got <= maybe
    # voxel graph: (query, reference) pairs within the radius, at most k per query
    e = ht.voxel_graph_step2(dev(qry), 0, radius=0.5, max_num_neighbors=64).cpu().numpy()
    pairs = set(zip(e[0].tolist(), e[1].tolist()))
    sure = set(zip(*np.nonzero(((d2 < 0.25 - 1e-5) & same).T)))
    loose = set(zip(*np.nonzero(((d2 < 0.25 + 1e-5) & same).T)))
    assert sure <= pairs <= loose
    # the one-call forms
    ht2 = HashTable(size=2 ** 15, device=DEV)
    a = ht2.find_corres(dev(ref), dev(qry), vs, pcr, 0)
    b = ht.find_corres_step2(dev(qry), 0)
    assert torch.equal(a, b)
    g = ht2.voxel_graph(dev(ref), dev(qry), vs, pcr, 0, radius=0.5, max_num_neighbors=64)
    assert set(zip(g[0].tolist(), g[1].tolist())) == pairs
    ht2.clear()
    assert ht2.find_corres_step2(dev(qry), 0).shape == (2, 0)
    with pytest.raises(ValueError):
        HashTable(size=1024, device=DEV).find_corres_step2(dev(qry), 0)


def test_one_nearest_neighbour_near_search_vs_cooperative_search(oracle):
    """The 1-NN path of pcs_radius_graph_search (the extraction's and the harvest's graph): survivors of the super-block
    filter first go one thread each through the 27 fine cells around them, and only the ones that search cannot settle
    reach the cooperative search -- against the cooperative search for every survivor
    (pcs_radius_graph_engine_only(2)) and against the oracle: the same edges and distances, on a cloud where most
    queries sit next to a reference point, some far from any, and some exactly between two."""
    from pcseqlearning_amd import lib as L
    rng = np.random.default_rng(77)
    ref = cloud(rng, 40000, 3, 12.0)
    near = ref[rng.integers(0, len(ref), 30000)].copy()
    near[:, 1:] += rng.normal(0, 0.03, (30000, 3)).astype(np.float32)
    far = cloud(rng, 15000, 3, 30.0)
    mid = (ref[:2000] + ref[1:2001]) / 2
    mid[:, 0] = ref[:2000, 0]
    # a crowd: 60 points within 5 cm of each of 40 places (cells of 20 cm hold 16 rows: the rest spills)
    crowd = np.repeat(ref[:40], 60, axis=0).copy()
    crowd[:, 1:] += rng.uniform(-0.05, 0.05, (2400, 3)).astype(np.float32)
    ref = np.concatenate([ref, crowd]).astype(np.float32)
    qry = np.concatenate([near, far, mid, crowd[::3] + np.float32(0.01)]).astype(np.float32)
    qry[:, 0] = np.round(qry[:, 0])
    lib = L.load()
    res = []
    for mode in (2, 0):
        L.check(lib.pcs_radius_graph_engine_only(mode), "mode")
        try:
            res.append(ops.radius_graph(dev(ref), dev(qry), 0.8, 1, True, 0, 0, cell_size=0.8 / 4 * 1.0005, want_dists=True))
        finally:
            L.check(lib.pcs_radius_graph_engine_only(0), "mode")
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    o_ref, o_q = oracle.radius_graph(ref, qry, 0.8, 1, True)
    np.testing.assert_array_equal(res[1][0].cpu().numpy(), o_ref)
    np.testing.assert_array_equal(res[1][1].cpu().numpy(), o_q)
