"""Pins the numpy registration oracle (oracle/reg_oracle.py) and the small segment helpers to
golden vectors produced by the reference's own Python (tests/golden/make_golden_reg.py). CPU."""
import glob
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


def test_segment_helpers_vs_reference(oracle):
    from oracle import reg_oracle as R
    g = np.load(os.path.join(G, "reg_utils.npz"))
    C = int(g["C"])
    np.testing.assert_allclose(R.robust_mean(g["data"], g["index"], C), g["robust_mean"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(R.segment_mean(g["data"], g["index"], C), g["eff_mean"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(R.truncated_robust_mean(np.abs(g["data"][:, 0]), g["index"], C),
                               g["trunc_mean"], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(oracle.robust_median(g["ints"], g["index"], C), g["median"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "reg_icp_*.npz"))))
def test_register_to_next_frame_vs_reference(oracle, path):
    """TLS transforms within 1e-5 of the reference's own register_to_next_frame."""
    from oracle import reg_oracle as R
    g = np.load(path)
    moving = dict(fxyz=g["mov_fxyz"], component=g["mov_comp"], frame=g["mov_frame"], stationary=g["mov_stationary"])
    ref = dict(fxyz=g["ref_fxyz"], frame=g["ref_frame"], stationary=g["ref_stationary"])
    graph = R.Graph(float(g["radius"]))
    moved, T, l1, ratio, iters = R.register_to_next_frame(graph, moving, ref, int(g["C"]), 10.0, 80, 0.05)
    np.testing.assert_allclose(T, g["T"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(l1, g["l1"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(ratio, g["ratio"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(moved, g["moved"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "reg_icp_*.npz"))))
@pytest.mark.parametrize("threads", [1, 3])
def test_cpp_register_to_next_frame_vs_numpy_oracle(oracle, path, threads):
    """The C++ restatement that bench.py's cpu_baseline times (oracle/pcs_oracle_icp.cpp, serial and
    OpenMP) against the numpy oracle: same iteration count, transforms to f64 round-off, and through
    it within 1e-5 of the reference's own function."""
    from oracle import reg_oracle as R
    g = np.load(path)
    moving = dict(fxyz=g["mov_fxyz"], component=g["mov_comp"], frame=g["mov_frame"], stationary=g["mov_stationary"])
    ref = dict(fxyz=g["ref_fxyz"], frame=g["ref_frame"], stationary=g["ref_stationary"])
    C = int(g["C"])
    want = R.register_to_next_frame(R.Graph(float(g["radius"])), moving, ref, C, 10.0, 80, 0.05)
    got = R.register_to_next_frame_cpp(float(g["radius"]), moving, ref, C, 10.0, 80, 0.05, threads=threads)
    assert got[4] == want[4]
    np.testing.assert_allclose(got[1], want[1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(got[2], want[2], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(got[3], want[3])
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got[1], g["T"], rtol=0, atol=1e-5)


def test_radius_graph_hand_case(oracle):
    """pcdet/ops/torch_hash/torch_hash_modules.py:146-151: three points on a diagonal, r=0.15,
    1-NN sorted: every point pairs with itself at distance 0."""
    pts = np.array([[0, 0, 0, 0], [0, .1, .1, 0], [0, .2, .2, 0]], np.float32)
    e_ref, e_q, d = oracle.radius_graph(pts, pts, 0.15, 1, True, want_dists=True)
    np.testing.assert_array_equal(e_ref, [0, 1, 2])
    np.testing.assert_array_equal(e_q, [0, 1, 2])
    np.testing.assert_array_equal(d, [0, 0, 0])


def test_radius_graph_vs_sklearn(oracle):
    """The reference's own known-answer protocol (torch_hash_utils.py:365-431): random clouds,
    last dim zeroed, compare with sklearn NearestNeighbors."""
    from sklearn.neighbors import NearestNeighbors
    rng = np.random.default_rng(1)
    ref = rng.normal(size=(20000, 4)).astype(np.float32)
    qry = rng.normal(size=(300, 4)).astype(np.float32)
    ref[:, 0] = 0
    qry[:, 0] = 0
    r = 0.35
    e_ref, e_q, d = oracle.radius_graph(ref, qry, r, 8, True, want_dists=True)
    nn = NearestNeighbors(n_neighbors=8, radius=r).fit(ref.astype(np.float64))
    dist, ind = nn.kneighbors(qry.astype(np.float64))
    for q in range(qry.shape[0]):
        want = ind[q][dist[q] <= r * (1 - 1e-6)]
        got = e_ref[e_q == q]
        assert set(want.tolist()) <= set(got.tolist())
        assert len(got) <= 8 and len(got) >= len(want)


def test_correspondence_and_points_in_radius_vs_brute_force(oracle):
    """The oracle's correspondence / points_in_radius (torch_hash_kernel.cu:96-222) against brute force
    in the reference's own protocol (torch_hash_utils.py:365-394: hash-table correspondences ==
    sklearn nearest neighbours when the neighbour lies within one cell)."""
    from sklearn.neighbors import NearestNeighbors
    rng = np.random.default_rng(11)
    ref = rng.uniform(-3, 3, (6000, 4)).astype(np.float32)
    qry = rng.uniform(-3, 3, (500, 4)).astype(np.float32)
    ref[:, 0] = 0
    qry[:, 0] = 0
    cell = 0.5
    got = oracle.correspondence(ref, qry, cell)
    dist, ind = NearestNeighbors(n_neighbors=1).fit(ref.astype(np.float64)).kneighbors(qry.astype(np.float64))
    near = dist[:, 0] < cell / 2          # then the nearest point is certainly inside the 27 cells
    assert near.sum() > 300
    np.testing.assert_array_equal(got[near], ind[near, 0])
    # points_in_radius: strict `<` on the fp32 squared distance, restricted to the walked cells
    r = 0.2
    want = np.zeros(ref.shape[0], np.int64)
    d2 = ((ref[None, :, :] - qry[:, None, :]) ** 2).astype(np.float32)
    d2 = ((d2[..., 0] + d2[..., 1]) + d2[..., 2]) + d2[..., 3]
    want[(d2 < np.float32(r) * np.float32(r)).any(0)] = 1
    np.testing.assert_array_equal(oracle.points_in_radius(ref, qry, cell, r), want)


def test_connected_components_vs_scipy(oracle):
    import scipy.sparse as sp
    rng = np.random.default_rng(2)
    n = 5000
    e0 = rng.integers(0, n, 4000)
    e1 = rng.integers(0, n, 4000)
    num, comp = oracle.connected_components(e0, e1, n)
    adj = sp.coo_matrix((np.ones(len(e0)), (e0, e1)), shape=(n, n))
    n2, c2 = sp.csgraph.connected_components(adj)
    assert num == n2
    np.testing.assert_array_equal(comp, c2)
