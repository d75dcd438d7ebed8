"""numpy restatement of the reference's registration maths.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py). Follows
pcdet/models/registration/preprocessors/registration_utils.py line by line in numpy:
robust_mean (:12-21), efficient_robust_mean/sum (:23-45), truncated_robust_mean (:44-58),
register_to_next_frame (:83-206). The graph object of the reference is replaced by the oracle's
literal radius graph (oracle.radius_graph). Pinned by tests/golden/reg_*.npz, produced by running
the reference's own register_to_next_frame (tests/golden/make_golden_reg.py).
"""
import numpy as np

from . import oracle as O


def robust_mean(data, index, C):
    """scatter sum / count where count > 0.5 (registration_utils.py:12-21)."""
    shape = (C,) + data.shape[1:]
    s = np.zeros(shape, data.dtype)
    np.add.at(s, index, data)
    cnt = np.bincount(index, minlength=C).astype(data.dtype)
    m = cnt > 0.5
    s[m] = s[m] / cnt[m].reshape((-1,) + (1,) * (data.ndim - 1))
    return s


def segment_mean(data, index, C):
    """efficient_robust_mean (:23-33): argsort + segment_coo(mean); empty segments -> 0.
    The reference sums fp32 inputs in an unspecified order; here the sum is carried in f64 and
    rounded ONCE to the input dtype (the order-free value every fp32 summation order scatters
    around)."""
    s = np.zeros((C,) + data.shape[1:], np.float64)
    np.add.at(s, index, data.astype(np.float64))
    cnt = np.bincount(index, minlength=C)
    m = cnt > 0
    s[m] = s[m] / cnt[m].reshape((-1,) + (1,) * (data.ndim - 1))
    return s.astype(data.dtype)


def truncated_robust_mean(data, index, C, trunc=0.3):
    """(:44-58): mean, clamp to mean +- trunc, mean again."""
    mean = segment_mean(data, index, C)
    cl = np.clip(data, mean[index] - trunc, mean[index] + trunc)
    return segment_mean(cl, index, C)


class Graph:
    """Stand-in for graph_utils.RadiusGraph driven by the oracle radius graph."""

    def __init__(self, radius, max_nbr=1, sort_by_dist=True):
        self.radius = radius
        self.max_nbr = max_nbr
        self.sort = sort_by_dist
        self.qmin = np.array([0, -1, -1, -1], np.int32)
        self.qmax = np.array([0, 1, 1, 1], np.int32)

    def __call__(self, ref_fxyz, query_fxyz):
        return O.radius_graph(ref_fxyz, query_fxyz, np.float32(self.radius), self.max_nbr, self.sort,
                              self.qmin.tolist(), self.qmax.tolist())


def register_to_next_frame(graph, moving, ref, C, angle_regularizer=10.0, max_iter=20,
                           stopping_delta=5e-2):
    """moving / ref: dicts with fxyz [n,4] f32, component [n] i64 (moving), frame [n], stationary
    [n] bool. Returns (moved fxyz, T [C,4,4] f64, l1 [C] f64, edge_ratio [C] f32, iterations)."""
    temp_radius = graph.radius
    fo = int(ref["frame"][0] - moving["frame"][0])
    graph.radius = (temp_radius ** 2 + fo ** 2) ** 0.5
    comp_deg = np.bincount(moving["component"], minlength=C)
    T = np.tile(np.eye(4), (C, 1, 1))
    nsm = ~moving["stationary"]
    nsr = ~ref["stationary"]
    m_fxyz = moving["fxyz"][nsm].copy()
    m_comp = moving["component"][nsm]
    r_fxyz = ref["fxyz"][nsr]
    last_error, countdown, iters = 1e10, 3, 0
    l1 = np.zeros(C)
    for itr in range(max_iter):
        iters += 1
        graph.qmin[0] = graph.qmax[0] = fo
        f_ref, f_mov = graph(r_fxyz, m_fxyz)
        graph.qmin[0] = graph.qmax[0] = -fo
        b_mov, b_ref = graph(m_fxyz, r_fxyz)
        e_mov = np.concatenate([f_mov, b_mov])
        e_ref = np.concatenate([f_ref, b_ref])
        e_comp = m_comp[e_mov]
        mc = segment_mean(m_fxyz[e_mov, 1:], e_comp, C).astype(np.float64)   # fp32 mean, then double
        rc = segment_mean(r_fxyz[e_ref, 1:], e_comp, C).astype(np.float64)
        P = m_fxyz[e_mov, 1:].astype(np.float64) - mc[e_comp]
        Q = r_fxyz[e_ref, 1:].astype(np.float64) - rc[e_comp]
        dist = np.linalg.norm(P - Q, axis=-1)
        l1 = truncated_robust_mean(dist, e_comp, C)
        loss = float(np.sum(dist ** 2))
        H = P[:, :, None] * Q[:, None, :]
        cov = robust_mean(H, e_comp, C)
        U, S, VT = np.linalg.svd(cov + T[:, :3, :3] * angle_regularizer)
        V = np.transpose(VT, (0, 2, 1))
        UT = np.transpose(U, (0, 2, 1))
        sign = np.ones_like(S)
        sign[:, -1] = np.linalg.det(V @ UT)
        R = V @ (sign[:, :, None] * UT)
        Ti = np.zeros((C, 4, 4))
        Ti[:, :3, :3] = R
        Ti[:, :3, 3] = rc - np.einsum("cb,cab->ca", mc, R)
        Ti[:, 3, 3] = 1.0
        T = Ti @ T
        m_fxyz[:, 1:] = (np.einsum("nb,nab->na", m_fxyz[:, 1:].astype(np.float64), R[m_comp]) +
                         Ti[m_comp, :3, 3]).astype(np.float32)
        if last_error - loss < stopping_delta:
            countdown -= 1
        else:
            countdown = 3
        if countdown <= 0:
            break
        last_error = loss
    graph.qmin[0] = graph.qmax[0] = fo
    _, f_mov = graph(ref["fxyz"], m_fxyz)
    cnt = np.bincount(m_comp[f_mov], minlength=C)
    ratio = cnt.astype(np.float32) / (comp_deg.astype(np.float32) + np.float32(1e-6))
    graph.radius = temp_radius
    out = moving["fxyz"].copy()
    out[nsm, 1:] = m_fxyz[:, 1:]
    return out, T, l1, ratio.astype(np.float32), iters


_ICP_LIB = None


def _icp_lib():
    global _ICP_LIB
    if _ICP_LIB is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpcs_oracle_icp.so")
        if not os.path.exists(path):
            O.build()
        _ICP_LIB = ctypes.CDLL(path)
    return _ICP_LIB


def cpu_threads():
    """OpenMP's thread count on this host (what `threads=0` below expands to)."""
    return int(_icp_lib().orc_max_threads())


def register_to_next_frame_cpp(radius, moving, ref, C, angle_regularizer=10.0, max_iter=20,
                               stopping_delta=5e-2, threads=1):
    """The same function in C++ (oracle/pcs_oracle_icp.cpp), serial (threads=1) or OpenMP
    (threads > 1; 0 = all cores): what bench.py's cpu_baseline times. Same arguments and results as
    register_to_next_frame, `radius` instead of the graph object."""
    import ctypes
    lib = _icp_lib()
    fo = int(ref["frame"][0] - moving["frame"][0])
    nsm, nsr = ~moving["stationary"], ~ref["stationary"]
    mov = np.ascontiguousarray(moving["fxyz"][nsm], np.float32).copy()
    comp = np.ascontiguousarray(moving["component"][nsm], np.int64)
    r_ns = np.ascontiguousarray(ref["fxyz"][nsr], np.float32)
    r_all = np.ascontiguousarray(ref["fxyz"], np.float32)
    deg = np.ascontiguousarray(np.bincount(moving["component"], minlength=C), np.int64)
    T = np.zeros((C, 4, 4), np.float64)
    l1 = np.zeros(C, np.float64)
    ratio = np.zeros(C, np.float32)
    if threads == 0:
        threads = cpu_threads()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    iters = lib.orc_register_to_next_frame(
        p(mov), p(comp), ctypes.c_int64(mov.shape[0]), p(r_ns), ctypes.c_int64(r_ns.shape[0]), p(r_all),
        ctypes.c_int64(r_all.shape[0]), p(deg), ctypes.c_int64(C), ctypes.c_double(radius), ctypes.c_int(fo),
        ctypes.c_double(angle_regularizer), ctypes.c_int(max_iter), ctypes.c_double(stopping_delta), p(T), p(l1),
        p(ratio), ctypes.c_int(threads))
    out = moving["fxyz"].copy()
    out[nsm, 1:] = mov[:, 1:]
    return out, T, l1, ratio, int(iters)
