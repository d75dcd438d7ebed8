"""numpy front-end of the CPU oracle (oracle/libpcs_oracle.so) and of the compiled reference
builders (oracle/_ref/*).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg. Nothing under pcseqlearning_amd/ imports this module.
"""
import ctypes
import importlib.util
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF_SPCONV = None


def build():
    subprocess.run(["make", "-C", HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libpcs_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        for name in ("orc_unique_i64", "orc_dynamic_voxelize", "orc_hard_voxelize",
                     "orc_get_indice_pairs", "orc_radius_graph", "orc_connected_components"):
            getattr(_LIB, name).restype = ctypes.c_int64
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else ctypes.c_void_p(0)


def _ints(v):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])


def _floats(v):
    return (ctypes.c_float * len(v))(*[float(x) for x in v])


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# ------------------------------------------------------------------------------------------
def grid_cluster_ids(points, size4, margin0=0.5):
    pts = _c(points, np.float32)
    ids = np.empty(pts.shape[0], np.int64)
    bounds = np.empty(8, np.float32)
    lib().orc_grid_cluster_ids(_p(pts), ctypes.c_int64(pts.shape[0]), _floats(size4),
                               ctypes.c_float(margin0), _p(ids), _p(bounds))
    return ids, bounds


def unique_i64(ids):
    ids = _c(ids, np.int64)
    n = ids.shape[0]
    uniq = np.empty(n, np.int64)
    inv = np.empty(n, np.int64)
    order = np.empty(n, np.int32)
    seg = np.empty(n + 1, np.int32)
    v = lib().orc_unique_i64(_p(ids), ctypes.c_int64(n), _p(uniq), _p(inv), _p(order), _p(seg))
    return uniq[:v].copy(), inv, order, seg[:v + 1].copy()


def segment_reduce(data, order, seg_start, mean=True):
    data = _c(data, np.float32)
    squeeze = data.ndim == 1
    d2 = data.reshape(data.shape[0], -1)
    v = seg_start.shape[0] - 1
    out = np.empty((v, d2.shape[1]), np.float32)
    lib().orc_segment_reduce_f32(_p(d2), ctypes.c_int(d2.shape[1]), _p(_c(order, np.int32)),
                                 _p(_c(seg_start, np.int32)), ctypes.c_int64(v),
                                 ctypes.c_int(int(mean)), _p(out))
    return out[:, 0] if squeeze else out


def grid_sample(points, grid_size3):
    """GridSampling3D.forward(points, return_inverse=True) -> (means [v,4], inv [n], order, seg)."""
    ids, _ = grid_cluster_ids(points, [1.0] + list(grid_size3), 0.5)
    uniq, inv, order, seg = unique_i64(ids)
    means = segment_reduce(points, order, seg, mean=True)
    return means, inv, order, seg


def dynamic_voxelize(points, pc_range, voxel_size, grid_size):
    pts = _c(points, np.float32)
    n, ncol = pts.shape
    feat = np.empty((n, ncol - 1), np.float32)
    coords = np.empty((n, 4), np.int32)
    inv = np.empty(n, np.int64)
    v = lib().orc_dynamic_voxelize(_p(pts), ctypes.c_int64(n), ctypes.c_int(ncol),
                                   _floats(pc_range), _floats(voxel_size),
                                   _p(_c(grid_size, np.int32)), _p(feat), _p(coords), _p(inv))
    return feat[:v].copy(), coords[:v].copy(), inv


def voxel_aggregation(point_bxyz, mean_attrs, median_attrs, voxel_size3, pc_range6=None):
    """VoxelAggregation.forward, pcdet/ops/voxel/voxel_modules.py:25-126, in numpy (fp32 quantisation like
    the reference's tensors; means accumulated in f64 and rounded once -- torch_scatter's fp32 sums are
    unordered). mean_attrs / median_attrs: dicts of per-point arrays (point_bxyz itself is averaged as
    'bxyz'). Returns (voxel dict, voxel_index [n'], point coords [n',4], num_voxels, out_of_boundary [n]).
    Pinned by tests/golden/voxel_agg_*.npz (the reference's own module run on CPU)."""
    bxyz = np.asarray(point_bxyz, np.float32)
    vs = np.asarray([1.0] + list(voxel_size3), np.float32)
    batch_size = int(bxyz[:, 0].max()) + 1
    if pc_range6 is None:
        mn = bxyz.min(0)
        vc = np.floor((bxyz - mn) / vs).astype(np.int64)
        vc = vc - vc.min(0)
        dims = vc.max(0) + 1
        oob = np.zeros(len(bxyz), bool)
    else:
        r = np.asarray([0] + list(pc_range6[:3]) + [1] + list(pc_range6[3:]), np.float32)
        mn, mx = r[:4].copy(), r[4:].copy()
        mn[0] = 0
        mx[0] = np.float32(batch_size - 1e-5)
        vc = np.floor((bxyz - mn) / vs).astype(np.int64)
        dims = np.ceil((mx - mn) / vs).astype(np.int64) + 1
        oob = (vc >= dims)[:, 1:4].any(-1) | (vc < 0)[:, 1:4].any(-1)
        vc, bxyz = vc[~oob], bxyz[~oob]
    merge = np.zeros(len(vc), np.int64)
    for i in range(4):
        merge = merge * dims[i] + vc[:, i]
    uniq, vidx = np.unique(merge, return_inverse=True)
    V = len(uniq)
    ucoords = np.zeros((V, 4), np.int64)
    u = uniq.copy()
    for i in range(3, -1, -1):
        ucoords[:, i] = u % dims[i]
        u = u // dims[i]
    center = (ucoords.astype(np.float32) * vs + mn)[:, 1:4]
    center = center + vs[1:4] / np.float32(2)
    cnt = np.bincount(vidx, minlength=V).astype(np.float64)
    out = {}
    attrs = dict(bxyz=bxyz)
    attrs.update({k: np.asarray(v)[~oob] for k, v in mean_attrs.items()})
    for key, val in attrs.items():
        v2 = val.reshape(len(val), -1).astype(np.float64)
        ssum = np.zeros((V, v2.shape[1]))
        np.add.at(ssum, vidx, v2)
        out["voxel_" + key] = (ssum / cnt[:, None]).astype(np.float32).reshape((V,) + val.shape[1:])
    deg = cnt.astype(np.int64)
    off = np.cumsum(deg) - deg
    for key, val in (median_attrs or {}).items():
        val = np.asarray(val)[~oob].astype(np.int64)
        tval = (val - val.min()) + vidx * val.max()
        out["voxel_" + key] = np.sort(tval)[off + deg // 2] - np.arange(V) * val.max()
    out["voxel_batch_index"] = np.round(out["voxel_bxyz"][:, 0]).astype(np.int64)
    out["voxel_xyz"] = out["voxel_bxyz"][:, 1:4]
    out["voxel_center"] = center
    out["voxel_bcenter"] = np.concatenate([out["voxel_bxyz"][:, :1], center], -1)
    out["voxel_bcoords"] = ucoords
    return out, vidx, vc, V, oob


def hard_voxelize(points, pc_range, voxel_size, max_points, max_voxels):
    pts = _c(points, np.float32)
    n, ncol = pts.shape
    voxels = np.zeros((max_voxels, max_points, ncol), np.float32)
    coords = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros(max_voxels, np.int32)
    v = lib().orc_hard_voxelize(_p(pts), ctypes.c_int64(n), ctypes.c_int(ncol), _floats(pc_range),
                                _floats(voxel_size), ctypes.c_int(max_points),
                                ctypes.c_int(max_voxels), _p(voxels), _p(coords), _p(num))
    return voxels[:v].copy(), coords[:v].copy(), num[:v].copy()


def conv_out_shape(spatial_shape, ksize, stride, padding, dilation):
    """get_conv_output_size, pcdet/ops/spconv/ops.py:20-30."""
    return [(spatial_shape[i] + 2 * padding[i] - dilation[i] * (ksize[i] - 1) - 1) // stride[i] + 1
            for i in range(3)]


def indice_conv_half(features, filters, pairs, num, num_out, inverse=False, subm=False, bias=None):
    """indiceConv<at::Half> / fusedIndiceConvBatchNorm<at::Half> (spconv_ops.h:260-361, fused_spconv_ops.h:29-132)
    in the reference's own order of roundings: fp16 storage, the output starts at zero (or at the bias
    rows), the subM centre is one GEMM into the output, every other offset gathers its rows, multiplies
    (fp32 accumulation, result rounded to fp16 as a half GEMM returns it) and scatter-adds in fp16."""
    f = np.asarray(features, np.float16)
    kv = pairs.shape[0]
    cin, cout = filters.shape[-2], filters.shape[-1]
    W = np.asarray(filters, np.float16).reshape(kv, cin, cout)
    out = np.zeros((num_out, cout), np.float16)
    if bias is not None:
        out[:] = np.asarray(bias, np.float16)[None, :]
    centre = -1
    if subm:
        centre = int(np.argmax(num))
        out = (out.astype(np.float32) + (f.astype(np.float32) @ W[centre].astype(np.float32)).astype(np.float16)
               .astype(np.float32)).astype(np.float16) if bias is not None else \
            (f.astype(np.float32) @ W[centre].astype(np.float32)).astype(np.float16)
    a, b = (1, 0) if inverse else (0, 1)
    for k in range(kv):
        n = int(num[k])
        if n == 0 or k == centre:
            continue
        src, dst = pairs[k, a, :n], pairs[k, b, :n]
        prod = (f[src].astype(np.float32) @ W[k].astype(np.float32)).astype(np.float16)
        for r in range(n):  # scatter-add in fp16 (a destination appears once per offset)
            out[dst[r]] = (out[dst[r]].astype(np.float32) + prod[r].astype(np.float32)).astype(np.float16)
    return out


def deconv_out_shape(spatial_shape, ksize, stride, padding, dilation, output_padding):
    """get_deconv_output_size, pcdet/ops/spconv/ops.py:33-44."""
    return [(spatial_shape[i] - 1) * stride[i] - 2 * padding[i] + ksize[i] + output_padding[i] for i in range(3)]


def get_indice_pairs(indices, out_shape, ksize, stride, padding, dilation, subm, transpose=False):
    """subM / strided / transposed (getIndicePairsDeConv, geometry.h:197-245) rulebook."""
    idx = _c(indices, np.int32)
    V = idx.shape[0]
    kv = int(np.prod(ksize))
    pairs = np.empty((kv, 2, V), np.int32)
    num = np.empty(kv, np.int32)
    out_ids = np.zeros((max(V * kv, 1), 4), np.int32)
    n = lib().orc_get_indice_pairs(_p(idx), ctypes.c_int64(V), _ints(out_shape), _ints(ksize),
                                   _ints(stride), _ints(padding), _ints(dilation),
                                   ctypes.c_int(2 if transpose else int(subm)), _p(out_ids), _p(pairs), _p(num))
    return (idx if subm else out_ids[:n].copy()), pairs, num


def indice_conv(features, filters, pairs, num, num_out, inverse=False, subm=False):
    feat = _c(features, np.float32)
    kv = pairs.shape[0]
    cin, cout = filters.shape[-2], filters.shape[-1]
    W = _c(filters, np.float32).reshape(kv, cin, cout)
    pairs = _c(pairs, np.int32)
    out = np.empty((num_out, cout), np.float32)
    lib().orc_indice_conv_f32(_p(feat), ctypes.c_int64(feat.shape[0]), ctypes.c_int(cin), _p(W),
                              ctypes.c_int(kv), ctypes.c_int(cout), _p(pairs),
                              ctypes.c_int64(pairs.shape[2]), _p(_c(num, np.int32)),
                              ctypes.c_int64(num_out), ctypes.c_int(int(inverse)),
                              ctypes.c_int(int(subm)), _p(out))
    return out


def indice_conv_backward(features, filters, out_grad, pairs, num, inverse=False, subm=False):
    """indiceConvBackward (pcdet/ops/spconv/include/spconv/spconv_ops.h:363-449): per kernel offset,
    filters_grad[k] = gathered_in^T @ gathered_out_grad and in_grad[i] += out_grad[o] @ W[k]^T over
    the rulebook pairs; for subM the centre offset (the one with the most pairs) is the plain
    product over all rows. Returns (input_grad [Nin,Cin], filters_grad like filters). Pinned by
    torch's dense conv3d autograd (tests/test_oracle_pinning.py)."""
    feat = _c(features, np.float32)
    og = _c(out_grad, np.float32)
    kv = pairs.shape[0]
    cin, cout = filters.shape[-2], filters.shape[-1]
    W = _c(filters, np.float32).reshape(kv, cin, cout)
    in_grad = np.zeros_like(feat)
    w_grad = np.zeros_like(W)
    num = np.asarray(num)
    centre = int(np.argmax(num)) if subm else -1
    if subm:
        w_grad[centre] = feat.T @ og
        in_grad += og @ W[centre].T
    a, b = (1, 0) if inverse else (0, 1)
    for k in range(kv):
        n = int(num[k])
        if n <= 0 or k == centre:
            continue
        i_in, i_out = pairs[k, a, :n], pairs[k, b, :n]
        gi, go = feat[i_in], og[i_out]
        w_grad[k] = gi.T @ go
        np.add.at(in_grad, i_in, go @ W[k].T)
    return in_grad, w_grad.reshape(filters.shape)


def indice_maxpool(features, pairs, num, num_out):
    """indiceMaxPool (pcdet/ops/spconv/include/spconv/pool_ops.h:26-60, src/maxpool.cc:22-40): the output
    starts at ZERO and takes every paired input that is larger, offset by offset."""
    feat = _c(features, np.float32)
    out = np.zeros((num_out, feat.shape[1]), np.float32)
    for k in range(pairs.shape[0]):
        n = int(num[k])
        if n > 0:
            np.maximum.at(out, pairs[k, 1, :n], feat[pairs[k, 0, :n]])
    return out


def indice_maxpool_backward(features, out_features, out_grad, pairs, num):
    """indiceMaxPoolBackward (pool_ops.h:62-98, src/maxpool.cc:43-64): every paired input equal to the
    pooled value receives the output's gradient (ties all receive it), offsets in ascending order."""
    feat = _c(features, np.float32)
    outf = _c(out_features, np.float32)
    og = _c(out_grad, np.float32)
    din = np.zeros_like(feat)
    for k in range(pairs.shape[0]):
        n = int(num[k])
        if n <= 0:
            continue
        i_in, i_out = pairs[k, 0, :n], pairs[k, 1, :n]
        hit = outf[i_out] == feat[i_in]
        np.add.at(din, i_in, np.where(hit, og[i_out], np.float32(0)))
    return din


def ref_maxpool():
    """The reference's own CPU functors compiled in place (oracle/_ref/libref_maxpool.so), or None."""
    path = os.path.join(HERE, "_ref", "libref_maxpool.so")
    return ctypes.CDLL(path) if os.path.exists(path) else None


def radius_graph(ref, query, radius, max_nbr, sort_by_dist=True, qmin=(0, -1, -1, -1),
                 qmax=(0, 1, 1, 1), want_dists=False):
    """RadiusGraph.build_graph: returns (e_ref, e_query) int64 (and squared dists)."""
    ref = _c(ref, np.float32)
    query = _c(query, np.float32)
    M = query.shape[0]
    cap = max(M * max_nbr, 1)
    e_ref = np.empty(cap, np.int64)
    e_qry = np.empty(cap, np.int64)
    d = np.empty(cap, np.float32)
    E = lib().orc_radius_graph(_p(ref), ctypes.c_int64(ref.shape[0]), _p(query), ctypes.c_int64(M),
                               ctypes.c_float(np.float32(radius)), _ints(qmin), _ints(qmax),
                               ctypes.c_int(max_nbr), ctypes.c_int(int(sort_by_dist)), _p(e_ref),
                               _p(e_qry), _p(d))
    if want_dists:
        return e_ref[:E].copy(), e_qry[:E].copy(), d[:E].copy()
    return e_ref[:E].copy(), e_qry[:E].copy()


def correspondence(ref, query, cell, qmin=(0, -1, -1, -1), qmax=(0, 1, 1, 1)):
    """correspondence_kernel (torch_hash_kernel.cu:96-155): nearest hashed point among the walked
    cells of size `cell`, -1 if none."""
    r, q = _c(ref, np.float32), _c(query, np.float32)
    out = np.empty(q.shape[0], np.int64)
    lib().orc_table_query(_p(r), ctypes.c_int64(r.shape[0]), _p(q), ctypes.c_int64(q.shape[0]),
                          ctypes.c_float(cell), _ints(qmin), _ints(qmax), ctypes.c_int(0), ctypes.c_float(0),
                          _p(out))
    return out


def points_in_radius(ref, query, cell, radius, qmin=(0, -1, -1, -1), qmax=(0, 1, 1, 1)):
    """points_in_radius_kernel (torch_hash_kernel.cu:160-222): visited[ref] = 1 where some query is
    closer than `radius` (strict)."""
    r, q = _c(ref, np.float32), _c(query, np.float32)
    out = np.zeros(r.shape[0], np.int64)
    lib().orc_table_query(_p(r), ctypes.c_int64(r.shape[0]), _p(q), ctypes.c_int64(q.shape[0]),
                          ctypes.c_float(cell), _ints(qmin), _ints(qmax), ctypes.c_int(1), ctypes.c_float(radius),
                          _p(out))
    return out


def connected_components(e0, e1, n):
    e0 = _c(e0, np.int64)
    e1 = _c(e1, np.int64)
    comp = np.empty(n, np.int64)
    num = lib().orc_connected_components(_p(e0), _p(e1), ctypes.c_int64(e0.shape[0]),
                                         ctypes.c_int64(n), _p(comp))
    return int(num), comp


def points_in_boxes(points, boxes):
    pts = _c(points, np.float32)
    bx = _c(boxes, np.float32)
    out = np.zeros((bx.shape[0], pts.shape[0]), np.int32)
    lib().orc_points_in_boxes(_p(pts), ctypes.c_int64(pts.shape[0]), _p(bx), ctypes.c_int(bx.shape[0]), _p(out))
    return out


def robust_median(data, index, num_segments):
    """registration_utils.robust_median (:60-81): sort index*range + (data - min), pick the
    element at exclusive_scan(deg) + deg // 2; empty segments -> int64(-1e10)."""
    data = np.asarray(data, np.int64)
    index = np.asarray(index, np.int64)
    dmin, dmax = data.min(), data.max()
    rng = dmax - dmin + 1
    key = np.sort(index * rng + (data - dmin))
    sidx = key // rng
    sdata = key - rng * sidx + dmin
    deg = np.bincount(index, minlength=num_segments)
    off = np.cumsum(deg) - deg + deg // 2
    out = np.full(num_segments, np.int64(-1e10), np.int64)
    m = deg > 0
    out[m] = sdata[off[m]]
    return out


# ------------------------------------------------------------------------------------------
# compiled reference (only where oracle/_ref was built, i.e. where /root/reference exists or the
# prebuilt .so travelled with the repo snapshot)
def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libref_spconv.so"))


def ref_get_indice_pairs(indices, batch, out_shape, ksize, stride, padding, dilation, subm):
    global _REF_SPCONV
    if _REF_SPCONV is None:
        _REF_SPCONV = ctypes.CDLL(os.path.join(HERE, "_ref", "libref_spconv.so"))
    idx = _c(indices, np.int32)
    V = idx.shape[0]
    kv = int(np.prod(ksize))
    pairs = np.empty((kv, 2, V), np.int32)
    num = np.empty(kv, np.int32)
    out_ids = np.zeros((max(V * kv, 1), 4), np.int32)
    n = _REF_SPCONV.ref_get_indice_pairs(_p(idx), ctypes.c_int(V), ctypes.c_int(batch),
                                         _ints(out_shape), _ints(ksize), _ints(stride),
                                         _ints(padding), _ints(dilation), ctypes.c_int(int(subm)),
                                         _p(out_ids), _p(pairs), _p(num))
    return (idx if subm else out_ids[:n].copy()), pairs, num


def ref_get_indice_pairs_deconv(indices, batch, out_shape, ksize, stride, padding, dilation):
    """The reference's own getIndicePairsDeConv (compiled in place, oracle/_ref/libref_spconv.so)."""
    global _REF_SPCONV
    if _REF_SPCONV is None:
        _REF_SPCONV = ctypes.CDLL(os.path.join(HERE, "_ref", "libref_spconv.so"))
    idx = _c(indices, np.int32)
    V = idx.shape[0]
    kv = int(np.prod(ksize))
    pairs = np.empty((kv, 2, V), np.int32)
    num = np.empty(kv, np.int32)
    out_ids = np.zeros((max(V * kv, 1), 4), np.int32)
    n = _REF_SPCONV.ref_get_indice_pairs_deconv(_p(idx), ctypes.c_int(V), ctypes.c_int(batch), _ints(out_shape),
                                                _ints(ksize), _ints(stride), _ints(padding), _ints(dilation),
                                                _p(out_ids), _p(pairs), _p(num))
    return out_ids[:n].copy(), pairs, num


def ref_hard_voxelize(points, pc_range, voxel_size, max_points, max_voxels):
    """The reference's points_to_voxel_3d_np<float,3> with the VoxelGeneratorV2 calling
    convention (zero-filled outputs, coor_to_voxelidx grid of -1)."""
    import glob
    so = glob.glob(os.path.join(HERE, "_ref", "ref_point2voxel*.so"))[0]
    spec = importlib.util.spec_from_file_location("ref_point2voxel", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pts = _c(points, np.float32)
    ncol = pts.shape[1]
    vs = np.asarray(voxel_size, np.float32)
    rng = np.asarray(pc_range, np.float32)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int32)
    voxels = np.zeros((max_voxels, max_points, ncol), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros(max_voxels, np.int32)
    c2v = -np.ones(tuple(grid[::-1]), np.int32)
    v = mod.points_to_voxel_3d_np(pts, voxels, coors, num, c2v, vs.tolist(), rng.tolist(),
                                  max_points, max_voxels)
    return voxels[:v].copy(), coors[:v].copy(), num[:v].copy()


# ------------------------------------------------------------------------------------------
# N-D rulebook (the reference instantiates NDim = 2, 3, 4: src/all.cc:22-29). A literal, loop-by-loop restatement
# of include/spconv/geometry.h:25-297 for small cases; pinned by running it at NDim = 3 against the compiled
# restatement above (itself pinned to the reference's own functors, tests/test_oracle_pinning.py).
def _valid_out_pos(pos, ksize, stride, padding, dilation, out_shape, transpose):
    """getValidOutPos / getValidOutPosTranspose (geometry.h:25-84, 86-143): (output position, kernel offset) of one
    input position, in the reference's order of enumeration."""
    D = len(pos)
    if transpose:
        lowers = [pos[i] * stride[i] - padding[i] for i in range(D)]
        uppers = [lowers[i] + (ksize[i] - 1) * dilation[i] for i in range(D)]
    else:
        # (C integer division truncates towards zero)
        tdiv = lambda a, b: int(a / b) if a * b < 0 else a // b  # noqa: E731
        lowers = [tdiv(pos[i] - (ksize[i] - 1) * dilation[i] - 1 + stride[i] + padding[i], stride[i]) for i in range(D)]
        uppers = [tdiv(pos[i] + padding[i], stride[i]) for i in range(D)]
    sizes = [(uppers[i] - lowers[i]) // dilation[i] + 1 for i in range(D)]
    n = 1
    for s in sizes:
        n *= s
    counter = [0] * D
    out = []
    for _ in range(max(n, 0)):
        valid, m, offset, vals = True, 1, 0, [0] * D
        for j in range(D - 1, -1, -1):
            val = uppers[j] - counter[j] * dilation[j]
            vals[j] = val
            if val < 0 or val > out_shape[j] - 1:
                valid = False
            if transpose:
                offset += m * (val - lowers[j]) // dilation[j]
            else:
                offset += m * (pos[j] - val * stride[j] + padding[j]) // dilation[j]
            m *= ksize[j]
        if valid:
            out.append((tuple(vals), offset))
        counter[D - 1] += 1
        for c in range(D - 1, 0, -1):
            if counter[c] == sizes[c]:
                counter[c - 1] += 1
                counter[c] = 0
    return out


def get_indice_pairs_nd_py(indices, out_shape, ksize, stride, padding, dilation, subm, transpose=False):
    """getIndicePairsSubM / Conv / DeConv (geometry.h:145-297) for indices [V, 1 + D]: (outids, pairs [KV,2,V] with -1
    beyond the counts, num [KV])."""
    idx = np.asarray(indices, np.int64)
    V, D = idx.shape[0], idx.shape[1] - 1
    kv = int(np.prod(ksize))
    pairs = np.full((kv, 2, V), -1, np.int32)
    num = np.zeros(kv, np.int32)
    grid = {}
    if subm:
        for j in range(V):
            grid[(int(idx[j, 0]),) + tuple(int(x) for x in idx[j, 1:])] = j
        for j in range(V):
            for vals, off in _valid_out_pos([int(x) for x in idx[j, 1:]], ksize, stride, padding, dilation, out_shape, False):
                o = grid.get((int(idx[j, 0]),) + vals, -1)
                if o > -1:
                    pairs[off, 0, num[off]] = j
                    pairs[off, 1, num[off]] = o
                    num[off] += 1
        return np.asarray(indices, np.int32), pairs, num
    outids = []
    for j in range(V):
        for vals, off in _valid_out_pos([int(x) for x in idx[j, 1:]], ksize, stride, padding, dilation, out_shape, transpose):
            key = (int(idx[j, 0]),) + vals
            if key not in grid:
                grid[key] = len(outids)
                outids.append(key)
            pairs[off, 0, num[off]] = j
            pairs[off, 1, num[off]] = grid[key]
            num[off] += 1
    return np.asarray(outids, np.int32).reshape(-1, D + 1), pairs, num
