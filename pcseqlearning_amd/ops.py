"""Tensor-level wrappers over the C ABI (include/pcs.h): torch owns the memory and the stream,
every computation below runs in the hand-written HIP kernels of csrc/. No CPU fallback."""
import ctypes

import torch

from . import lib as L


def _dev_scalar(device):
    return torch.empty(1, dtype=torch.int64, device=device)


# ---------------------------------------------------------------- grid sampling primitives
def minmax4(points):
    """[n,4] f32 -> [8] f32 (min0..3, max0..3), on device."""
    L.require_gpu(points, "minmax4")
    lib = L.load()
    out = torch.empty(8, dtype=torch.float32, device=points.device)
    L.check(lib.pcs_minmax4_f32(L.ptr(points), L.i64(points.shape[0]), L.ptr(out), L.stream()),
            "pcs_minmax4_f32")
    return out


def grid_cluster_ids(points, size4, margin0=0.5, bounds8=None):
    """torch_cluster.grid_cluster as GridSampling3D calls it (grid_sampling.py:32-37)."""
    L.require_gpu(points, "grid_cluster_ids")
    lib = L.load()
    points = points.contiguous()
    if bounds8 is None:
        bounds8 = minmax4(points)
    ids = torch.empty(points.shape[0], dtype=torch.int64, device=points.device)
    L.check(lib.pcs_grid_cluster_ids(L.ptr(points), L.i64(points.shape[0]), L.ptr(bounds8),
                                     L.host_floats(size4), L.f32(margin0), L.ptr(ids), L.stream()),
            "pcs_grid_cluster_ids")
    return ids


class Segments:
    """Result of a sorted unique: `order` (original indices sorted by (id, index)), `seg_start`
    ([v+1] boundaries into order), `inv` ([n] id rank), `uniq` ([v]), `num` (v, host int)."""
    __slots__ = ("order", "seg_start", "inv", "uniq", "num")


def unique_i64(ids, key_bits=64, want_uniq=True, want_inv=True):
    """torch.unique(ids, sorted=True, return_inverse=True) + segment structure. One host sync
    (the unique count), as torch.unique has."""
    L.require_gpu(ids, "unique_i64")
    lib = L.load()
    n = ids.shape[0]
    dev = ids.device
    seg = Segments()
    if n == 0:
        seg.num = 0
        seg.order = torch.empty(0, dtype=torch.int32, device=dev)
        seg.seg_start = torch.zeros(1, dtype=torch.int32, device=dev)
        seg.inv = torch.empty(0, dtype=torch.int64, device=dev) if want_inv else None
        seg.uniq = torch.empty(0, dtype=torch.int64, device=dev) if want_uniq else None
        return seg
    uniq = torch.empty(n, dtype=torch.int64, device=dev) if want_uniq else None
    inv = torch.empty(n, dtype=torch.int64, device=dev) if want_inv else None
    order = torch.empty(n, dtype=torch.int32, device=dev)
    seg_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
    d_num = _dev_scalar(dev)
    ws = L.workspace(lib.pcs_unique_i64_workspace_bytes(L.i64(n)), dev)
    L.check(lib.pcs_unique_i64(L.ptr(ids.contiguous()), L.i64(n), L.i32(key_bits), L.ptr(uniq),
                               L.ptr(inv), L.ptr(order), L.ptr(seg_start), L.ptr(d_num), L.ptr(ws),
                               L.i64(ws.numel()), L.stream()), "pcs_unique_i64")
    v = int(d_num.item())
    seg.num = v
    seg.order = order
    seg.seg_start = seg_start[:v + 1]
    seg.inv = inv
    seg.uniq = uniq[:v] if want_uniq else None
    return seg


def segment_reduce(data, seg, mean=True, wave=False, block=False):
    """scatter(data, inv, reduce='mean'/'sum'). data [n] or [n,d]. wave=False: serial ascending
    original index per segment (small segments); wave=True: one wave per segment (large ones)."""
    lib = L.load()
    squeeze = data.dim() == 1
    d2 = data.reshape(data.shape[0], -1).float().contiguous()
    out = torch.empty(seg.num, d2.shape[1], dtype=torch.float32, device=data.device)
    L.check(lib.pcs_segment_reduce_f32(L.ptr(d2), L.i32(d2.shape[1]), L.ptr(seg.order),
                                       L.ptr(seg.seg_start), L.i64(seg.num),
                                       L.i32(int(mean) | (2 if (wave and d2.shape[1] <= 16) else 0)
                                             | (4 if (block and d2.shape[1] <= 16) else 0)),
                                       L.ptr(out), L.stream()), "pcs_segment_reduce_f32")
    return out[:, 0] if squeeze else out


def segment_pick(seg, last=True):
    lib = L.load()
    out = torch.empty(seg.num, dtype=torch.int64, device=seg.order.device)
    L.check(lib.pcs_segment_pick(L.ptr(seg.order), L.ptr(seg.seg_start), L.i64(seg.num),
                                 L.i32(int(last)), L.ptr(out), L.stream()), "pcs_segment_pick")
    return out


# ---------------------------------------------------------------- voxelizers
def dynamic_voxelize(points, point_cloud_range, voxel_size, grid_size):
    """DynamicMeanVFE (dynamic_mean_vfe.py:51-76). points [n, 1+3+C] (batch, x, y, z, feat).
    Returns (voxel_features [V,3+C], voxel_coords [V,4] int32 (b,z,y,x), inv [n] int64)."""
    L.require_gpu(points, "dynamic_voxelize")
    lib = L.load()
    points = points.float().contiguous()
    n, ncol = points.shape
    dev = points.device
    feat = torch.empty(n, ncol - 1, dtype=torch.float32, device=dev)
    coords = torch.empty(n, 4, dtype=torch.int32, device=dev)
    inv = torch.empty(n, dtype=torch.int64, device=dev)
    d_num = _dev_scalar(dev)
    ws = L.workspace(lib.pcs_dynamic_voxelize_workspace_bytes(L.i64(n)), dev)
    L.check(lib.pcs_dynamic_voxelize(L.ptr(points), L.i64(n), L.i32(ncol),
                                     L.host_floats(point_cloud_range), L.host_floats(voxel_size),
                                     L.host_ints(grid_size), L.ptr(feat), L.ptr(coords),
                                     L.ptr(inv), L.ptr(d_num), L.ptr(ws), L.i64(ws.numel()),
                                     L.stream()), "pcs_dynamic_voxelize")
    v = int(d_num.item())
    return feat[:v], coords[:v], inv


def hard_voxelize(points, point_cloud_range, voxel_size, max_points, max_voxels):
    """VoxelGeneratorV2 / points_to_voxel_3d_np (point2voxel.h:33-92). points [n, 3+C].
    Returns (voxels [V,max_points,3+C], coords [V,3] int32 (z,y,x), num_points [V] int32)."""
    L.require_gpu(points, "hard_voxelize")
    lib = L.load()
    points = points.float().contiguous()
    n, ncol = points.shape
    dev = points.device
    voxels = torch.empty(max_voxels, max_points, ncol, dtype=torch.float32, device=dev)
    coords = torch.empty(max_voxels, 3, dtype=torch.int32, device=dev)
    num = torch.empty(max_voxels, dtype=torch.int32, device=dev)
    d_num = _dev_scalar(dev)
    ws = L.workspace(lib.pcs_hard_voxelize_workspace_bytes(L.i64(n)), dev)
    L.check(lib.pcs_hard_voxelize(L.ptr(points), L.i64(n), L.i32(ncol),
                                  L.host_floats(point_cloud_range), L.host_floats(voxel_size),
                                  L.i32(max_points), L.i32(max_voxels), L.ptr(voxels),
                                  L.ptr(coords), L.ptr(num), L.ptr(d_num), L.ptr(ws),
                                  L.i64(ws.numel()), L.stream()), "pcs_hard_voxelize")
    v = int(d_num.item())
    return voxels[:v], coords[:v], num[:v]


# ---------------------------------------------------------------- rulebook + conv
def get_indice_pairs(indices, batch_size, out_shape, spatial_shape, ksize, stride, padding,
                     dilation, subm, transpose=False, with_table=False):
    """sparse_conv_ext.get_indice_pairs_3d. Returns (out_ids [V',4] int32, pairs [KV,2,V] int32,
    num [KV] int32); with_table: a fourth value, the output-major neighbour table [V',KV] the conv kernels
    read when the rulebook's own probing pass can leave it (submanifold, odd kernel, no dilation), else None
    (build it from the pairs: build_neighbor_table)."""
    L.require_gpu(indices, "get_indice_pairs")
    lib = L.load()
    if indices.dtype != torch.int32:
        indices = indices.int()
    indices = indices.contiguous()
    V = indices.shape[0]
    dev = indices.device
    kv = int(ksize[0] * ksize[1] * ksize[2])
    ks, st = L.host_ints(ksize), L.host_ints(stride)
    kind = 2 if transpose else int(bool(subm))  # capacity / workspace: 0 strided, 1 submanifold, 2 transposed
    cap = int(lib.pcs_indice_pairs_out_capacity(L.i64(V), ks, st, L.i32(kind)))
    pairs = torch.empty(kv, 2, V, dtype=torch.int32, device=dev)
    num = torch.empty(kv, dtype=torch.int32, device=dev)
    out_ids = None if subm else torch.empty(max(cap, 1), 4, dtype=torch.int32, device=dev)
    d_num = _dev_scalar(dev)
    ws = L.workspace(lib.pcs_get_indice_pairs_workspace_bytes(L.i64(V), ks, st, L.i32(kind)), dev)
    dl = L.host_ints(dilation)
    h_num = ctypes.c_int64(-1)
    table = None
    if with_table and V > 0 and lib.pcs_indice_pairs_gives_table(ks, dl, L.i32(int(subm))):
        table = torch.empty(V, kv, dtype=torch.int32, device=dev)
    L.check(lib.pcs_get_indice_pairs_table(L.ptr(indices), L.i64(V), L.i32(batch_size),
                                           L.host_ints(out_shape), L.host_ints(spatial_shape), ks, st,
                                           L.host_ints(padding), dl, L.i32(int(subm)),
                                           L.i32(int(transpose)), L.ptr(out_ids), L.i64(cap),
                                           L.ptr(pairs), L.ptr(num), L.ptr(d_num), ctypes.byref(h_num), L.ptr(table),
                                           L.ptr(ws), L.i64(ws.numel()), L.stream()), "pcs_get_indice_pairs_table")
    # (the strided form hands the output count back on the host: no second device read)
    n_out = int(h_num.value) if h_num.value >= 0 else int(d_num.item())
    out = (indices, pairs, num) if subm else (out_ids[:n_out], pairs, num)
    return (*out, table) if with_table else out


def indice_conv_backward(features, filters, out_bp, pairs, num, inverse=False, subm=False):
    """sparse_conv_ext.indice_conv_backward_fp32 (spconv_ops.h:363-449): returns (input_bp like
    features, filters_bp like filters)."""
    L.require_gpu(features, "indice_conv_backward")
    lib = L.load()
    cin, cout = filters.shape[-2], filters.shape[-1]
    kv = pairs.shape[0]
    W = filters.detach().reshape(kv, cin, cout).float().contiguous()
    features = features.detach().float().contiguous()
    out_bp = out_bp.detach().float().contiguous()
    pairs = pairs.contiguous()
    num_in, num_out, stride = features.shape[0], out_bp.shape[0], pairs.shape[2]
    in_bp = torch.empty_like(features)
    w_bp = torch.empty(kv, cin, cout, dtype=torch.float32, device=features.device)
    ws = L.workspace(lib.pcs_indice_conv_backward_workspace_bytes(L.i64(num_in), L.i64(stride), L.i32(kv),
                                                                  L.i32(cin), L.i32(cout)), features.device)
    L.check(lib.pcs_indice_conv_backward_f32(L.ptr(features), L.i32(cin), L.ptr(W), L.i32(kv), L.i32(cout),
                                             L.ptr(out_bp), L.ptr(pairs), L.i64(stride), L.ptr(num),
                                             L.i64(num_in), L.i64(num_out), L.i32(int(inverse)),
                                             L.i32(int(subm)), L.ptr(in_bp), L.ptr(w_bp), L.ptr(ws),
                                             L.i64(ws.numel()), L.stream()),
            "pcs_indice_conv_backward_f32")
    return in_bp, w_bp.reshape(filters.shape)


def indice_maxpool(features, pairs, num, num_out):
    """sparse_conv_ext.indice_maxpool_fp32 (pool_ops.h:26-60): [num_out, C], zero floor."""
    L.require_gpu(features, "indice_maxpool")
    lib = L.load()
    features = features.detach().float().contiguous()
    pairs = pairs.contiguous()
    kv, C = pairs.shape[0], features.shape[1]
    out = torch.empty(num_out, C, dtype=torch.float32, device=features.device)
    ws = L.workspace(lib.pcs_neighbor_table_bytes(L.i64(num_out), L.i32(kv)), features.device)
    L.check(lib.pcs_indice_maxpool_f32(L.ptr(features), L.i32(C), L.ptr(pairs), L.i64(pairs.shape[2]), L.ptr(num),
                                       L.i32(kv), L.i64(num_out), L.ptr(out), L.ptr(ws), L.i64(ws.numel()),
                                       L.stream()), "pcs_indice_maxpool_f32")
    return out


def indice_maxpool_backward(features, out_features, out_bp, pairs, num):
    """sparse_conv_ext.indice_maxpool_backward_fp32 (pool_ops.h:62-98)."""
    L.require_gpu(features, "indice_maxpool_backward")
    lib = L.load()
    features = features.detach().float().contiguous()
    out_features = out_features.detach().float().contiguous()
    out_bp = out_bp.detach().float().contiguous()
    pairs = pairs.contiguous()
    kv, C, n_in = pairs.shape[0], features.shape[1], features.shape[0]
    in_bp = torch.empty_like(features)
    ws = L.workspace(lib.pcs_neighbor_table_bytes(L.i64(n_in), L.i32(kv)), features.device)
    L.check(lib.pcs_indice_maxpool_backward_f32(L.ptr(features), L.ptr(out_features), L.ptr(out_bp), L.i32(C),
                                                L.ptr(pairs), L.i64(pairs.shape[2]), L.ptr(num), L.i32(kv),
                                                L.i64(n_in), L.ptr(in_bp), L.ptr(ws), L.i64(ws.numel()),
                                                L.stream()), "pcs_indice_maxpool_backward_f32")
    return in_bp


def sparse_to_dense(features, indices, batch_size, spatial_shape):
    """SparseConvTensor.dense(channels_first=True): features [V,C] f32 at indices [V,4] int32
    (b,z,y,x) -> [B,C,D,H,W] f32."""
    L.require_gpu(features, "sparse_to_dense")
    lib = L.load()
    features = features.float().contiguous()
    indices = indices.int().contiguous()
    V, C = features.shape
    D, H, W = (int(x) for x in spatial_shape)
    out = torch.empty(int(batch_size), C, D, H, W, dtype=torch.float32, device=features.device)
    L.check(lib.pcs_sparse_to_dense(L.ptr(features), L.ptr(indices), L.i64(V), L.i32(C), L.i32(int(batch_size)),
                                    L.i32(D), L.i32(H), L.i32(W), L.ptr(out), L.ptr(None), L.stream()),
            "pcs_sparse_to_dense")
    return out


def build_neighbor_table(pairs, num, num_out, inverse=False):
    lib = L.load()
    kv = pairs.shape[0]
    table = torch.empty(max(num_out, 1), kv, dtype=torch.int32, device=pairs.device)
    L.check(lib.pcs_build_neighbor_table(L.ptr(pairs), L.i64(pairs.shape[2]), L.ptr(num),
                                         L.i32(kv), L.i32(int(inverse)), L.i64(num_out),
                                         L.ptr(table), L.stream()), "pcs_build_neighbor_table")
    return table[:num_out]


def bn_fold(weight, bias, running_mean, running_var, eps):
    """scale / shift [C] f32 of an eval-mode BatchNorm1d (device tensors; weight / bias may be None)."""
    L.require_gpu(running_mean, "bn_fold")
    lib = L.load()
    C = running_mean.numel()
    f = lambda t: None if t is None else t.detach().float().contiguous()
    weight, bias, mean = f(weight), f(bias), f(running_mean)
    inv = torch.rsqrt(running_var.detach().float() + eps)
    out = torch.empty(2, C, dtype=torch.float32, device=mean.device)
    L.check(lib.pcs_bn_fold_f32(L.ptr(weight), L.ptr(bias), L.ptr(mean), L.ptr(inv), L.i32(C),
                                L.ptr(out[0]), L.ptr(out[1]), L.stream()), "pcs_bn_fold_f32")
    return out[0], out[1]


def neighbor_table_order(table, num_out):
    """pcs_neighbor_table_order: the output rows sorted (stably) by the set of offsets they have a partner at, int32
    [num_out] -- the order in which the matrix-core conv kernels take the rows of a cached table (a 16-row fragment
    whose rows want the same offsets multiplies no padding). kv <= 32."""
    L.require_gpu(table, "neighbor_table_order")
    lib = L.load()
    kv = table.shape[1]
    order = torch.empty(max(num_out, 1), dtype=torch.int32, device=table.device)
    ws = L.workspace(lib.pcs_neighbor_table_order_workspace_bytes(L.i64(num_out)), table.device)
    L.check(lib.pcs_neighbor_table_order(L.ptr(table.contiguous()), L.i64(num_out), L.i32(kv), L.ptr(order), L.ptr(ws),
                                         L.i64(ws.numel()), L.stream()), "pcs_neighbor_table_order")
    return order[:num_out]


def indice_conv_table(features, filters, table, num_out, subm, bias=None, scale=None, shift=None,
                      relu=False, order=None, residual=None):
    """out = epilogue(sum_k feat[table[:,k]] @ W[k]); filters [..., Cin, Cout]. order: neighbor_table_order(table) or
    None (the same results either way). residual [num_out, Cout] f32: added after scale / shift, before the ReLU
    (pcs_indice_conv_table_residual_f32: a residual block's `+ identity` in the conv's epilogue)."""
    L.require_gpu(features, "indice_conv")
    lib = L.load()
    cin, cout = filters.shape[-2], filters.shape[-1]
    kv = table.shape[1]
    W = filters.reshape(kv, cin, cout).float().contiguous()
    features = features.float().contiguous()
    out = torch.empty(num_out, cout, dtype=torch.float32, device=features.device)
    centre = kv // 2 if (subm and kv % 2 == 1) else -1
    if residual is not None:
        L.require_gpu(residual, "indice_conv residual")
        residual = residual.float().contiguous()
        assert tuple(residual.shape) == (num_out, cout), "residual must be [num_out, Cout]"
        L.check(lib.pcs_indice_conv_table_residual_f32(L.ptr(features), L.i32(cin), L.ptr(W), L.i32(kv),
                                                       L.i32(cout), L.i32(centre), L.ptr(table.contiguous()),
                                                       L.i64(num_out), L.ptr(bias), L.ptr(scale), L.ptr(shift),
                                                       L.ptr(residual), L.i32(int(relu)), L.ptr(out), L.stream()),
                "pcs_indice_conv_table_residual_f32")
        return out
    L.check(lib.pcs_indice_conv_table_f32(L.ptr(features), L.i32(cin), L.ptr(W), L.i32(kv),
                                          L.i32(cout), L.i32(centre), L.ptr(table.contiguous()),
                                          L.i64(num_out), L.ptr(bias), L.ptr(scale), L.ptr(shift),
                                          L.i32(int(relu)), L.ptr(out), L.stream()),
            "pcs_indice_conv_table_f32")
    return out


def indice_conv(features, filters, pairs, num, num_out, inverse=False, subm=False):
    """sparse_conv_ext.indice_conv_fp32 (spconv_ops.h:260-361)."""
    L.require_gpu(features, "indice_conv")
    lib = L.load()
    cin, cout = filters.shape[-2], filters.shape[-1]
    kv = pairs.shape[0]
    W = filters.reshape(kv, cin, cout).float().contiguous()
    features = features.float().contiguous()
    out = torch.empty(num_out, cout, dtype=torch.float32, device=features.device)
    ws = L.workspace(lib.pcs_neighbor_table_bytes(L.i64(num_out), L.i32(kv)), features.device)
    L.check(lib.pcs_indice_conv_f32(L.ptr(features), L.i32(cin), L.ptr(W), L.i32(kv), L.i32(cout),
                                    L.ptr(pairs.contiguous()), L.i64(pairs.shape[2]), L.ptr(num),
                                    L.i64(num_out), L.i32(int(inverse)), L.i32(int(subm)),
                                    L.ptr(out), L.ptr(ws), L.i64(ws.numel()), L.stream()),
            "pcs_indice_conv_f32")
    return out


def indice_conv_half(features, filters, pairs, num, num_out, inverse=False, subm=False, bias=None):
    """sparse_conv_ext.indice_conv_half / fused_indice_conv_half (src/all.cc:35-41): fp16 in, fp16 out,
    fp32 accumulation (pcs_indice_conv_f16). `bias`: the fused form's [cout] fp16 vector."""
    L.require_gpu(features, "indice_conv_half")
    lib = L.load()
    cin, cout = filters.shape[-2], filters.shape[-1]
    kv = pairs.shape[0]
    W = filters.reshape(kv, cin, cout).half().contiguous()
    features = features.half().contiguous()
    b = bias.half().contiguous() if bias is not None else None
    num_in = features.shape[0]
    out = torch.empty(num_out, cout, dtype=torch.float16, device=features.device)
    ws = L.workspace(lib.pcs_indice_conv_f16_workspace_bytes(L.i64(num_in), L.i32(cin), L.i32(kv), L.i32(cout),
                                                             L.i64(num_out)), features.device)
    L.check(lib.pcs_indice_conv_f16(L.ptr(features), L.i64(num_in), L.i32(cin), L.ptr(W), L.i32(kv), L.i32(cout),
                                    L.ptr(b), L.ptr(pairs.contiguous()), L.i64(pairs.shape[2]), L.ptr(num),
                                    L.i64(num_out), L.i32(int(inverse)), L.i32(int(subm)), L.ptr(out), L.ptr(ws),
                                    L.i64(ws.numel()), L.stream()), "pcs_indice_conv_f16")
    return out


def indice_maxpool_half(features, pairs, num, num_out):
    """sparse_conv_ext.indice_maxpool_half (src/all.cc:46-47): exact (a maximum of fp16 values is one of them)."""
    L.require_gpu(features, "indice_maxpool_half")
    lib = L.load()
    features = features.half().contiguous()
    pairs = pairs.contiguous()
    kv, C, num_in = pairs.shape[0], features.shape[1], features.shape[0]
    out = torch.empty(num_out, C, dtype=torch.float16, device=features.device)
    ws = L.workspace(lib.pcs_indice_maxpool_f16_workspace_bytes(L.i64(num_in), L.i32(C), L.i32(kv), L.i64(num_out)),
                     features.device)
    L.check(lib.pcs_indice_maxpool_f16(L.ptr(features), L.i64(num_in), L.i32(C), L.ptr(pairs), L.i64(pairs.shape[2]),
                                       L.ptr(num), L.i32(kv), L.i64(num_out), L.ptr(out), L.ptr(ws), L.i64(ws.numel()),
                                       L.stream()), "pcs_indice_maxpool_f16")
    return out


def indice_conv_table_half(features, filters, table, num_out, subm, bias=None, relu=False, order=None, residual=None):
    """pcs_indice_conv_table_ordered_f16: the fp16 conv on a cached neighbour table; filters [..., Cin, Cout], bias
    [Cout]; order: neighbor_table_order(table) or None. residual [num_out, Cout] fp16: added to the conv's fp16 result
    before the ReLU (pcs_indice_conv_table_residual_f16)."""
    L.require_gpu(features, "indice_conv_table_half")
    lib = L.load()
    cin, cout = filters.shape[-2], filters.shape[-1]
    kv = table.shape[1]
    W = filters.reshape(kv, cin, cout).half().contiguous()
    features = features.half().contiguous()
    b = bias.half().contiguous() if bias is not None else None
    out = torch.empty(num_out, cout, dtype=torch.float16, device=features.device)
    centre = kv // 2 if (subm and kv % 2 == 1) else -1
    if residual is not None:
        L.require_gpu(residual, "indice_conv_table_half residual")
        residual = residual.half().contiguous()
        assert tuple(residual.shape) == (num_out, cout), "residual must be [num_out, Cout]"
        L.check(lib.pcs_indice_conv_table_residual_f16(L.ptr(features), L.i32(cin), L.ptr(W), L.i32(kv), L.i32(cout),
                                                       L.i32(centre), L.ptr(table.contiguous()), L.i64(num_out), L.ptr(b),
                                                       L.ptr(residual), L.i32(int(relu)), L.ptr(out), L.stream()),
                "pcs_indice_conv_table_residual_f16")
        return out
    L.check(lib.pcs_indice_conv_table_ordered_f16(L.ptr(features), L.i32(cin), L.ptr(W), L.i32(kv), L.i32(cout),
                                                  L.i32(centre), L.ptr(table.contiguous()), L.i64(num_out), L.ptr(b),
                                                  L.i32(int(relu)), L.ptr(out), L.ptr(order), L.stream()),
            "pcs_indice_conv_table_ordered_f16")
    return out


def indice_conv_backward_half(features, filters, out_grad, pairs, num, inverse=False, subm=False):
    """sparse_conv_ext.indice_conv_backward_half (src/all.cc:36-37): (in_grad [num_in,cin], filters_grad like
    filters), fp16, from fp32 accumulations rounded once (pcs_indice_conv_backward_f16)."""
    L.require_gpu(features, "indice_conv_backward_half")
    lib = L.load()
    cin, cout = filters.shape[-2], filters.shape[-1]
    kv = pairs.shape[0]
    W = filters.reshape(kv, cin, cout).half().contiguous()
    features = features.half().contiguous()
    out_grad = out_grad.half().contiguous()
    pairs = pairs.contiguous()
    num_in, num_out = features.shape[0], out_grad.shape[0]
    in_grad = torch.empty(num_in, cin, dtype=torch.float16, device=features.device)
    w_grad = torch.empty(kv, cin, cout, dtype=torch.float16, device=features.device)
    ws = L.workspace(lib.pcs_indice_conv_backward_f16_workspace_bytes(L.i64(num_in), L.i64(pairs.shape[2]), L.i32(kv),
                                                                      L.i32(cin), L.i32(cout)), features.device)
    L.check(lib.pcs_indice_conv_backward_f16(L.ptr(features), L.i32(cin), L.ptr(W), L.i32(kv), L.i32(cout),
                                             L.ptr(out_grad), L.ptr(pairs), L.i64(pairs.shape[2]), L.ptr(num),
                                             L.i64(num_in), L.i64(num_out), L.i32(int(inverse)), L.i32(int(subm)),
                                             L.ptr(in_grad), L.ptr(w_grad), L.ptr(ws), L.i64(ws.numel()), L.stream()),
            "pcs_indice_conv_backward_f16")
    return in_grad, w_grad.reshape(filters.shape)


def indice_maxpool_backward_half(features, out_features, out_grad, pairs, num):
    """sparse_conv_ext.indice_maxpool_backward_half (src/all.cc:48-50): fp16 in / out, the selected gradients added
    in fp16 in ascending offset like the reference's functor on at::Half."""
    L.require_gpu(features, "indice_maxpool_backward_half")
    lib = L.load()
    features = features.half().contiguous()
    out_features = out_features.half().contiguous()
    out_grad = out_grad.half().contiguous()
    pairs = pairs.contiguous()
    kv, C, num_in = pairs.shape[0], features.shape[1], features.shape[0]
    in_grad = torch.empty(num_in, C, dtype=torch.float16, device=features.device)
    ws = L.workspace(lib.pcs_neighbor_table_bytes(L.i64(num_in), L.i32(kv)), features.device)
    L.check(lib.pcs_indice_maxpool_backward_f16(L.ptr(features), L.ptr(out_features), L.ptr(out_grad), L.i32(C),
                                                L.ptr(pairs), L.i64(pairs.shape[2]), L.ptr(num), L.i32(kv),
                                                L.i64(num_in), L.ptr(in_grad), L.ptr(ws), L.i64(ws.numel()),
                                                L.stream()), "pcs_indice_maxpool_backward_f16")
    return in_grad


# ---------------------------------------------------------------- radius graph
def radius_graph(ref, query, radius, max_nbr, sort_by_dist=True, dframe_min=0, dframe_max=0,
                 cell_size=0.0, want_dists=False):
    """Edges (e_ref, e_query) int64 of RadiusGraph.build_graph (graph_utils.py:149-209):
    grouped by ascending query, within a query ascending (dist2, ref index). One host sync (E)."""
    L.require_gpu(ref, "radius_graph")
    lib = L.load()
    ref = ref.float().contiguous()
    query = query.float().contiguous()
    N, M = ref.shape[0], query.shape[0]
    dev = ref.device
    d_num = _dev_scalar(dev)
    nbytes = lib.pcs_radius_graph_workspace_bytes(L.i64(N), L.i64(M), L.i32(max_nbr))
    ws = L.workspace(nbytes, dev)
    L.check(lib.pcs_radius_graph_search(L.ptr(ref), L.i64(N), L.ptr(query), L.i64(M), L.f32(radius),
                                        L.f32(cell_size), L.i32(dframe_min), L.i32(dframe_max),
                                        L.i32(max_nbr), L.i32(int(sort_by_dist)), L.ptr(d_num),
                                        L.ptr(ws), L.i64(ws.numel()), L.stream()),
            "pcs_radius_graph_search")
    E = int(d_num.item())
    if E < 0 and max_nbr == 1 and cell_size > 0:
        # the two-level 1-NN grid did not cover the search ball with this cell size: general engine
        cell_size = -float(cell_size)
        L.check(lib.pcs_radius_graph_search(L.ptr(ref), L.i64(N), L.ptr(query), L.i64(M), L.f32(radius),
                                            L.f32(cell_size), L.i32(dframe_min), L.i32(dframe_max),
                                            L.i32(max_nbr), L.i32(int(sort_by_dist)), L.ptr(d_num),
                                            L.ptr(ws), L.i64(ws.numel()), L.stream()),
                "pcs_radius_graph_search")
        E = int(d_num.item())
    if E < 0:
        raise L.PcsError("radius_graph: extent / cell_size exceeds the 17-bit cell range")
    e_ref = torch.empty(E, dtype=torch.int64, device=dev)
    e_query = torch.empty(E, dtype=torch.int64, device=dev)
    dists = torch.empty(E, dtype=torch.float32, device=dev) if want_dists else None
    if E > 0:
        L.check(lib.pcs_radius_graph_emit(L.i64(N), L.i64(M), L.i32(max_nbr), L.ptr(e_ref),
                                          L.ptr(e_query), L.ptr(dists), L.ptr(ws),
                                          L.i64(ws.numel()), L.stream()), "pcs_radius_graph_emit")
    if want_dists:
        return e_ref, e_query, dists
    return e_ref, e_query


def knn_components(points, radii, max_nbr=32):
    """ClusterProposal's unit of work (cluster_proposal.py:57-83 of the reference) for several radius keys at once: the
    max_nbr-nearest radius graph of `points` [n,4] on itself at the LARGEST radius (graph_utils.py:149-209), then the
    connected components (graph_utils.py:40-53) of its cut at every radius of `radii` -- the k nearest within r,
    nearest first, are the k nearest within R >= r cut at r -- straight from the neighbour table: no edge list.
    Returns (component [len(radii), n] int64, num [len(radii)] int64 on the device, E of the largest graph). One host
    synchronisation (the search's status / edge count)."""
    L.require_gpu(points, "knn_components")
    lib = L.load()
    pts = points.float().contiguous()
    n = pts.shape[0]
    dev = pts.device
    r32 = [float(torch.tensor(float(r), dtype=torch.float32)) for r in radii]
    big = max(r32)
    d_num = _dev_scalar(dev)
    ws = L.workspace(lib.pcs_radius_graph_workspace_bytes(L.i64(n), L.i64(n), L.i32(max_nbr)), dev)
    L.check(lib.pcs_radius_graph_search(L.ptr(pts), L.i64(n), L.ptr(pts), L.i64(n), L.f32(big), L.f32(-big / 2),
                                        L.i32(0), L.i32(0), L.i32(max_nbr), L.i32(1), L.ptr(d_num), L.ptr(ws),
                                        L.i64(ws.numel()), L.stream()), "pcs_radius_graph_search")
    comp = torch.empty(len(r32), n, dtype=torch.int64, device=dev)
    num = torch.empty(len(r32), dtype=torch.int64, device=dev)
    ws_cc = L.workspace(lib.pcs_radius_graph_components_workspace_bytes(L.i64(n), L.i32(len(r32))), dev)
    L.check(lib.pcs_radius_graph_components(L.i64(n), L.i64(n), L.i32(max_nbr), L.host_floats(r32), L.i32(len(r32)),
                                            L.ptr(comp), L.ptr(num), L.ptr(ws), L.i64(ws.numel()), L.ptr(ws_cc),
                                            L.i64(ws_cc.numel()), L.stream()), "pcs_radius_graph_components")
    E = int(d_num.item())
    if E < 0:
        raise L.PcsError("knn_components: extent / cell_size exceeds the 17-bit cell range")
    return comp, num, E


def hash_insert_gpu(keys, values, reverse_indices, dims, insert_keys, insert_values):
    """torch_hash_cuda.hash_insert_gpu (torch_hash_kernel.cu:411-442), same argument order."""
    L.require_gpu(keys, "hash_insert_gpu")
    lib = L.load()
    D = insert_values.shape[1]
    L.check(lib.pcs_hash_insert(L.ptr(keys), L.ptr(values), L.ptr(reverse_indices),
                                L.i64(keys.shape[0]), L.ptr(dims.contiguous()), L.i32(D),
                                L.ptr(insert_keys.contiguous()), L.ptr(insert_values.contiguous()),
                                L.i64(insert_keys.shape[0]), L.stream()), "pcs_hash_insert")


def radius_graph_gpu(keys, values, reverse_indices, dims, query_keys, query_values, qmin, qmax,
                     radius, max_num_neighbors, sort_by_dist):
    """torch_hash_cuda.radius_graph_gpu (torch_hash_kernel.cu:487-561): returns edges [E,2]."""
    L.require_gpu(keys, "radius_graph_gpu")
    lib = L.load()
    M, D = query_values.shape
    dev = keys.device
    degree = torch.empty(M, dtype=torch.int32, device=dev)
    offset = torch.empty(M, dtype=torch.int32, device=dev)
    d_num = _dev_scalar(dev)
    ws = L.workspace(lib.pcs_radius_graph_table_workspace_bytes(L.i64(M)), dev)
    args = (L.ptr(keys), L.ptr(values), L.ptr(reverse_indices), L.i64(keys.shape[0]),
            L.ptr(dims.contiguous()), L.i32(D), L.ptr(query_keys.contiguous()),
            L.ptr(query_values.contiguous()), L.i64(M), L.ptr(qmin.contiguous()),
            L.ptr(qmax.contiguous()), L.ptr(radius.contiguous()))
    L.check(lib.pcs_radius_graph_table_count(*args, L.i32(max_num_neighbors), L.ptr(degree),
                                             L.ptr(offset), L.ptr(d_num), L.ptr(ws),
                                             L.i64(ws.numel()), L.stream()),
            "pcs_radius_graph_table_count")
    E = int(d_num.item())
    edges = torch.zeros(E, 2, dtype=torch.int64, device=dev)
    dists = torch.zeros(E, dtype=torch.float32, device=dev)
    if E == 0:
        return edges
    L.check(lib.pcs_radius_graph_table_fill(*args, L.i32(int(sort_by_dist)), L.ptr(degree),
                                            L.ptr(offset), L.ptr(edges), L.ptr(dists), L.stream()),
            "pcs_radius_graph_table_fill")
    return edges


def correspondence(keys, values, reverse_indices, dims, query_keys, query_values, qmin, qmax,
                   corres_indices):
    """torch_hash_cuda.correspondence (torch_hash_kernel.cu:444-485), same argument order; fills
    corres_indices [M] int64 in place (nearest hashed point among the walked cells, -1 if none)."""
    L.require_gpu(keys, "correspondence")
    lib = L.load()
    M, D = query_values.shape
    assert corres_indices.is_contiguous() and corres_indices.dtype == torch.int64 and corres_indices.shape[0] == M
    L.check(lib.pcs_correspondence(
        L.ptr(keys), L.ptr(values), L.ptr(reverse_indices), L.i64(keys.shape[0]), L.ptr(dims.contiguous()),
        L.i32(D), L.ptr(query_keys.contiguous()), L.ptr(query_values.contiguous()), L.i64(M),
        L.ptr(qmin.contiguous()), L.ptr(qmax.contiguous()), L.ptr(corres_indices), L.stream()),
        "pcs_correspondence")


def points_in_radius_gpu(keys, values, reverse_indices, dims, query_keys, query_values, qmin, qmax,
                         radius, visited):
    """torch_hash_cuda.points_in_radius_gpu (torch_hash_kernel.cu:563-605), same argument order; sets
    visited [N] int64 to 1 for hashed points strictly closer than `radius` to some query."""
    L.require_gpu(keys, "points_in_radius_gpu")
    lib = L.load()
    M, D = query_values.shape
    assert visited.is_contiguous() and visited.dtype == torch.int64
    L.check(lib.pcs_points_in_radius(
        L.ptr(keys), L.ptr(values), L.ptr(reverse_indices), L.i64(keys.shape[0]), L.ptr(dims.contiguous()),
        L.i32(D), L.ptr(query_keys.contiguous()), L.ptr(query_values.contiguous()), L.i64(M),
        L.ptr(qmin.contiguous()), L.ptr(qmax.contiguous()), L.f32(radius), L.ptr(visited), L.stream()),
        "pcs_points_in_radius")


# ---------------------------------------------------------------- components / medians
def connected_components(e0, e1, num_nodes, host_count=True):
    """graph_utils.connected_components: (num_components, component [N] int64). host_count=False: the count stays
    on the device (a one-element int64 tensor): no host synchronisation."""
    L.require_gpu(e0, "connected_components")
    lib = L.load()
    dev = e0.device
    comp = torch.empty(num_nodes, dtype=torch.int64, device=dev)
    d_num = _dev_scalar(dev)
    ws = L.workspace(lib.pcs_connected_components_workspace_bytes(L.i64(num_nodes)), dev)
    L.check(lib.pcs_connected_components(L.ptr(e0.contiguous()), L.ptr(e1.contiguous()),
                                         L.i64(e0.shape[0]), L.i64(num_nodes), L.ptr(comp),
                                         L.ptr(d_num), L.ptr(ws), L.i64(ws.numel()), L.stream()),
            "pcs_connected_components")
    return (int(d_num.item()) if host_count else d_num), comp


def segment_median(data, seg):
    lib = L.load()
    data = data.long().contiguous()
    out = torch.empty(seg.num, dtype=torch.int64, device=data.device)
    L.check(lib.pcs_segment_median_i64(L.ptr(data), L.ptr(seg.order), L.ptr(seg.seg_start),
                                       L.i64(seg.num), L.ptr(out), L.stream()),
            "pcs_segment_median_i64")
    return out


def segment_minmax(data, seg, is_max=True):
    lib = L.load()
    squeeze = data.dim() == 1
    d2 = data.reshape(data.shape[0], -1).float().contiguous()
    out = torch.empty(seg.num, d2.shape[1], dtype=torch.float32, device=data.device)
    L.check(lib.pcs_segment_minmax_f32(L.ptr(d2), L.i32(d2.shape[1]), L.ptr(seg.order),
                                       L.ptr(seg.seg_start), L.i64(seg.num), L.i32(int(is_max)),
                                       L.ptr(out), L.stream()), "pcs_segment_minmax_f32")
    return out[:, 0] if squeeze else out


# ---------------------------------------------------------------- registration
def register_to_next_frame(mov_fxyz, mov_comp, ref_fxyz, ref_all, comp_deg, num_components, radius,
                           frame_offset, angle_regularizer=10.0, max_iter=80, stopping_delta=5e-2,
                           cell_size=0.0, chunk=8):
    """Fused ICP of registration_utils.register_to_next_frame on the non-stationary subsets.
    mov_fxyz is updated in place. Returns (T [C,4,4] f64, l1_err [C] f64, edge_ratio [C] f32,
    iterations)."""
    import ctypes
    L.require_gpu(mov_fxyz, "register_to_next_frame")
    lib = L.load()
    dev = mov_fxyz.device
    assert mov_fxyz.is_contiguous() and mov_fxyz.dtype == torch.float32
    comp32 = mov_comp.int().contiguous()
    ref_fxyz = ref_fxyz.float().contiguous()
    ref_all = ref_all.float().contiguous()
    C = int(num_components)
    T = torch.empty(C, 4, 4, dtype=torch.float64, device=dev)
    l1 = torch.empty(C, dtype=torch.float64, device=dev)
    ratio = torch.empty(C, dtype=torch.float32, device=dev)
    info = (ctypes.c_int32 * 4)()
    Mn, Rn, Ra = mov_fxyz.shape[0], ref_fxyz.shape[0], ref_all.shape[0]
    ws = L.workspace(lib.pcs_register_to_next_frame_workspace_bytes(L.i64(Mn), L.i64(Rn), L.i64(Ra),
                                                                    L.i64(C)), dev)
    L.check(lib.pcs_register_to_next_frame(
        L.ptr(mov_fxyz), L.ptr(comp32), L.i64(Mn), L.ptr(ref_fxyz), L.i64(Rn), L.ptr(ref_all),
        L.i64(Ra), L.ptr(comp_deg.long().contiguous()), L.i64(C), ctypes.c_double(radius),
        L.i32(frame_offset), ctypes.c_double(angle_regularizer), L.i32(max_iter),
        ctypes.c_double(stopping_delta), L.f32(cell_size), L.i32(chunk), L.ptr(T), L.ptr(l1),
        L.ptr(ratio), info, L.ptr(ws), L.i64(ws.numel()), L.stream()),
        "pcs_register_to_next_frame")
    return T, l1, ratio, int(info[0])


def smooth_velo(comp_velos, comp_center_diffs, frame_id, next_frame_id, weight0=1.0, weight=10.0,
                num_itr=300, stopping=1e-3):
    """cluster_tracking.smooth_velo: in place on comp_velos [C,F,3] f32 (and returns it)."""
    if frame_id == next_frame_id:
        return comp_velos
    L.require_gpu(comp_velos, "smooth_velo")
    lib = L.load()
    lo, hi = (frame_id, next_frame_id) if frame_id < next_frame_id else (next_frame_id, frame_id)
    assert comp_velos.is_contiguous() and comp_velos.dtype == torch.float32
    C, F, _ = comp_velos.shape
    ws = L.workspace(lib.pcs_smooth_velo_workspace_bytes(L.i64(C), L.i32(F)), comp_velos.device)
    L.check(lib.pcs_smooth_velo(L.ptr(comp_velos), L.ptr(comp_center_diffs.float().contiguous()),
                                L.i64(C), L.i32(F), L.i32(lo), L.i32(hi), L.f32(weight0),
                                L.f32(weight), L.i32(num_itr), L.f32(stopping), L.ptr(ws),
                                L.i64(ws.numel()), L.stream()), "pcs_smooth_velo")
    return comp_velos


def smooth_velo_batch(problems, weight0=1.0, weight=10.0, num_itr=300, stopping=1e-3):
    """smooth_velo for several problems in one call: problems = [(comp_velos [C,F,3] f32 (in place),
    comp_center_diffs, frame_id, next_frame_id), ...]; what the sequence tracker does at every frame step."""
    import ctypes
    lib = L.load()
    n = len(problems)
    if n == 0:
        return []
    keep, ws_list = [], []
    vel = (ctypes.c_void_p * n)()
    dif = (ctypes.c_void_p * n)()
    wsp = (ctypes.c_void_p * n)()
    Cs = (ctypes.c_int64 * n)()
    Fs = (ctypes.c_int * n)()
    lo = (ctypes.c_int * n)()
    hi = (ctypes.c_int * n)()
    wsb = (ctypes.c_size_t * n)()
    for e, (v, d, a, b) in enumerate(problems):
        L.require_gpu(v, "smooth_velo_batch")
        assert v.is_contiguous() and v.dtype == torch.float32
        d = d.float().contiguous()
        keep.append(d)
        C, F, _ = v.shape
        ws = L.workspace(lib.pcs_smooth_velo_workspace_bytes(L.i64(C), L.i32(F)), v.device)
        ws_list.append(ws)
        vel[e], dif[e], wsp[e] = v.data_ptr(), d.data_ptr(), ws.data_ptr()
        Cs[e], Fs[e], lo[e], hi[e], wsb[e] = C, F, min(a, b), max(a, b), ws.numel()
    L.check(lib.pcs_smooth_velo_batch(L.i32(n), vel, dif, Cs, Fs, lo, hi, L.f32(weight0), L.f32(weight),
                                      L.i32(num_itr), L.f32(stopping), wsp, wsb, L.stream()),
            "pcs_smooth_velo_batch")
    return [p[0] for p in problems]


def points_in_boxes(points, boxes):
    """roiaware_pool3d_utils.points_in_boxes_cpu semantics on the device: [B, N] int32."""
    L.require_gpu(points, "points_in_boxes")
    lib = L.load()
    points = points.float().contiguous()
    boxes = boxes.float().contiguous()
    assert boxes.shape[1] == 7 and points.shape[1] == 3
    out = torch.zeros(boxes.shape[0], points.shape[0], dtype=torch.int32, device=points.device)
    L.check(lib.pcs_points_in_boxes(L.ptr(points), L.i64(points.shape[0]), L.ptr(boxes),
                                    L.i32(boxes.shape[0]), L.ptr(out), L.stream()),
            "pcs_points_in_boxes")
    return out


def points_in_frame_boxes(points, point_frame, boxes, box_offsets, bmax, want_member=True, want_first=True):
    """points_in_boxes of every point against the boxes of its own frame (boxes grouped by frame,
    `box_offsets` device int32 [F+1]). Returns (member [N,bmax] int32 or None, first [N] int32 or None)."""
    L.require_gpu(points, "points_in_frame_boxes")
    lib = L.load()
    points = points.float().contiguous()
    frame = point_frame.long().contiguous()
    boxes = boxes.float().contiguous()
    assert points.shape[1] == 3 and (boxes.numel() == 0 or boxes.shape[1] == 7)
    assert box_offsets.dtype == torch.int32 and box_offsets.is_cuda
    n, F = points.shape[0], box_offsets.shape[0] - 1
    member = torch.empty(n, bmax, dtype=torch.int32, device=points.device) if want_member else None
    first = torch.empty(n, dtype=torch.int32, device=points.device) if want_first else None
    L.check(lib.pcs_points_in_frame_boxes(L.ptr(points), L.ptr(frame), L.i64(n), L.ptr(boxes),
                                          L.ptr(box_offsets.contiguous()), L.i32(F), L.i32(bmax),
                                          L.ptr(member), L.ptr(first), L.stream()),
            "pcs_points_in_frame_boxes")
    return member, first


def frame_box_counts(points, point_frame, boxes, box_offsets, bmax, group, num_groups):
    """counts [num_groups, bmax] int32: how many points of every group lie in every box of their frame (the test of
    points_in_frame_boxes, the sum of component_box_iou_frames' index_add) without the [N, bmax] membership matrix."""
    L.require_gpu(points, "frame_box_counts")
    lib = L.load()
    points = points.float().contiguous()
    frame = point_frame.long().contiguous()
    boxes = boxes.float().contiguous()
    group = group.long().contiguous()
    assert points.shape[1] == 3 and box_offsets.dtype == torch.int32 and box_offsets.is_cuda
    counts = torch.zeros(num_groups, bmax, dtype=torch.int32, device=points.device)
    L.check(lib.pcs_frame_box_counts(L.ptr(points), L.ptr(frame), L.i64(points.shape[0]), L.ptr(boxes),
                                     L.ptr(box_offsets.contiguous()), L.i32(box_offsets.shape[0] - 1), L.i32(bmax),
                                     L.ptr(group), L.i64(num_groups), L.ptr(counts), L.stream()), "pcs_frame_box_counts")
    return counts


def assemble_sweeps(raw, seg, sweep_offsets, sweep_T, sweep_ids, batch_index=0):
    """Device side of the sequence loader (WaymoDataset.get_lidar / __getitem__ + the point_xyz
    branch of collate_batch): raw [N,8] f32 rows of all sweeps -> (point_bxyz [N,4] f32, point_feat
    [N,3] f32, point_sweep [N,1] int32, segmentation_label [N] int32, instance_label [N] int32);
    the two label arrays are None when `seg` is None. sweep_offsets int64 [F+1], sweep_T f64
    [F,3,4], sweep_ids int32 [F], all on the device."""
    L.require_gpu(raw, "assemble_sweeps")
    lib = L.load()
    dev = raw.device
    assert raw.dtype == torch.float32 and raw.dim() == 2 and raw.shape[1] == 8 and raw.is_contiguous()
    N, F = raw.shape[0], sweep_ids.shape[0]
    assert sweep_offsets.dtype == torch.int64 and sweep_offsets.shape[0] == F + 1 and sweep_offsets.is_cuda
    assert sweep_T.dtype == torch.float64 and tuple(sweep_T.shape) == (F, 3, 4) and sweep_T.is_cuda
    assert sweep_ids.dtype == torch.int32 and sweep_ids.is_cuda
    if seg is not None:
        assert seg.dtype == torch.int32 and tuple(seg.shape) == (N, 2) and seg.is_contiguous() and seg.is_cuda
    bxyz = torch.empty(N, 4, dtype=torch.float32, device=dev)
    feat = torch.empty(N, 3, dtype=torch.float32, device=dev)
    sweep = torch.empty(N, 1, dtype=torch.int32, device=dev)
    seg_label = torch.empty(N, dtype=torch.int32, device=dev) if seg is not None else None
    inst_label = torch.empty(N, dtype=torch.int32, device=dev) if seg is not None else None
    L.check(lib.pcs_assemble_sweeps(L.ptr(raw), L.ptr(seg), L.ptr(sweep_offsets.contiguous()),
                                    L.ptr(sweep_T.contiguous()), L.ptr(sweep_ids.contiguous()), L.i32(F),
                                    L.i64(N), L.i32(batch_index), L.ptr(bxyz), L.ptr(feat), L.ptr(sweep),
                                    L.ptr(seg_label), L.ptr(inst_label), L.stream()),
            "pcs_assemble_sweeps")
    return bxyz, feat, sweep, seg_label, inst_label


def l1_minimization(pillar_min_z, weight, lr, decay_steps, rigid_weight, max_iters):
    """preprocessor_utils.l1_minimization: returns pillar_height [X,Y] f32."""
    L.require_gpu(pillar_min_z, "l1_minimization")
    lib = L.load()
    X, Y = pillar_min_z.shape
    mz = pillar_min_z.float().contiguous()
    w = weight.float().reshape(X, Y).contiguous()
    out = torch.empty(X, Y, dtype=torch.float32, device=mz.device)
    ws = L.workspace(lib.pcs_l1_minimization_workspace_bytes(L.i32(X), L.i32(Y)), mz.device)
    import ctypes
    L.check(lib.pcs_l1_minimization(L.ptr(mz), L.ptr(w), L.i32(X), L.i32(Y), ctypes.c_double(lr),
                                    L.host_ints(list(decay_steps)), L.i32(len(decay_steps)),
                                    L.f32(rigid_weight), L.i32(max_iters), L.ptr(out),
                                    ctypes.c_void_p(0), L.ptr(ws), L.i64(ws.numel()), L.stream()),
            "pcs_l1_minimization")
    return out


def irls_plane_fit(xyz, pillar_idx, w0, origin, num_pillars, sigma2, stopping_delta=1e-2, max_rounds=50,
                   seg_start=None):
    """preprocessor_utils.iterative_reweighted_ransac with the convergence test on the device (no
    host sync). `seg_start` ([P+1] int32 row ranges) says the rows are grouped by pillar: one launch
    per round with fixed-order sums; without it the rows are grouped here first (same sums).
    Returns (center [P,3], normal [P,3], plane_fitting_error [n])."""
    import ctypes
    L.require_gpu(xyz, "irls_plane_fit")
    lib = L.load()
    xyz = xyz.float().contiguous()
    w0 = w0.float().reshape(-1).contiguous()
    origin = origin.float().contiguous()
    n, P = xyz.shape[0], int(num_pillars)
    plane = torch.zeros(P, 6, dtype=torch.float32, device=xyz.device)
    perr = torch.empty(n, dtype=torch.float32, device=xyz.device)
    if seg_start is not None:
        assert seg_start.dtype == torch.int32 and seg_start.shape[0] == P + 1
        ws = L.workspace(lib.pcs_irls_plane_fit_sorted_workspace_bytes(L.i64(n), L.i64(P)), xyz.device)
        L.check(lib.pcs_irls_plane_fit_sorted(L.ptr(xyz), L.ptr(seg_start.contiguous()), L.i64(n), L.ptr(w0),
                                              L.ptr(origin), L.i64(P), L.f32(sigma2), L.f32(stopping_delta),
                                              L.i32(max_rounds), L.ptr(plane), L.ptr(perr),
                                              ctypes.c_void_p(0), L.ptr(ws), L.i64(ws.numel()),
                                              L.stream()), "pcs_irls_plane_fit_sorted")
        return plane[:, :3], plane[:, 3:], perr
    # rows in any order: grouped by pillar here (stable, so the rows of a pillar keep their order) and given to
    # the sorted form -- its sums have a fixed order, the unsorted entry (pcs_irls_plane_fit) adds f64 terms with
    # atomics and can differ in the last bits from run to run
    pidx = pillar_idx.long().contiguous()
    order = torch.argsort(pidx, stable=True)
    counts = torch.bincount(pidx, minlength=P)
    seg = torch.zeros(P + 1, dtype=torch.int32, device=xyz.device)
    seg[1:] = torch.cumsum(counts, 0).to(torch.int32)
    center, normal, perr_sorted = irls_plane_fit(xyz[order], pidx[order], w0[order], origin, P, sigma2, stopping_delta,
                                                 max_rounds, seg_start=seg)
    perr[order] = perr_sorted
    return center, normal, perr


def ransac_plane_hypotheses(xyz, seg_start, min_z, max_z, ratios, origin, sigma2, stopping_delta=1e-2,
                            max_rounds=50, hit_below=None):
    """All height-ratio hypotheses of compute_min_height_from_ransac in one call (the R IRLS fits run
    side by side). Returns (best_center [P,3], best_normal [P,3], best_confidence [P])."""
    L.require_gpu(xyz, "ransac_plane_hypotheses")
    lib = L.load()
    xyz = xyz.float().contiguous()
    n, P, R = xyz.shape[0], min_z.shape[0], len(ratios)
    assert seg_start.dtype == torch.int32 and seg_start.shape[0] == P + 1
    dev = xyz.device
    center = torch.empty(P, 3, dtype=torch.float32, device=dev)
    normal = torch.empty(P, 3, dtype=torch.float32, device=dev)
    conf = torch.empty(P, dtype=torch.float32, device=dev)
    if hit_below is None:
        hit_below = float(sigma2) ** 0.5
    ws = L.workspace(lib.pcs_ransac_plane_hypotheses_workspace_bytes(L.i64(n), L.i64(P), L.i32(R)), dev)
    L.check(lib.pcs_ransac_plane_hypotheses(
        L.ptr(xyz), L.ptr(seg_start.contiguous()), L.i64(n), L.ptr(min_z.float().contiguous()),
        L.ptr(max_z.float().contiguous()), L.host_floats(ratios), L.i32(R), L.ptr(origin.float().contiguous()),
        L.i64(P), L.f32(sigma2), L.f32(stopping_delta), L.i32(max_rounds), L.f32(hit_below),
        L.ptr(center), L.ptr(normal), L.ptr(conf), L.ptr(ws), L.i64(ws.numel()), L.stream()),
        "pcs_ransac_plane_hypotheses")
    return center, normal, conf


def prune_plane_hypotheses(xyz, normal, thresholds, k=8):
    """The 100-threshold pruning loop of compute_min_height_from_ransac in one launch: keep mask [H] bool.
    Raises PcsError(PCS_ERR_RANGE) for more than 2048 hypotheses."""
    L.require_gpu(xyz, "prune_plane_hypotheses")
    lib = L.load()
    xyz = xyz.float().contiguous()
    normal = normal.float().contiguous()
    H = xyz.shape[0]
    keep = torch.empty(H, dtype=torch.uint8, device=xyz.device)
    th = [float(t) for t in thresholds]
    L.check(lib.pcs_prune_plane_hypotheses(L.ptr(xyz), L.ptr(normal), L.i64(H), L.i32(k), L.host_floats(th),
                                           L.i32(len(th)), L.ptr(keep), L.stream()), "pcs_prune_plane_hypotheses")
    return keep.bool()


# ---------------------------------------------------------------- native tracker
class TrackParams(__import__("ctypes").Structure):
    """Mirror of PcsTrackParams (include/pcs.h)."""
    import ctypes as _c
    _fields_ = [
        ("num_levels", _c.c_int32), ("max_iter", _c.c_int32), ("min_move_frame", _c.c_int32),
        ("smooth_num_itr", _c.c_int32),
        ("radius", _c.c_double * 4), ("stopping_delta", _c.c_double * 4),
        ("angle_regularizer", _c.c_double),
        ("voxel_size", (_c.c_float * 3) * 4),
        ("nn_radius", _c.c_float), ("reg_error_coeff", _c.c_float), ("angle_threshold", _c.c_float),
        ("max_diameter", _c.c_float), ("smooth_weight0", _c.c_float), ("smooth_weight", _c.c_float),
        ("smooth_stopping", _c.c_float), ("groups", _c.c_int32),
    ]


def icp_set_schedule(wide_iters=-1, use_tiles=True):
    """pcs_icp_set_schedule: how the iterations of the fused ICP are run (process-wide). wide_iters < 0: every
    iteration as wide launches over all registrations in flight (default, fastest measured); >= 1: that many wide
    iterations, then one resident workgroup per registration -- with LDS-staged searches (use_tiles) or the
    hash-grid searches of the cache path. Every schedule computes the same iterations, bit for bit."""
    lib = L.load()
    L.check(lib.pcs_icp_set_schedule(L.i32(int(wide_iters)), L.i32(1 if use_tiles else 0)), "pcs_icp_set_schedule")


def icp_set_grid_mode(rebuild_always=False):
    """pcs_icp_set_grid_mode: the moving points' grid updated in place between iterations (default) or rebuilt at
    every iteration (cross-check; same results)."""
    lib = L.load()
    L.check(lib.pcs_icp_set_grid_mode(L.i32(1 if rebuild_always else 0)), "pcs_icp_set_grid_mode")


def track_configure(engines=-1, chunk_us=-1):
    """pcs_track_configure: tracker engines per device (before they are created) and the chunk budget of the
    resident ICP schedules; negative = unchanged."""
    lib = L.load()
    L.check(lib.pcs_track_configure(L.i32(int(engines)), L.i32(int(chunk_us))), "pcs_track_configure")


def track_engines_init(device=None):
    """pcs_track_engines_init: create the device's lock-step engines now (before other streams are created:
    the runtime deals streams to its hardware queues in creation order). Cheap once they exist."""
    lib = L.load()
    index = -1 if device is None else torch.device(device).index
    L.check(lib.pcs_track_engines_init(L.i32(-1 if index is None else index)), "pcs_track_engines_init")


def track_params(radius_list, voxel_size_list, stopping_delta, angle_regularizer, nn_radius,
                 reg_error_coeff, angle_threshold, min_move_frame, max_iter=80, max_diameter=12.5,
                 smooth=(1.0, 10.0, 300, 1e-3)):
    prm = TrackParams()
    prm.groups = 0  # (ignored by the library: the units of all calls share the device's lock-step engine)
    L_ = len(radius_list)
    if not (1 <= L_ <= 4 and len(voxel_size_list) == L_ and len(stopping_delta) == L_):
        raise L.PcsError("track_params: 1..4 registration levels with matching voxel sizes / deltas")
    prm.num_levels = L_
    prm.max_iter = int(max_iter)
    prm.min_move_frame = int(min_move_frame)
    prm.smooth_num_itr = int(smooth[2])
    for i in range(L_):
        prm.radius[i] = float(radius_list[i])
        prm.stopping_delta[i] = float(stopping_delta[i])
        for k in range(3):
            prm.voxel_size[i][k] = float(voxel_size_list[i][k])
    prm.angle_regularizer = float(angle_regularizer)
    prm.nn_radius = float(nn_radius)
    prm.reg_error_coeff = float(reg_error_coeff)
    prm.angle_threshold = float(angle_threshold)
    prm.max_diameter = float(max_diameter)
    prm.smooth_weight0, prm.smooth_weight, prm.smooth_stopping = float(smooth[0]), float(smooth[1]), float(smooth[3])
    return prm


class TrackUnit(__import__("ctypes").Structure):
    """Mirror of PcsTrackUnit (include/pcs.h)."""
    import ctypes as _c
    _fields_ = [
        ("seq_component", _c.c_void_p), ("seq_stationary", _c.c_void_p),
        ("anchor_frame", _c.c_int32), ("min_frame_id", _c.c_int32), ("max_frame_id", _c.c_int32),
        ("pad_", _c.c_int32),
        ("comp_base", _c.c_int64), ("C", _c.c_int64),
        ("ex_index", _c.c_void_p), ("ex_local", _c.c_void_p), ("ex_component", _c.c_void_p),
        ("ex_cap", _c.c_int64),
        ("transforms", _c.c_void_p),
        ("reg_errors", _c.c_void_p), ("comp_edge_ratios", _c.c_void_p), ("comp_velos", _c.c_void_p),
        ("comp_min_frame_id", _c.c_void_p), ("comp_max_frame_id", _c.c_void_p),
        ("moving", _c.c_void_p), ("valid_comp", _c.c_void_p),
        ("info", _c.c_int64 * 4),
    ]


MAX_TRACK_UNITS = 32  # units one sequence may hand over (kMaxCallUnits of the native tracker)


def _track_unit_array(dev, frame_offsets, units):
    """ctypes TrackUnit array + output tensors + the tensors that must outlive the call."""
    arr = (TrackUnit * len(units))()
    outs, keep = [], []
    for i, u in enumerate(units):
        a, lo, hi = int(u["anchor_frame"]), int(u["min_frame_id"]), int(u["max_frame_id"])
        C, F, W = int(u["num_components"]), hi + 1, hi - lo + 1
        comp = u["component"].long().contiguous()
        stat = u["stationary"].to(torch.uint8).contiguous()
        cap = int(frame_offsets[hi + 1] - frame_offsets[lo])
        out = dict(
            ex_index=torch.empty(cap, dtype=torch.int64, device=dev),
            ex_local=torch.empty(cap, dtype=torch.int64, device=dev),
            ex_component=torch.empty(cap, dtype=torch.int64, device=dev),
            transforms=torch.empty(C, W, 4, 4, dtype=torch.float64, device=dev),
            reg_errors=torch.empty(C, F, dtype=torch.float32, device=dev),
            comp_edge_ratios=torch.empty(C, F, dtype=torch.float32, device=dev),
            comp_velos=torch.empty(C, F, 3, dtype=torch.float32, device=dev),
            comp_min_frame_id=torch.empty(C, dtype=torch.int64, device=dev),
            comp_max_frame_id=torch.empty(C, dtype=torch.int64, device=dev),
            moving=torch.empty(C, dtype=torch.uint8, device=dev),
            valid_comp=torch.empty(C, dtype=torch.uint8, device=dev),
        )
        keep.append((comp, stat))
        t = arr[i]
        t.seq_component, t.seq_stationary = comp.data_ptr(), stat.data_ptr()
        t.anchor_frame, t.min_frame_id, t.max_frame_id = a, lo, hi
        t.comp_base, t.C, t.ex_cap = int(u["comp_base"]), C, cap
        for k in ("ex_index", "ex_local", "ex_component", "transforms", "reg_errors", "comp_edge_ratios",
                  "comp_velos", "comp_min_frame_id", "comp_max_frame_id", "moving", "valid_comp"):
            setattr(t, k, out[k].data_ptr())
        outs.append(out)
    return arr, outs, keep


def _track_results(arr, outs):
    res = []
    for i, out in enumerate(outs):
        info = arr[i].info
        n_ex = int(info[0])
        for k in ("ex_index", "ex_local", "ex_component"):
            out[k] = out[k][:n_ex]
        out["moving"] = out["moving"].bool()
        out["valid_comp"] = out["valid_comp"].bool()
        out["_ticks"] = int(info[3])
        res.append((out, (n_ex, int(info[1]), int(info[2]))))
    return res


def track_sequence(prm, seq_fxyz, frame_offsets, units):
    """All tracking units of a sequence in ONE native call (pcs_track_sequence): they join the
    device's lock-step engine (one driver thread and stream per device: every launch serves all units of
    all calls that are in the same phase) and the call returns when they are done.
    `units`: list of dicts with keys component (int64 [N]), stationary (bool/uint8 [N]),
    anchor_frame, min_frame_id, max_frame_id, comp_base, num_components. Returns one
    (out dict, (rows, registration calls, frame steps)) per unit, as ops.track_frame would."""
    import ctypes
    L.require_gpu(seq_fxyz, "track_sequence")
    lib = L.load()
    dev = seq_fxyz.device
    assert seq_fxyz.is_contiguous() and seq_fxyz.dtype == torch.float32 and seq_fxyz.shape[1] == 4
    if not (1 <= len(units) <= MAX_TRACK_UNITS):
        raise L.PcsError(f"track_sequence: 1..{MAX_TRACK_UNITS} units per call")
    nf = len(frame_offsets) - 1
    offs = (ctypes.c_int64 * (nf + 1))(*[int(v) for v in frame_offsets])
    arr, outs, keep = _track_unit_array(dev, frame_offsets, units)
    nbytes = lib.pcs_track_sequence_workspace_bytes(offs, L.i32(nf), arr, L.i32(len(units)))
    if nbytes == 0:
        raise L.PcsError("pcs_track_sequence_workspace_bytes: bad unit description")
    ws = L.workspace(nbytes, dev)
    L.check(lib.pcs_track_sequence(ctypes.byref(prm), L.ptr(seq_fxyz), offs, L.i32(nf), arr,
                                   L.i32(len(units)), L.ptr(ws), L.i64(ws.numel()), L.stream()),
            "pcs_track_sequence")
    del keep
    return _track_results(arr, outs)


def track_batch(prm, sequences):
    """The tracking units of SEVERAL sequences in one native call (pcs_track_batch), from one host
    thread. `sequences`: list of (seq_fxyz, frame_offsets, units) as for track_sequence. Returns one
    result list per sequence; each unit's results are those of track_sequence / track_frame, bit for bit."""
    import ctypes
    lib = L.load()
    n = len(sequences)
    if n < 1:
        return []
    dev = sequences[0][0].device
    p_fxyz = (ctypes.c_void_p * n)()
    p_offs = (ctypes.POINTER(ctypes.c_int64) * n)()
    p_units = (ctypes.POINTER(TrackUnit) * n)()
    nfs, nus = (ctypes.c_int * n)(), (ctypes.c_int * n)()
    held = []
    for q, (seq_fxyz, frame_offsets, units) in enumerate(sequences):
        L.require_gpu(seq_fxyz, "track_batch")
        assert seq_fxyz.is_contiguous() and seq_fxyz.dtype == torch.float32 and seq_fxyz.shape[1] == 4
        if not (1 <= len(units) <= MAX_TRACK_UNITS):
            raise L.PcsError(f"track_batch: 1..{MAX_TRACK_UNITS} units per sequence")
        nf = len(frame_offsets) - 1
        offs = (ctypes.c_int64 * (nf + 1))(*[int(v) for v in frame_offsets])
        arr, outs, keep = _track_unit_array(dev, frame_offsets, units)
        held.append((offs, arr, outs, keep))
        p_fxyz[q] = seq_fxyz.data_ptr()
        p_offs[q] = ctypes.cast(offs, ctypes.POINTER(ctypes.c_int64))
        p_units[q] = ctypes.cast(arr, ctypes.POINTER(TrackUnit))
        nfs[q], nus[q] = nf, len(units)
    nbytes = lib.pcs_track_batch_workspace_bytes(L.i32(n), p_offs, nfs, p_units, nus)
    if nbytes == 0:
        raise L.PcsError("pcs_track_batch_workspace_bytes: bad unit description")
    ws = L.workspace(nbytes, dev)
    L.check(lib.pcs_track_batch(ctypes.byref(prm), L.i32(n), p_fxyz, p_offs, nfs, p_units, nus, L.ptr(ws),
                                L.i64(ws.numel()), L.stream()), "pcs_track_batch")
    return [_track_results(arr, outs) for _, arr, outs, _ in held]


def track_frame(prm, seq_fxyz, seq_component, seq_stationary, frame_offsets, anchor_frame,
                min_frame_id, max_frame_id, comp_base, num_components):
    """ClusterTracking.track_frame in one native call (pcs_track_frame). `frame_offsets`: host list
    [num_frames + 1] of row ranges of the frame-grouped sequence. Returns a dict of tensors
    (ex_index / ex_local / ex_component already cut to the extracted count) and the info triple."""
    import ctypes
    L.require_gpu(seq_fxyz, "track_frame")
    lib = L.load()
    dev = seq_fxyz.device
    assert seq_fxyz.is_contiguous() and seq_fxyz.dtype == torch.float32 and seq_fxyz.shape[1] == 4
    comp = seq_component.long().contiguous()
    stat = seq_stationary.to(torch.uint8).contiguous()
    nf = len(frame_offsets) - 1
    offs = (ctypes.c_int64 * (nf + 1))(*[int(v) for v in frame_offsets])
    C = int(num_components)
    F = max_frame_id + 1
    W = max_frame_id - min_frame_id + 1
    sizes = [frame_offsets[f + 1] - frame_offsets[f] for f in range(min_frame_id, max_frame_id + 1)]
    na = frame_offsets[anchor_frame + 1] - frame_offsets[anchor_frame]
    nmax = max([1] + [frame_offsets[f + 1] - frame_offsets[f]
                      for f in range(min_frame_id, max_frame_id + 1) if f != anchor_frame])
    cap = int(sum(sizes))
    out = dict(
        ex_index=torch.empty(cap, dtype=torch.int64, device=dev),
        ex_local=torch.empty(cap, dtype=torch.int64, device=dev),
        ex_component=torch.empty(cap, dtype=torch.int64, device=dev),
        transforms=torch.empty(C, W, 4, 4, dtype=torch.float64, device=dev),
        reg_errors=torch.empty(C, F, dtype=torch.float32, device=dev),
        comp_edge_ratios=torch.empty(C, F, dtype=torch.float32, device=dev),
        comp_velos=torch.empty(C, F, 3, dtype=torch.float32, device=dev),
        comp_min_frame_id=torch.empty(C, dtype=torch.int64, device=dev),
        comp_max_frame_id=torch.empty(C, dtype=torch.int64, device=dev),
        moving=torch.empty(C, dtype=torch.uint8, device=dev),
        valid_comp=torch.empty(C, dtype=torch.uint8, device=dev),
    )
    info = (ctypes.c_int64 * 4)()
    ws = L.workspace(lib.pcs_track_frame_workspace_bytes(L.i64(na), L.i64(nmax), L.i64(C), L.i32(F)), dev)
    L.check(lib.pcs_track_frame(
        ctypes.byref(prm), L.ptr(seq_fxyz), L.ptr(comp), L.ptr(stat), offs, L.i32(nf),
        L.i32(anchor_frame), L.i32(min_frame_id), L.i32(max_frame_id), L.i64(comp_base), L.i64(C),
        L.ptr(out["ex_index"]), L.ptr(out["ex_local"]), L.ptr(out["ex_component"]), L.i64(cap),
        L.ptr(out["transforms"]), L.ptr(out["reg_errors"]), L.ptr(out["comp_edge_ratios"]),
        L.ptr(out["comp_velos"]), L.ptr(out["comp_min_frame_id"]), L.ptr(out["comp_max_frame_id"]),
        L.ptr(out["moving"]), L.ptr(out["valid_comp"]), info, L.ptr(ws), L.i64(ws.numel()),
        L.stream()), "pcs_track_frame")
    n_ex = int(info[0])
    for k in ("ex_index", "ex_local", "ex_component"):
        out[k] = out[k][:n_ex]
    out["moving"] = out["moving"].bool()
    out["valid_comp"] = out["valid_comp"].bool()
    out["_ticks"] = int(info[3])
    return out, (n_ex, int(info[1]), int(info[2]))
