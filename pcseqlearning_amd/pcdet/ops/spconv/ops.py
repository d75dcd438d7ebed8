"""pcdet/ops/spconv/ops.py: output-size helpers, get_indice_pairs, indice_conv on the HIP kernels."""
from .... import ops as _ops


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    out = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        out.append(1 if kernel_size[i] == -1 else size)
    return out


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    out = []
    for i in range(len(input_size)):
        if kernel_size[i] == -1:
            raise ValueError("deconv don't support kernel_size < 0")
        out.append((input_size[i] - 1) * stride[i] - 2 * padding[i] + kernel_size[i] + output_padding[i])
    return out


def _list(v, ndim):
    return list(v) if isinstance(v, (list, tuple)) else [v] * ndim


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                     out_padding=0, subm=False, transpose=False, grid=None, with_table=False):
    ndim = indices.shape[1] - 1
    if ndim == 2:
        # get_indice_pairs_2d (src/all.cc:22-25 of the reference): a plane is a volume one cell deep -- indices
        # (b, y, x) -> (b, 0, y, x), window (1, ky, kx). The kernel offsets' row-major numbers, the outputs' linear
        # indices and the order in which outputs first appear are the 2-D ones, so pairs / counts are the 2-D
        # rulebook's and the output ids are its ids with the zero column dropped.
        import torch
        wide = torch.cat([indices[:, :1], torch.zeros_like(indices[:, :1]), indices[:, 1:]], dim=1).contiguous()
        pad3 = lambda v, fill: [fill] + _list(v, 2)  # noqa: E731
        res = get_indice_pairs(wide, batch_size, [1] + list(spatial_shape), pad3(ksize, 1), pad3(stride, 1),
                               pad3(padding, 0), pad3(dilation, 1), pad3(out_padding, 0), subm, transpose, grid,
                               with_table)
        outids = res[0]
        outids = indices if subm else torch.cat([outids[:, :1], outids[:, 2:]], dim=1).contiguous()
        return (outids, *res[1:])
    if ndim not in (3, 4):
        raise NotImplementedError("2-D, 3-D and 4-D rulebooks only (the reference instantiates no others, src/all.cc:22-29)")
    ksize, stride, padding = _list(ksize, ndim), _list(stride, ndim), _list(padding, ndim)
    dilation, out_padding = _list(dilation, ndim), _list(out_padding, ndim)
    for d, s in zip(dilation, stride):
        assert any([s == 1, d == 1]), "don't support this."
    if subm:
        out_shape = list(spatial_shape)
    elif transpose:
        out_shape = get_deconv_output_size(spatial_shape, ksize, stride, padding, dilation, out_padding)
    else:
        out_shape = get_conv_output_size(spatial_shape, ksize, stride, padding, dilation)
    if ndim == 4:
        res = _indice_pairs_nd(indices, out_shape, ksize, stride, padding, dilation, subm, transpose)
        return (*res, None) if with_table else res
    return _ops.get_indice_pairs(indices, batch_size, out_shape, list(spatial_shape), ksize, stride,
                                 padding, dilation, subm, transpose, with_table=with_table)


def _indice_pairs_nd(indices, out_shape, ksize, stride, padding, dilation, subm, transpose):
    """get_indice_pairs_4d (src/all.cc:28-29 of the reference -> getIndicePairsSubM / Conv / DeConv<4>,
    include/spconv/geometry.h:145-297) for any number of dimensions, as tensor operations on the device: the reference
    instantiates it, nothing on the hot path calls it, so it has no kernel of its own -- the convolutions over its
    pairs are the HIP ones (any kernel volume). Semantics restated from geometry.h:25-143: for input j (ascending) and
    kernel position k (getValidOutPos walks the output positions downwards, i.e. the kernel positions in ascending
    row-major order; getValidOutPosTranspose in descending order) the output position is
    (pos + pad - k * dil) / stride where that division is exact (transposed: pos * stride - pad + k * dil), kept when
    it lies inside out_shape; output ids are dealt in the order outputs first appear in that walk; the pairs of an
    offset are in ascending j. Returns (outids [n_out, 1 + D] int32, pairs [KV, 2, V] int32 (-1 beyond the counts),
    num [KV] int32)."""
    import torch
    dev = indices.device
    D, V = len(ksize), indices.shape[0]
    kv = 1
    for k in ksize:
        kv *= k
    idx = indices.long()
    b, pos = idx[:, 0], idx[:, 1:]
    t = lambda v: torch.tensor(v, dtype=torch.long, device=dev)  # noqa: E731
    ks, st, pd, dl, osz = t(ksize), t(stride), t(padding), t(dilation), t(out_shape)
    grids = torch.stack(torch.meshgrid(*[torch.arange(k, device=dev) for k in ksize], indexing="ij"), -1).reshape(kv, D)
    if transpose:
        val = pos[:, None, :] * st - pd + grids[None] * dl
        valid = torch.ones(V, kv, dtype=torch.bool, device=dev)
    else:
        numer = pos[:, None, :] + pd - grids[None] * dl
        val = torch.div(numer, st, rounding_mode="floor")
        valid = (numer - val * st == 0).all(-1)
    valid = valid & ((val >= 0) & (val < osz)).all(-1)
    mult = torch.ones(D, dtype=torch.long, device=dev)
    for d in range(D - 2, -1, -1):
        mult[d] = mult[d + 1] * out_shape[d + 1]
    vol = int(mult[0]) * out_shape[0]
    key = (val * mult).sum(-1) + b[:, None] * vol  # [V, KV] linear output index (rowArrayIdx + volume * batch)
    pairs = torch.full((kv, 2, V), -1, dtype=torch.int32, device=dev)
    if subm:
        # gridsOut[index of input j] = j (a later duplicate overwrites an earlier one), outputs = inputs
        in_key = (pos * mult).sum(-1) + b * vol
        skey, order = torch.sort(in_key, stable=True)
        at = torch.searchsorted(skey, key.reshape(-1), right=True) - 1  # the LAST input with that key
        hit = (at >= 0) & (skey[at.clamp(min=0)] == key.reshape(-1))
        valid = valid & hit.reshape(V, kv)
        out_of = order[at.clamp(min=0)].reshape(V, kv)
        outids = indices
    else:
        # output ids by first appearance in the walk: j ascending, kernel positions ascending (transposed: descending)
        walk = torch.arange(kv, device=dev)
        if transpose:
            walk = kv - 1 - walk
        seq = torch.arange(V, device=dev)[:, None] * kv + walk[None]
        vk, vs = key[valid], seq[valid]
        uniq, inv = torch.unique(vk, return_inverse=True)
        first = torch.full((uniq.shape[0],), V * kv, dtype=torch.long, device=dev)
        first.scatter_reduce_(0, inv, vs, reduce="amin")
        rank = torch.empty_like(first)
        rank[torch.argsort(first)] = torch.arange(uniq.shape[0], device=dev)
        out_of = torch.zeros(V, kv, dtype=torch.long, device=dev)
        out_of[valid] = rank[inv]
        by_rank = torch.argsort(first)
        ukey = uniq[by_rank]
        coords = [torch.div(ukey, vol, rounding_mode="floor")]
        rem = ukey - coords[0] * vol
        for d in range(D):
            c = torch.div(rem, mult[d], rounding_mode="floor")
            coords.append(c)
            rem = rem - c * mult[d]
        outids = torch.stack(coords, 1).to(torch.int32)
    num = valid.sum(0).to(torch.int32)
    kk, jj = valid.t().nonzero(as_tuple=True)  # grouped by offset, j ascending inside
    start = torch.cumsum(num.long(), 0) - num.long()
    slot = torch.arange(kk.shape[0], device=dev) - start[kk]
    pairs[kk, 0, slot] = jj.to(torch.int32)
    pairs[kk, 1, slot] = out_of[jj, kk].to(torch.int32)
    return outids.contiguous(), pairs, num


def indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse=False,
                subm=False):
    """ops.py:105-120 of the reference: dispatch on the filters' dtype (fp32 / half)."""
    import torch
    if filters.dtype == torch.float16:
        return _ops.indice_conv_half(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse, subm)
    return _ops.indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out,
                            inverse, subm)


def fused_indice_conv(features, filters, bias, indice_pairs, indice_pair_num, num_activate_out, inverse, subm):
    """ops.py:123-135 of the reference (fused_indice_conv_fp32 / _half): conv + bias in one call."""
    import torch
    if filters.dtype == torch.float16:
        return _ops.indice_conv_half(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse, subm,
                                     bias=bias)
    table = _ops.build_neighbor_table(indice_pairs, indice_pair_num, num_activate_out, inverse)
    return _ops.indice_conv_table(features, filters, table, num_activate_out, subm and not inverse, bias=bias)


def indice_conv_backward(features, filters, out_bp, indice_pairs, indice_pair_num, inverse=False, subm=False):
    return _ops.indice_conv_backward(features, filters, out_bp, indice_pairs, indice_pair_num, inverse, subm)


def indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out):
    import torch
    if features.dtype == torch.float16:
        return _ops.indice_maxpool_half(features, indice_pairs, indice_pair_num, num_activate_out)
    return _ops.indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out)


def indice_maxpool_backward(features, out_features, out_bp, indice_pairs, indice_pair_num):
    return _ops.indice_maxpool_backward(features, out_features, out_bp, indice_pairs, indice_pair_num)
