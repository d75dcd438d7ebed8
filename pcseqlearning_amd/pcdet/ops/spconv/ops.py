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
    if ndim != 3:
        raise NotImplementedError("2-D and 3-D rulebooks only (the reference's 4-D instantiation, src/all.cc:28-29, "
                                  "is not on the hot path)")
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
    return _ops.get_indice_pairs(indices, batch_size, out_shape, list(spatial_shape), ksize, stride,
                                 padding, dilation, subm, transpose, with_table=with_table)


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
