/* pcs.h -- C ABI of libpcs_hip.so: the MI355X (gfx950) implementation of the PCSeqLearning
 * hot path (per-frame voxelization + sparse-conv backbone; radius-graph / connected-component
 * cluster proposal; per-component ICP registration and cluster tracking).
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer unless the parameter is documented "host".
 *  - Every function returns 0 (PCS_OK) or a negative PCS_ERR_* code. Nothing exits, throws or
 *    allocates device memory: scratch comes from the caller's workspace (`*_workspace_bytes`
 *    queries are host-only and cheap), outputs are caller-allocated at their documented capacity.
 *  - Data-dependent output counts are written to a device scalar (`d_num*`, int64); the caller
 *    decides when to read it back (the reference syncs at the same places with `.item()`).
 *  - `stream` is a hipStream_t; kernels are enqueued, never synchronised.
 *  - Arrays are row-major and contiguous. Index outputs are int64 where the reference returns
 *    int64 tensors and int32 where it returns int32 (rulebooks, voxel coords).
 *
 * Each entry cites the reference interface it replaces (paths relative to the reference root).
 */
#ifndef PCS_H_
#define PCS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* pcs_stream_t; /* == hipStream_t */

#define PCS_OK 0
#define PCS_ERR_ARG (-1)
#define PCS_ERR_WORKSPACE (-2)
#define PCS_ERR_HIP (-3)
#define PCS_ERR_RANGE (-4)

/* ------------------------------------------------------------------ voxel keys / grid sampling */

/* out8 = {min over rows (4), max over rows (4)} of a [n,4] f32 array.
 * Replaces points.min(0)/max(0) in pcdet/models/model_utils/grid_sampling.py:32-35 and
 * graph_utils.py:171-173. */
int pcs_minmax4_f32(const float* pts, int64_t n, float* out8, pcs_stream_t stream);

/* torch_cluster.grid_cluster(points, size, start, end) as called by
 * pcdet/models/model_utils/grid_sampling.py:37 with start[0]-=margin0, end[0]+=margin0 applied to
 * bounds8 (= pcs_minmax4_f32 output). size4 is a host array. ids[n] int64. */
int pcs_grid_cluster_ids(const float* pts, int64_t n, const float* bounds8, const float* size4,
                         float margin0, int64_t* ids, pcs_stream_t stream);

/* torch.unique(ids, sorted=True, return_inverse=True) (grid_sampling.py:38) plus the segment
 * structure the callers rebuild with scatter ops: order[n] = original indices sorted by
 * (id, index); seg_start[v+1] = segment boundaries in `order`; uniq[v]; inv[n]; *d_num = v.
 * uniq / inv / order may be NULL. ids must be non-negative; key_bits = significant bits (64 if
 * unknown). Capacities: uniq[n], seg_start[n+1]. */
size_t pcs_unique_i64_workspace_bytes(int64_t n);
int pcs_unique_i64(const int64_t* ids, int64_t n, int key_bits, int64_t* uniq, int64_t* inv,
                   int32_t* order, int32_t* seg_start, int64_t* d_num, void* ws, size_t ws_bytes,
                   pcs_stream_t stream);

/* torch_scatter.scatter(data, inv, reduce='mean'|'sum') (grid_sampling.py:42,
 * cluster_tracking.py:44) over the segments of pcs_unique_i64. mean bit 0: 1 = mean, 0 = sum.
 * mean bit 1 clear: summed serially in ascending original index (voxel-sized segments);
 * set (d <= 16): one wave per segment, lane-strided partial sums + butterfly (pillar- and
 * component-sized segments); mean bit 2 set (d <= 16): one 256-thread workgroup per segment (few,
 * very large segments: ground super-pillars). All orders are fixed: reproducible run to run.
 * data [n,d] f32 -> out [v,d]. */
int pcs_segment_reduce_f32(const float* data, int d, const int32_t* order, const int32_t* seg_start,
                           int64_t v, int mean, float* out, pcs_stream_t stream);

/* scatter(arange(n), inv, reduce='max') of pcdet/models/registration/simple_reg.py:122-124
 * (pick_last=1) or the min variant (pick_last=0). out[v] int64. */
int pcs_segment_pick(const int32_t* order, const int32_t* seg_start, int64_t v, int pick_last,
                     int64_t* out, pcs_stream_t stream);

/* DynamicMeanVFE.forward, pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:51-76.
 * pts [n,ncol] = (batch, x, y, z, feat...); range6/voxel3/grid3 host arrays (x,y,z order).
 * voxel_feat [n,ncol-1] capacity, voxel_coords [n,4] int32 (b,z,y,x), inv[n] (-1 = out of range,
 * may be NULL), *d_num = voxel count. Voxels come out in ascending b*XYZ + x*YZ + y*Z + z. */
size_t pcs_dynamic_voxelize_workspace_bytes(int64_t n);
int pcs_dynamic_voxelize(const float* pts, int64_t n, int ncol, const float* range6,
                         const float* voxel3, const int32_t* grid3, float* voxel_feat,
                         int32_t* voxel_coords, int64_t* inv, int64_t* d_num, void* ws,
                         size_t ws_bytes, pcs_stream_t stream);

/* points_to_voxel_3d_np, pcdet/ops/spconv/include/spconv/point2voxel.h:33-92 (the spconv-1.x
 * VoxelGeneratorV2 used by pcdet/datasets/processor/data_processor.py:23-68).
 * pts [n,ncol] = (x, y, z, feat...). voxels [max_voxels,max_points,ncol], coords [max_voxels,3]
 * (z,y,x), num [max_voxels]; rows past *d_num are zero. First-come voxel order. */
size_t pcs_hard_voxelize_workspace_bytes(int64_t n);
int pcs_hard_voxelize(const float* pts, int64_t n, int ncol, const float* range6,
                      const float* voxel3, int max_points, int max_voxels, float* voxels,
                      int32_t* coords, int32_t* num, int64_t* d_num, void* ws, size_t ws_bytes,
                      pcs_stream_t stream);

/* ------------------------------------------------------------------ sparse-conv rulebook + conv */

/* sparse_conv_ext.get_indice_pairs_3d, pcdet/ops/spconv/src/all.cc:21-23 ->
 * getIndicePair<3>, include/spconv/spconv_ops.h:27-141. indices [V,4] int32 (b,z,y,x).
 * pairs [KV,2,V] int32 (filled with -1 then written), num [KV]; strided conv also writes
 * out_ids [out_cap,4] with out_cap >= pcs_indice_pairs_out_capacity() and *d_num_out.
 * Pair order per offset = ascending input index; output voxel order = first encounter
 * (the reference CPU functor order, geometry.h:145-194). transpose != 0 (with subm == 0): the
 * rulebook of a transposed conv (getIndicePairsDeConv, geometry.h:197-245: out = in * stride - pad
 * + k * dilation, out_shape3 = get_deconv_output_size), same orders. The capacity / workspace
 * queries take `subm` = 0 strided, 1 submanifold, 2 transposed.
 * All *3 arrays are host int[3] in (z,y,x) order. KV <= 32.
 * The strided and transposed forms synchronise `stream` once inside the call: the number of output voxels
 * decides the size of the remaining work (and whether the first, small coordinate table was large enough;
 * if not the claims are redone on a table of two slots per possible output). The reference's caller reads
 * that count on the host straight after the call in any case (ops.py:97-99). */
int64_t pcs_indice_pairs_out_capacity(int64_t V, const int* ksize3, const int* stride3, int subm);
size_t pcs_get_indice_pairs_workspace_bytes(int64_t V, const int* ksize3, const int* stride3,
                                            int subm);
int pcs_get_indice_pairs(const int32_t* indices, int64_t V, int batch, const int* out_shape3,
                         const int* spatial_shape3, const int* ksize3, const int* stride3,
                         const int* pad3, const int* dil3, int subm, int transpose,
                         int32_t* out_ids, int64_t out_cap, int32_t* pairs, int32_t* num,
                         int64_t* d_num_out, void* ws, size_t ws_bytes, pcs_stream_t stream);
/* The same call, and -- where pcs_indice_pairs_gives_table() returns 1: submanifold, odd kernel sizes, no
 * dilation -- row j of `table` [V,KV] int32 = the inputs of output j by kernel offset, -1 where none:
 * what pcs_build_neighbor_table makes of the pairs, taken from the rulebook's own probing pass. `table`
 * may be NULL. `h_num_out` (HOST pointer, may be NULL) receives the number of output voxels where the call
 * knows it on return (submanifold: V; strided / transposed: the call has synchronised the stream for it),
 * else -1 = read *d_num_out. With both NULL this IS pcs_get_indice_pairs. */
int pcs_indice_pairs_gives_table(const int* ksize3, const int* dil3, int subm);
int pcs_get_indice_pairs_table(const int32_t* indices, int64_t V, int batch, const int* out_shape3,
                               const int* spatial_shape3, const int* ksize3, const int* stride3,
                               const int* pad3, const int* dil3, int subm, int transpose,
                               int32_t* out_ids, int64_t out_cap, int32_t* pairs, int32_t* num,
                               int64_t* d_num_out, int64_t* h_num_out, int32_t* table, void* ws,
                               size_t ws_bytes, pcs_stream_t stream);

/* sparse_conv_ext.indice_conv_fp32, src/all.cc:30-33 -> indiceConv<float>,
 * spconv_ops.h:260-361. features [num_in,cin], filters [kv,cin,cout], out [num_out,cout].
 * ws >= pcs_neighbor_table_bytes(num_out, kv). */
size_t pcs_neighbor_table_bytes(int64_t num_out, int kv);
int pcs_indice_conv_f32(const float* features, int cin, const float* filters, int kv, int cout,
                        const int32_t* pairs, int64_t pair_stride, const int32_t* num,
                        int64_t num_out, int inverse, int subm, float* out, void* ws,
                        size_t ws_bytes, pcs_stream_t stream);

/* The two halves of the above, so a rulebook cached under an indice_key
 * (pcdet/ops/spconv/conv.py:146-171) keeps its output-major table too, and bias / eval-mode
 * BatchNorm / ReLU (fused_spconv_ops.h:29-132, modules.py:125-187) fold into the epilogue. */
int pcs_build_neighbor_table(const int32_t* pairs, int64_t pair_stride, const int32_t* num, int kv,
                             int inverse, int64_t num_out, int32_t* table, pcs_stream_t stream);
int pcs_conv_use_mfma(int enable); /* 1 (default): matrix cores for Cin >= 16 (16 / 32 / 64 input channels and 100 000 rows or
                                    * more: partner rows gathered straight into the operand registers); 2: matrix cores on
                                    * round 4's staged 64-row tiles everywhere; 3: the direct-gather form at any size (tests);
                                    * 0: FMA tiles only */
/* scale / shift [C] f32 of an eval-mode BatchNorm1d for the conv epilogue above (torch.nn.BatchNorm1d in
 * eval mode after every conv of spconv_backbone.py:6-27) from inv_std = rsqrt(running_var + eps):
 * scale = inv_std * weight, shift = -mean * scale + bias, each product / sum rounded to fp32 like the
 * tensor ops they replace; weight / bias may be NULL. */
int pcs_bn_fold_f32(const float* weight, const float* bias, const float* running_mean,
                    const float* inv_std, int C, float* scale, float* shift, pcs_stream_t stream);
int pcs_indice_conv_table_f32(const float* features, int cin, const float* filters, int kv,
                              int cout, int centre, const int32_t* table, int64_t num_out,
                              const float* bias, const float* scale, const float* shift, int relu,
                              float* out, pcs_stream_t stream);
/* The same with a residual row added before the ReLU: out = relu(((conv + bias) * scale + shift) + residual[o,:]) --
 * the tail of SparseBasicBlock.forward (pcdet/models/blocks/spconv_blocks.py:47-64: bn2(conv2(.)) + identity, relu) in
 * the second conv's epilogue instead of two more passes over the block's output; the same operations in the same order
 * as the separate add and ReLU, hence the same bits. residual [num_out, cout] f32, must not be `out`; NULL = the
 * entry above. */
int pcs_indice_conv_table_residual_f32(const float* features, int cin, const float* filters, int kv,
                                       int cout, int centre, const int32_t* table, int64_t num_out,
                                       const float* bias, const float* scale, const float* shift,
                                       const float* residual, int relu, float* out, pcs_stream_t stream);

/* sparse_conv_ext.indice_conv_backward_fp32 (src/all.cc:34-37; indiceConvBackward,
 * include/spconv/spconv_ops.h:363-449): features [num_in,cin], filters [kv,cin,cout],
 * out_grad [num_out,cout], the forward call's rulebook (pairs [kv,2,pair_stride], num [kv]) and its
 * inverse / subm flags -> in_grad [num_in,cin], filters_grad [kv,cin,cout]. Deterministic (no atomics:
 * the input gradient is a conv over the transposed rulebook, the filter gradient a two-stage sum). */
size_t pcs_indice_conv_backward_workspace_bytes(int64_t num_in, int64_t pair_stride, int kv, int cin,
                                                int cout);
int pcs_indice_conv_backward_f32(const float* features, int cin, const float* filters, int kv, int cout,
                                 const float* out_grad, const int32_t* pairs, int64_t pair_stride,
                                 const int32_t* num, int64_t num_in, int64_t num_out, int inverse,
                                 int subm, float* in_grad, float* filters_grad, void* ws,
                                 size_t ws_bytes, pcs_stream_t stream);

/* sparse_conv_ext.indice_maxpool_fp32 / indice_maxpool_backward_fp32 (src/all.cc:42-47; pool_ops.h:26-98,
 * src/maxpool.cc:22-64): features [num_in,C] and the pooling window's rulebook ->
 * out [num_out,C] = max(0, max over paired inputs) (the reference's output starts at zero); backward:
 * in_grad [num_in,C] = sum over paired outputs whose pooled value equals the input (ties all receive).
 * ws: pcs_neighbor_table_bytes(num_out | num_in, kv). */
int pcs_indice_maxpool_f32(const float* features, int C, const int32_t* pairs, int64_t pair_stride,
                           const int32_t* num, int kv, int64_t num_out, float* out, void* ws,
                           size_t ws_bytes, pcs_stream_t stream);
int pcs_indice_maxpool_backward_f32(const float* features, const float* out_features,
                                    const float* out_grad, int C, const int32_t* pairs,
                                    int64_t pair_stride, const int32_t* num, int kv, int64_t num_in,
                                    float* in_grad, void* ws, size_t ws_bytes, pcs_stream_t stream);

/* sparse_conv_ext.indice_conv_half / fused_indice_conv_half / indice_conv_backward_half / indice_maxpool_half /
 * indice_maxpool_backward_half (src/all.cc:35-50: the reference's templates instantiated for at::Half).
 * features / filters / bias / gradients / outputs hold IEEE fp16 bit patterns. The fp16 rows are gathered as
 * they are (half the bytes of the fp32 path), the products run on the fp16 matrix pipe
 * (v_mfma_f32_16x16x32_f16; Cin % 16 == 0 and Cout in {16,32,64,128}, other shapes one thread per output
 * element) with fp32 accumulation over ALL offsets, and the result is rounded to fp16 once -- where the
 * reference rounds each offset's GEMM to fp16 and adds in fp16. The pooling is exact; its backward adds the
 * selected gradients in fp16 in ascending offset, as the reference's functor does on at::Half.
 * bias: [cout] or NULL (NULL = indice_conv_half). ws: *_workspace_bytes. */
size_t pcs_indice_conv_f16_workspace_bytes(int64_t num_in, int cin, int kv, int cout, int64_t num_out);
int pcs_indice_conv_f16(const void* features, int64_t num_in, int cin, const void* filters, int kv, int cout,
                        const void* bias, const int32_t* pairs, int64_t pair_stride, const int32_t* num,
                        int64_t num_out, int inverse, int subm, void* out, void* ws, size_t ws_bytes,
                        pcs_stream_t stream);
/* the same on a cached output-major neighbour table (pcs_build_neighbor_table), with an optional ReLU: the form the
 * module path uses (an eval-mode BatchNorm folds into filters and bias on the host side) */
int pcs_indice_conv_table_f16(const void* features, int cin, const void* filters, int kv, int cout, int centre,
                              const int32_t* table, int64_t num_out, const void* bias, int relu, void* out,
                              pcs_stream_t stream);
/* the conv on a cached table with its output rows taken in a given order: order [num_out] int32 from
 * pcs_neighbor_table_order -- the rows sorted (stably) by the set of offsets they have a partner at, so that the sixteen
 * rows of a matrix-core fragment want the same offsets (in table order about half of the (fragment, offset) products
 * of VoxelBackBone8x's layers multiply padding). Every output row is the same sum in the same order: the same bits
 * as pcs_indice_conv_table_f32 / _f16. order == NULL: table order. kv <= 32. */
size_t pcs_neighbor_table_order_workspace_bytes(int64_t num_out);
int pcs_neighbor_table_order(const int32_t* table, int64_t num_out, int kv, int32_t* order, void* ws, size_t ws_bytes,
                             pcs_stream_t stream);
int pcs_indice_conv_table_ordered_f32(const float* features, int cin, const float* filters, int kv, int cout, int centre,
                                      const int32_t* table, int64_t num_out, const float* bias, const float* scale,
                                      const float* shift, int relu, float* out, const int32_t* order,
                                      pcs_stream_t stream);
int pcs_indice_conv_table_ordered_f16(const void* features, int cin, const void* filters, int kv, int cout, int centre,
                                      const int32_t* table, int64_t num_out, const void* bias, int relu, void* out,
                                      const int32_t* order, pcs_stream_t stream);
/* test hook, forms of the fp16 convs: 0 (default) = partner rows gathered straight into the MFMA fragments, several
 * offsets' weights in LDS per round (256-thread workgroups; up to 8 input channels: one thread per output row);
 * 3 = the same on 512-thread workgroups; 2 = 128-row tiles staged through LDS, the next offset prefetched;
 * 1 = round 4's kernels (64-row tiles, one thread per output element). The same bits in every form. */
int pcs_indice_conv_f16_short_tiles(int on);
size_t pcs_indice_conv_backward_f16_workspace_bytes(int64_t num_in, int64_t pair_stride, int kv, int cin, int cout);
int pcs_indice_conv_backward_f16(const void* features, int cin, const void* filters, int kv, int cout,
                                 const void* out_grad, const int32_t* pairs, int64_t pair_stride,
                                 const int32_t* num, int64_t num_in, int64_t num_out, int inverse, int subm,
                                 void* in_grad, void* filters_grad, void* ws, size_t ws_bytes, pcs_stream_t stream);
size_t pcs_indice_maxpool_f16_workspace_bytes(int64_t num_in, int C, int kv, int64_t num_out);
int pcs_indice_maxpool_f16(const void* features, int64_t num_in, int C, const int32_t* pairs, int64_t pair_stride,
                           const int32_t* num, int kv, int64_t num_out, void* out, void* ws, size_t ws_bytes,
                           pcs_stream_t stream);
/* ws: pcs_neighbor_table_bytes(num_in, kv) */
/* pcs_indice_conv_table_f16 with a residual row: fp16(relu(fp16(conv + bias) + residual[o,:])), the sum of the two fp16
 * rows formed as the tensor library forms it (fp32, one rounding) -- SparseBasicBlock's `+ identity` and ReLU for fp16
 * features, the bits of the separate tensor ops. residual [num_out, cout] fp16, must not be `out`; NULL = no residual. */
int pcs_indice_conv_table_residual_f16(const void* features, int cin, const void* filters, int kv, int cout, int centre,
                                       const int32_t* table, int64_t num_out, const void* bias, const void* residual,
                                       int relu, void* out, pcs_stream_t stream);
int pcs_indice_maxpool_backward_f16(const void* features, const void* out_features, const void* out_grad, int C,
                                    const int32_t* pairs, int64_t pair_stride, const int32_t* num, int kv,
                                    int64_t num_in, void* in_grad, void* ws, size_t ws_bytes, pcs_stream_t stream);

/* SparseConvTensor.dense() (pcdet/ops/spconv/structure.py:5-18,55-64) in its channels-first form, as
 * HeightCompression consumes it (pcdet/models/backbones_2d/map_to_bev/height_compression.py:10-27;
 * SURVEY 8f row f3): features [V,C] at indices [V,4] (b,z,y,x) -> out [B,C,D,H,W] f32, zeros
 * elsewhere ([B, C*D, H, W] when viewed). status: optional device int, set to 1 if an index lies
 * outside the shape (that row is skipped). */
int pcs_sparse_to_dense(const float* features, const int32_t* indices, int64_t V, int C, int B, int D,
                        int H, int W, float* out, int* status, pcs_stream_t stream);

/* ------------------------------------------------------------------ radius graph */

/* RadiusGraph.build_graph, pcdet/models/model_utils/graph_utils.py:149-209 (hash_insert_gpu +
 * radius_graph_gpu, pcdet/ops/torch_hash/src/torch_hash_kernel.cu:411-561) as ONE engine call
 * pair: ref [N,4], query [M,4] f32 (frame, x, y, z); `radius` is the reference's fp32 radius
 * (already including the frame offset term, registration_utils.py:111-112); frame cells follow
 * the reference quantisation and the query looks at frame cells fq + dframe_min..dframe_max
 * (qmin[0], qmax[0]); spatial qmin/qmax are the reference defaults (-1..1: the whole ball).
 * Edges: grouped by ascending query, within a query ascending (dist2, ref index); at most
 * max_nbr (1..32) per query. *d_num_edges = E, or -1 when extent/cell_size overflows the 17-bit
 * cell coordinates. cell_size <= 0 -> radius/2. search and emit must see the same ws. */
size_t pcs_radius_graph_workspace_bytes(int64_t N, int64_t M, int max_nbr);
int pcs_radius_graph_search(const float* ref, int64_t N, const float* query, int64_t M,
                            float radius, float cell_size, int dframe_min, int dframe_max,
                            int max_nbr, int sort_by_dist, int64_t* d_num_edges, void* ws,
                            size_t ws_bytes, pcs_stream_t stream);
int pcs_radius_graph_emit(int64_t N, int64_t M, int max_nbr, int64_t* e_ref, int64_t* e_query,
                          float* dists, const void* ws, size_t ws_bytes, pcs_stream_t stream);
/* ClusterProposal.propose_cluster, pcdet/models/registration/preprocessors/cluster_proposal.py:57-83: per
 * radius key a 32-nearest radius graph of the chunk on itself (graph_utils.py:149-209) and its weakly connected
 * components (graph_utils.py:40-53). The k nearest within r, nearest first, are the k nearest within R >= r cut at
 * r, so ONE search at the largest radius serves every key: after pcs_radius_graph_search(ref == query, N == M,
 * sort_by_dist = 1) this entry labels the components of the graph still held in `ws`, cut at each of the n_cut
 * (<= 8) radii in radius_cut (host array; an edge counts when dist2 <= fl32(r) * fl32(r), the search's own test),
 * with no edge list in between: component [n_cut, M] int64 = rank of the node's component by its smallest node
 * id, d_num [n_cut] = number of components (device). */
size_t pcs_radius_graph_components_workspace_bytes(int64_t M, int n_cut);
int pcs_radius_graph_components(int64_t N, int64_t M, int max_nbr, const float* radius_cut, int n_cut,
                                int64_t* component, int64_t* d_num, const void* ws, size_t ws_bytes,
                                void* ws_cc, size_t ws_cc_bytes, pcs_stream_t stream);
/* test hooks of the search engines behind pcs_radius_graph_search (both exact, same edges): 1 = every search on
 * the one-thread-per-query engine (the cell-sorted wave search of max_nbr > 1 switched off); 1 = every query of
 * the cell-sorted search through its one-thread fallback. */
int pcs_radius_graph_engine_only(int on);
/* diagnostics: host_out[8] = the counters the last cell-sorted search (max_nbr > 1) left in `ws`: query runs, work
 * cursor, queries that took the one-thread fallback, its cursor, four build-dependent statistics. Synchronises. */
int pcs_radius_graph_counters(int64_t N, int64_t M, int max_nbr, const void* ws, size_t ws_bytes,
                              int32_t* host_out, pcs_stream_t stream);
int pcs_knn_force_hard(int on);

/* Literal drop-ins for torch_hash_cuda (pcdet/ops/torch_hash/src/torch_hash_api.cpp:9-15,
 * torch_hash.h:16-32) on the reference's own table layout: keys[H] int64 (-1 empty),
 * values[H,D] f32, reverse[H] int64, dims[D] int64; all caller-allocated, H chosen by the caller
 * (graph_utils.py:179-183). hash_insert_gpu -> pcs_hash_insert; radius_graph_gpu ->
 * pcs_radius_graph_table_count (degree kernel + cumsum) then pcs_radius_graph_table_fill.
 * Unlike the reference kernels, query_keys is not mutated. max_nbr = -1 means unbounded. */
int pcs_hash_insert(int64_t* keys, float* values, int64_t* reverse, int64_t H, const int64_t* dims,
                    int D, const int64_t* insert_keys, const float* insert_values, int64_t n,
                    pcs_stream_t stream);
size_t pcs_radius_graph_table_workspace_bytes(int64_t M);
int pcs_radius_graph_table_count(const int64_t* keys, const float* values, const int64_t* reverse,
                                 int64_t H, const int64_t* dims, int D, const int64_t* query_keys,
                                 const float* query_values, int64_t M, const int32_t* qmin,
                                 const int32_t* qmax, const float* radius, int max_nbr,
                                 int32_t* degree, int32_t* offset, int64_t* d_num_edges, void* ws,
                                 size_t ws_bytes, pcs_stream_t stream);
int pcs_radius_graph_table_fill(const int64_t* keys, const float* values, const int64_t* reverse,
                                int64_t H, const int64_t* dims, int D, const int64_t* query_keys,
                                const float* query_values, int64_t M, const int32_t* qmin,
                                const int32_t* qmax, const float* radius, int sort_by_dist,
                                const int32_t* degree, const int32_t* offset, int64_t* edges,
                                float* dists, pcs_stream_t stream);

/* correspondence (pcdet/ops/torch_hash/src/torch_hash.h:20-22, torch_hash_kernel.cu:96-155,444-485)
 * on the reference's table layout, its argument order: corres_indices [M] int64 = index into the
 * inserted array of the nearest hashed point among the walked cells (strict `<`, no radius; ties by
 * the lower index), -1 if none. */
int pcs_correspondence(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                       const int64_t* dims, int D, const int64_t* query_keys,
                       const float* query_values, int64_t M, const int32_t* qmin,
                       const int32_t* qmax, int64_t* corres_indices, pcs_stream_t stream);
/* points_in_radius_gpu (torch_hash.h:29-32, torch_hash_kernel.cu:160-222,563-605): visited [N] int64,
 * zeroed by the caller, gets 1 for every hashed point strictly closer than `radius` to a query. */
int pcs_points_in_radius(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                         const int64_t* dims, int D, const int64_t* query_keys,
                         const float* query_values, int64_t M, const int32_t* qmin,
                         const int32_t* qmax, float radius, int64_t* visited, pcs_stream_t stream);

/* ------------------------------------------------------------------ components, medians */

/* connected_components, pcdet/models/model_utils/graph_utils.py:40-53 (scipy csgraph on the
 * host in the reference): component[n] int64 = rank of the node's weakly connected component by
 * its smallest node id; *d_num = number of components. */
size_t pcs_connected_components_workspace_bytes(int64_t n);
int pcs_connected_components(const int64_t* e0, const int64_t* e1, int64_t E, int64_t n,
                             int64_t* component, int64_t* d_num, void* ws, size_t ws_bytes,
                             pcs_stream_t stream);

/* robust_median, pcdet/models/registration/preprocessors/registration_utils.py:60-81: per
 * segment the element at sorted position deg//2. data[n] int64 -> out[v]. */
int pcs_segment_median_i64(const int64_t* data, const int32_t* order, const int32_t* seg_start,
                           int64_t v, int64_t* out, pcs_stream_t stream);

/* torch_scatter.scatter(reduce='max'|'min') over segments (cluster_tracking.py:104,325-328). */
int pcs_segment_minmax_f32(const float* data, int d, const int32_t* order, const int32_t* seg_start,
                           int64_t v, int is_max, float* out, pcs_stream_t stream);

/* ------------------------------------------------------------------ registration */

/* register_to_next_frame, pcdet/models/registration/preprocessors/registration_utils.py:83-206
 * (the "TLS solve"): per-component regularised Kabsch ICP with the loss-delta / 3-strike stop
 * rule evaluated on the device. BLOCKING: synchronises `stream` once per `chunk` iterations and
 * at the end (the reference synchronises every iteration). mov_fxyz [Mn,4] (non-stationary
 * moving points) is updated in place; mov_comp [Mn] int32 in [0,C); ref_fxyz [Rn,4]
 * non-stationary reference points; ref_all [Ra,4] every reference point (coverage ratio);
 * comp_deg [C] int64 moving points per component (stationary included). Outputs: T [C,16] f64
 * row-major 4x4, l1_err [C] f64 (truncated mean of the last iteration), edge_ratio [C] f32.
 * info: HOST int32[4] = {iterations, stopped by countdown, 0, 0}. `radius` is the graph's base
 * radius; the frame-offset term is added inside as the reference does.
 * RANGE: the matched pairs' moments are summed as 64-bit fixed-point integers about each component's first moving
 * point, which bounds a component's extent: a moving point or its partner further than 64 m from that origin (4096 m^2;
 * less for registrations of more than 2^14 queries: 2^26 m^2 / (Mn + Rn)) fails the call with PCS_ERR_RANGE instead
 * of returning wrong moments. The reference sums in float64 and has no such bound; the tracker's components are at
 * most 12.5 m across (cluster_tracking.py:123-148). */
size_t pcs_register_to_next_frame_workspace_bytes(int64_t Mn, int64_t Rn, int64_t Ra, int64_t C);
int pcs_register_to_next_frame(float* mov_fxyz, const int32_t* mov_comp, int64_t Mn,
                               const float* ref_fxyz, int64_t Rn, const float* ref_all, int64_t Ra,
                               const int64_t* comp_deg, int64_t C, double radius, int frame_offset,
                               double angle_regularizer, int max_iter, double stopping_delta,
                               float cell_size, int chunk, double* T, double* l1_err,
                               float* edge_ratio, int32_t* info, void* ws, size_t ws_bytes,
                               pcs_stream_t stream);

/* smooth_velo, pcdet/models/registration/preprocessors/cluster_tracking.py:162-199 (<=300 AdamW
 * steps with MultiStepLR and a 3-strike stop, one persistent kernel). comp_velos [C,F,3] f32 is
 * updated in place (the reference's Parameter aliases its input the same way);
 * comp_center_diffs [C,F,3]; frame_lo <= frame_hi ordered ends of the range. */
size_t pcs_smooth_velo_workspace_bytes(int64_t C, int F);
int pcs_smooth_velo(float* comp_velos, const float* comp_center_diffs, int64_t C, int F,
                    int frame_lo, int frame_hi, float weight0, float weight, int num_itr,
                    float stopping, void* ws, size_t ws_bytes, pcs_stream_t stream);
/* The same for n problems in one call (what the sequence tracker does at every frame step, one problem
 * per tracking unit; host arrays of n device pointers / sizes; ws[e] sized by
 * pcs_smooth_velo_workspace_bytes(C[e], F[e])). Results are those of n separate calls. The call
 * returns after the stream has drained. (A many-component problem whose multi-workgroup form could
 * not get its workgroups resident within ~0.2 s is run again in the single-workgroup form; stderr
 * says so.) */
int pcs_smooth_velo_batch(int n, float* const* comp_velos, const float* const* comp_center_diffs,
                          const int64_t* C, const int* F, const int* frame_lo, const int* frame_hi,
                          float weight0, float weight, int num_itr, float stopping, void* const* ws,
                          const size_t* ws_bytes, pcs_stream_t stream);

/* ClusterTracking.track_frame, pcdet/models/registration/preprocessors/cluster_tracking.py:430-787
 * (with sample_frame :39-51, component_center / component_diameter / filter_components :89-160,
 * dist_compensate :80-87, smooth_velo :162-199 and the harvest through NN_GRAPH :702-741): tracks
 * the components of one anchor frame backward and forward through the sequence. The whole loop
 * runs inside this call on `stream` (BLOCKING: it reads a few words per chunk of ICP iterations).
 * pcs_track_sequence below is the form for all anchors / component keys of a sequence at once.
 *   seq_fxyz [N,4] f32, seq_component [N] int64, seq_stationary [N] uint8: the sub-sampled
 *     sequence, rows grouped by ascending frame; frame_offsets: HOST [num_frames+1] row ranges.
 *   anchor_frame, [min_frame_id, max_frame_id]: the anchor and the window to track through.
 *   comp_base: subtracted from the anchor rows' component ids (the caller's `component.min()`);
 *   C: number of (re-based) components.
 *   outputs: ex_index / ex_local / ex_component [ex_cap] int64 = extracted rows as (row in the
 *     sequence, row within its frame, component), anchor rows first, then per visited frame in
 *     visiting order; transforms [C, W, 4, 4] f64 with W = max_frame_id - min_frame_id + 1;
 *     reg_errors, comp_edge_ratios [C, F] f32 and comp_velos [C, F, 3] f32 with F = max_frame_id + 1;
 *     comp_min_frame_id / comp_max_frame_id [C] int64; moving, valid_comp [C] uint8 (valid_comp =
 *     filter_components of the anchor). info: HOST int64[4] = {rows extracted, registration calls,
 *     frame steps, host ticks}. Returns PCS_ERR_RANGE when a search grid overflowed (the caller may
 *     fall back to the op-by-op host path, and must say so). */
#define PCS_TRACK_MAX_LEVELS 4
typedef struct PcsTrackParams {
  int32_t num_levels;      /* REGISTRATION.GRAPH.RADIUS / VOXEL_SIZE / STOPPING_DELTA entries */
  int32_t max_iter;        /* register_to_next_frame max_iter (80 in track_frame) */
  int32_t min_move_frame;  /* TRACKING_PARAMS.MIN_MOVE_FRAME */
  int32_t smooth_num_itr;  /* smooth_velo num_itr (300) */
  double radius[PCS_TRACK_MAX_LEVELS];
  double stopping_delta[PCS_TRACK_MAX_LEVELS];
  double angle_regularizer;
  float voxel_size[PCS_TRACK_MAX_LEVELS][3];
  float nn_radius;         /* NN_GRAPH.RADIUS */
  float reg_error_coeff;   /* TRACKING_PARAMS.REGISTRATION_ERROR_COEFFICIENT */
  float angle_threshold;   /* TRACKING_PARAMS.ANGLE_THRESHOLD (degrees) */
  float max_diameter;      /* filter_components max_diameter (12.5) */
  float smooth_weight0, smooth_weight, smooth_stopping; /* smooth_velo (1, 10, 1e-3) */
  int32_t groups;          /* ignored (round 2: lock-step groups per call); units of all calls now share the device's lock-step engines */
} PcsTrackParams;
size_t pcs_track_frame_workspace_bytes(int64_t n_anchor, int64_t n_frame_max, int64_t C,
                                       int num_frames);
int pcs_track_frame(const PcsTrackParams* prm, const float* seq_fxyz, const int64_t* seq_component,
                    const uint8_t* seq_stationary, const int64_t* frame_offsets, int num_frames,
                    int anchor_frame, int min_frame_id, int max_frame_id, int64_t comp_base,
                    int64_t C, int64_t* ex_index, int64_t* ex_local, int64_t* ex_component,
                    int64_t ex_cap, double* transforms, float* reg_errors, float* comp_edge_ratios,
                    float* comp_velos, int64_t* comp_min_frame_id, int64_t* comp_max_frame_id,
                    uint8_t* moving, uint8_t* valid_comp, int64_t* info, void* ws, size_t ws_bytes,
                    pcs_stream_t stream);

/* The tracking loop of ClusterTracking.forward (cluster_tracking.py:853-884: for every component
 * key, for every anchor frame, track_frame) for ALL units of a sequence in one call. The units are
 * independent, and each is a chain of small latency-bound launches, so they advance in lock-step:
 * every launch serves all units that are in the same phase, the ICP iterations of all units share
 * their four launches. The lock-step loop belongs to the DEVICE, not to the call: the device's engines
 * (three by default, PCS_TRACK_ENGINES; one driver thread and one stream each) advance every unit that
 * any call (any host thread, any sequence of a batch: SimpleReg.forward loops over independent sequences,
 * simple_reg.py:105-147) has handed over; units join and leave at tick boundaries. The call prepares its units on `stream`, hands them
 * over and BLOCKS until they are done. A unit's results are those of pcs_track_frame on the same
 * inputs, bit for bit, whatever other units share its launches. Calls whose PcsTrackParams differ
 * from those of the units in flight wait until the engine is empty.
 *   seq_fxyz / frame_offsets / num_frames: as for pcs_track_frame (shared by all units);
 *   units[n_units] (HOST array, n_units <= 32): per unit its key's component ids and stationary
 *     flags, the anchor and window, comp_base / C, and the output pointers of pcs_track_frame;
 *     info is written by the call.
 * pcs_track_batch: the same for the sequences of a batch from ONE host thread (sequence q's units are
 * handed over as soon as they are prepared; at most 128 units are in flight on a device, further
 * ones wait for free table slots); arrays of n_seqs HOST pointers / counts; `ws` is one workspace of
 * pcs_track_batch_workspace_bytes for all of them. */
/* Creates the device's lock-step engines (driver threads + streams) now instead of at the first tracking
 * call; device < 0 = the current device. Speed only: the HIP runtime deals streams to its hardware queues in
 * creation order, and engines created between the streams of a dozen host threads can end up on one queue
 * (their chains then run one after the other). Call it before creating other streams. */
int pcs_track_engines_init(int device);
/* Engines per device (1..8, default 3; for devices whose engines do not exist yet) and the time budget, in
 * microseconds, of a chunk of the unit-resident ICP schedules (pcs_icp_set_schedule; default 3000, 0: none).
 * Arguments < 0 keep the current setting. Results never depend on either. */
int pcs_track_configure(int engines, int chunk_us);
typedef struct PcsTrackUnit {
  const int64_t* seq_component;   /* [N] int64 */
  const uint8_t* seq_stationary;  /* [N] uint8 */
  int32_t anchor_frame, min_frame_id, max_frame_id, pad_;
  int64_t comp_base, C;
  int64_t *ex_index, *ex_local, *ex_component;
  int64_t ex_cap;
  double* transforms;
  float *reg_errors, *comp_edge_ratios, *comp_velos;
  int64_t *comp_min_frame_id, *comp_max_frame_id;
  uint8_t *moving, *valid_comp;
  int64_t info[4];                /* out: rows extracted, registration calls, frame steps, host ticks */
} PcsTrackUnit;
size_t pcs_track_sequence_workspace_bytes(const int64_t* frame_offsets, int num_frames,
                                          const PcsTrackUnit* units, int n_units);
int pcs_track_sequence(const PcsTrackParams* prm, const float* seq_fxyz, const int64_t* frame_offsets,
                       int num_frames, PcsTrackUnit* units, int n_units, void* ws, size_t ws_bytes,
                       pcs_stream_t stream);
size_t pcs_track_batch_workspace_bytes(int n_seqs, const int64_t* const* frame_offsets, const int* num_frames,
                                       const PcsTrackUnit* const* units, const int* n_units);
int pcs_track_batch(const PcsTrackParams* prm, int n_seqs, const float* const* seq_fxyz,
                    const int64_t* const* frame_offsets, const int* num_frames, PcsTrackUnit* const* units,
                    const int* n_units, void* ws, size_t ws_bytes, pcs_stream_t stream);

/* points_in_boxes_cpu, pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-168 (evaluation
 * side-channel of ClusterProposal / ClusterTracking): points [N,3], boxes [B,7]
 * (x,y,z,dx,dy,dz,heading) -> out [B,N] int32. */
int pcs_points_in_boxes(const float* points, int64_t N, const float* boxes, int B, int32_t* out,
                        pcs_stream_t stream);

/* The per-frame points_in_boxes_cpu calls of the evaluation loops (cluster_proposal.py:200-215,
 * cluster_tracking.py:388-400) for all frames at once. boxes [B,7] grouped by ascending frame,
 * box_offsets: DEVICE int32 [F+1]; point_frame [N] int64. A point is tested against the boxes of
 * its own frame only. member [N,Bmax] int32 (column = box index within the frame; 0 beyond the
 * frame's count) and / or first [N] int32 (lowest containing box index within the frame, -1 none);
 * either may be NULL. */
int pcs_points_in_frame_boxes(const float* points, const int64_t* point_frame, int64_t N,
                              const float* boxes, const int32_t* box_offsets, int F, int Bmax,
                              int32_t* member, int32_t* first, pcs_stream_t stream);

/* The same test summed on the fly: counts [G, Bmax] int32 += 1 for every (point j, box of its frame) that holds, under
 * row group[j] (int64 [N], values in [0, G): the (frame, component) group of component_box_iou_frames,
 * cluster_tracking.py:388-407 of the reference) -- the [N, Bmax] membership matrix is never written. The caller zeroes
 * counts. Integer adds: the table does not depend on the order of the points. */
int pcs_frame_box_counts(const float* points, const int64_t* point_frame, int64_t N, const float* boxes,
                         const int32_t* box_offsets, int F, int Bmax, const int64_t* group, int64_t G,
                         int32_t* counts, pcs_stream_t stream);

/* Sequence loader, device side: WaymoDataset.get_lidar / load_data / __getitem__
 * (pcdet/datasets/waymo/waymo_dataset.py:334-343, 383-391, 584-630) + the point_xyz branch of
 * collate_batch (pcdet/datasets/dataset.py:222-232) for all sweeps of one sample in one pass.
 * raw [N,8] f32 rows [x,y,z,intensity,elongation,range,rimage_w,rimage_h] of the kept lidars, sweeps
 * concatenated in output order; seg [N,2] int32 [instance, segmentation] or NULL; sweep_offsets
 * DEVICE int64 [F+1] (rows of sweep s are [off[s], off[s+1])); sweep_T DEVICE f64 [F,3,4] = rows of
 * T0^-1 T1; sweep_ids DEVICE int32 [F] (sample index of the sweep, int(frame_id[-3:])).
 * Out: point_bxyz [N,4] f32 (batch index, aligned xyz), point_feat [N,3] f32 (tanh intensity,
 * elongation, range/75), point_sweep [N] int32, segmentation_label / instance_label [N] int32
 * (untouched when seg is NULL). num_sweeps <= 1024. */
int pcs_assemble_sweeps(const float* raw, const int32_t* seg, const int64_t* sweep_offsets,
                        const double* sweep_T, const int32_t* sweep_ids, int num_sweeps, int64_t N,
                        int batch_index, float* point_bxyz, float* point_feat, int32_t* point_sweep,
                        int32_t* segmentation_label, int32_t* instance_label, pcs_stream_t stream);

/* l1_minimization, pcdet/models/registration/preprocessors/preprocessor_utils.py:313-350 (ground
 * height per 2 m pillar: L1 data term + 4-direction second-difference smoothness, AdamW with
 * MultiStepLR, <= max_iters steps, 3-strike stop at 1e-4), one persistent kernel.
 * pillar_min_z, weight, height_out: [X,Y] f32 (X*Y <= 12288); decay_steps: HOST int array;
 * d_iters: optional device int32 receiving the number of steps taken. */
size_t pcs_l1_minimization_workspace_bytes(int X, int Y);
int pcs_l1_minimization(const float* pillar_min_z, const float* weight, int X, int Y, double lr,
                        const int* decay_steps, int n_decay, float rigid_weight, int max_iters,
                        float* height_out, int32_t* d_iters, void* ws, size_t ws_bytes,
                        pcs_stream_t stream);

/* Measurement support for bench.py (not on the data path): HIP-event timing, on the launch stream, of the
 * iteration launches of the fused ICP, chunk by chunk (first iteration launch's start to the last one's stop:
 * the HIP events ride on the dispatches, hipExtLaunchKernelGGL, which is what rocprofv3 reports as kernel
 * begin / end). enable=1 resets and starts accumulating; read returns HOST doubles: [0] milliseconds, [1] chunks
 * timed, [2] algorithmic bytes, [3] unit-iterations of those chunks; [4..7] microseconds the resident loops'
 * workgroups spent in their four phases (resident schedules only); [8] 0. Chunks in which no registration
 * iterated are not counted. */
int pcs_profile_icp(int enable);
int pcs_profile_icp_read(double* out9);
/* the same nine values and, behind them: [9] the chunks' algorithmic bytes by SURVEY 8(d)'s formula, 112 (Mn + Rn) +
 * 144 C_live per unit and iteration ([2] is the implementation's own traffic model), [10] the unit-iterations of
 * registrations whose key numbers more than 5000 components (the 0.25 m key), [11] their 8(d) bytes; n >= 9 doubles. */
int pcs_profile_icp_read_ex(double* out, int n);

/* Schedule of the iterations of the fused ICP (pcs_register_to_next_frame and the tracker), process-wide;
 * takes effect for the chunks enqueued afterwards. Every schedule computes the same iterations (the same
 * neighbours, the same integer moments), so results do not depend on it; it is a performance choice.
 *   wide_iters < 0 (default): every iteration = four wide launches that serve all registrations in flight
 *     (blockIdx.y = registration), eight iterations per host round trip;
 *   wide_iters >= 1: that many wide iterations per registration (the full searches of a first iteration want
 *     the whole chip), then ONE resident workgroup per registration runs its iterations back to back (no
 *     launch between iterations): use_tiles != 0 with the candidates staged in LDS (clouds sorted by block
 *     row, bands of rows copied into LDS sorted by cell bucket, one lane or one wave per failed query;
 *     registrations that do not qualify take the cache path), use_tiles == 0 with the hash-grid searches. */
int pcs_icp_set_schedule(int wide_iters, int use_tiles);
/* How the grid of the MOVING points follows them between iterations (process-wide, registrations described after the
 * call): 0 (default) in place -- a point that stays in its fine cell has its stored row overwritten, cell changes leave
 * tombstones, the grid is rebuilt when they add up; 1: rebuilt at every iteration (the form of rounds 1-3, kept as the
 * cross-check of the in-place form: same results, bit for bit). */
int pcs_icp_set_grid_mode(int rebuild_always);
/* every search of the fused ICP looks radius / radius_divisor further than it must, so that it can say how long its
 * answer stays exact (default 16; any value >= 1 gives the same results: it trades cells per search against searches). */
int pcs_icp_set_widening(double radius_divisor);
/* iterations per chunk of the wide schedule's launches (default 10, 1..64): the lock-step engines read their units'
 * states once per chunk. A scheduling parameter: results do not depend on it. */
int pcs_icp_set_chunk_iterations(int n);
/* the verified cache of the fused ICP: 2 (default) = budgets and the second level (a query whose budget has run out
 * re-ranks its four cached points before it is searched), 1 = budgets only, the form of rounds 1-4 -- the same results
 * bit for bit (cross-check of the tests). Takes effect for registrations described afterwards. */
int pcs_icp_set_cache_levels(int levels);
/* diagnostics: counters of the fused ICP's verified cache since the last call (all zero unless the library was built
 * with -DPCS_ICP_STATS): host_out[16] = queries tested, budgets that had run out, of those kept by the second level
 * (the four cached points re-ranked), queries handed to a search; then the same four counts of the backward queries
 * (reference -> moving cloud) alone; [8], [9] = queries served by the cooperative / the eight-lane seeded search.
 * Synchronises the device. */
int pcs_icp_cache_stats(unsigned long long* host_out);

/* iterative_reweighted_ransac, pcdet/models/registration/preprocessors/preprocessor_utils.py:32-80
 * (per-pillar IRLS plane fit, <= max_rounds rounds, convergence test on the device; two kernels per
 * round instead of ~25 + eigh + a host sync). xyz [n,3] f32; pillar_idx [n] int64 in [0,P);
 * w0 [n] f32; origin [P,3] f32 (a point near each pillar, conditions the raw moments). Outputs:
 * plane [P,6] f32 (centre, unit normal with n_z >= 0), plane_err [n] f32, optional *d_rounds.
 * This form takes the rows in any order and adds its f64 moments with atomics: the last bits can differ from
 * run to run. pcs_irls_plane_fit_sorted (rows grouped by pillar) has fixed-order sums and is the one the
 * pipeline uses; ops.irls_plane_fit groups unsorted rows and calls that. */
size_t pcs_irls_plane_fit_workspace_bytes(int64_t n, int64_t P);
int pcs_irls_plane_fit(const float* xyz, const int64_t* pillar_idx, int64_t n, const float* w0,
                       const float* origin, int64_t P, float sigma2, float stopping_delta,
                       int max_rounds, float* plane, float* plane_err, int32_t* d_rounds, void* ws,
                       size_t ws_bytes, pcs_stream_t stream);

/* The same fit when the rows are grouped by pillar, as the reference has them after its
 * sort-by-pillar (preprocessor_utils.py:112-118): seg_start [P+1] int32 row ranges replace
 * pillar_idx. One launch per round (a workgroup per pillar, fixed-order sums: bit-reproducible). */
size_t pcs_irls_plane_fit_sorted_workspace_bytes(int64_t n, int64_t P);
int pcs_irls_plane_fit_sorted(const float* xyz, const int32_t* seg_start, int64_t n, const float* w0,
                              const float* origin, int64_t P, float sigma2, float stopping_delta,
                              int max_rounds, float* plane, float* plane_err, int32_t* d_rounds,
                              void* ws, size_t ws_bytes, pcs_stream_t stream);

/* The hypothesis loop of compute_min_height_from_ransac, preprocessor_utils.py:120-135: for each of
 * R height ratios (HOST float array, R <= 64) the plane fit above from the initial weights
 * sigma2 / ((min_z*ratio + max_z*(1-ratio) - z)^2 + sigma2), its hit count per pillar (rows with
 * point-plane error < hit_below) and, per pillar, the first ratio with the most hits. The R fits run
 * side by side (grid = pillars x ratios, one launch per round). min_z / max_z [P]; outputs
 * best_center / best_normal [P,3] (defaults 0 / (0,0,1) when nothing hits), best_conf [P]. */
size_t pcs_ransac_plane_hypotheses_workspace_bytes(int64_t n, int64_t P, int R);
int pcs_ransac_plane_hypotheses(const float* xyz, const int32_t* seg_start, int64_t n,
                                const float* min_z, const float* max_z, const float* ratios, int R,
                                const float* origin, int64_t P, float sigma2, float stopping_delta,
                                int max_rounds, float hit_below, float* best_center,
                                float* best_normal, float* best_conf, void* ws, size_t ws_bytes,
                                pcs_stream_t stream);
/* test hook: 1 = the height-ratio fits of pcs_ransac_plane_hypotheses on one workgroup per (pillar, fit) and round
 * (the form of rounds 3-4) instead of one per chunk of rows and round; a different summation tree. */
int pcs_irls_per_fit_rounds(int on);

/* The "truncated least squares" pruning of the plane hypotheses (preprocessor_utils.py:137-160: for
 * each of the T shrinking thresholds (host array, T <= 128) above which nothing lies the loop skips;
 * otherwise the hypotheses whose mean curvature over their K (<= 8) nearest hypotheses, themselves
 * included, is not below the threshold are dropped and the curvatures are recomputed) as ONE launch.
 * xyz / normal [H,3] f32 device; keep [H] uint8 device out (1 = survives). PCS_ERR_RANGE when H > 2048
 * (the caller then uses its own loop). */
int pcs_prune_plane_hypotheses(const float* xyz, const float* normal, int64_t H, int K,
                               const float* thresholds, int T, uint8_t* keep, pcs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PCS_H_ */
