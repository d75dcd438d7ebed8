/* CPU twins of the hot-path entries of pcs.h ("plus a CPU twin of every symbol
 * (pcs_cpu_<op>) for the oracle / CPU baseline").
 *
 * Same names with the prefix pcs_cpu_, the SAME argument lists as their pcs_* counterparts -- every
 * pointer is a HOST pointer, `ws` / `ws_bytes` / `stream` are accepted and ignored (pass NULL / 0) --
 * same return codes. They are exported by oracle/libpcs_oracle.so and oracle/libpcs_oracle_icp.so:
 * TEST INFRASTRUCTURE and CPU baseline only. The product library (libpcs_hip.so) contains no CPU
 * path and never calls these; a maintainer who wants the CPU baseline behind the same header links
 * the oracle libraries instead of libpcs_hip.so and renames the prefix.
 * The arithmetic is the oracle's (oracle/pcs_oracle.cpp: each function cites the reference lines it
 * restates); tests/test_cpu_twins.py checks the twins against the oracle's own entry points. */
#ifndef PCS_CPU_H
#define PCS_CPU_H
#include "pcs.h"

#ifdef __cplusplus
extern "C" {
#endif

int pcs_cpu_minmax4_f32(const float* pts, int64_t n, float* out8, pcs_stream_t stream);
int pcs_cpu_grid_cluster_ids(const float* pts, int64_t n, const float* bounds8, const float* size4,
                             float margin0, int64_t* ids, pcs_stream_t stream);
int pcs_cpu_unique_i64(const int64_t* ids, int64_t n, int key_bits, int64_t* uniq, int64_t* inv,
                       int32_t* order, int32_t* seg_start, int64_t* d_num, void* ws, size_t ws_bytes,
                       pcs_stream_t stream);
int pcs_cpu_segment_reduce_f32(const float* data, int d, const int32_t* order, const int32_t* seg_start,
                               int64_t v, int mean, float* out, pcs_stream_t stream);
int pcs_cpu_dynamic_voxelize(const float* pts, int64_t n, int ncol, const float* range6,
                             const float* voxel3, const int32_t* grid3, float* voxel_feat,
                             int32_t* voxel_coords, int64_t* inv, int64_t* d_num, void* ws,
                             size_t ws_bytes, pcs_stream_t stream);
int pcs_cpu_hard_voxelize(const float* pts, int64_t n, int ncol, const float* range6,
                          const float* voxel3, int max_points, int max_voxels, float* voxels,
                          int32_t* coords, int32_t* num, int64_t* d_num, void* ws, size_t ws_bytes,
                          pcs_stream_t stream);
int pcs_cpu_get_indice_pairs(const int32_t* indices, int64_t V, int batch, const int* out_shape3,
                             const int* spatial_shape3, const int* ksize3, const int* stride3,
                             const int* pad3, const int* dil3, int subm, int transpose,
                             int32_t* out_ids, int64_t out_cap, int32_t* pairs, int32_t* num,
                             int64_t* d_num_out, void* ws, size_t ws_bytes, pcs_stream_t stream);
int pcs_cpu_indice_conv_f32(const float* features, int cin, const float* filters, int kv, int cout,
                            const int32_t* pairs, int64_t pair_stride, const int32_t* num,
                            int64_t num_out, int inverse, int subm, float* out, void* ws,
                            size_t ws_bytes, pcs_stream_t stream);
int pcs_cpu_connected_components(const int64_t* e0, const int64_t* e1, int64_t E, int64_t n,
                                 int64_t* component, int64_t* d_num, void* ws, size_t ws_bytes,
                                 pcs_stream_t stream);
int pcs_cpu_points_in_boxes(const float* points, int64_t N, const float* boxes, int B, int32_t* out,
                            pcs_stream_t stream);
/* (exported by oracle/libpcs_oracle_icp.so; `cell_size` and `chunk` do not change any result) */
int pcs_cpu_register_to_next_frame(float* mov_fxyz, const int32_t* mov_comp, int64_t Mn,
                                   const float* ref_fxyz, int64_t Rn, const float* ref_all, int64_t Ra,
                                   const int64_t* comp_deg, int64_t C, double radius, int frame_offset,
                                   double angle_regularizer, int max_iter, double stopping_delta,
                                   float cell_size, int chunk, double* T, double* l1_err,
                                   float* edge_ratio, int32_t* info, void* ws, size_t ws_bytes,
                                   pcs_stream_t stream);

/* the ground stage's entries (round 5): oracle/pcs_oracle_ground.cpp, the canonical restatement -- the operations of
 * the HIP kernels in their order, so the twins return the entries' own bits; ws may be NULL */
int pcs_cpu_ransac_plane_hypotheses(const float* xyz, const int32_t* seg_start, int64_t n,
                                    const float* min_z, const float* max_z, const float* ratios, int R,
                                    const float* origin, int64_t P, float sigma2, float stopping_delta,
                                    int max_rounds, float hit_below, float* best_center,
                                    float* best_normal, float* best_conf, void* ws, size_t ws_bytes,
                                    pcs_stream_t stream);
int pcs_cpu_prune_plane_hypotheses(const float* xyz, const float* normal, int64_t H, int K,
                                   const float* thresholds, int T, uint8_t* keep, pcs_stream_t stream);
int pcs_cpu_l1_minimization(const float* pillar_min_z, const float* weight, int X, int Y, double lr,
                            const int* decay_steps, int n_decay, float rigid_weight, int max_iters,
                            float* height_out, int32_t* d_iters, void* ws, size_t ws_bytes,
                            pcs_stream_t stream);

/* the radius graph's two phases (pcs_radius_graph_search / _emit): the workspace carries the edges between them */
size_t pcs_cpu_radius_graph_workspace_bytes(int64_t N, int64_t M, int max_nbr);
int pcs_cpu_radius_graph_search(const float* ref, int64_t N, const float* query, int64_t M, float radius,
                                float cell_size, int dframe_min, int dframe_max, int max_nbr, int sort_by_dist,
                                int64_t* d_num_edges, void* ws, size_t ws_bytes, pcs_stream_t stream);
int pcs_cpu_radius_graph_emit(int64_t N, int64_t M, int max_nbr, int64_t* e_ref, int64_t* e_query, float* dists,
                              const void* ws, size_t ws_bytes, pcs_stream_t stream);
/* the components of that graph cut at n_cut radii (pcs_radius_graph_components; needs sort_by_dist = 1 in the search:
 * the workspace then holds the squared distances) */
size_t pcs_cpu_radius_graph_components_workspace_bytes(int64_t M, int n_cut);
int pcs_cpu_radius_graph_components(int64_t N, int64_t M, int max_nbr, const float* radius_cut, int n_cut,
                                    int64_t* component, int64_t* d_num, const void* ws, size_t ws_bytes, void* ws_cc,
                                    size_t ws_cc_bytes, pcs_stream_t stream);

/* second set (round 5, oracle/pcs_cpu_twins2.cpp): serial restatements of the reference lines each one cites there;
 * workspaces are not needed on the host (the *_workspace_bytes twins return 0; pass NULL / 0) */
int pcs_cpu_segment_pick(const int32_t* order, const int32_t* seg_start, int64_t v, int pick_last, int64_t* out,
                         pcs_stream_t stream);
int pcs_cpu_segment_median_i64(const int64_t* data, const int32_t* order, const int32_t* seg_start, int64_t v,
                               int64_t* out, pcs_stream_t stream);
int pcs_cpu_segment_minmax_f32(const float* data, int d, const int32_t* order, const int32_t* seg_start, int64_t v,
                               int is_max, float* out, pcs_stream_t stream);
int pcs_cpu_hash_insert(int64_t* keys, float* values, int64_t* reverse, int64_t H, const int64_t* dims, int D,
                        const int64_t* insert_keys, const float* insert_values, int64_t n, pcs_stream_t stream);
size_t pcs_cpu_radius_graph_table_workspace_bytes(int64_t M);
int pcs_cpu_radius_graph_table_count(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                                     const int64_t* dims, int D, const int64_t* query_keys, const float* query_values,
                                     int64_t M, const int32_t* qmin, const int32_t* qmax, const float* radius,
                                     int max_nbr, int32_t* degree, int32_t* offset, int64_t* d_num_edges, void* ws,
                                     size_t ws_bytes, pcs_stream_t stream);
int pcs_cpu_radius_graph_table_fill(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                                    const int64_t* dims, int D, const int64_t* query_keys, const float* query_values,
                                    int64_t M, const int32_t* qmin, const int32_t* qmax, const float* radius,
                                    int sort_by_dist, const int32_t* degree, const int32_t* offset, int64_t* edges,
                                    float* dists, pcs_stream_t stream);
int pcs_cpu_correspondence(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                           const int64_t* dims, int D, const int64_t* query_keys, const float* query_values, int64_t M,
                           const int32_t* qmin, const int32_t* qmax, int64_t* corres_indices, pcs_stream_t stream);
int pcs_cpu_points_in_radius(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                             const int64_t* dims, int D, const int64_t* query_keys, const float* query_values, int64_t M,
                             const int32_t* qmin, const int32_t* qmax, float radius, int64_t* visited,
                             pcs_stream_t stream);
size_t pcs_cpu_neighbor_table_bytes(int64_t num_out, int kv);
int pcs_cpu_build_neighbor_table(const int32_t* pairs, int64_t pair_stride, const int32_t* num, int kv, int inverse,
                                 int64_t num_out, int32_t* table, pcs_stream_t stream);
int pcs_cpu_indice_conv_table_f32(const float* features, int cin, const float* filters, int kv, int cout, int centre,
                                  const int32_t* table, int64_t num_out, const float* bias, const float* scale,
                                  const float* shift, int relu, float* out, pcs_stream_t stream);
int pcs_cpu_indice_conv_table_residual_f32(const float* features, int cin, const float* filters, int kv, int cout,
                                           int centre, const int32_t* table, int64_t num_out, const float* bias,
                                           const float* scale, const float* shift, const float* residual, int relu,
                                           float* out, pcs_stream_t stream);
int pcs_cpu_bn_fold_f32(const float* weight, const float* bias, const float* running_mean, const float* inv_std, int C,
                        float* scale, float* shift, pcs_stream_t stream);
size_t pcs_cpu_indice_conv_backward_workspace_bytes(int64_t num_in, int64_t pair_stride, int kv, int cin, int cout);
int pcs_cpu_indice_conv_backward_f32(const float* features, int cin, const float* filters, int kv, int cout,
                                     const float* out_grad, const int32_t* pairs, int64_t pair_stride,
                                     const int32_t* num, int64_t num_in, int64_t num_out, int inverse, int subm,
                                     float* in_grad, float* filters_grad, void* ws, size_t ws_bytes,
                                     pcs_stream_t stream);
int pcs_cpu_indice_maxpool_f32(const float* features, int C, const int32_t* pairs, int64_t pair_stride,
                               const int32_t* num, int kv, int64_t num_out, float* out, void* ws, size_t ws_bytes,
                               pcs_stream_t stream);
int pcs_cpu_indice_maxpool_backward_f32(const float* features, const float* out_features, const float* out_grad, int C,
                                        const int32_t* pairs, int64_t pair_stride, const int32_t* num, int kv,
                                        int64_t num_in, float* in_grad, void* ws, size_t ws_bytes,
                                        pcs_stream_t stream);
int pcs_cpu_sparse_to_dense(const float* features, const int32_t* indices, int64_t V, int C, int B, int D, int H, int W,
                            float* out, int* status, pcs_stream_t stream);
int pcs_cpu_points_in_frame_boxes(const float* points, const int64_t* point_frame, int64_t N, const float* boxes,
                                  const int32_t* box_offsets, int F, int Bmax, int32_t* member, int32_t* first,
                                  pcs_stream_t stream);
int pcs_cpu_frame_box_counts(const float* points, const int64_t* point_frame, int64_t N, const float* boxes,
                             const int32_t* box_offsets, int F, int Bmax, const int64_t* group, int64_t G,
                             int32_t* counts, pcs_stream_t stream);
int pcs_cpu_assemble_sweeps(const float* raw, const int32_t* seg, const int64_t* sweep_offsets, const double* sweep_T,
                            const int32_t* sweep_ids, int num_sweeps, int64_t N, int batch_index, float* point_bxyz,
                            float* point_feat, int32_t* point_sweep, int32_t* segmentation_label,
                            int32_t* instance_label, pcs_stream_t stream);
size_t pcs_cpu_smooth_velo_workspace_bytes(int64_t C, int F);
int pcs_cpu_smooth_velo(float* comp_velos, const float* comp_center_diffs, int64_t C, int F, int frame_lo, int frame_hi,
                        float weight0, float weight, int num_itr, float stopping, void* ws, size_t ws_bytes,
                        pcs_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif
