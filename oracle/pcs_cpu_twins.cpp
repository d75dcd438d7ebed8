// pcs_cpu_<op>: the oracle's entry points behind the argument lists of include/pcs.h (see
// include/pcs_cpu.h). TEST INFRASTRUCTURE / CPU baseline only: thin adapters, no arithmetic of their own.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/pcs_cpu.h"

extern "C" {
// oracle/pcs_oracle.cpp
void orc_grid_cluster_ids(const float* pts, int64_t n, const float* size4, float margin0, int64_t* ids,
                          float* bounds8_out);
int64_t orc_unique_i64(const int64_t* ids, int64_t n, int64_t* uniq, int64_t* inv, int32_t* order,
                       int32_t* seg_start);
void orc_segment_reduce_f32_mode(const float* data, int d, const int32_t* order, const int32_t* seg_start,
                                 int64_t v, int mean, int mode, float* out);
int64_t orc_dynamic_voxelize(const float* pts, int64_t n, int ncol, const float* range6, const float* voxel3,
                             const int32_t* grid3, float* feat, int32_t* coords, int64_t* inv);
int64_t orc_hard_voxelize(const float* pts, int64_t n, int ncol, const float* range6, const float* voxel3,
                          int max_points, int max_voxels, float* voxels, int32_t* coords, int32_t* num);
int64_t orc_get_indice_pairs(const int32_t* indices, int64_t V, const int* out_shape3, const int* ksize3,
                             const int* stride3, const int* pad3, const int* dil3, int subm, int32_t* out_ids,
                             int32_t* pairs, int32_t* num);
void orc_indice_conv_f32(const float* feat, int64_t num_in, int cin, const float* W, int kv, int cout,
                         const int32_t* pairs, int64_t pair_stride, const int32_t* num, int64_t num_out,
                         int inverse, int subm, float* out);
int64_t orc_connected_components(const int64_t* e0, const int64_t* e1, int64_t E, int64_t n, int64_t* component);
void orc_points_in_boxes(const float* pts, int64_t N, const float* boxes, int B, int32_t* out);
int64_t orc_radius_graph(const float* ref, int64_t N, const float* qry, int64_t M, float radius, const int* qmin,
                         const int* qmax, int max_nbr, int sort_by_dist, int64_t* e_ref, int64_t* e_qry, float* dists);
void orc_ransac_plane_hypotheses(const float* xyz, const int32_t* seg_start, int64_t n, const float* min_z,
                                 const float* max_z, const float* ratios, int R, const float* origin, int64_t P,
                                 float sigma2, float delta, int max_rounds, float hit_below, float* best_center,
                                 float* best_normal, float* best_conf);
void orc_prune_plane_hypotheses(const float* xyz, const float* normal, int H, int K, const float* thr, int T,
                                uint8_t* keep);
int orc_l1_minimization(const float* min_z, const float* weight, int X, int Y, double lr0, const int* decay_steps,
                        int n_decay, float rigid, int max_iters, float* h_out);

#define CPU_OK 0
#define CPU_ERR_ARG (-1)
#define CPU_ERR_RANGE (-4)

int pcs_cpu_minmax4_f32(const float* pts, int64_t n, float* out8, pcs_stream_t) {
  if (!pts || !out8 || n < 1) return CPU_ERR_ARG;
  for (int d = 0; d < 4; ++d) { out8[d] = INFINITY; out8[4 + d] = -INFINITY; }
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < 4; ++d) {
      out8[d] = std::fmin(out8[d], pts[i * 4 + d]);
      out8[4 + d] = std::fmax(out8[4 + d], pts[i * 4 + d]);
    }
  return CPU_OK;
}

int pcs_cpu_grid_cluster_ids(const float* pts, int64_t n, const float* bounds8, const float* size4, float margin0,
                             int64_t* ids, pcs_stream_t) {
  if (!pts || !size4 || !ids || n < 0) return CPU_ERR_ARG;
  (void)bounds8;  // (the oracle recomputes the bounds: bounds8 is pcs_minmax4_f32 of the same points)
  orc_grid_cluster_ids(pts, n, size4, margin0, ids, nullptr);
  return CPU_OK;
}

int pcs_cpu_unique_i64(const int64_t* ids, int64_t n, int key_bits, int64_t* uniq, int64_t* inv, int32_t* order,
                       int32_t* seg_start, int64_t* d_num, void*, size_t, pcs_stream_t) {
  (void)key_bits;
  if (!ids || !seg_start || !d_num || n < 0) return CPU_ERR_ARG;
  std::vector<int64_t> u(uniq ? 0 : (size_t)n), iv(inv ? 0 : (size_t)n);
  std::vector<int32_t> od(order ? 0 : (size_t)n);
  *d_num = orc_unique_i64(ids, n, uniq ? uniq : u.data(), inv ? inv : iv.data(), order ? order : od.data(), seg_start);
  return CPU_OK;
}

int pcs_cpu_segment_reduce_f32(const float* data, int d, const int32_t* order, const int32_t* seg_start, int64_t v,
                               int mean, float* out, pcs_stream_t) {
  if (!data || !order || !seg_start || !out || d < 1 || v < 0) return CPU_ERR_ARG;
  const int mode = ((mean & 4) && d <= 16) ? 2 : (((mean & 2) && d <= 16) ? 1 : 0);
  orc_segment_reduce_f32_mode(data, d, order, seg_start, v, mean & 1, mode, out);
  return CPU_OK;
}

int pcs_cpu_dynamic_voxelize(const float* pts, int64_t n, int ncol, const float* range6, const float* voxel3,
                             const int32_t* grid3, float* voxel_feat, int32_t* voxel_coords, int64_t* inv,
                             int64_t* d_num, void*, size_t, pcs_stream_t) {
  if (!pts || !range6 || !voxel3 || !grid3 || !voxel_feat || !voxel_coords || !d_num || ncol < 4 || n < 0)
    return CPU_ERR_ARG;
  *d_num = orc_dynamic_voxelize(pts, n, ncol, range6, voxel3, grid3, voxel_feat, voxel_coords, inv);
  return CPU_OK;
}

int pcs_cpu_hard_voxelize(const float* pts, int64_t n, int ncol, const float* range6, const float* voxel3,
                          int max_points, int max_voxels, float* voxels, int32_t* coords, int32_t* num,
                          int64_t* d_num, void*, size_t, pcs_stream_t) {
  if (!pts || !range6 || !voxel3 || !voxels || !coords || !num || !d_num || ncol < 3 || n < 0) return CPU_ERR_ARG;
  std::memset(voxels, 0, sizeof(float) * (size_t)max_voxels * max_points * ncol);
  std::memset(coords, 0, sizeof(int32_t) * (size_t)max_voxels * 3);
  std::memset(num, 0, sizeof(int32_t) * (size_t)max_voxels);
  *d_num = orc_hard_voxelize(pts, n, ncol, range6, voxel3, max_points, max_voxels, voxels, coords, num);
  return CPU_OK;
}

int pcs_cpu_get_indice_pairs(const int32_t* indices, int64_t V, int batch, const int* out_shape3,
                             const int* spatial_shape3, const int* ksize3, const int* stride3, const int* pad3,
                             const int* dil3, int subm, int transpose, int32_t* out_ids, int64_t out_cap,
                             int32_t* pairs, int32_t* num, int64_t* d_num_out, void*, size_t, pcs_stream_t) {
  (void)batch; (void)spatial_shape3; (void)out_cap;
  if (!out_shape3 || !ksize3 || !stride3 || !pad3 || !dil3 || !num || !d_num_out || V < 0 ||
      (V > 0 && (!indices || !pairs)) || (transpose && subm))
    return CPU_ERR_ARG;
  std::vector<int32_t> tmp;
  if (!out_ids) { tmp.resize((size_t)(V > 0 ? V : 1) * 27 * 4); out_ids = tmp.data(); }
  *d_num_out = orc_get_indice_pairs(indices, V, out_shape3, ksize3, stride3, pad3, dil3,
                                    transpose ? 2 : (subm ? 1 : 0), out_ids, pairs, num);
  return CPU_OK;
}

int pcs_cpu_indice_conv_f32(const float* features, int cin, const float* filters, int kv, int cout,
                            const int32_t* pairs, int64_t pair_stride, const int32_t* num, int64_t num_out,
                            int inverse, int subm, float* out, void*, size_t, pcs_stream_t) {
  if (!features || !filters || !pairs || !num || !out) return CPU_ERR_ARG;
  orc_indice_conv_f32(features, 0, cin, filters, kv, cout, pairs, pair_stride, num, num_out, inverse, subm, out);
  return CPU_OK;
}

int pcs_cpu_connected_components(const int64_t* e0, const int64_t* e1, int64_t E, int64_t n, int64_t* component,
                                 int64_t* d_num, void*, size_t, pcs_stream_t) {
  if (!component || !d_num || n < 0 || E < 0 || (E > 0 && (!e0 || !e1))) return CPU_ERR_ARG;
  *d_num = orc_connected_components(e0, e1, E, n, component);
  return CPU_OK;
}

int pcs_cpu_points_in_boxes(const float* points, int64_t N, const float* boxes, int B, int32_t* out, pcs_stream_t) {
  if (!points || !boxes || !out || N < 0 || B < 0) return CPU_ERR_ARG;
  orc_points_in_boxes(points, N, boxes, B, out);
  return CPU_OK;
}

// ---- ground stage (round 5): the canonical restatement of oracle/pcs_oracle_ground.cpp behind the entries' own
// argument lists (workspaces are not needed on the host: may be NULL)
int pcs_cpu_ransac_plane_hypotheses(const float* xyz, const int32_t* seg_start, int64_t n, const float* min_z,
                                    const float* max_z, const float* ratios, int R, const float* origin, int64_t P,
                                    float sigma2, float stopping_delta, int max_rounds, float hit_below,
                                    float* best_center, float* best_normal, float* best_conf, void*, size_t,
                                    pcs_stream_t) {
  if (!xyz || !seg_start || !min_z || !max_z || !ratios || !origin || !best_center || !best_normal || !best_conf ||
      n < 1 || P < 1 || R < 1 || R > 64 || max_rounds < 1)
    return PCS_ERR_ARG;
  orc_ransac_plane_hypotheses(xyz, seg_start, n, min_z, max_z, ratios, R, origin, P, sigma2, stopping_delta, max_rounds,
                              hit_below, best_center, best_normal, best_conf);
  return PCS_OK;
}

int pcs_cpu_prune_plane_hypotheses(const float* xyz, const float* normal, int64_t H, int K, const float* thresholds,
                                   int T, uint8_t* keep, pcs_stream_t) {
  if (H < 0 || K < 1 || K > 8 || T < 0 || T > 128 || (H > 0 && (!xyz || !normal || !keep)) || (T > 0 && !thresholds))
    return PCS_ERR_ARG;
  if (H > 2048) return PCS_ERR_RANGE;
  if (H > 0) orc_prune_plane_hypotheses(xyz, normal, (int)H, K, thresholds, T, keep);
  return PCS_OK;
}

int pcs_cpu_l1_minimization(const float* pillar_min_z, const float* weight, int X, int Y, double lr,
                            const int* decay_steps, int n_decay, float rigid_weight, int max_iters, float* height_out,
                            int32_t* d_iters, void*, size_t, pcs_stream_t) {
  if (!pillar_min_z || !weight || !height_out || X < 3 || Y < 3 || n_decay < 0 || (n_decay > 0 && !decay_steps))
    return PCS_ERR_ARG;
  const int it = orc_l1_minimization(pillar_min_z, weight, X, Y, lr, decay_steps, n_decay, rigid_weight, max_iters,
                                     height_out);
  if (d_iters) *d_iters = it;
  return PCS_OK;
}


// ---- radius graph (round 5): pcs_radius_graph_search / _emit's two phases on the host. The workspace holds the
// edges between the calls: [E][e_ref M * max_nbr][e_query M * max_nbr][dists M * max_nbr]. cell_size is the device
// engine's tuning knob and has no effect here (the restatement walks the reference's own cells of one radius).
size_t pcs_cpu_radius_graph_workspace_bytes(int64_t N, int64_t M, int max_nbr) {
  (void)N;
  const size_t cap = (size_t)(M > 0 ? M : 1) * (size_t)(max_nbr > 0 ? max_nbr : 1);
  return sizeof(int64_t) + cap * (2 * sizeof(int64_t) + sizeof(float));
}
int pcs_cpu_radius_graph_search(const float* ref, int64_t N, const float* query, int64_t M, float radius, float,
                                int dframe_min, int dframe_max, int max_nbr, int sort_by_dist, int64_t* d_num_edges,
                                void* ws, size_t ws_bytes, pcs_stream_t) {
  if (!d_num_edges || !ws || N < 0 || M < 0 || (N > 0 && !ref) || (M > 0 && !query) || !(radius > 0.f) ||
      dframe_min > dframe_max || max_nbr < 1 || max_nbr > 32)
    return CPU_ERR_ARG;
  if (ws_bytes < pcs_cpu_radius_graph_workspace_bytes(N, M, max_nbr)) return PCS_ERR_WORKSPACE;
  const size_t cap = (size_t)(M > 0 ? M : 1) * (size_t)max_nbr;
  int64_t* head = (int64_t*)ws;
  int64_t* e_ref = head + 1;
  int64_t* e_qry = e_ref + cap;
  float* dists = (float*)(e_qry + cap);
  const int qmin[4] = {dframe_min, -1, -1, -1}, qmax[4] = {dframe_max, 1, 1, 1};
  head[0] = orc_radius_graph(ref, N, query, M, radius, qmin, qmax, max_nbr, sort_by_dist, e_ref, e_qry, dists);
  *d_num_edges = head[0];
  return CPU_OK;
}
int pcs_cpu_radius_graph_emit(int64_t N, int64_t M, int max_nbr, int64_t* e_ref, int64_t* e_query, float* dists,
                              const void* ws, size_t ws_bytes, pcs_stream_t) {
  if (!ws || M < 0 || max_nbr < 1 || max_nbr > 32) return CPU_ERR_ARG;
  if (ws_bytes < pcs_cpu_radius_graph_workspace_bytes(N, M, max_nbr)) return PCS_ERR_WORKSPACE;
  const size_t cap = (size_t)(M > 0 ? M : 1) * (size_t)max_nbr;
  const int64_t* head = (const int64_t*)ws;
  const int64_t E = head[0];
  if (E <= 0) return CPU_OK;
  if (!e_ref || !e_query) return CPU_ERR_ARG;
  std::memcpy(e_ref, head + 1, (size_t)E * sizeof(int64_t));
  std::memcpy(e_query, head + 1 + cap, (size_t)E * sizeof(int64_t));
  if (dists) std::memcpy(dists, (const float*)(head + 1 + 2 * cap), (size_t)E * sizeof(float));
  return CPU_OK;
}

// ClusterProposal.propose_cluster's graphs + components (cluster_proposal.py:57-83, graph_utils.py:40-53) from the edges
// pcs_cpu_radius_graph_search left in `ws`: per cut radius the edges with dist2 <= fl32(r) * fl32(r), then the
// components' ranks by smallest node id. component [n_cut, M]; d_num [n_cut].
size_t pcs_cpu_radius_graph_components_workspace_bytes(int64_t, int) { return 0; }
int pcs_cpu_radius_graph_components(int64_t N, int64_t M, int max_nbr, const float* radius_cut, int n_cut,
                                    int64_t* component, int64_t* d_num, const void* ws, size_t ws_bytes, void*, size_t,
                                    pcs_stream_t) {
  if (!ws || !radius_cut || !component || !d_num || M < 0 || N != M || max_nbr < 1 || max_nbr > 32 || n_cut < 1 ||
      n_cut > 8)
    return CPU_ERR_ARG;
  if (ws_bytes < pcs_cpu_radius_graph_workspace_bytes(N, M, max_nbr)) return PCS_ERR_WORKSPACE;
  const size_t cap = (size_t)(M > 0 ? M : 1) * (size_t)max_nbr;
  const int64_t* head = (const int64_t*)ws;
  const int64_t E = head[0];
  const int64_t* e_ref = head + 1;
  const int64_t* e_qry = e_ref + cap;
  const float* dists = (const float*)(e_qry + cap);
  std::vector<int64_t> a, b;
  for (int c = 0; c < n_cut; ++c) {
    const float r2 = radius_cut[c] * radius_cut[c];
    a.clear();
    b.clear();
    for (int64_t e = 0; e < E; ++e)
      if (dists[e] <= r2) { a.push_back(e_ref[e]); b.push_back(e_qry[e]); }
    d_num[c] = orc_connected_components(a.data(), b.data(), (int64_t)a.size(), M, component + (int64_t)c * M);
  }
  return CPU_OK;
}

}  // extern "C"
