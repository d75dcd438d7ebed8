// pcs_cpu_<op>, second set (round 5): host twins of the segment helpers, the torch_hash table drop-ins, the
// sparse-conv table / backward / pooling entries, the evaluation's box tests, the loader's sweep assembly and
// smooth_velo -- each behind the argument list of its pcs_* counterpart (include/pcs_cpu.h).
// TEST INFRASTRUCTURE / CPU baseline only, like everything under oracle/: the product library never calls these.
// Each function restates the reference lines it cites (serial loops, fp32 where the reference computes in fp32);
// nothing here is derived from the HIP kernels except where a comment says which canonical order it adopts.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/pcs_cpu.h"

extern "C" {

void orc_points_in_boxes(const float* pts, int64_t N, const float* boxes, int B, int32_t* out);

// ---------------------------------------------------------------------------------------------------------------
// segments of pcs_unique_i64 (order = original indices grouped by segment, ascending inside; seg_start [v+1])

// scatter(arange(n), inv, reduce='max' | 'min'), pcdet/models/registration/simple_reg.py:122-124
int pcs_cpu_segment_pick(const int32_t* order, const int32_t* seg_start, int64_t v, int pick_last, int64_t* out,
                         pcs_stream_t) {
  if (!order || !seg_start || !out || v < 0) return PCS_ERR_ARG;
  for (int64_t s = 0; s < v; ++s) out[s] = pick_last ? order[seg_start[s + 1] - 1] : order[seg_start[s]];
  return PCS_OK;
}

// robust_median, pcdet/models/registration/preprocessors/registration_utils.py:60-81: the segment's values sorted,
// the element at position deg // 2
int pcs_cpu_segment_median_i64(const int64_t* data, const int32_t* order, const int32_t* seg_start, int64_t v,
                               int64_t* out, pcs_stream_t) {
  if (v < 0 || (v > 0 && (!data || !order || !seg_start || !out))) return PCS_ERR_ARG;
  std::vector<int64_t> tmp;
  for (int64_t s = 0; s < v; ++s) {
    const int b = seg_start[s], e = seg_start[s + 1];
    tmp.resize((size_t)(e - b));
    for (int p = b; p < e; ++p) tmp[(size_t)(p - b)] = data[order[p]];
    std::nth_element(tmp.begin(), tmp.begin() + (e - b) / 2, tmp.end());
    out[s] = tmp[(size_t)((e - b) / 2)];
  }
  return PCS_OK;
}

// torch_scatter.scatter(reduce='max' | 'min'), cluster_tracking.py:104,325-328
int pcs_cpu_segment_minmax_f32(const float* data, int d, const int32_t* order, const int32_t* seg_start, int64_t v,
                               int is_max, float* out, pcs_stream_t) {
  if (v < 0 || d < 1 || (v > 0 && (!data || !order || !seg_start || !out))) return PCS_ERR_ARG;
  for (int64_t s = 0; s < v; ++s)
    for (int c = 0; c < d; ++c) {
      float acc = is_max ? -INFINITY : INFINITY;
      for (int p = seg_start[s]; p < seg_start[s + 1]; ++p) {
        const float x = data[(int64_t)order[p] * d + c];
        acc = is_max ? std::fmax(acc, x) : std::fmin(acc, x);
      }
      out[s * d + c] = acc;
    }
  return PCS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// torch_hash_cuda on the reference's own table (pcdet/ops/torch_hash/src/torch_hash_kernel.cu)

static inline int64_t map2key(const int64_t* c, const int64_t* dims, int D) {  // :31-47 (clamps to dims[i], not - 1)
  int64_t ans = 0;
  for (int i = 0; i < D; ++i) {
    int64_t k = c[i];
    if (k >= dims[i]) k = dims[i];
    if (k < 0) k = 0;
    ans = ans * dims[i] + k;
  }
  return ans;
}
static inline int64_t hashkey(int64_t key, int64_t H) { return ((key % H) * 999269 + 999437) % H; }  // :27-28,49-51

// hash_insert_gpu (:54-91,411-442): every point takes the first free slot of its probe chain (inserted serially here;
// the reference's slot assignment follows its atomicCAS race -- no query result depends on it)
int pcs_cpu_hash_insert(int64_t* keys, float* values, int64_t* reverse, int64_t H, const int64_t* dims, int D,
                        const int64_t* insert_keys, const float* insert_values, int64_t n, pcs_stream_t) {
  if (!keys || !values || !reverse || !dims || H < 1 || D < 1 || D > 8 || n < 0) return PCS_ERR_ARG;
  if (n == 0) return PCS_OK;
  if (!insert_keys || !insert_values || n > H) return PCS_ERR_ARG;
  int64_t used = 0;
  for (int64_t s = 0; s < H; ++s) used += keys[s] != -1;
  if (used + n > H) return PCS_ERR_ARG;  // (a full table: the probe would never end)
  for (int64_t i = 0; i < n; ++i) {
    const int64_t key = map2key(insert_keys + i * D, dims, D);
    int64_t slot = hashkey(key, H);
    while (keys[slot] != -1) slot = (slot + 1) % H;
    keys[slot] = key;
    for (int d = 0; d < D; ++d) values[slot * D + d] = insert_values[i * D + d];
    reverse[slot] = i;
  }
  return PCS_OK;
}

// the cell walk of every query kernel of the file: combinations with dim 0 fastest, probe chain until an empty slot
extern "C++" {
template <class Hit>
static void walk(const int64_t* keys, const float* values, int64_t H, const int64_t* dims, int D, const int64_t* qk,
                 const float* qv, const int32_t* qmin, const int32_t* qmax, Hit hit) {
  int w[8], ncomb = 1;
  for (int d = 0; d < D; ++d) { w[d] = qmax[d] - qmin[d] + 1; ncomb *= w[d]; }
  for (int c = 0; c < ncomb; ++c) {
    int64_t cc[8];
    int tmp = c;
    for (int d = 0; d < D; ++d) { cc[d] = qk[d] + (tmp % w[d] + qmin[d]); tmp /= w[d]; }
    const int64_t key = map2key(cc, dims, D);
    int64_t slot = hashkey(key, H);
    while (keys[slot] != -1) {
      if (keys[slot] == key) {
        float dist2 = 0.f;
        for (int d = 0; d < D; ++d) {
          const float di = values[slot * D + d] - qv[d];
          dist2 = dist2 + di * di;
        }
        hit(slot, dist2);
      }
      slot = (slot + 1) % H;
    }
  }
}
}  // extern "C++"

size_t pcs_cpu_radius_graph_table_workspace_bytes(int64_t) { return 0; }

// radius_graph_gpu, phase 1 (count_radius_graph_degree_kernel :224-300 + cumsum :487-520)
int pcs_cpu_radius_graph_table_count(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                                     const int64_t* dims, int D, const int64_t* query_keys, const float* query_values,
                                     int64_t M, const int32_t* qmin, const int32_t* qmax, const float* radius,
                                     int max_nbr, int32_t* degree, int32_t* offset, int64_t* d_num_edges, void*, size_t,
                                     pcs_stream_t) {
  if (!keys || !values || !reverse || !dims || !qmin || !qmax || !radius || !degree || !offset || !d_num_edges ||
      D < 1 || D > 8 || M < 0 || H < 1)
    return PCS_ERR_ARG;
  int64_t E = 0;
  for (int64_t q = 0; q < M; ++q) {
    const float r2 = radius[q] * radius[q];
    int cnt = 0;
    walk(keys, values, H, dims, D, query_keys + q * D, query_values + q * D, qmin, qmax, [&](int64_t, float dist2) {
      if (dist2 <= r2 && (max_nbr == -1 || cnt < max_nbr)) ++cnt;
    });
    degree[q] = cnt;
    offset[q] = (int32_t)E;
    E += cnt;
  }
  *d_num_edges = E;
  return PCS_OK;
}

// phase 2 (radius_graph_kernel :302-409): edges [E,2] (ref, query) and squared distances; which `degree[q]` of the
// points in range are kept is race-ordered in the reference -- canonical here as in the device entry: the nearest by
// (dist2, ref index) when sort_by_dist, the lowest ref indices otherwise
int pcs_cpu_radius_graph_table_fill(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                                    const int64_t* dims, int D, const int64_t* query_keys, const float* query_values,
                                    int64_t M, const int32_t* qmin, const int32_t* qmax, const float* radius,
                                    int sort_by_dist, const int32_t* degree, const int32_t* offset, int64_t* edges,
                                    float* dists, pcs_stream_t) {
  if (!keys || !values || !reverse || !dims || !qmin || !qmax || !radius || !degree || !offset || D < 1 || D > 8 ||
      M < 0 || H < 1)
    return PCS_ERR_ARG;
  if (M == 0) return PCS_OK;
  if (!edges || !dists) return PCS_ERR_ARG;
  std::vector<std::pair<float, int64_t>> found;
  for (int64_t q = 0; q < M; ++q) {
    const float r2 = radius[q] * radius[q];
    found.clear();
    walk(keys, values, H, dims, D, query_keys + q * D, query_values + q * D, qmin, qmax, [&](int64_t slot, float dist2) {
      if (dist2 <= r2) found.emplace_back(dist2, reverse[slot]);
    });
    std::sort(found.begin(), found.end(), [&](const std::pair<float, int64_t>& a, const std::pair<float, int64_t>& b) {
      if (sort_by_dist && a.first != b.first) return a.first < b.first;
      return a.second < b.second;
    });
    for (int e = 0; e < degree[q] && e < (int)found.size(); ++e) {
      edges[((int64_t)offset[q] + e) * 2] = found[(size_t)e].second;
      edges[((int64_t)offset[q] + e) * 2 + 1] = q;
      dists[(int64_t)offset[q] + e] = found[(size_t)e].first;
    }
  }
  return PCS_OK;
}

// correspondence (:96-155,444-485): nearest hashed point among the walked cells, strict `<` from 1e10, ties (the
// reference: insertion race) by the lower index; -1 if none
int pcs_cpu_correspondence(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                           const int64_t* dims, int D, const int64_t* query_keys, const float* query_values, int64_t M,
                           const int32_t* qmin, const int32_t* qmax, int64_t* corres_indices, pcs_stream_t) {
  if (!keys || !values || !reverse || !dims || !qmin || !qmax || D < 1 || D > 8 || M < 0 || H < 1) return PCS_ERR_ARG;
  if (M == 0) return PCS_OK;
  if (!query_keys || !query_values || !corres_indices) return PCS_ERR_ARG;
  for (int64_t q = 0; q < M; ++q) {
    float best = 1e10f;
    int64_t bi = -1;
    walk(keys, values, H, dims, D, query_keys + q * D, query_values + q * D, qmin, qmax, [&](int64_t slot, float dist2) {
      const int64_t ri = reverse[slot];
      if (dist2 < best || (dist2 == best && bi >= 0 && ri < bi)) { best = dist2; bi = ri; }
    });
    corres_indices[q] = bi;
  }
  return PCS_OK;
}

// points_in_radius_gpu (:160-222,563-605): visited[ref] = 1 for every hashed point strictly closer than radius
int pcs_cpu_points_in_radius(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                             const int64_t* dims, int D, const int64_t* query_keys, const float* query_values, int64_t M,
                             const int32_t* qmin, const int32_t* qmax, float radius, int64_t* visited, pcs_stream_t) {
  if (!keys || !values || !reverse || !dims || !qmin || !qmax || D < 1 || D > 8 || M < 0 || H < 1) return PCS_ERR_ARG;
  if (M == 0) return PCS_OK;
  if (!query_keys || !query_values || !visited) return PCS_ERR_ARG;
  const float r2 = radius * radius;
  for (int64_t q = 0; q < M; ++q)
    walk(keys, values, H, dims, D, query_keys + q * D, query_values + q * D, qmin, qmax, [&](int64_t slot, float dist2) {
      if (dist2 < r2) visited[reverse[slot]] = 1;
    });
  return PCS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// sparse conv: output-major table, fused epilogue, backward, pooling, dense()

size_t pcs_cpu_neighbor_table_bytes(int64_t num_out, int kv) {
  return (size_t)(num_out > 0 ? num_out : 1) * (size_t)kv * sizeof(int32_t);
}

// the rulebook's pairs by output row: table[o][k] = the input paired with output o at offset k, -1 where none (an
// output appears at most once per offset: what makes the reference's plain scatter-add legal,
// include/spconv/reordering.cu.h:147-156)
int pcs_cpu_build_neighbor_table(const int32_t* pairs, int64_t pair_stride, const int32_t* num, int kv, int inverse,
                                 int64_t num_out, int32_t* table, pcs_stream_t) {
  if (!pairs || !num || !table || kv < 1 || num_out < 0 || pair_stride < 0) return PCS_ERR_ARG;
  for (int64_t i = 0; i < num_out * kv; ++i) table[i] = -1;
  for (int k = 0; k < kv; ++k) {
    const int32_t* pin = pairs + ((int64_t)k * 2 + (inverse ? 1 : 0)) * pair_stride;
    const int32_t* pout = pairs + ((int64_t)k * 2 + (inverse ? 0 : 1)) * pair_stride;
    for (int p = 0; p < num[k]; ++p)
      if (pout[p] >= 0 && pout[p] < num_out) table[(int64_t)pout[p] * kv + k] = pin[p];
  }
  return PCS_OK;
}

// indiceConv / fusedIndiceConvBatchNorm on the table (spconv_ops.h:260-361, fused_spconv_ops.h:29-132; SparseSequential's
// conv + BatchNorm1d(eval) + ReLU, modules.py:125-187): per output row the centre offset first (subM), then the
// offsets ascending, every offset's product a fresh fp32 dot product over ascending input channels added to the row --
// the order of the reference's gather / GEMM / scatter-add loop; then (+bias), (*scale + shift), relu
static int conv_table_host(const float* features, int cin, const float* filters, int kv, int cout, int centre,
                           const int32_t* table, int64_t num_out, const float* bias, const float* scale,
                           const float* shift, const float* residual, int relu, float* out) {
  if (!features || !filters || !table || !out || cin < 1 || cout < 1 || kv < 1 || num_out < 0 || centre >= kv ||
      (scale && !shift))
    return PCS_ERR_ARG;
  for (int64_t o = 0; o < num_out; ++o) {
    float* row = out + o * cout;
    for (int co = 0; co < cout; ++co) row[co] = 0.f;
    bool first = true;
    for (int step = (centre >= 0 ? -1 : 0); step < kv; ++step) {
      const int k = step < 0 ? centre : step;
      if (step >= 0 && k == centre) continue;
      const int32_t i = table[o * kv + k];
      if (i < 0) continue;
      const float* f = features + (int64_t)i * cin;
      for (int co = 0; co < cout; ++co) {
        float acc = 0.f;
        for (int c = 0; c < cin; ++c) acc = acc + f[c] * filters[((int64_t)k * cin + c) * cout + co];
        row[co] = first ? acc : row[co] + acc;
      }
      first = false;
    }
    for (int co = 0; co < cout; ++co) {
      float x = row[co];
      if (bias) x = x + bias[co];
      if (scale) x = x * scale[co] + shift[co];
      if (residual) x = x + residual[o * cout + co];
      if (relu) x = x > 0.f ? x : 0.f;
      row[co] = x;
    }
  }
  return PCS_OK;
}
int pcs_cpu_indice_conv_table_f32(const float* features, int cin, const float* filters, int kv, int cout, int centre,
                                  const int32_t* table, int64_t num_out, const float* bias, const float* scale,
                                  const float* shift, int relu, float* out, pcs_stream_t) {
  return conv_table_host(features, cin, filters, kv, cout, centre, table, num_out, bias, scale, shift, nullptr, relu, out);
}
// SparseBasicBlock's tail (pcdet/models/blocks/spconv_blocks.py:47-64): relu(bn2(conv2(.)) + identity)
int pcs_cpu_indice_conv_table_residual_f32(const float* features, int cin, const float* filters, int kv, int cout,
                                           int centre, const int32_t* table, int64_t num_out, const float* bias,
                                           const float* scale, const float* shift, const float* residual, int relu,
                                           float* out, pcs_stream_t) {
  if (residual && residual == out) return PCS_ERR_ARG;
  return conv_table_host(features, cin, filters, kv, cout, centre, table, num_out, bias, scale, shift, residual, relu, out);
}

// scale / shift of an eval-mode BatchNorm1d (torch.nn.functional.batch_norm on running statistics, as after every
// conv of spconv_backbone.py:6-27): each product / sum rounded to fp32
int pcs_cpu_bn_fold_f32(const float* weight, const float* bias, const float* running_mean, const float* inv_std, int C,
                        float* scale, float* shift, pcs_stream_t) {
  if (!running_mean || !inv_std || !scale || !shift || C < 1) return PCS_ERR_ARG;
  for (int c = 0; c < C; ++c) {
    const float sc = weight ? inv_std[c] * weight[c] : inv_std[c];
    float sh = -running_mean[c] * sc;
    if (bias) sh = sh + bias[c];
    scale[c] = sc;
    shift[c] = sh;
  }
  return PCS_OK;
}

size_t pcs_cpu_indice_conv_backward_workspace_bytes(int64_t, int64_t, int, int, int) { return 0; }

// indiceConvBackward (spconv_ops.h:363-449) with the CPU functors (src/reordering.cc:20-49): per offset
// filters_grad[k] = gathered_in^T . gathered_out_grad, in_grad[i] += out_grad[o] . W[k]^T; subM's centre offset (the
// one with the most pairs) is the plain product over all rows. fp32, pairs in rulebook order.
int pcs_cpu_indice_conv_backward_f32(const float* features, int cin, const float* filters, int kv, int cout,
                                     const float* out_grad, const int32_t* pairs, int64_t pair_stride,
                                     const int32_t* num, int64_t num_in, int64_t num_out, int inverse, int subm,
                                     float* in_grad, float* filters_grad, void*, size_t, pcs_stream_t) {
  if (!features || !filters || !out_grad || !pairs || !num || !in_grad || !filters_grad || cin < 1 || cout < 1 ||
      kv < 1 || num_in < 0 || num_out < 0)
    return PCS_ERR_ARG;
  for (int64_t i = 0; i < num_in * cin; ++i) in_grad[i] = 0.f;
  for (int64_t i = 0; i < (int64_t)kv * cin * cout; ++i) filters_grad[i] = 0.f;
  const int centre = subm ? (int)(std::max_element(num, num + kv) - num) : -1;
  auto one = [&](int k, int64_t i, int64_t o) {
    const float* f = features + i * cin;
    const float* g = out_grad + o * cout;
    float* wg = filters_grad + (int64_t)k * cin * cout;
    const float* w = filters + (int64_t)k * cin * cout;
    for (int c = 0; c < cin; ++c) {
      float acc = 0.f;
      for (int co = 0; co < cout; ++co) {
        wg[c * cout + co] += f[c] * g[co];
        acc = acc + g[co] * w[c * cout + co];
      }
      in_grad[i * cin + c] += acc;
    }
  };
  if (subm)
    for (int64_t r = 0; r < std::min(num_in, num_out); ++r) one(centre, r, r);
  for (int k = 0; k < kv; ++k) {
    if (num[k] <= 0 || k == centre) continue;
    const int32_t* pin = pairs + ((int64_t)k * 2 + (inverse ? 1 : 0)) * pair_stride;
    const int32_t* pout = pairs + ((int64_t)k * 2 + (inverse ? 0 : 1)) * pair_stride;
    for (int p = 0; p < num[k]; ++p) one(k, pin[p], pout[p]);
  }
  return PCS_OK;
}

// indiceMaxPool (pool_ops.h:26-60, src/maxpool.cc:22-40): the output starts at ZERO and takes every paired input
// that is larger, offset by offset
int pcs_cpu_indice_maxpool_f32(const float* features, int C, const int32_t* pairs, int64_t pair_stride,
                               const int32_t* num, int kv, int64_t num_out, float* out, void*, size_t, pcs_stream_t) {
  if (!features || !pairs || !num || !out || C < 1 || kv < 1 || num_out < 0) return PCS_ERR_ARG;
  for (int64_t i = 0; i < num_out * C; ++i) out[i] = 0.f;
  for (int k = 0; k < kv; ++k) {
    const int32_t* pin = pairs + ((int64_t)k * 2) * pair_stride;
    const int32_t* pout = pin + pair_stride;
    for (int p = 0; p < num[k]; ++p) {
      const float* f = features + (int64_t)pin[p] * C;
      float* o = out + (int64_t)pout[p] * C;
      for (int c = 0; c < C; ++c)
        if (f[c] > o[c]) o[c] = f[c];
    }
  }
  return PCS_OK;
}

// indiceMaxPoolBackward (pool_ops.h:62-98, src/maxpool.cc:43-64): every paired input equal to the pooled value
// receives the output's gradient, offsets ascending
int pcs_cpu_indice_maxpool_backward_f32(const float* features, const float* out_features, const float* out_grad, int C,
                                        const int32_t* pairs, int64_t pair_stride, const int32_t* num, int kv,
                                        int64_t num_in, float* in_grad, void*, size_t, pcs_stream_t) {
  if (!features || !out_features || !out_grad || !pairs || !num || !in_grad || C < 1 || kv < 1 || num_in < 0)
    return PCS_ERR_ARG;
  for (int64_t i = 0; i < num_in * C; ++i) in_grad[i] = 0.f;
  for (int k = 0; k < kv; ++k) {
    const int32_t* pin = pairs + ((int64_t)k * 2) * pair_stride;
    const int32_t* pout = pin + pair_stride;
    for (int p = 0; p < num[k]; ++p) {
      const int64_t i = pin[p], o = pout[p];
      for (int c = 0; c < C; ++c)
        if (features[i * C + c] == out_features[o * C + c]) in_grad[i * C + c] += out_grad[o * C + c];
    }
  }
  return PCS_OK;
}

// SparseConvTensor.dense() (pcdet/ops/spconv/structure.py:5-18,55-64), channels first: out [B,C,D,H,W]
int pcs_cpu_sparse_to_dense(const float* features, const int32_t* indices, int64_t V, int C, int B, int D, int H, int W,
                            float* out, int* status, pcs_stream_t) {
  if (V < 0 || C < 1 || B < 1 || D < 1 || H < 1 || W < 1 || !out) return PCS_ERR_ARG;
  if (V > 0 && (!features || !indices)) return PCS_ERR_ARG;
  std::memset(out, 0, sizeof(float) * (size_t)B * C * D * H * W);
  if (status) *status = 0;
  for (int64_t v = 0; v < V; ++v) {
    const int32_t* q = indices + v * 4;
    if (q[0] < 0 || q[0] >= B || q[1] < 0 || q[1] >= D || q[2] < 0 || q[2] >= H || q[3] < 0 || q[3] >= W) {
      if (status) *status = 1;
      continue;
    }
    for (int c = 0; c < C; ++c) out[((((int64_t)q[0] * C + c) * D + q[1]) * H + q[2]) * W + q[3]] = features[v * C + c];
  }
  return PCS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// evaluation: points_in_boxes_cpu (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-168) frame by frame, as the
// loops of cluster_proposal.py:200-215 and cluster_tracking.py:388-400 call it

int pcs_cpu_points_in_frame_boxes(const float* points, const int64_t* point_frame, int64_t N, const float* boxes,
                                  const int32_t* box_offsets, int F, int Bmax, int32_t* member, int32_t* first,
                                  pcs_stream_t) {
  if (N < 0 || F < 0 || Bmax < 0 || (N > 0 && (!points || !point_frame || !box_offsets))) return PCS_ERR_ARG;
  for (int64_t j = 0; j < N; ++j) {
    const int64_t f = point_frame[j];
    int b0 = 0, nb = 0;
    if (f >= 0 && f < F) { b0 = box_offsets[f]; nb = box_offsets[f + 1] - b0; }
    int fst = -1;
    for (int bl = 0; bl < Bmax; ++bl) {
      int32_t in = 0;
      if (bl < nb) orc_points_in_boxes(points + j * 3, 1, boxes + (int64_t)(b0 + bl) * 7, 1, &in);
      if (member) member[j * Bmax + bl] = in;
      if (in && fst < 0) fst = bl;
    }
    if (first) first[j] = fst;
  }
  return PCS_OK;
}

// the same test summed per (group, box): the [G, Bmax] table of component_box_iou_frames (cluster_tracking.py:388-407)
int pcs_cpu_frame_box_counts(const float* points, const int64_t* point_frame, int64_t N, const float* boxes,
                             const int32_t* box_offsets, int F, int Bmax, const int64_t* group, int64_t G,
                             int32_t* counts, pcs_stream_t) {
  if (N < 0 || F < 0 || Bmax < 0 || G < 0 || (N > 0 && (!points || !point_frame || !box_offsets || !group || !counts)))
    return PCS_ERR_ARG;
  for (int64_t j = 0; j < N; ++j) {
    const int64_t f = point_frame[j], g = group[j];
    if (f < 0 || f >= F || g < 0 || g >= G) continue;
    const int b0 = box_offsets[f], nb = box_offsets[f + 1] - b0;
    for (int bl = 0; bl < nb && bl < Bmax; ++bl) {
      int32_t in = 0;
      orc_points_in_boxes(points + j * 3, 1, boxes + (int64_t)(b0 + bl) * 7, 1, &in);
      counts[g * Bmax + bl] += in;
    }
  }
  return PCS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// loader: WaymoDataset.get_lidar / __getitem__ (pcdet/datasets/waymo/waymo_dataset.py:334-343,383-391,584-630) + the
// point_xyz branch of collate_batch (pcdet/datasets/dataset.py:222-232) for all sweeps of a sample
int pcs_cpu_assemble_sweeps(const float* raw, const int32_t* seg, const int64_t* sweep_offsets, const double* sweep_T,
                            const int32_t* sweep_ids, int num_sweeps, int64_t N, int batch_index, float* point_bxyz,
                            float* point_feat, int32_t* point_sweep, int32_t* segmentation_label,
                            int32_t* instance_label, pcs_stream_t) {
  if (N < 0 || num_sweeps < 1 || num_sweeps > 1024) return PCS_ERR_ARG;
  if (N == 0) return PCS_OK;
  if (!raw || !sweep_offsets || !sweep_T || !sweep_ids || !point_bxyz || !point_feat || !point_sweep) return PCS_ERR_ARG;
  if (seg && (!segmentation_label || !instance_label)) return PCS_ERR_ARG;
  for (int s = 0; s < num_sweeps; ++s) {
    const double* t = sweep_T + (size_t)s * 12;
    for (int64_t i = sweep_offsets[s]; i < sweep_offsets[s + 1] && i < N; ++i) {
      const float* r = raw + i * 8;
      const double x = r[0], y = r[1], z = r[2];
      // numpy: (xyz f32 -> f64) @ R^T accumulated k = 0,1,2 (the BLAS's fused multiply-adds) + t, stored as f32
      point_bxyz[i * 4 + 0] = (float)batch_index;
      point_bxyz[i * 4 + 1] = (float)(std::fma(z, t[2], std::fma(y, t[1], x * t[0])) + t[3]);
      point_bxyz[i * 4 + 2] = (float)(std::fma(z, t[6], std::fma(y, t[5], x * t[4])) + t[7]);
      point_bxyz[i * 4 + 3] = (float)(std::fma(z, t[10], std::fma(y, t[9], x * t[8])) + t[11]);
      point_feat[i * 3 + 0] = (float)std::tanh((double)r[3]);
      point_feat[i * 3 + 1] = r[4];
      point_feat[i * 3 + 2] = r[5] / 75.f;
      point_sweep[i] = sweep_ids[s];
      if (seg) {
        instance_label[i] = seg[i * 2];
        segmentation_label[i] = seg[i * 2 + 1];
      }
    }
  }
  return PCS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// smooth_velo (cluster_tracking.py:162-199): AdamW(lr 1e-2, betas (.9, .999), eps 1e-8, weight decay 1e-2) with
// MultiStepLR([100, 200, 300]) on the whole [C,F,3] Parameter -- the gradient reaches the xy columns of the frame range
// only, the decoupled weight decay every element --, loss = mean((v - d)^2) * weight0 + mean(|v_t - v_t+1|) * weight,
// stop after three consecutive steps that gain less than `stopping`. fp32 state, the loss summed in double.
size_t pcs_cpu_smooth_velo_workspace_bytes(int64_t, int) { return 0; }
int pcs_cpu_smooth_velo(float* comp_velos, const float* comp_center_diffs, int64_t C, int F, int frame_lo, int frame_hi,
                        float weight0, float weight, int num_itr, float stopping, void*, size_t, pcs_stream_t) {
  if (!comp_velos || !comp_center_diffs || C < 0 || F < 1 || frame_lo < 0 || frame_hi >= F || frame_lo > frame_hi)
    return PCS_ERR_ARG;
  if (C == 0 || frame_lo == frame_hi) return PCS_OK;
  const int a = frame_lo, nf = frame_hi - frame_lo + 1;
  const float inv_fit = 1.0f / (float)(C * nf * 2), inv_sm = 1.0f / (float)(C * (nf - 1) * 2);
  const size_t n = (size_t)C * nf * 2;
  std::vector<float> m(n, 0.f), s(n, 0.f), g(n), sgn(n);
  auto X = [&](int64_t c, int t, int k) -> float& { return comp_velos[(c * F + a + t) * 3 + k]; };
  auto Dd = [&](int64_t c, int t, int k) { return comp_center_diffs[(c * F + a + t) * 3 + k]; };
  float last_loss = 1e10f;
  int countdown = 3;
  for (int it = 0; it < num_itr; ++it) {
    const double lr = 1e-2 * std::pow(0.1, (it >= 100) + (it >= 200) + (it >= 300));
    double fit_sum = 0.0, sm_sum = 0.0;
    for (int64_t c = 0; c < C; ++c)
      for (int t = 0; t < nf; ++t)
        for (int k = 0; k < 2; ++k) {
          const float fit = X(c, t, k) - Dd(c, t, k);
          fit_sum += (double)(fit * fit);
          g[((size_t)c * nf + t) * 2 + k] = fit * 2.f * (inv_fit * weight0);
        }
    for (int64_t c = 0; c < C; ++c)
      for (int t = 0; t + 1 < nf; ++t)
        for (int k = 0; k < 2; ++k) {
          const float sm = X(c, t, k) - X(c, t + 1, k);
          sm_sum += (double)std::fabs(sm);
          sgn[((size_t)c * nf + t) * 2 + k] = (float)((sm > 0.f) - (sm < 0.f)) * (inv_sm * weight);
        }
    // (autograd adds the two branches of |v_t - v_t+1| in this order: every +, then every -)
    for (int64_t c = 0; c < C; ++c)
      for (int t = 0; t + 1 < nf; ++t)
        for (int k = 0; k < 2; ++k) g[((size_t)c * nf + t) * 2 + k] += sgn[((size_t)c * nf + t) * 2 + k];
    for (int64_t c = 0; c < C; ++c)
      for (int t = 0; t + 1 < nf; ++t)
        for (int k = 0; k < 2; ++k) g[((size_t)c * nf + t + 1) * 2 + k] -= sgn[((size_t)c * nf + t) * 2 + k];
    const float loss = (float)(fit_sum * inv_fit * weight0 + sm_sum * inv_sm * weight);
    const double bc1 = 1.0 - std::pow(0.9, it + 1), bc2 = 1.0 - std::pow(0.999, it + 1);
    const float step = (float)(lr / bc1), bc2s = (float)std::sqrt(bc2), decay = (float)(1.0 - lr * 1e-2);
    for (int64_t i = 0; i < C * F * 3; ++i) {
      const int64_t c = i / (F * 3);
      const int t = (int)((i / 3) % F) - a, k = (int)(i % 3);
      if (t >= 0 && t < nf && k < 2) {
        const size_t j = ((size_t)c * nf + t) * 2 + k;
        m[j] = m[j] + 0.1f * (g[j] - m[j]);
        s[j] = s[j] * 0.999f + 0.001f * g[j] * g[j];
        const float denom = std::sqrt(s[j]) / bc2s + 1e-8f;
        comp_velos[i] = comp_velos[i] * decay - step * (m[j] / denom);
      } else {
        comp_velos[i] *= decay;
      }
    }
    if (last_loss - loss < stopping) --countdown; else countdown = 3;
    if (countdown <= 0) break;
    last_loss = loss;
  }
  return PCS_OK;
}

}  // extern "C"
