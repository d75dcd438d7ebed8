// Connected components, per-segment medians and min/max for gfx950.
//
// Reference behaviour restated from:
//   pcdet/models/model_utils/graph_utils.py:40-53  (connected_components: edges -> CPU ->
//       scipy.sparse.csgraph.connected_components -> back; label = rank of the component's
//       smallest node id)
//   pcdet/models/registration/preprocessors/registration_utils.py:60-81 (robust_median: per-index
//       median of integers by sorting index*range + data, element offset + deg//2)
//   torch_scatter.scatter(reduce='max'|'min') call sites (cluster_tracking.py:104,325-328)
#include "pcs_common.hpp"

namespace pcs {

// ------------------------------------------------------------------ connected components
__global__ void __launch_bounds__(kBlock) cc_init_kernel(int32_t* __restrict__ parent, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    parent[i] = (int32_t)i;
}

__device__ __forceinline__ int cc_find(const int32_t* parent, int x) {
  // parent pointers only ever decrease, so a stale (cached) read is an older, still valid ancestor
  int p = parent[x];
  while (p != x) {
    x = p;
    p = parent[x];
  }
  return x;
}

// Lock-free union, larger root hooked under the smaller one: the final root of every set is its
// smallest node id, independent of edge order.
__global__ void __launch_bounds__(kBlock)
cc_hook_kernel(const int64_t* __restrict__ e0, const int64_t* __restrict__ e1, int64_t E,
               int64_t n, int32_t* __restrict__ parent) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t u = e0[e], v = e1[e];
    if (u < 0 || v < 0 || u >= n || v >= n || u == v) continue;
    int a = cc_find(parent, (int)u), b = cc_find(parent, (int)v);
    while (a != b) {
      if (a < b) { int t = a; a = b; b = t; }
      int old = atomicCAS(&parent[a], a, b);
      if (old == a) break;
      a = cc_find(parent, old);
      b = cc_find(parent, b);
    }
  }
}

__global__ void __launch_bounds__(kBlock)
cc_root_kernel(int32_t* __restrict__ parent, int64_t n, int32_t* __restrict__ is_root) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int r = cc_find(parent, (int)i);
    is_root[i] = (r == (int)i);
  }
}

// second pass after the kernel boundary (so every CU sees final parents): root -> rank
__global__ void __launch_bounds__(kBlock)
cc_label_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ rank, int64_t n,
                const int32_t* __restrict__ is_root, int64_t* __restrict__ label,
                int64_t* __restrict__ d_num) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int r = cc_find(parent, (int)i);
    label[i] = rank[r];
    if (i == n - 1) *d_num = (int64_t)rank[i] + is_root[i];
  }
}

// ---- the same over a neighbour TABLE (the k-nearest search's output) instead of an edge list, for several
// distance cuts at once: ClusterProposal takes the components of the 32-NN graph at three radii
// (cluster_proposal.py:57-83 of the reference), and the graph of a smaller radius is the larger one's cut at
// dist2 <= r^2. Cut c has its own parent array [c * n, (c + 1) * n); one pass over the table hooks all of them.
struct CcCuts {
  float r2[8];
  int n;
};
__device__ __forceinline__ void cc_union(int32_t* __restrict__ parent, int u, int v) {
  int a = cc_find(parent, u), b = cc_find(parent, v);
  while (a != b) {
    if (a < b) { int t = a; a = b; b = t; }
    int old = atomicCAS(&parent[a], a, b);
    if (old == a) break;
    a = cc_find(parent, old);
    b = cc_find(parent, b);
  }
}
__global__ void __launch_bounds__(kBlock)
cc_init_multi_kernel(int32_t* __restrict__ parent, int64_t n, int n_cut) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * n_cut;
       i += (int64_t)gridDim.x * blockDim.x)
    parent[i] = (int32_t)(i % n);
}
__global__ void __launch_bounds__(kBlock)
cc_hook_table_kernel(const int32_t* __restrict__ nbr_idx, const float* __restrict__ nbr_d,
                     const int32_t* __restrict__ deg, int64_t n, int K, CcCuts cuts, int32_t* __restrict__ parent) {
  const int64_t total = n * K;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(t / K), e = (int)(t - (int64_t)q * K);
    if (e >= deg[q]) continue;
    const int v = nbr_idx[t];
    if (v < 0 || v >= n || v == q) continue;
    const float d = nbr_d[t];
    for (int c = 0; c < cuts.n; ++c)
      if (d <= cuts.r2[c]) cc_union(parent + (int64_t)c * n, q, v);
  }
}
__global__ void __launch_bounds__(kBlock)
cc_root_multi_kernel(const int32_t* __restrict__ parent, int64_t n, int n_cut, int32_t* __restrict__ is_root) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * n_cut;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i / n;
    const int x = (int)(i - c * n);
    is_root[i] = cc_find(parent + c * n, x) == x;
  }
}
__global__ void __launch_bounds__(kBlock)
cc_label_multi_kernel(const int32_t* __restrict__ parent, const int32_t* __restrict__ rank, int64_t n, int n_cut,
                      const int32_t* __restrict__ is_root, int64_t* __restrict__ label, int64_t* __restrict__ d_num) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * n_cut;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i / n;
    const int x = (int)(i - c * n);
    const int r = cc_find(parent + c * n, x);
    const int base = rank[c * n];  // (the scan runs over all cuts: ranks restart at every cut)
    label[i] = rank[c * n + r] - base;
    if (x == n - 1) d_num[c] = (int64_t)(rank[i] - base) + is_root[i];
  }
}

size_t cc_table_workspace_bytes(int64_t n, int n_cut) {
  if (n < 1) n = 1;
  if (n_cut < 1) n_cut = 1;
  Arena a(nullptr, 0);
  a.take<int32_t>(n * n_cut);
  a.take<int32_t>(n * n_cut);
  a.take<int32_t>(n * n_cut);
  a.take<char>(scan_i32_temp_bytes(n * n_cut));
  return a.off;
}

int cc_table_components(const int32_t* nbr_idx, const float* nbr_d, const int32_t* deg, int64_t n, int K,
                        const float* r2cut, int n_cut, int64_t* component, int64_t* d_num, void* ws,
                        size_t ws_bytes, hipStream_t s) {
  Arena a(ws, ws_bytes);
  const int64_t m = n * n_cut;
  int32_t* parent = a.take<int32_t>(m);
  int32_t* is_root = a.take<int32_t>(m);
  int32_t* rank = a.take<int32_t>(m);
  const size_t tb = scan_i32_temp_bytes(m);
  void* tmp = a.take<char>(tb);
  if (!a.ok()) return PCS_ERR_WORKSPACE;
  CcCuts cuts;
  cuts.n = n_cut;
  for (int i = 0; i < 8; ++i) cuts.r2[i] = i < n_cut ? r2cut[i] : -1.f;
  cc_init_multi_kernel<<<grid_for(m), kBlock, 0, s>>>(parent, n, n_cut);
  cc_hook_table_kernel<<<grid_for(n * K), kBlock, 0, s>>>(nbr_idx, nbr_d, deg, n, K, cuts, parent);
  cc_root_multi_kernel<<<grid_for(m), kBlock, 0, s>>>(parent, n, n_cut, is_root);
  int rc = exclusive_scan_i32(tmp, tb, is_root, rank, m, s);
  if (rc) return rc;
  cc_label_multi_kernel<<<grid_for(m), kBlock, 0, s>>>(parent, rank, n, n_cut, is_root, component, d_num);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// ------------------------------------------------------------------ per-segment median (ints)
// k-th smallest with k = deg // 2 (upper median), O(deg^2) per segment: segments are voxels.
__global__ void __launch_bounds__(kBlock)
segment_median_kernel(const int64_t* __restrict__ data, const int32_t* __restrict__ order,
                      const int32_t* __restrict__ seg_start, int64_t v,
                      int64_t* __restrict__ out) {
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < v;
       s += (int64_t)gridDim.x * blockDim.x) {
    int b = seg_start[s], e = seg_start[s + 1];
    int k = (e - b) / 2;
    int64_t first = data[order[b]];
    // fast path: all equal (one component / one frame per voxel is the common case)
    bool same = true;
    for (int p = b + 1; p < e; ++p) if (data[order[p]] != first) { same = false; break; }
    int64_t res = first;
    if (!same) {
      for (int p = b; p < e; ++p) {
        int64_t x = data[order[p]];
        int lt = 0, eq = 0;
        for (int q = b; q < e; ++q) {
          int64_t y = data[order[q]];
          lt += (y < x);
          eq += (y == x);
        }
        if (lt <= k && k < lt + eq) { res = x; break; }
      }
    }
    out[s] = res;
  }
}

// ------------------------------------------------------------------ per-segment max / min (f32)
// one wave per (segment, column): lane-strided scan + butterfly (min / max are order-free)
__global__ void __launch_bounds__(kBlock)
segment_minmax_kernel(const float* __restrict__ data, int d, const int32_t* __restrict__ order,
                      const int32_t* __restrict__ seg_start, int64_t v, int is_max,
                      float* __restrict__ out) {
  int lane = threadIdx.x & 63;
  int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int64_t total = v * d;
  for (int64_t t = wave; t < total; t += nwaves) {
    int64_t seg = t / d;
    int c = (int)(t - seg * d);
    int b = seg_start[seg], e = seg_start[seg + 1];
    float acc = is_max ? -__builtin_inff() : __builtin_inff();
    for (int p = b + lane; p < e; p += 64) {
      float x = data[(int64_t)order[p] * d + c];
      acc = is_max ? fmaxf(acc, x) : fminf(acc, x);
    }
    acc = is_max ? wave_max(acc) : wave_min(acc);
    if (lane == 0) out[t] = acc;
  }
}

}  // namespace pcs

using namespace pcs;

extern "C" {

size_t pcs_connected_components_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  Arena a(nullptr, 0);
  a.take<int32_t>(n);
  a.take<int32_t>(n);
  a.take<int32_t>(n);
  a.take<char>(scan_i32_temp_bytes(n));
  return a.off;
}

// component[n] int64 = rank of the node's component by smallest member id; *d_num = count.
int pcs_connected_components(const int64_t* e0, const int64_t* e1, int64_t E, int64_t n,
                             int64_t* component, int64_t* d_num, void* ws, size_t ws_bytes,
                             hipStream_t s) {
  if (!d_num || !ws || n < 0 || E < 0 || (E > 0 && (!e0 || !e1)) || (n > 0 && !component))
    return PCS_ERR_ARG;
  if (n >= (1LL << 31)) return PCS_ERR_RANGE;
  if (n == 0) { PCS_CHECK_HIP(hipMemsetAsync(d_num, 0, sizeof(int64_t), s)); return PCS_OK; }
  Arena a(ws, ws_bytes);
  int32_t* parent = a.take<int32_t>(n);
  int32_t* is_root = a.take<int32_t>(n);
  int32_t* rank = a.take<int32_t>(n);
  size_t tb = scan_i32_temp_bytes(n);
  void* tmp = a.take<char>(tb);
  if (!a.ok()) return PCS_ERR_WORKSPACE;
  cc_init_kernel<<<grid_for(n), kBlock, 0, s>>>(parent, n);
  if (E > 0) cc_hook_kernel<<<grid_for(E), kBlock, 0, s>>>(e0, e1, E, n, parent);
  cc_root_kernel<<<grid_for(n), kBlock, 0, s>>>(parent, n, is_root);
  int rc = exclusive_scan_i32(tmp, tb, is_root, rank, n, s);
  if (rc) return rc;
  cc_label_kernel<<<grid_for(n), kBlock, 0, s>>>(parent, rank, n, is_root, component, d_num);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

int pcs_segment_median_i64(const int64_t* data, const int32_t* order, const int32_t* seg_start,
                           int64_t v, int64_t* out, hipStream_t s) {
  if (v < 0 || (v > 0 && (!data || !order || !seg_start || !out))) return PCS_ERR_ARG;
  if (v == 0) return PCS_OK;
  segment_median_kernel<<<grid_for(v), kBlock, 0, s>>>(data, order, seg_start, v, out);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

int pcs_segment_minmax_f32(const float* data, int d, const int32_t* order, const int32_t* seg_start,
                           int64_t v, int is_max, float* out, hipStream_t s) {
  if (v < 0 || d < 1 || (v > 0 && (!data || !order || !seg_start || !out))) return PCS_ERR_ARG;
  if (v == 0) return PCS_OK;
  {
    int64_t blocks = (v * d * 64 + kBlock - 1) / kBlock;
    segment_minmax_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), kBlock, 0, s>>>(data, d, order, seg_start, v,
                                                                                is_max, out);
  }
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

}  // extern "C"
