// pcs_oracle.cpp -- CPU restatement of the reference algorithms on the hot path.
//
// TEST INFRASTRUCTURE ONLY. Nothing under pcseqlearning_amd/ may import, link or call this file;
// it exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check and
// time the HIP path against a serial implementation of what the reference computes.
//
// Pinning (see DESIGN.md "Oracle"):
//  - rulebooks are checked against the reference's own CPU builders compiled from
//    /root/reference (oracle/_ref/libref_spconv.so, built by oracle/Makefile) and against golden
//    vectors generated from them (tests/golden/rulebook_*.npz);
//  - the hard voxelizer against the reference's point2voxel.h compiled the same way;
//  - the radius graph against brute force / sklearn in the reference's own protocol
//    (pcdet/ops/torch_hash/torch_hash_utils.py:365-431) and its hand case
//    (torch_hash_modules.py:146-151);
//  - connected components against scipy.sparse.csgraph (the reference's call, graph_utils.py:52);
//  - registration maths against the imported reference Python (tests/golden/reg_*.npz).
//  - grid_cluster / torch_scatter orderings restate upstream documentation only:
//    "parity unpinned" for those two third-party leaves.
//
// Plain serial C++ (no torch, no OpenMP), compiled with -ffp-contract=off so fp32 expressions
// round exactly as written.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <unordered_map>
#include <vector>

extern "C" {

// ---------------------------------------------------------------------------------------------
// GridSampling3D ids: pcdet/models/model_utils/grid_sampling.py:32-37 + torch_cluster.grid_cluster
// (upstream semantics: c_d = (int64)((p_d - start_d) / size_d), strides from
// (int64)((end_d - start_d) / size_d) + 1, dim 0 fastest).
void orc_grid_cluster_ids(const float* pts, int64_t n, const float* size4, float margin0,
                          int64_t* ids, float* bounds8_out) {
  float st[4], en[4];
  for (int d = 0; d < 4; ++d) { st[d] = INFINITY; en[d] = -INFINITY; }
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < 4; ++d) {
      st[d] = std::fmin(st[d], pts[i * 4 + d]);
      en[d] = std::fmax(en[d], pts[i * 4 + d]);
    }
  if (bounds8_out) for (int d = 0; d < 4; ++d) { bounds8_out[d] = st[d]; bounds8_out[4 + d] = en[d]; }
  st[0] = st[0] - margin0;
  en[0] = en[0] + margin0;
  int64_t stride[4];
  stride[0] = 1;
  for (int d = 0; d < 3; ++d) stride[d + 1] = stride[d] * ((int64_t)((en[d] - st[d]) / size4[d]) + 1);
  for (int64_t i = 0; i < n; ++i) {
    int64_t id = 0;
    for (int d = 0; d < 4; ++d) id += (int64_t)((pts[i * 4 + d] - st[d]) / size4[d]) * stride[d];
    ids[i] = id;
  }
}

// torch.unique(sorted=True, return_inverse=True) (grid_sampling.py:38) + segment structure.
// Returns the number of unique ids.
int64_t orc_unique_i64(const int64_t* ids, int64_t n, int64_t* uniq, int64_t* inv, int32_t* order,
                       int32_t* seg_start) {
  std::vector<int32_t> ord(n);
  std::iota(ord.begin(), ord.end(), 0);
  std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return ids[a] < ids[b]; });
  int64_t v = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (i == 0 || ids[ord[i]] != ids[ord[i - 1]]) {
      if (uniq) uniq[v] = ids[ord[i]];
      seg_start[v] = (int32_t)i;
      ++v;
    }
    if (inv) inv[ord[i]] = v - 1;
    if (order) order[i] = ord[i];
  }
  seg_start[v] = (int32_t)n;
  return v;
}

// scatter(data, inv, reduce='mean'|'sum') in ascending original index per segment.
void orc_segment_reduce_f32(const float* data, int d, const int32_t* order,
                            const int32_t* seg_start, int64_t v, int mean, float* out) {
  for (int64_t s = 0; s < v; ++s)
    for (int c = 0; c < d; ++c) {
      float acc = 0.f;
      for (int p = seg_start[s]; p < seg_start[s + 1]; ++p) acc = acc + data[(int64_t)order[p] * d + c];
      if (mean) acc = acc / (float)(seg_start[s + 1] - seg_start[s]);
      out[s * d + c] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// DynamicMeanVFE.forward, pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:51-76.
// Returns the number of voxels; feat [*, ncol-1], coords [*,4] (b,z,y,x), inv[n] (-1 dropped).
int64_t orc_dynamic_voxelize(const float* pts, int64_t n, int ncol, const float* range6,
                             const float* voxel3, const int32_t* grid3, float* feat,
                             int32_t* coords, int64_t* inv) {
  int64_t gx = grid3[0], gy = grid3[1], gz = grid3[2];
  std::vector<int64_t> key(n);
  std::vector<int32_t> keep;
  for (int64_t i = 0; i < n; ++i) {
    const float* p = pts + i * ncol;
    int c[3];
    bool ok = true;
    for (int d = 0; d < 3; ++d) {
      c[d] = (int)std::floor((p[1 + d] - range6[d]) / voxel3[d]);
      if (c[d] < 0 || c[d] >= grid3[d]) ok = false;
    }
    key[i] = -1;
    if (ok) {
      key[i] = (int64_t)(int)p[0] * gx * gy * gz + (int64_t)c[0] * gy * gz + (int64_t)c[1] * gz + c[2];
      keep.push_back((int32_t)i);
    }
    if (inv) inv[i] = -1;
  }
  std::stable_sort(keep.begin(), keep.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
  int64_t v = 0;
  size_t i = 0;
  int d = ncol - 1;
  while (i < keep.size()) {
    size_t j = i;
    while (j < keep.size() && key[keep[j]] == key[keep[i]]) ++j;
    for (int c = 0; c < d; ++c) {
      float acc = 0.f;
      for (size_t q = i; q < j; ++q) acc = acc + pts[(int64_t)keep[q] * ncol + 1 + c];
      feat[v * d + c] = acc / (float)(j - i);
    }
    int64_t k = key[keep[i]];
    int64_t xyz = gx * gy * gz, yz = gy * gz;
    coords[v * 4 + 0] = (int32_t)(k / xyz);
    coords[v * 4 + 3] = (int32_t)((k % xyz) / yz);
    coords[v * 4 + 2] = (int32_t)((k % yz) / gz);
    coords[v * 4 + 1] = (int32_t)(k % gz);
    if (inv) for (size_t q = i; q < j; ++q) inv[keep[q]] = v;
    ++v;
    i = j;
  }
  return v;
}

// points_to_voxel_3d_np, pcdet/ops/spconv/include/spconv/point2voxel.h:33-92 (serial scan,
// first-come voxel ids, max_points per voxel, max_voxels). The dense coor_to_voxelidx grid of the
// reference is a hash map here. voxels/coords/num must be zero-initialised by the caller.
int64_t orc_hard_voxelize(const float* pts, int64_t n, int ncol, const float* range6,
                          const float* voxel3, int max_points, int max_voxels, float* voxels,
                          int32_t* coords, int32_t* num) {
  int grid[3];
  for (int d = 0; d < 3; ++d) grid[d] = (int)std::round((range6[3 + d] - range6[d]) / voxel3[d]);
  std::unordered_map<int64_t, int> cell2vox;
  int voxel_num = 0;
  for (int64_t i = 0; i < n; ++i) {
    int coor[3];
    bool failed = false;
    for (int j = 0; j < 3; ++j) {
      int c = (int)std::floor((pts[i * ncol + j] - range6[j]) / voxel3[j]);
      if (c < 0 || c >= grid[j]) { failed = true; break; }
      coor[2 - j] = c;
    }
    if (failed) continue;
    int64_t key = ((int64_t)coor[0] * grid[1] + coor[1]) * grid[0] + coor[2];
    auto it = cell2vox.find(key);
    int voxelidx;
    if (it == cell2vox.end()) {
      voxelidx = voxel_num;
      if (voxel_num >= max_voxels) continue;
      voxel_num += 1;
      cell2vox.emplace(key, voxelidx);
      for (int k = 0; k < 3; ++k) coords[voxelidx * 3 + k] = coor[k];
    } else {
      voxelidx = it->second;
    }
    int cnt = num[voxelidx];
    if (cnt < max_points) {
      for (int k = 0; k < ncol; ++k) voxels[((int64_t)voxelidx * max_points + cnt) * ncol + k] = pts[i * ncol + k];
      num[voxelidx] += 1;
    }
  }
  return voxel_num;
}

// ---------------------------------------------------------------------------------------------
// Rulebook. getValidOutPos (geometry.h:25-85) restated: for one input position enumerate output
// positions from `uppers` downward, last dim fastest; offset = kernel offset id.
static int valid_out_pos(const int* in, const int* ks, const int* st, const int* pd, const int* dl,
                         const int* oshape, int* out /* [kv][4] */) {
  int lowers[3], uppers[3], counter[3], csize[3];
  int npts = 1, cnt = 0;
  for (int i = 0; i < 3; ++i) {
    lowers[i] = (in[i] - (ks[i] - 1) * dl[i] - 1 + st[i] + pd[i]) / st[i];
    uppers[i] = (in[i] + pd[i]) / st[i];
  }
  for (int i = 0; i < 3; ++i) {
    csize[i] = (uppers[i] - lowers[i]) / dl[i] + 1;
    npts *= csize[i];
    counter[i] = 0;
  }
  for (int i = 0; i < npts; ++i) {
    bool valid = true;
    int m = 1, offset = 0;
    for (int j = 2; j >= 0; --j) {
      int val = uppers[j] - counter[j] * dl[j];
      out[cnt * 4 + j] = val;
      if (val < 0 || val > oshape[j] - 1) valid = false;
      offset += m * (in[j] - val * st[j] + pd[j]) / dl[j];
      m *= ks[j];
    }
    out[cnt * 4 + 3] = offset;
    if (valid) ++cnt;
    counter[2] += 1;
    for (int c = 2; c >= 0; --c)
      if (counter[c] == csize[c] && c > 0) { counter[c - 1] += 1; counter[c] = 0; }
  }
  return cnt;
}

// getValidOutPosTranspose (geometry.h:87-142) restated: the kernel footprint of one input position of a
// transposed conv, enumerated from `uppers` downward, last dim fastest; offset = kernel offset id.
static int valid_out_pos_transpose(const int* in, const int* ks, const int* st, const int* pd, const int* dl,
                                   const int* oshape, int* out /* [kv][4] */) {
  int lowers[3], uppers[3], counter[3], csize[3];
  int npts = 1, cnt = 0;
  for (int i = 0; i < 3; ++i) {
    lowers[i] = in[i] * st[i] - pd[i];
    uppers[i] = lowers[i] + (ks[i] - 1) * dl[i];
  }
  for (int i = 0; i < 3; ++i) {
    csize[i] = (uppers[i] - lowers[i]) / dl[i] + 1;
    npts *= csize[i];
    counter[i] = 0;
  }
  for (int i = 0; i < npts; ++i) {
    bool valid = true;
    int m = 1, offset = 0;
    for (int j = 2; j >= 0; --j) {
      int val = uppers[j] - counter[j] * dl[j];
      out[cnt * 4 + j] = val;
      if (val < 0 || val > oshape[j] - 1) valid = false;
      offset += m * (val - lowers[j]) / dl[j];
      m *= ks[j];
    }
    out[cnt * 4 + 3] = offset;
    if (valid) ++cnt;
    counter[2] += 1;
    for (int c = 2; c >= 0; --c)
      if (counter[c] == csize[c] && c > 0) { counter[c - 1] += 1; counter[c] = 0; }
  }
  return cnt;
}

// getIndicePairsSubM (geometry.h:248-297) / getIndicePairsConv (geometry.h:145-194) with the
// parameter handling of getIndicePair (spconv_ops.h:73-85). pairs [KV,2,V] pre-filled with -1,
// num [KV] zeroed here. Returns the number of output voxels; out_ids [*,4] for conv. subm: 0 strided conv,
// 1 submanifold, 2 transposed conv (getIndicePairsDeConv, geometry.h:197-245).
int64_t orc_get_indice_pairs(const int32_t* indices, int64_t V, const int* out_shape3,
                             const int* ksize3, const int* stride3, const int* pad3,
                             const int* dil3, int subm, int32_t* out_ids, int32_t* pairs,
                             int32_t* num) {
  int ks[3], st[3], pd[3], dl[3];
  int kv = 1;
  for (int d = 0; d < 3; ++d) {
    ks[d] = ksize3[d];
    dl[d] = dil3[d];
    st[d] = subm == 1 ? 1 : stride3[d];
    pd[d] = subm == 1 ? ksize3[d] / 2 : pad3[d];
    kv *= ks[d];
  }
  for (int64_t i = 0; i < (int64_t)kv * 2 * V; ++i) pairs[i] = -1;
  for (int k = 0; k < kv; ++k) num[k] = 0;
  int64_t vol = (int64_t)out_shape3[0] * out_shape3[1] * out_shape3[2];
  std::unordered_map<int64_t, int32_t> grid;
  grid.reserve((size_t)V * 2);
  std::vector<int> vp((size_t)kv * 4);
  auto flat = [&](const int* p, int b) {
    return ((int64_t)p[0] * out_shape3[1] + p[1]) * out_shape3[2] + p[2] + vol * b;
  };
  if (subm == 1) {
    for (int64_t j = 0; j < V; ++j) grid[flat(indices + j * 4 + 1, indices[j * 4])] = (int32_t)j;
    for (int64_t j = 0; j < V; ++j) {
      int cnt = valid_out_pos(indices + j * 4 + 1, ks, st, pd, dl, out_shape3, vp.data());
      for (int i = 0; i < cnt; ++i) {
        const int* p = vp.data() + i * 4;
        auto it = grid.find(flat(p, indices[j * 4]));
        if (it != grid.end()) {
          int off = p[3];
          pairs[((int64_t)off * 2 + 0) * V + num[off]] = (int32_t)j;
          pairs[((int64_t)off * 2 + 1) * V + num[off]++] = it->second;
        }
      }
    }
    return V;
  }
  const bool transpose = subm == 2;  // getIndicePairsDeConv (geometry.h:197-245)
  int32_t num_act = 0;
  for (int64_t j = 0; j < V; ++j) {
    int b = indices[j * 4];
    int cnt = transpose ? valid_out_pos_transpose(indices + j * 4 + 1, ks, st, pd, dl, out_shape3, vp.data())
                        : valid_out_pos(indices + j * 4 + 1, ks, st, pd, dl, out_shape3, vp.data());
    for (int i = 0; i < cnt; ++i) {
      const int* p = vp.data() + i * 4;
      int off = p[3];
      int64_t idx = flat(p, b);
      auto it = grid.find(idx);
      if (it == grid.end()) {
        out_ids[num_act * 4 + 0] = b;
        for (int k = 0; k < 3; ++k) out_ids[num_act * 4 + 1 + k] = p[k];
        it = grid.emplace(idx, num_act++).first;
      }
      pairs[((int64_t)off * 2 + 0) * V + num[off]] = (int32_t)j;
      pairs[((int64_t)off * 2 + 1) * V + num[off]++] = it->second;
    }
  }
  return num_act;
}

// indiceConv<float> (spconv_ops.h:260-361) with the CPU gather / scatter-add functors
// (src/reordering.cc:20-49): subM centre = argmax(num) as a dense feat.W[centre], then per offset
// gather -> GEMM (fresh dot product per row) -> scatter-add. fp32 throughout.
void orc_indice_conv_f32(const float* feat, int64_t num_in, int cin, const float* W, int kv,
                         int cout, const int32_t* pairs, int64_t pair_stride, const int32_t* num,
                         int64_t num_out, int inverse, int subm, float* out) {
  (void)num_in;
  for (int64_t i = 0; i < num_out * cout; ++i) out[i] = 0.f;
  int centre = -1;
  if (subm) {
    centre = (int)(std::max_element(num, num + kv) - num);
    for (int64_t r = 0; r < num_out; ++r)
      for (int co = 0; co < cout; ++co) {
        float acc = 0.f;
        for (int c = 0; c < cin; ++c) acc = acc + feat[r * cin + c] * W[((int64_t)centre * cin + c) * cout + co];
        out[r * cout + co] = acc;
      }
  }
  std::vector<float> buf(cout);
  for (int k = 0; k < kv; ++k) {
    int nhot = num[k];
    if (nhot <= 0 || (subm && k == centre)) continue;
    const int32_t* pin = pairs + ((int64_t)k * 2 + (inverse ? 1 : 0)) * pair_stride;
    const int32_t* pout = pairs + ((int64_t)k * 2 + (inverse ? 0 : 1)) * pair_stride;
    for (int p = 0; p < nhot; ++p) {
      const float* f = feat + (int64_t)pin[p] * cin;
      for (int co = 0; co < cout; ++co) {
        float acc = 0.f;
        for (int c = 0; c < cin; ++c) acc = acc + f[c] * W[((int64_t)k * cin + c) * cout + co];
        buf[co] = acc;
      }
      float* o = out + (int64_t)pout[p] * cout;
      for (int co = 0; co < cout; ++co) o[co] += buf[co];
    }
  }
}


// ---------------------------------------------------------------------------------------------
// RadiusGraph.build_graph (pcdet/models/model_utils/graph_utils.py:149-209) + hash_insert_gpu /
// radius_graph_gpu (pcdet/ops/torch_hash/src/torch_hash_kernel.cu:31-91, 224-409, 487-561),
// followed literally: voxel coords by round-half-even, table of H = int(N/0.5) slots with linear
// probing (inserted serially here, so deterministic), per query the cell combinations with dim 0
// fastest, probe chain until an empty slot, fp32 dist2 over all 4 dims, `<=` radius^2, at most
// max_nbr kept. Tie rule (the reference's depends on an atomicCAS race): (dist2, ref index)
// ascending when sort_by_dist, ref index ascending otherwise.
// Returns E; e_ref/e_qry/dists have capacity M*max_nbr.
static inline int64_t orc_map2key(const int64_t* c, const int64_t* dims) {
  int64_t ans = 0;
  for (int i = 0; i < 4; ++i) {
    int64_t k = c[i];
    if (k >= dims[i]) k = dims[i];
    if (k < 0) k = 0;
    ans = ans * dims[i] + k;
  }
  return ans;
}
static inline int64_t orc_hashkey(int64_t key, int64_t H) { return ((key % H) * 999269 + 999437) % H; }

int64_t orc_radius_graph(const float* ref, int64_t N, const float* qry, int64_t M, float radius,
                         const int* qmin, const int* qmax, int max_nbr, int sort_by_dist,
                         int64_t* e_ref, int64_t* e_qry, float* dists) {
  if (N == 0 || M == 0) return 0;
  float vs[4] = {(float)(1 - 1e-3), radius, radius, radius};
  float lo[4], hi[4];
  for (int d = 0; d < 4; ++d) { lo[d] = INFINITY; hi[d] = -INFINITY; }
  for (int64_t i = 0; i < N; ++i)
    for (int d = 0; d < 4; ++d) { lo[d] = std::fmin(lo[d], ref[i * 4 + d]); hi[d] = std::fmax(hi[d], ref[i * 4 + d]); }
  for (int64_t i = 0; i < M; ++i)
    for (int d = 0; d < 4; ++d) { lo[d] = std::fmin(lo[d], qry[i * 4 + d]); hi[d] = std::fmax(hi[d], qry[i * 4 + d]); }
  int64_t dims[4];
  for (int d = 0; d < 4; ++d) {
    lo[d] = lo[d] - vs[d] * 2.f;
    hi[d] = hi[d] + vs[d] * 2.f;
    dims[d] = (int64_t)std::nearbyint((hi[d] - lo[d]) / vs[d]) + 3;
  }
  auto coord = [&](const float* p, int64_t* c) {
    for (int d = 0; d < 4; ++d) c[d] = (int64_t)std::nearbyint((p[d] - lo[d]) / vs[d]) + 1;
  };
  int64_t H = (int64_t)((double)N / 0.5);
  std::vector<int64_t> keys(H, -1), rev(H, 0);
  std::vector<float> vals((size_t)H * 4);
  for (int64_t i = 0; i < N; ++i) {
    int64_t c[4];
    coord(ref + i * 4, c);
    int64_t key = orc_map2key(c, dims);
    int64_t slot = orc_hashkey(key, H);
    while (keys[slot] != -1) slot = (slot + 1) % H;
    keys[slot] = key;
    for (int d = 0; d < 4; ++d) vals[slot * 4 + d] = ref[i * 4 + d];
    rev[slot] = i;
  }
  int w[4], ncomb = 1;
  for (int d = 0; d < 4; ++d) { w[d] = qmax[d] - qmin[d] + 1; ncomb *= w[d]; }
  float r2 = radius * radius;
  int64_t E = 0;
  std::vector<float> bd(max_nbr);
  std::vector<int64_t> bi(max_nbr);
  for (int64_t q = 0; q < M; ++q) {
    int64_t qc[4];
    coord(qry + q * 4, qc);
    int cnt = 0;
    for (int c = 0; c < ncomb; ++c) {
      int64_t cc[4];
      int tmp = c;
      for (int d = 0; d < 4; ++d) { cc[d] = qc[d] + (tmp % w[d] + qmin[d]); tmp /= w[d]; }
      int64_t key = orc_map2key(cc, dims);
      int64_t slot = orc_hashkey(key, H);
      while (keys[slot] != -1) {
        if (keys[slot] == key) {
          float dist2 = 0.f;
          for (int d = 0; d < 4; ++d) {
            float di = vals[slot * 4 + d] - qry[q * 4 + d];
            dist2 = dist2 + di * di;
          }
          if (dist2 <= r2) {
            float kk = sort_by_dist ? dist2 : 0.f;
            int64_t idx = rev[slot];
            int pos = cnt;
            while (pos > 0) {
              float kd = sort_by_dist ? bd[pos - 1] : 0.f;
              bool less = (kk < kd) || (kk == kd && idx < bi[pos - 1]);
              if (!less) break;
              if (pos < max_nbr) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; }
              --pos;
            }
            if (pos < max_nbr) { bd[pos] = dist2; bi[pos] = idx; if (cnt < max_nbr) ++cnt; }
          }
        }
        slot = (slot + 1) % H;
      }
    }
    for (int e = 0; e < cnt; ++e) {
      e_ref[E] = bi[e];
      e_qry[E] = q;
      if (dists) dists[E] = bd[e];
      ++E;
    }
  }
  return E;
}

// correspondence_kernel / points_in_radius_kernel (torch_hash_kernel.cu:96-155, 160-222) on the
// same literal table and cell walk as orc_radius_graph (voxel coordinates of graph_utils.py:169-183
// for cells of `cell` metres, frame cells of 1 - 1e-3). mode 0: corres[q] = the nearest hashed point
// among the walked cells (strict `<`, no radius; ties -- race-ordered in the reference -- by the lower
// ref index), -1 if none. mode 1: visited[ref] = 1 for every hashed point with dist2 < radius^2 of
// some query.
void orc_table_query(const float* ref, int64_t N, const float* qry, int64_t M, float cell,
                     const int* qmin, const int* qmax, int mode, float radius, int64_t* out) {
  if (mode == 0) for (int64_t q = 0; q < M; ++q) out[q] = -1;
  if (N == 0 || M == 0) return;
  float vs[4] = {(float)(1 - 1e-3), cell, cell, cell};
  float lo[4], hi[4];
  for (int d = 0; d < 4; ++d) { lo[d] = INFINITY; hi[d] = -INFINITY; }
  for (int64_t i = 0; i < N; ++i)
    for (int d = 0; d < 4; ++d) { lo[d] = std::fmin(lo[d], ref[i * 4 + d]); hi[d] = std::fmax(hi[d], ref[i * 4 + d]); }
  for (int64_t i = 0; i < M; ++i)
    for (int d = 0; d < 4; ++d) { lo[d] = std::fmin(lo[d], qry[i * 4 + d]); hi[d] = std::fmax(hi[d], qry[i * 4 + d]); }
  int64_t dims[4];
  for (int d = 0; d < 4; ++d) {
    lo[d] = lo[d] - vs[d] * 2.f;
    hi[d] = hi[d] + vs[d] * 2.f;
    dims[d] = (int64_t)std::nearbyint((hi[d] - lo[d]) / vs[d]) + 3;
  }
  auto coord = [&](const float* p, int64_t* c) {
    for (int d = 0; d < 4; ++d) c[d] = (int64_t)std::nearbyint((p[d] - lo[d]) / vs[d]) + 1;
  };
  int64_t H = (int64_t)((double)N / 0.5);
  std::vector<int64_t> keys(H, -1), rev(H, 0);
  std::vector<float> vals((size_t)H * 4);
  for (int64_t i = 0; i < N; ++i) {
    int64_t c[4];
    coord(ref + i * 4, c);
    int64_t key = orc_map2key(c, dims);
    int64_t slot = orc_hashkey(key, H);
    while (keys[slot] != -1) slot = (slot + 1) % H;
    keys[slot] = key;
    for (int d = 0; d < 4; ++d) vals[slot * 4 + d] = ref[i * 4 + d];
    rev[slot] = i;
  }
  int w[4], ncomb = 1;
  for (int d = 0; d < 4; ++d) { w[d] = qmax[d] - qmin[d] + 1; ncomb *= w[d]; }
  const float r2 = radius * radius;
  for (int64_t q = 0; q < M; ++q) {
    int64_t qc[4];
    coord(qry + q * 4, qc);
    float best = 1e10f;
    int64_t bi = -1;
    for (int c = 0; c < ncomb; ++c) {
      int64_t cc[4];
      int tmp = c;
      for (int d = 0; d < 4; ++d) { cc[d] = qc[d] + (tmp % w[d] + qmin[d]); tmp /= w[d]; }
      int64_t key = orc_map2key(cc, dims);
      int64_t slot = orc_hashkey(key, H);
      while (keys[slot] != -1) {
        if (keys[slot] == key) {
          float dist2 = 0.f;
          for (int d = 0; d < 4; ++d) {
            float di = vals[slot * 4 + d] - qry[q * 4 + d];
            dist2 = dist2 + di * di;
          }
          if (mode == 0) {
            if (dist2 < best || (dist2 == best && bi >= 0 && rev[slot] < bi)) { best = dist2; bi = rev[slot]; }
          } else if (dist2 < r2) {
            out[rev[slot]] = 1;
          }
        }
        slot = (slot + 1) % H;
      }
    }
    if (mode == 0) out[q] = bi;
  }
}

// connected_components (graph_utils.py:40-53): weakly connected components, label = order of
// first appearance scanning nodes 0..n-1 (scipy.sparse.csgraph labelling). Returns the count.
int64_t orc_connected_components(const int64_t* e0, const int64_t* e1, int64_t E, int64_t n,
                                 int64_t* component) {
  std::vector<int64_t> parent(n);
  std::iota(parent.begin(), parent.end(), 0);
  auto find = [&](int64_t x) {
    while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
    return x;
  };
  for (int64_t e = 0; e < E; ++e) {
    int64_t a = find(e0[e]), b = find(e1[e]);
    if (a != b) parent[a > b ? a : b] = a > b ? b : a;
  }
  std::vector<int64_t> label(n, -1);
  int64_t num = 0;
  for (int64_t i = 0; i < n; ++i) {
    int64_t r = find(i);
    if (label[r] < 0) label[r] = num++;
    component[i] = label[r];
  }
  return num;
}

// points_in_boxes_cpu / check_pt_in_box3d_cpu, pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-168
void orc_points_in_boxes(const float* pts, int64_t N, const float* boxes, int B, int32_t* out) {
  for (int i = 0; i < B; ++i)
    for (int64_t j = 0; j < N; ++j) {
      const float* bx = boxes + i * 7;
      float x = pts[j * 3], y = pts[j * 3 + 1], z = pts[j * 3 + 2];
      float cx = bx[0], cy = bx[1], cz = bx[2], dx = bx[3], dy = bx[4], dz = bx[5], rz = bx[6];
      int flag = 0;
      if (!(std::fabs(z - cz) > dz / 2.0)) {
        const float MARGIN = 1e-2;
        float cosa = std::cos((double)-rz), sina = std::sin((double)-rz);
        float sx = x - cx, sy = y - cy;
        float lx = sx * cosa + sy * (-sina);
        float ly = sx * sina + sy * cosa;
        flag = (std::fabs((double)lx) < dx / 2.0 + MARGIN) & (std::fabs((double)ly) < dy / 2.0 + MARGIN);
      }
      out[(int64_t)i * N + j] = flag;
    }
}

}  // extern "C"
