// Radius graph (the pairwise point-correspondence primitive of the whole pipeline) for gfx950.
//
// Reference: pcdet/ops/torch_hash/src/torch_hash_kernel.cu (hash_insert_gpu_kernel :54-91,
// count_radius_graph_degree_kernel :224-288, radius_graph_kernel :290-409, host wrappers :411-561)
// driven by RadiusGraph.build_graph, pcdet/models/model_utils/graph_utils.py:149-209.
//
// What the reference computes (restated): with cell size = radius the 3x3x3
// cell walk covers the whole ball, so for every query the candidate set is exactly
//   { ref i : framecell(ref i) == framecell(query) + df, df in [qmin0, qmax0],
//             dist2 = sum_{d=0..3} (ref_d - query_d)^2 (fp32, d ascending)  <=  radius^2 }
// and the output keeps the max_nbr nearest (sort_by_dist) in ascending dist2. The reference's
// tie order depends on an atomicCAS race; the canonical rule here is (dist2, ref index) ascending.
//
// Two implementations live in this file:
//  1. pcs_radius_graph_*: the engine used by the host-side RadiusGraph module. It does NOT walk
//     radius-sized cells: reference points are chained into an open-addressing table of FINER cells
//     (cell size h <= radius) and every query runs an expanding-ring search that stops as soon as
//     the ring's lower-bound distance exceeds its current k-th best. Same edges, a fraction of the
//     distance evaluations (1-NN at radius 2.5 m over 0.4 m voxels examines tens of candidates
//     instead of thousands). One search pass (no count pass + fill pass re-walk): per-query top-k
//     lives in LDS, results are compacted by an exclusive scan of the degrees.
//  2. pcs_hash_insert / pcs_radius_graph_table_*: literal drop-ins for the reference's
//     torch_hash_cuda entry points on the reference's own table layout (keys/values/reverse,
//     caller-allocated), for callers that bind at the pcdet/ops/torch_hash level.
#include "pcs_grid2.hpp"
#include "pcs_knn.hpp"

namespace pcs {

constexpr int kMaxK = 32;

// Per-query top-K (sorted ascending by (key, idx)) kept in LDS columns: element e of thread t at
// [e * BLOCK + t] -> consecutive lanes hit consecutive banks. An element is ONE 64-bit word, key bits above the
// index: squared distances are non-negative floats, whose bit patterns order like their values, so (key, idx)
// ascending is the word ascending -- one LDS read, one compare and one write per shifted element instead of two each.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
radius_search_kernel(const float4* __restrict__ ref, const float4* __restrict__ qry, int M,
                     const GridDesc* __restrict__ gp, const uint64_t* __restrict__ keys,
                     const int32_t* __restrict__ head, const int32_t* __restrict__ next,
                     uint32_t hmask, const float* __restrict__ fr, int df_min, int df_max, int K,
                     int sort_by_dist, int32_t* __restrict__ nbr_idx, float* __restrict__ nbr_d,
                     int32_t* __restrict__ deg) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* sk = (unsigned long long*)smem;
  int t = threadIdx.x;
  int q = blockIdx.x * BLOCK + t;
  if (q >= M) return;
  GridDesc g = *gp;
  float4 p = qry[q];
  int fq = frame_cell(g, p.x);
  int cx = (int)floorf((p.y - g.lo[0]) / g.h);
  int cy = (int)floorf((p.z - g.lo[1]) / g.h);
  int cz = (int)floorf((p.w - g.lo[2]) / g.h);
  int cnt = 0;
  const float margin = 1e-5f * g.radius2 + 1e-6f;
  for (int df = df_min; df <= df_max; ++df) {
    int fc = fq + df;
    if (fc < 0 || fc >= kFrameCells) continue;
    float fa = fr[fc], fb = fr[kFrameCells + fc];
    if (fa > fb) continue;  // no reference point in this frame cell
    // xyz budget: dist2 = dframe^2 + |dxyz|^2 and |dframe| >= distance from the query's frame
    // value to the range of frame values present in the cell
    float dfl = fmaxf(fmaxf(fa - p.x, p.x - fb), 0.f);
    float dfsq = dfl * dfl;
    for (int s = 0; s <= g.rings; ++s) {
      if (s >= 1) {
        float lb = (float)(s - 1) * g.h;
        float bound = (sort_by_dist && cnt == K) ? __uint_as_float((unsigned)(sk[(K - 1) * BLOCK + t] >> 32)) : g.radius2;
        if (lb * lb > bound - dfsq + margin) break;
      }
      for (int dz = -s; dz <= s; ++dz) {
        int z = cz + dz;
        if (z < 0 || z >= kCoordMax) continue;
        for (int dy = -s; dy <= s; ++dy) {
          int y = cy + dy;
          if (y < 0 || y >= kCoordMax) continue;
          bool shell = (dz == -s || dz == s || dy == -s || dy == s);
          int step = shell ? 1 : (s > 0 ? 2 * s : 1);
          for (int dx = -s; dx <= s; dx += step) {
            int x = cx + dx;
            if (x < 0 || x >= kCoordMax) continue;
            int j = grid_head(keys, head, hmask, pack_cell(fc, x, y, z));
            while (j >= 0) {
              float4 c = ref[j];
              // torch_hash_kernel.cu:364-368: dims 0..3 in order, starting from 0
              float d0 = c.x - p.x, d1 = c.y - p.y, d2 = c.z - p.z, d3 = c.w - p.w;
              float dist2 = 0.f;
              dist2 = dist2 + d0 * d0;
              dist2 = dist2 + d1 * d1;
              dist2 = dist2 + d2 * d2;
              dist2 = dist2 + d3 * d3;
              if (dist2 <= g.radius2) {
                const unsigned long long word =
                    ((unsigned long long)(sort_by_dist ? __float_as_uint(dist2) : 0u) << 32) | (unsigned)j;
                // position among the kept ones: strict (key, idx) order
                int pos = cnt;
                while (pos > 0) {
                  const unsigned long long kw = sk[(pos - 1) * BLOCK + t];
                  if (!(word < kw)) break;
                  if (pos < K) sk[pos * BLOCK + t] = kw;
                  --pos;
                }
                if (pos < K) {
                  sk[pos * BLOCK + t] = word;
                  if (cnt < K) ++cnt;
                }
              }
              j = next[j];
            }
          }
        }
      }
    }
  }
  deg[q] = cnt;
  for (int e = 0; e < cnt; ++e) {
    const unsigned long long w = sk[e * BLOCK + t];
    nbr_idx[(int64_t)q * K + e] = (int32_t)(unsigned)(w & 0xFFFFFFFFULL);
    if (nbr_d) nbr_d[(int64_t)q * K + e] = __uint_as_float((unsigned)(w >> 32));
  }
}

// max_nbr == 1, sorted: the two-level grid + 32-lanes-per-query search of the fused ICP
__global__ void __launch_bounds__(kBlock)
nn1_search_kernel(const float4* __restrict__ qry, int M, const GridDesc* __restrict__ gp, Grid2 g,
                  int df, int32_t* __restrict__ err, int32_t* __restrict__ nbr_idx,
                  float* __restrict__ nbr_d, int32_t* __restrict__ deg) {
  GridDesc d = *gp;
  int qid = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  bool valid = qid < M;
  float4 p = valid ? qry[qid] : make_float4(0.f, 0.f, 0.f, 0.f);
  float bd = 0.f;
  int r = g2_nearest_coop(d, g, p, df, valid, err, &bd);
  if (valid && (threadIdx.x & 31) == 0) {
    deg[qid] = r >= 0;
    nbr_idx[qid] = r;
    nbr_d[qid] = bd;
  }
}

// ---- round 5: most queries of the tracker's extraction graph (every full-resolution point of a unit's frames
// against the unit's tracked voxels, cluster_tracking.py:356-358 of the reference) have NOTHING within the radius,
// and the cooperative search spends 32 lanes and 27 mask probes on each to find that out. A query can only have a
// neighbour if one of the 3 x 3 x 3 coarse blocks around it is occupied; those 27 blocks lie in 2 x 2 x 2 SUPER
// blocks (two coarse blocks per axis), so one thread decides with eight probes of a set of occupied super blocks.
// Exact: a survivor list that is a superset of the queries with a neighbour; the others get degree 0 here.
__global__ void __launch_bounds__(kBlock)
nn1_super_insert_kernel(const float4* __restrict__ ref, int N, const GridDesc* __restrict__ gp,
                        uint64_t* __restrict__ skeys, uint32_t smask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const GridDesc d = *gp;
  const float4 p = ref[i];
  const int fc = frame_cell(d, p.x);
  const int x = (int)floorf((p.y - d.lo[0]) / d.h), y = (int)floorf((p.z - d.lo[1]) / d.h),
            z = (int)floorf((p.w - d.lo[2]) / d.h);
  if (x < 0 || y < 0 || z < 0 || x >= kCoordMax || y >= kCoordMax || z >= kCoordMax || fc < 0 || fc >= kFrameCells)
    return;  // (g2_insert_kernel raises the call's error word for such a point)
  const uint64_t key = pack_cell(fc, x >> 3, y >> 3, z >> 3);
  uint32_t s = mix64(key) & smask;
  for (uint32_t t = 0; t <= smask; ++t) {
    const uint64_t prev = atomicCAS((unsigned long long*)&skeys[s], (unsigned long long)kEmpty, (unsigned long long)key);
    if (prev == kEmpty || prev == key) return;
    s = (s + 1) & smask;
  }
}
__global__ void __launch_bounds__(kBlock)
nn1_filter_kernel(const float4* __restrict__ qry, int M, const GridDesc* __restrict__ gp,
                  const uint64_t* __restrict__ skeys, uint32_t smask, int df, int32_t* __restrict__ list,
                  int32_t* __restrict__ cnt, int32_t* __restrict__ nbr_idx, float* __restrict__ nbr_d,
                  int32_t* __restrict__ deg) {
  const GridDesc d = *gp;
  const int lane = threadIdx.x & 63;
  const int nround = (M + kBlock - 1) / kBlock;
  for (int r = blockIdx.x; r < nround; r += gridDim.x) {  // whole waves stay together (ballot)
    const int q = r * kBlock + threadIdx.x;
    bool keep = false;
    if (q < M) {
      const float4 p = qry[q];
      const int fc = frame_cell(d, p.x) + df;
      const int bx = (int)floorf((p.y - d.lo[0]) / d.h) >> 2, by = (int)floorf((p.z - d.lo[1]) / d.h) >> 2,
                bz = (int)floorf((p.w - d.lo[2]) / d.h) >> 2;
      if (fc >= 0 && fc < kFrameCells) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int sx = ((bx - 1) >> 1) + (c & 1), sy = ((by - 1) >> 1) + ((c >> 1) & 1), sz = ((bz - 1) >> 1) + (c >> 2);
          if (sx < 0 || sy < 0 || sz < 0) continue;
          const uint64_t key = pack_cell(fc, sx, sy, sz);
          uint32_t s = mix64(key) & smask;
          for (uint32_t t = 0; t <= smask; ++t) {
            const uint64_t k = skeys[s];
            if (k == key) { keep = true; break; }
            if (k == kEmpty) break;
            s = (s + 1) & smask;
          }
        }
      }
      if (!keep) {
        deg[q] = 0;
        nbr_idx[q] = -1;
        nbr_d[q] = d.radius2;
      }
    }
    const unsigned long long m = __ballot(keep);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(cnt, __popcll(m));
      base = __shfl(base, 0, 64);
      if (keep) list[base + __popcll(m & ((1ULL << lane) - 1ULL))] = q;
    }
  }
}
// The static grid of the 1-NN path after its build: the points that did not fit their cell's kCellCap rows (voxels
// of several sampling levels piled into one 22 cm cell: a few hundred of a million) are sorted by cell -- ONE
// workgroup, word = slot << 12 | position, bitonic in LDS -- and every full cell's slot learns where its spilled
// points start (pad = start + 1). flag[0] = 1: done (at most 4096 spilled points; beyond that the searches keep
// scanning the whole list, as the ICP's grids do).
constexpr int kSpillSort = 4096;
__global__ void __launch_bounds__(1024)
nn1_spill_sort_kernel(const GridDesc* __restrict__ gp, Grid2 g, int32_t* __restrict__ flag) {
  __shared__ uint64_t key[kSpillSort];
  __shared__ int s_bad;
  const int t = threadIdx.x;
  const int n = *g.ovf;
  if (t == 0) s_bad = 0;
  __syncthreads();
  if (n <= 0 || n > kSpillSort || n > (int)g.ovf_cap) {
    if (t == 0) flag[0] = n <= 0 ? 1 : 0;
    return;
  }
  const GridDesc d = *gp;
  for (int i = t; i < kSpillSort; i += 1024) {
    uint64_t k = ~0ULL;
    if (i < n) {
      const float4 p = g.ovf_pts[i];
      const int fc = frame_cell(d, p.x);
      const int x = (int)floorf((p.y - d.lo[0]) / d.h), y = (int)floorf((p.z - d.lo[1]) / d.h),
                z = (int)floorf((p.w - d.lo[2]) / d.h);
      uint32_t c = 0, pw = 0;
      const int fs = g2_find(g.fslot, g.Hf - 1, pack_cell(fc, x, y, z), &c, &pw);
      if (fs < 0 || (int)c <= kCellCap) s_bad = 1;  // (cannot be: a spilled point's cell exists and is full)
      k = ((uint64_t)(uint32_t)max(fs, 0) << 12) | (uint64_t)i;
    }
    key[i] = k;
  }
  __syncthreads();
  if (s_bad) {  // leave the list as it is: the searches scan all of it
    if (t == 0) flag[0] = 0;
    return;
  }
  for (int size = 2; size <= kSpillSort; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < kSpillSort / 2; i += 1024) {
        const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint64_t a = key[lo], b = key[hi];
        if ((a > b) == up) { key[lo] = b; key[hi] = a; }
      }
      __syncthreads();
    }
  // the points in sorted order (read all, then write), the cells' start offsets
  float4 pp[kSpillSort / 1024];
  int32_t ii[kSpillSort / 1024];
#pragma unroll
  for (int q = 0; q < kSpillSort / 1024; ++q) {
    const int i = t + q * 1024;
    if (i < n) { const int src = (int)(key[i] & 4095ULL); pp[q] = g.ovf_pts[src]; ii[q] = g.ovf_idx[src]; }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kSpillSort / 1024; ++q) {
    const int i = t + q * 1024;
    if (i < n) {
      g.ovf_pts[i] = pp[q];
      g.ovf_idx[i] = ii[q];
      const uint32_t fs = (uint32_t)(key[i] >> 12);
      if (i == 0 || (uint32_t)(key[i - 1] >> 12) != fs) g.fslot[fs].pad = i + 1;
    }
  }
  if (t == 0) flag[0] = 1;
}

// Most queries that pass the filter lie ON a tracked object: a voxel of it sits in their own fine cell or the one
// beside it, nearer than one cell width h -- and then nothing outside the 3 x 3 x 3 fine cells around the query can be
// nearer (it is at least h away along some axis): the cooperative search stops after its first phase too, but it
// spends 32 lanes, a dozen shuffles and two ballots on that one query (3600 SIMD cycles per query measured: the
// wave's chain of dependent loads serves two queries). Here ONE THREAD takes a query through the same 27 cells --
// 27 independent probe chains in flight per lane, 64 queries per wave -- and settles it when the search's own stop
// rule (g2_nearest_coop: h^2 > dist2 - frame term + margin) holds for the nearest point it found; everything else
// (no point in the 27 cells, or the nearest not nearer than h, a cell that spilled, a query outside the frame range)
// goes on to the cooperative search unchanged. Same canonical choice (dist2, original index): same edges.
__global__ void __launch_bounds__(kBlock)
nn1_near_kernel(const float4* __restrict__ qry, const int32_t* __restrict__ list, const int32_t* __restrict__ cnt,
                const GridDesc* __restrict__ gp, Grid2 g, int df, int32_t* __restrict__ list2,
                int32_t* __restrict__ cnt2, int32_t* __restrict__ nbr_idx, float* __restrict__ nbr_d,
                int32_t* __restrict__ deg, const int32_t* __restrict__ spill_sorted) {
  const GridDesc d = *gp;
  const int n = *cnt;
  const int lane = threadIdx.x & 63;
  // (spilled points that are not sorted by cell: only the cooperative search knows how to scan that list)
  const bool spilled = *g.ovf > 0 && !*spill_sorted;
  const int nround = (n + kBlock - 1) / kBlock;
  const float margin = 1e-5f * d.radius2 + 1e-6f;
  for (int r = blockIdx.x; r < nround; r += gridDim.x) {  // whole waves stay together (ballot)
    const int w = r * kBlock + threadIdx.x;
    const bool valid = w < n;
    bool later = false;
    if (valid) {
      const int qid = list[w];
      const float4 p = qry[qid];
      const int fc = frame_cell(d, p.x) + df;
      const int cx = (int)floorf((p.y - d.lo[0]) / d.h), cy = (int)floorf((p.z - d.lo[1]) / d.h),
                cz = (int)floorf((p.w - d.lo[2]) / d.h);
      later = spilled || fc < 0 || fc >= kFrameCells || cx < 1 || cy < 1 || cz < 1;
      if (!later) {
        const float fa = g.fr[fc], fb = g.fr[kFrameCells + fc];
        const float dfl = fmaxf(fmaxf(fa - p.x, p.x - fb), 0.f);
        const float dfsq = dfl * dfl;
        later = !(fa <= fb) || !(d.radius2 - dfsq + margin >= 0.f);
        int cand = -1;
        float cd = 0.f;
        if (!later) {
          // the query's own cell first; a neighbour cell is skipped when its nearest face is farther than the
          // nearest point found so far (most queries sit within a few centimetres of a voxel: 4-8 cells of 27)
          const float fx = (p.y - d.lo[0]) / d.h - (float)cx, fy = (p.z - d.lo[1]) / d.h - (float)cy,
                      fz = (p.w - d.lo[2]) / d.h - (float)cz;
#pragma unroll 1
          for (int qq = 0; qq < 27; ++qq) {
            const int q = qq == 0 ? 13 : (qq <= 13 ? qq - 1 : qq);
            const int ox = q % 3 - 1, oy = (q / 3) % 3 - 1, oz = q / 9 - 1;
            if (cand >= 0) {
              const float ax = ox < 0 ? fx : (ox > 0 ? 1.f - fx : 0.f), ay = oy < 0 ? fy : (oy > 0 ? 1.f - fy : 0.f),
                          az = oz < 0 ? fz : (oz > 0 ? 1.f - fz : 0.f);
              // (0.999: the cell coordinates are fp32 quotients; a cell that close is simply looked at)
              const float lb = ((ax * ax + ay * ay) + az * az) * (d.h * d.h) * 0.999f;
              if (lb > (cd - dfsq) + margin) continue;
            }
            const int x = cx + ox, y = cy + oy, z = cz + oz;
            uint32_t cnt_w = 0, pad_w = 0;
            const int fs = g2_find(g.fslot, g.Hf - 1, pack_cell(fc, x, y, z), &cnt_w, &pad_w);
            if (fs < 0) continue;
            // (a full cell: its kCellCap rows, then its spilled points where the slot says they start)
            const float4* run = g.spts + (size_t)fs * kCellCap;
            const int32_t* idx = g.sidx + (size_t)fs * kCellCap;
            const int rows = min((int)cnt_w, kCellCap);
            const int extra = (int)cnt_w > kCellCap ? min((int)cnt_w - kCellCap, (int)g.ovf_cap - ((int)pad_w - 1)) : 0;
            for (int j = 0; j < rows + extra; ++j) {
              const float4 c = j < rows ? run[j] : g.ovf_pts[(int)pad_w - 1 + (j - rows)];
              const int oi = j < rows ? idx[j] : g.ovf_idx[(int)pad_w - 1 + (j - rows)];
              const float d0 = c.x - p.x, d1 = c.y - p.y, d2 = c.z - p.z, d3 = c.w - p.w;
              float dist2 = 0.f;
              dist2 = dist2 + d0 * d0;
              dist2 = dist2 + d1 * d1;
              dist2 = dist2 + d2 * d2;
              dist2 = dist2 + d3 * d3;
              if (dist2 <= d.radius2 && (cand < 0 || dist2 < cd || (dist2 == cd && oi < cand))) {
                cand = oi;
                cd = dist2;
              }
            }
          }
          // the search's own stop rule after its first phase: nothing outside the 27 cells can be nearer
          if (!later) later = !(cand >= 0 && d.h * d.h > (cd - dfsq) + margin);
        }
        if (!later) {
          deg[qid] = 1;
          nbr_idx[qid] = cand;
          nbr_d[qid] = cd;
        }
      }
    }
    const unsigned long long m = __ballot(later);
    if (m) {
      int base = 0;
      if (lane == 0) base = atomicAdd(cnt2, __popcll(m));
      base = __shfl(base, 0, 64);
      if (later) list2[base + __popcll(m & ((1ULL << lane) - 1ULL))] = list[w];
    }
  }
}

__global__ void __launch_bounds__(kBlock)
nn1_search_list_kernel(const float4* __restrict__ qry, const int32_t* __restrict__ list, const int32_t* __restrict__ cnt,
                       const GridDesc* __restrict__ gp, Grid2 g, int df, int32_t* __restrict__ err,
                       int32_t* __restrict__ nbr_idx, float* __restrict__ nbr_d, int32_t* __restrict__ deg,
                       const int32_t* __restrict__ spill_sorted) {
  const GridDesc d = *gp;
  const int n = *cnt;
  const bool by_cell = *spill_sorted != 0;
  const int per = kBlock / 32;
  const int nround = (n + per - 1) / per;
  for (int r = blockIdx.x; r < nround; r += gridDim.x) {  // whole waves stay together (shuffles)
    const int w = r * per + (threadIdx.x >> 5);
    const bool valid = w < n;
    const int qid = valid ? list[w] : 0;
    const float4 p = valid ? qry[qid] : make_float4(0.f, 0.f, 0.f, 0.f);
    float bd = 0.f;
    const int res = g2_nearest_coop(d, g, p, df, valid, err, &bd, 0.f, nullptr, __builtin_inff(), nullptr, by_cell);
    if (valid && (threadIdx.x & 31) == 0) {
      deg[qid] = res >= 0;
      nbr_idx[qid] = res;
      nbr_d[qid] = bd;
    }
  }
}

__global__ void __launch_bounds__(kBlock)
edge_total_kernel(const int32_t* __restrict__ deg, const int32_t* __restrict__ off, int M,
                  const int32_t* __restrict__ err, int64_t* __restrict__ d_num) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int64_t e = M > 0 ? (int64_t)off[M - 1] + deg[M - 1] : 0;
    if (err && *err) e = -1;  // fine-cell coordinates overflowed: the caller must not use the result
    *d_num = e;
  }
}

__global__ void __launch_bounds__(kBlock)
edge_emit_kernel(const int32_t* __restrict__ nbr_idx, const float* __restrict__ nbr_d,
                 const int32_t* __restrict__ deg, const int32_t* __restrict__ off, int M, int K,
                 int64_t* __restrict__ e_ref, int64_t* __restrict__ e_qry,
                 float* __restrict__ dists) {
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < M; q += gridDim.x * blockDim.x) {
    int n = deg[q], o = off[q];
    for (int e = 0; e < n; ++e) {
      e_ref[o + e] = nbr_idx[(int64_t)q * K + e];
      e_qry[o + e] = q;
      if (dists) dists[o + e] = nbr_d[(int64_t)q * K + e];
    }
  }
}

// test hook (pcs_radius_graph_engine_only): 1 = every search on the one-thread-per-query engine
static int g_radius_engine_only = 0;
// ... 2 = the 1-NN path without its one-thread near search (every survivor of the filter searched cooperatively)
static int g_nn1_coop_only = 0;

struct RadiusPlan {
  GridDesc* grid;
  float *b_ref, *b_qry;
  uint64_t* keys;
  int32_t *head, *next, *nbr_idx, *deg, *off, *err;
  float *nbr_d, *fr;
  Grid2 g2;          // max_nbr == 1 path
  int32_t* g2_words; // [0] cursor, [1] "stop" word (INT_MAX: the gated grid kernels always run)
  void* tmp;
  size_t tmp_bytes;
  uint32_t H;
  KnnPlan knn;       // max_nbr > 1: the cell-sorted search (pcs_knn.hip)
  uint64_t* skeys;   // max_nbr == 1: occupied super blocks (nn1_filter_kernel), [Hs]
  uint32_t Hs;
  int32_t* nn1_list; // [M] queries that may have a neighbour
  int32_t* nn1_list2; // [M] ... of those, the ones the one-thread near search could not settle
};

static inline uint32_t pow2_ge(uint64_t x) {
  uint32_t h = 64;
  while (h < x) h <<= 1;
  return h;
}

static bool plan_radius(Arena& a, int64_t N, int64_t M, int K, RadiusPlan& p) {
  if (N < 1) N = 1;
  if (M < 1) M = 1;
  p.grid = a.take<GridDesc>(1);
  p.b_ref = a.take<float>(8);
  p.b_qry = a.take<float>(8);
  p.err = a.take<int32_t>(1);
  p.fr = a.take<float>(2 * kFrameCells);
  p.H = pow2_ge(2 * (uint64_t)N);
  p.keys = a.take<uint64_t>(p.H);
  p.head = a.take<int32_t>(p.H);
  p.next = a.take<int32_t>(N);
  p.nbr_idx = a.take<int32_t>(M * K);
  p.nbr_d = a.take<float>(M * K);
  p.deg = a.take<int32_t>(M);
  p.off = a.take<int32_t>(M);
  p.tmp_bytes = scan_i32_temp_bytes(M);
  p.tmp = a.take<char>(p.tmp_bytes);
  if (K == 1) {
    plan_grid2(a, N, p.g2);
    p.g2_words = a.take<int32_t>(4);  // ... [2] the 1-NN path: spilled points sorted by cell
    p.Hs = pow2_ge(2 * (uint64_t)N);
    p.skeys = a.take<uint64_t>(p.Hs);
    p.nn1_list = a.take<int32_t>(M);
    p.nn1_list2 = a.take<int32_t>(M);
  } else {
    plan_knn(a, N, M, p.knn);
  }
  return a.ok();
}

// view of the 1-NN search result left in `ws` by pcs_radius_graph_search(max_nbr = 1): per query
// the neighbour index (nbr_idx, valid where deg == 1) -- used by the native tracker
// (pcs_tracker.hip), which filters the edges on the device instead of emitting them
int radius_nn1_view(int64_t N, int64_t M, void* ws, size_t ws_bytes, const int32_t** nbr_idx,
                    const int32_t** deg, const int32_t** err) {
  Arena a(ws, ws_bytes);
  RadiusPlan p;
  if (!plan_radius(a, N, M, 1, p)) return PCS_ERR_WORKSPACE;
  *nbr_idx = p.nbr_idx;
  *deg = p.deg;
  *err = p.err;
  return PCS_OK;
}

// ------------------------------------------------------------------------------------------
// Literal torch_hash drop-ins on the reference table layout
// (keys[H] i64 = -1 empty, values[H,D] f32, reverse[H] i64; H = int(N / 0.5) chosen by the caller).
__device__ __forceinline__ int64_t ref_map2key(const int64_t* c, const int64_t* dims, int D) {
  int64_t ans = 0;
  for (int i = 0; i < D; ++i) {
    int64_t k = c[i];
    if (k >= dims[i]) k = dims[i];  // torch_hash_kernel.cu:36-38 clamps to dims[i], not dims[i]-1
    if (k < 0) k = 0;
    ans = ans * dims[i] + k;
  }
  return ans;
}
__device__ __forceinline__ int64_t ref_hashkey(int64_t key, int64_t H) {
  return ((key % H) * 999269 + 999437) % H;  // torch_hash_kernel.cu:27-28,49-51
}

__global__ void __launch_bounds__(kBlock)
table_insert_kernel(int64_t* __restrict__ keys, float* __restrict__ values,
                    int64_t* __restrict__ reverse, int64_t H, const int64_t* __restrict__ dims,
                    int D, const int64_t* __restrict__ ikeys, const float* __restrict__ ivals,
                    int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t c[8];
  for (int d = 0; d < D; ++d) c[d] = ikeys[i * D + d];
  int64_t key = ref_map2key(c, dims, D);
  int64_t slot = ref_hashkey(key, H);
  // every point takes its own slot (duplicates are not merged), torch_hash_kernel.cu:72-80
  while (atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)(-1LL),
                   (unsigned long long)key) != (unsigned long long)(-1LL))
    slot = (slot + 1) % H;
  for (int d = 0; d < D; ++d) values[slot * D + d] = ivals[i * D + d];
  reverse[slot] = i;
}

// One walk per query over the reference's cell combinations (dim 0 fastest), MODE 0 = count,
// 1 = fill (sorted or first-come by ascending (walk order is racy in the reference -> canonical:
// (dist2, ref index) when sorted, ref index when not)).
template <int MODE>
__global__ void __launch_bounds__(kBlock)
table_walk_kernel(const int64_t* __restrict__ keys, const float* __restrict__ values,
                  const int64_t* __restrict__ reverse, int64_t H, const int64_t* __restrict__ dims,
                  int D, const int64_t* __restrict__ qkeys, const float* __restrict__ qvals,
                  int64_t M, const int32_t* __restrict__ qmin, const int32_t* __restrict__ qmax,
                  const float* __restrict__ radius, int max_nbr, int sort_by_dist,
                  int32_t* __restrict__ degree, const int32_t* __restrict__ offset,
                  int64_t* __restrict__ edges, float* __restrict__ dists) {
  int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= M) return;
  int w[8], ncomb = 1;
  for (int d = 0; d < D; ++d) { w[d] = qmax[d] - qmin[d] + 1; ncomb *= w[d]; }
  float r = radius[q], r2 = r * r;
  int cap = (MODE == 1) ? degree[q] : max_nbr;
  int cnt = 0;
  int64_t* eptr = (MODE == 1) ? edges + (int64_t)offset[q] * 2 : nullptr;
  float* dptr = (MODE == 1) ? dists + offset[q] : nullptr;
  for (int c = 0; c < ncomb; ++c) {
    int64_t cc[8];
    int tmp = c;
    for (int d = 0; d < D; ++d) { cc[d] = qkeys[q * D + d] + (tmp % w[d] + qmin[d]); tmp /= w[d]; }
    int64_t key = ref_map2key(cc, dims, D);
    int64_t slot = ref_hashkey(key, H);
    while (keys[slot] != -1) {
      if (keys[slot] == key) {
        float dist2 = 0.f;
        for (int d = 0; d < D; ++d) {
          float di = values[slot * D + d] - qvals[q * D + d];
          dist2 = dist2 + di * di;
        }
        if (dist2 <= r2) {
          if (MODE == 0) {
            if (max_nbr == -1 || cnt < max_nbr) ++cnt;
          } else {
            int64_t ridx = reverse[slot];
            float kkey = sort_by_dist ? dist2 : 0.f;
            int pos = cnt;
            while (pos > 0) {
              float kd = sort_by_dist ? dptr[pos - 1] : 0.f;
              int64_t ki = eptr[(pos - 1) * 2];
              bool less = (kkey < kd) || (kkey == kd && ridx < ki);
              if (!less) break;
              if (pos < cap) { dptr[pos] = dptr[pos - 1]; eptr[pos * 2] = ki; eptr[pos * 2 + 1] = q; }
              --pos;
            }
            if (pos < cap) {
              eptr[pos * 2] = ridx;
              eptr[pos * 2 + 1] = q;
              dptr[pos] = dist2;
              if (cnt < cap) ++cnt;
            }
          }
        }
      }
      slot = (slot + 1) % H;
    }
  }
  if (MODE == 0) degree[q] = cnt;
}

// correspondence_kernel / points_in_radius_kernel of the reference (torch_hash_kernel.cu:96-155,
// 160-222) on its own table layout: one thread per query walks the cell combinations (dim 0
// fastest) and every probe chain. MODE 0: corres[q] = nearest hashed point, strict `<` against
// 1e10 (no radius); ties -- broken by the atomicCAS race of the reference's insert -- by the lower
// reference index. MODE 1: visited[ref] = 1 for every hashed point closer than `radius`.
template <int MODE>
__global__ void __launch_bounds__(kBlock)
table_near_kernel(const int64_t* __restrict__ keys, const float* __restrict__ values,
                  const int64_t* __restrict__ reverse, int64_t H, const int64_t* __restrict__ dims,
                  int D, const int64_t* __restrict__ qkeys, const float* __restrict__ qvals, int64_t M,
                  const int32_t* __restrict__ qmin, const int32_t* __restrict__ qmax, float radius,
                  int64_t* __restrict__ out) {
  int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (q >= M) return;
  int w[8], ncomb = 1;
  for (int d = 0; d < D; ++d) { w[d] = qmax[d] - qmin[d] + 1; ncomb *= w[d]; }
  const float r2 = radius * radius;
  float best = 1e10f;
  int64_t bi = -1;
  for (int c = 0; c < ncomb; ++c) {
    int64_t cc[8];
    int tmp = c;
    for (int d = 0; d < D; ++d) { cc[d] = qkeys[q * D + d] + (tmp % w[d] + qmin[d]); tmp /= w[d]; }
    int64_t key = ref_map2key(cc, dims, D);
    int64_t slot = ref_hashkey(key, H);
    while (keys[slot] != -1) {
      if (keys[slot] == key) {
        float dist2 = 0.f;
        for (int d = 0; d < D; ++d) {
          float di = values[slot * D + d] - qvals[q * D + d];
          dist2 = dist2 + di * di;
        }
        if (MODE == 0) {
          int64_t ri = reverse[slot];
          if (dist2 < best || (dist2 == best && bi >= 0 && ri < bi)) { best = dist2; bi = ri; }
        } else if (dist2 < r2) {
          out[reverse[slot]] = 1;
        }
      }
      slot = (slot + 1) % H;
    }
  }
  if (MODE == 0) out[q] = bi;
}

}  // namespace pcs

using namespace pcs;

extern "C" {

int pcs_minmax4_f32(const float* pts, int64_t n, float* out8, hipStream_t s);

size_t pcs_radius_graph_workspace_bytes(int64_t N, int64_t M, int max_nbr) {
  Arena a(nullptr, 0);
  RadiusPlan p;
  int K = max_nbr < 1 ? 1 : (max_nbr > kMaxK ? kMaxK : max_nbr);
  plan_radius(a, N, M, K, p);
  return a.off;
}

// Phase 1: grid build + search + degree scan. *d_num_edges = E (or -1 if the extent / cell_size
// ratio overflowed the 17-bit fine-cell coordinates: discard and retry with a larger cell).
// cell_size <= 0 picks radius/2.
int pcs_radius_graph_search(const float* ref, int64_t N, const float* query, int64_t M,
                            float radius, float cell_size, int dframe_min, int dframe_max,
                            int max_nbr, int sort_by_dist, int64_t* d_num_edges, void* ws,
                            size_t ws_bytes, hipStream_t s) {
  if (!d_num_edges || !ws || N < 0 || M < 0 || (N > 0 && !ref) || (M > 0 && !query) ||
      !(radius > 0.f) || dframe_min > dframe_max)
    return PCS_ERR_ARG;
  if (max_nbr < 1 || max_nbr > kMaxK) return PCS_ERR_ARG;  // unbounded / >32: table_* entry points
  if (N >= (1LL << 30) || M >= (1LL << 30) || M * (int64_t)max_nbr >= (1LL << 31)) return PCS_ERR_RANGE;
  if (N == 0 || M == 0) { PCS_CHECK_HIP(hipMemsetAsync(d_num_edges, 0, sizeof(int64_t), s)); return PCS_OK; }
  int K = max_nbr;
  Arena a(ws, ws_bytes);
  RadiusPlan p;
  if (!plan_radius(a, N, M, K, p)) return PCS_ERR_WORKSPACE;
  // cell_size > 0: fine cell; for max_nbr == 1 (sorted, one frame offset) it selects the two-level
  // grid, which needs cell_size >= (spatial radius) / 4. cell_size < 0: |cell_size| on the general
  // engine whatever max_nbr is. 0: radius / 2 on the general engine.
  float cell = cell_size != 0.f ? fabsf(cell_size) : radius * 0.5f;
  if (cell > radius) cell = radius;
  // several sorted neighbours under one frame offset (the 32-NN graphs of ClusterProposal): the cell-sorted,
  // wave-cooperative search of pcs_knn.hip on cells of its own choice (radius / 4), whatever cell_size says
  const bool tiled = K > 1 && sort_by_dist && dframe_min == dframe_max && !g_radius_engine_only;
  if (tiled) cell = knn_cell(radius);
  int rc = pcs_minmax4_f32(ref, N, p.b_ref, s);
  if (rc) return rc;
  if ((const void*)query == (const void*)ref && M == N) {
    PCS_CHECK_HIP(hipMemcpyAsync(p.b_qry, p.b_ref, 8 * sizeof(float), hipMemcpyDeviceToDevice, s));
  } else {
    rc = pcs_minmax4_f32(query, M, p.b_qry, s);
    if (rc) return rc;
  }
  grid_setup_kernel<<<1, 64, 0, s>>>(p.b_ref, p.b_qry, radius, cell, p.grid);
  PCS_CHECK_HIP(hipMemsetAsync(p.err, 0, sizeof(int32_t), s));
  if (tiled) {
    rc = knn_tiled_search((const float4*)ref, N, (const float4*)query, M, p.grid, dframe_min, K, p.knn, p.err,
                          p.nbr_idx, p.nbr_d, p.deg, s);
    if (rc) return rc;
    rc = exclusive_scan_i32(p.tmp, p.tmp_bytes, p.deg, p.off, M, s);
    if (rc) return rc;
    edge_total_kernel<<<1, 64, 0, s>>>(p.deg, p.off, (int)M, p.err, d_num_edges);
    PCS_LAUNCH_CHECK();
    return PCS_OK;
  }
  if (K == 1 && sort_by_dist && dframe_min == dframe_max && cell_size > 0.f) {
    // cell_size must be (spatial radius) / 4 or more so that the 3x3x3 coarse blocks cover the ball;
    // if the data say otherwise the search raises the overflow flag and *d_num_edges becomes -1
    g2_clear_kernel<<<grid_for((int64_t)p.g2.slot_words), kBlock, 0, s>>>(p.g2);
    int nb = (int)((N + kBlock - 1) / kBlock);
    g2_insert_kernel<<<nb, kBlock, 0, s>>>((const float4*)ref, (int)N, p.grid, p.g2, 0, nullptr, p.err);
    // queries with an empty neighbourhood leave after eight probes of one thread; the rest is searched
    PCS_CHECK_HIP(hipMemsetAsync(p.skeys, 0xFF, (size_t)p.Hs * sizeof(uint64_t), s));
    PCS_CHECK_HIP(hipMemsetAsync(p.g2_words, 0, 4 * sizeof(int32_t), s));
    nn1_super_insert_kernel<<<nb, kBlock, 0, s>>>((const float4*)ref, (int)N, p.grid, p.skeys, p.Hs - 1);
    nn1_filter_kernel<<<grid_for(M), kBlock, 0, s>>>((const float4*)query, (int)M, p.grid, p.skeys, p.Hs - 1, dframe_min,
                                                    p.nn1_list, p.g2_words, p.nbr_idx, p.nbr_d, p.deg);
    {
      // the survivors: one thread each through the 27 fine cells around them (settles the queries whose nearest point
      // is nearer than a cell width), the cooperative search for the rest (g2_words[1] counts them)
      int64_t blocks = (M * 32 + kBlock - 1) / kBlock;
      if (blocks > 4096) blocks = 4096;
      const int32_t* todo = p.nn1_list;
      const int32_t* todo_n = p.g2_words;
      if (!g_nn1_coop_only) {
        // (the spilled points by cell: the searches then look at them with their cells instead of scanning the list)
        nn1_spill_sort_kernel<<<1, 1024, 0, s>>>(p.grid, p.g2, p.g2_words + 2);
        nn1_near_kernel<<<grid_for(M), kBlock, 0, s>>>((const float4*)query, p.nn1_list, p.g2_words, p.grid, p.g2, dframe_min,
                                                      p.nn1_list2, p.g2_words + 1, p.nbr_idx, p.nbr_d, p.deg,
                                                      p.g2_words + 2);
        todo = p.nn1_list2;
        todo_n = p.g2_words + 1;
      }
      nn1_search_list_kernel<<<(int)blocks, kBlock, 0, s>>>((const float4*)query, todo, todo_n, p.grid, p.g2,
                                                            dframe_min, p.err, p.nbr_idx, p.nbr_d, p.deg, p.g2_words + 2);
    }
    rc = exclusive_scan_i32(p.tmp, p.tmp_bytes, p.deg, p.off, M, s);
    if (rc) return rc;
    edge_total_kernel<<<1, 64, 0, s>>>(p.deg, p.off, (int)M, p.err, d_num_edges);
    PCS_LAUNCH_CHECK();
    return PCS_OK;
  }
  PCS_CHECK_HIP(hipMemsetAsync(p.keys, 0xFF, (size_t)p.H * sizeof(uint64_t), s));
  PCS_CHECK_HIP(hipMemsetAsync(p.head, 0xFF, (size_t)p.H * sizeof(int32_t), s));
  uint32_t hmask = p.H - 1;
  frame_range_init_kernel<<<4, kBlock, 0, s>>>(p.fr);
  grid_insert_kernel<<<(int)((N + kBlock - 1) / kBlock), kBlock, 0, s>>>(
      (const float4*)ref, (int)N, p.grid, p.keys, p.head, p.next, hmask, p.fr, p.err);
  if (K <= 8) {
    constexpr int B = 256;
    size_t sh = (size_t)K * B * 8;
    radius_search_kernel<B><<<(int)((M + B - 1) / B), B, sh, s>>>(
        (const float4*)ref, (const float4*)query, (int)M, p.grid, p.keys, p.head, p.next, hmask,
        p.fr, dframe_min, dframe_max, K, sort_by_dist, p.nbr_idx, p.nbr_d, p.deg);
  } else {
    constexpr int B = 128;
    size_t sh = (size_t)K * B * 8;
    radius_search_kernel<B><<<(int)((M + B - 1) / B), B, sh, s>>>(
        (const float4*)ref, (const float4*)query, (int)M, p.grid, p.keys, p.head, p.next, hmask,
        p.fr, dframe_min, dframe_max, K, sort_by_dist, p.nbr_idx, p.nbr_d, p.deg);
  }
  rc = exclusive_scan_i32(p.tmp, p.tmp_bytes, p.deg, p.off, M, s);
  if (rc) return rc;
  edge_total_kernel<<<1, 64, 0, s>>>(p.deg, p.off, (int)M, p.err, d_num_edges);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// Phase 2: write the E edges found by the search still held in `ws` (same N, M, max_nbr).
// e_ref[E], e_query[E] int64; dists[E] (squared, as the reference's dists buffer) may be NULL.
int pcs_radius_graph_emit(int64_t N, int64_t M, int max_nbr, int64_t* e_ref, int64_t* e_query,
                          float* dists, const void* ws, size_t ws_bytes, hipStream_t s) {
  if (!ws || M < 0 || max_nbr < 1 || max_nbr > kMaxK) return PCS_ERR_ARG;
  if (M == 0 || N == 0) return PCS_OK;
  if (!e_ref || !e_query) return PCS_ERR_ARG;
  Arena a((void*)ws, ws_bytes);
  RadiusPlan p;
  if (!plan_radius(a, N, M, max_nbr, p)) return PCS_ERR_WORKSPACE;
  edge_emit_kernel<<<grid_for(M), kBlock, 0, s>>>(p.nbr_idx, p.nbr_d, p.deg, p.off, (int)M, max_nbr,
                                                  e_ref, e_query, dists);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

int pcs_radius_graph_engine_only(int on) {
  g_nn1_coop_only = on == 2 ? 1 : 0;
  g_radius_engine_only = on == 1 ? 1 : 0;
  return PCS_OK;
}

// diagnostics: the counters the last cell-sorted search left in `ws` (host_out[8]: runs, work cursor, queries that
// took the one-thread fallback, its cursor, then build-dependent statistics); synchronises the stream
int pcs_radius_graph_counters(int64_t N, int64_t M, int max_nbr, const void* ws, size_t ws_bytes, int32_t* host_out,
                              hipStream_t s) {
  if (!ws || !host_out || max_nbr < 1 || max_nbr > kMaxK) return PCS_ERR_ARG;
  Arena a((void*)ws, ws_bytes);
  RadiusPlan p;
  if (!plan_radius(a, N, M, max_nbr, p)) return PCS_ERR_WORKSPACE;
  if (max_nbr == 1) {  // the 1-NN path: [0] = queries that passed the super-block filter (searched cooperatively)
    for (int i = 0; i < 8; ++i) host_out[i] = 0;
    PCS_CHECK_HIP(hipMemcpyAsync(host_out, p.g2_words, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    PCS_CHECK_HIP(hipMemcpyAsync(host_out + 2, p.g2.ovf, sizeof(int32_t), hipMemcpyDeviceToHost, s));  // spilled points
    PCS_CHECK_HIP(hipStreamSynchronize(s));
    return PCS_OK;
  }
  PCS_CHECK_HIP(hipMemcpyAsync(host_out, p.knn.cnt, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PCS_CHECK_HIP(hipStreamSynchronize(s));
  return PCS_OK;
}

size_t pcs_radius_graph_components_workspace_bytes(int64_t M, int n_cut) {
  return cc_table_workspace_bytes(M, n_cut);
}

// Connected components of the graph the last search left in `ws` (ref and query the SAME cloud, N == M), cut at
// n_cut radii: an edge (neighbour, query) counts for cut i when its dist2 <= radius_cut[i]^2 (fp32 product of the
// fp32 radius, as the host's `dists <= r * r` test forms it). component [n_cut, M], d_num [n_cut].
int pcs_radius_graph_components(int64_t N, int64_t M, int max_nbr, const float* radius_cut, int n_cut,
                                int64_t* component, int64_t* d_num, const void* ws, size_t ws_bytes, void* ws_cc,
                                size_t ws_cc_bytes, hipStream_t s) {
  if (!ws || !ws_cc || !radius_cut || !d_num || N != M || M < 0 || max_nbr < 1 || max_nbr > kMaxK || n_cut < 1 ||
      n_cut > 8 || (M > 0 && !component))
    return PCS_ERR_ARG;
  if (M >= (1LL << 31) / 8) return PCS_ERR_RANGE;
  if (M == 0) { PCS_CHECK_HIP(hipMemsetAsync(d_num, 0, (size_t)n_cut * sizeof(int64_t), s)); return PCS_OK; }
  Arena a((void*)ws, ws_bytes);
  RadiusPlan p;
  if (!plan_radius(a, N, M, max_nbr, p)) return PCS_ERR_WORKSPACE;
  float r2[8];
  for (int i = 0; i < n_cut; ++i) r2[i] = radius_cut[i] * radius_cut[i];
  return cc_table_components(p.nbr_idx, p.nbr_d, p.deg, M, max_nbr, r2, n_cut, component, d_num, ws_cc, ws_cc_bytes, s);
}

// ---- torch_hash_cuda drop-ins (pcdet/ops/torch_hash/src/torch_hash_api.cpp:9-15) ----------
int pcs_hash_insert(int64_t* keys, float* values, int64_t* reverse, int64_t H, const int64_t* dims,
                    int D, const int64_t* insert_keys, const float* insert_values, int64_t n,
                    hipStream_t s) {
  if (!keys || !values || !reverse || !dims || H < 1 || D < 1 || D > 8 || n < 0) return PCS_ERR_ARG;
  if (n == 0) return PCS_OK;
  if (!insert_keys || !insert_values || n > H) return PCS_ERR_ARG;  // n > H would never terminate
  table_insert_kernel<<<(int)((n + kBlock - 1) / kBlock), kBlock, 0, s>>>(
      keys, values, reverse, H, dims, D, insert_keys, insert_values, n);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

size_t pcs_radius_graph_table_workspace_bytes(int64_t M) { return scan_i32_temp_bytes(M > 0 ? M : 1); }

// radius_graph_gpu phase 1 (count_radius_graph_degree_kernel + cumsum): degree[M], offset[M],
// *d_num_edges.
int pcs_radius_graph_table_count(const int64_t* keys, const float* values, const int64_t* reverse,
                                 int64_t H, const int64_t* dims, int D, const int64_t* query_keys,
                                 const float* query_values, int64_t M, const int32_t* qmin,
                                 const int32_t* qmax, const float* radius, int max_nbr,
                                 int32_t* degree, int32_t* offset, int64_t* d_num_edges, void* ws,
                                 size_t ws_bytes, hipStream_t s) {
  if (!keys || !values || !reverse || !dims || !qmin || !qmax || !radius || !degree || !offset ||
      !d_num_edges || !ws || D < 1 || D > 8 || M < 0 || H < 1)
    return PCS_ERR_ARG;
  if (M == 0) { PCS_CHECK_HIP(hipMemsetAsync(d_num_edges, 0, sizeof(int64_t), s)); return PCS_OK; }
  if (ws_bytes < scan_i32_temp_bytes(M)) return PCS_ERR_WORKSPACE;
  table_walk_kernel<0><<<(int)((M + kBlock - 1) / kBlock), kBlock, 0, s>>>(
      keys, values, reverse, H, dims, D, query_keys, query_values, M, qmin, qmax, radius, max_nbr,
      0, degree, nullptr, nullptr, nullptr);
  int rc = exclusive_scan_i32(ws, ws_bytes, degree, offset, M, s);
  if (rc) return rc;
  edge_total_kernel<<<1, 64, 0, s>>>(degree, offset, (int)M, nullptr, d_num_edges);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// radius_graph_gpu phase 2 (radius_graph_kernel): edges [E,2] int64 (ref, query), dists [E].
int pcs_radius_graph_table_fill(const int64_t* keys, const float* values, const int64_t* reverse,
                                int64_t H, const int64_t* dims, int D, const int64_t* query_keys,
                                const float* query_values, int64_t M, const int32_t* qmin,
                                const int32_t* qmax, const float* radius, int sort_by_dist,
                                const int32_t* degree, const int32_t* offset, int64_t* edges,
                                float* dists, hipStream_t s) {
  if (!keys || !values || !reverse || !dims || !qmin || !qmax || !radius || !degree || !offset ||
      D < 1 || D > 8 || M < 0 || H < 1)
    return PCS_ERR_ARG;
  if (M == 0) return PCS_OK;
  if (!edges || !dists) return PCS_ERR_ARG;
  table_walk_kernel<1><<<(int)((M + kBlock - 1) / kBlock), kBlock, 0, s>>>(
      keys, values, reverse, H, dims, D, query_keys, query_values, M, qmin, qmax, radius, 0,
      sort_by_dist, (int32_t*)degree, offset, edges, dists);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// correspondence (torch_hash.h:20-22, torch_hash_kernel.cu:444-485), the reference's argument order:
// corres_indices [M] int64 = index (into the inserted array) of the nearest hashed point among the
// walked cells, -1 if there is none. query_keys is read only here (the reference adds the cell
// offsets in place and takes them back).
int pcs_correspondence(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                       const int64_t* dims, int D, const int64_t* query_keys, const float* query_values,
                       int64_t M, const int32_t* qmin, const int32_t* qmax, int64_t* corres_indices,
                       hipStream_t s) {
  if (!keys || !values || !reverse || !dims || !qmin || !qmax || D < 1 || D > 8 || M < 0 || H < 1)
    return PCS_ERR_ARG;
  if (M == 0) return PCS_OK;
  if (!query_keys || !query_values || !corres_indices) return PCS_ERR_ARG;
  table_near_kernel<0><<<(int)((M + kBlock - 1) / kBlock), kBlock, 0, s>>>(
      keys, values, reverse, H, dims, D, query_keys, query_values, M, qmin, qmax, 0.f, corres_indices);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// points_in_radius_gpu (torch_hash.h:29-32, torch_hash_kernel.cu:563-605): visited [N] int64 (the
// caller zeroes it) gets 1 for every hashed point strictly closer than `radius` to some query.
int pcs_points_in_radius(const int64_t* keys, const float* values, const int64_t* reverse, int64_t H,
                         const int64_t* dims, int D, const int64_t* query_keys, const float* query_values,
                         int64_t M, const int32_t* qmin, const int32_t* qmax, float radius, int64_t* visited,
                         hipStream_t s) {
  if (!keys || !values || !reverse || !dims || !qmin || !qmax || D < 1 || D > 8 || M < 0 || H < 1)
    return PCS_ERR_ARG;
  if (M == 0) return PCS_OK;
  if (!query_keys || !query_values || !visited) return PCS_ERR_ARG;
  table_near_kernel<1><<<(int)((M + kBlock - 1) / kBlock), kBlock, 0, s>>>(
      keys, values, reverse, H, dims, D, query_keys, query_values, M, qmin, qmax, radius, visited);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

}  // extern "C"
