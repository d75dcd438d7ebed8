// k nearest neighbours within a radius, block-sorted and wave-cooperative, for gfx950.
//
// Reference: RadiusGraph.build_graph with SORT_BY_DIST and MAX_NUM_NEIGHBORS = 32
// (pcdet/models/model_utils/graph_utils.py:149-209; count_radius_graph_degree_kernel / radius_graph_kernel,
// pcdet/ops/torch_hash/src/torch_hash_kernel.cu:224-409), as ClusterProposal calls it on every 10-frame
// chunk (pcdet/models/registration/preprocessors/cluster_proposal.py:57-83). The edge set is the one
// pcs_radius_graph.hip states: per query the max_nbr nearest hashed points of frame cell fq + df with
// dist2 = sum_{d=0..3} (ref_d - query_d)^2 <= radius^2 (fp32, d ascending), ascending (dist2, index).
//
// The one-thread-per-query engine walked chained cells of radius / 2 through a hash (two dependent loads
// per candidate, ~1700 L1 accesses per query) and kept its top 32 by insertion into LDS, the wave taking
// the insertion path whenever ANY lane accepted. Here:
//   * both clouds are SORTED BY BLOCK, then by fine cell inside the block (h = radius / 4, a block = 4 x 4 x 4
//     cells = one radius cubed; one radix sort): a block is one contiguous run of 16-byte rows, found with one
//     16-byte table probe, and the 3 x 3 x 3 blocks around a block hold every point within the radius of any
//     point of it;
//   * a RUN = the queries of one block. A wave owns a run: 27 probes, then the points of the 27 blocks stream
//     through the wave 64 at a time -- every lane loads ONE row (coalesced) and its cell bits, the lanes whose
//     row belongs to the cells wanted now are ballotted, and those rows are broadcast one by one (v_readlane)
//     while every lane tests its own query against them: no LDS and no memory traffic in the inner loop, 64
//     queries per instruction;
//   * the cells wanted are taken RING by ring (Chebyshev distance 1, 2, 3, 4 in cells around the block); the
//     wave stops at the first ring whose box holds the k-th neighbour of every query of the run -- one or two
//     rings on a LiDAR surface, all four where a query has fewer than k points within the radius; from the
//     second ring on only the cells inside the box around the balls of the queries still open are taken;
//   * selection is by THRESHOLD, not by insertion. Pass A counts every query's in-radius candidates into a
//     private histogram of dist2 (8 bins per octave: LDS adds without return); the bin in which the count
//     reaches k gives tau >= the k-th distance, and the ring test compares tau with the distance from the
//     query to the faces of the box seen so far. Pass B streams the same cells again and appends the
//     candidates with dist2 <= tau (k plus the few that share the last bin) to the query's row in LDS; the
//     row is sorted once and its first k entries leave in 128-byte rows.
// Exact by construction: every point with dist2 <= tau has been seen, and (dist2, index) orders them.
// A row that fills up (more than kKnnCap - k candidates in the k-th one's bin: a few queries in a thousand) keeps
// its kKnnCap smallest entries. Queries with more in-radius candidates than the 16-bit counters hold (60 000: no
// LiDAR cloud) go to a list and are searched by one thread each (knn_hard_kernel) on the same sorted layout.
#include "pcs_knn.hpp"

#include <climits>

namespace pcs {

constexpr int kKnnCap = 40;                    // candidates a query's row takes in pass B
constexpr int kKnnRow = kKnnCap + 1;           // 64-bit words per lane (odd: the lanes' rows start in different banks)
constexpr int kKnnBins = 96;                   // 12 octaves of dist2 below radius^2, 8 bins each
constexpr int kKnnHistRow = kKnnBins / 2 + 1;  // dwords per lane: two 16-bit counters per dword, odd stride
constexpr int kKnnShift = 20;                  // float bits >> 20: exponent and three mantissa bits
constexpr int kKnnCountLimit = 60000;          // in-radius candidates per query the 16-bit counters are good for

void plan_knn(Arena& a, int64_t N, int64_t M, KnnPlan& p) {
  if (N < 1) N = 1;
  if (M < 1) M = 1;
  for (int b = 0; b < 2; ++b) {
    p.rkey[b] = a.take<uint64_t>(N);
    p.rval[b] = a.take<int32_t>(N);
    p.qkey[b] = a.take<uint64_t>(M);
    p.qval[b] = a.take<int32_t>(M);
  }
  p.rpts = a.take<float4>(N);
  p.qpts = a.take<float4>(M);
  uint32_t h = 64;
  while (h < 2 * (uint64_t)N) h <<= 1;
  p.H = h;
  p.slot = a.take<KnnSlot>(h);
  p.runflag = a.take<int32_t>(M);
  p.runid = a.take<int32_t>(M);
  p.run_start = a.take<int32_t>(M + 1);
  p.cnt = a.take<int32_t>(8);
  p.hard = a.take<int32_t>(M);
  size_t t1 = sort_pairs_u64_temp_bytes((size_t)(N > M ? N : M));
  size_t t2 = scan_i32_temp_bytes((size_t)M);
  p.tmp_bytes = t1 > t2 ? t1 : t2;
  p.tmp = a.take<char>(p.tmp_bytes);
}

// sort key: frame cell, block (z, y, x: 15 bits each), cell inside the block (z, y, x: 2 bits each) -- the points
// of a block are consecutive in the sort order, its cells in ascending order inside it
__device__ __forceinline__ uint64_t knn_key(int fc, int x, int y, int z) {
  return ((uint64_t)(uint32_t)fc << 51) | ((uint64_t)(z >> 2) << 36) | ((uint64_t)(y >> 2) << 21) |
         ((uint64_t)(x >> 2) << 6) | (uint64_t)(((z & 3) << 4) | ((y & 3) << 2) | (x & 3));
}
__device__ __forceinline__ uint64_t knn_block_key(int fc, int bx, int by, int bz) {
  return ((uint64_t)(uint32_t)fc << 45) | ((uint64_t)bz << 30) | ((uint64_t)by << 15) | (uint64_t)bx;
}

__global__ void __launch_bounds__(kBlock)
knn_key_kernel(const float4* __restrict__ pts, int n, const GridDesc* __restrict__ gp, uint64_t* __restrict__ keys,
               int32_t* __restrict__ vals, int32_t* __restrict__ err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const GridDesc g = *gp;
  const float4 p = pts[i];
  const int fc = frame_cell(g, p.x);
  const int x = (int)floorf((p.y - g.lo[0]) / g.h), y = (int)floorf((p.z - g.lo[1]) / g.h),
            z = (int)floorf((p.w - g.lo[2]) / g.h);
  uint64_t key = ~0ULL >> 1;
  if (x < 0 || y < 0 || z < 0 || x >= kCoordMax || y >= kCoordMax || z >= kCoordMax || fc < 0 || fc >= kFrameCells)
    atomicExch(err, 1);  // extent / cell-size ratio beyond the 17-bit cell coordinates: the call's result is void
  else
    key = knn_key(fc, x, y, z);
  keys[i] = key;
  vals[i] = i;
}

// the hashed cloud in block order; every block's first point claims the block's table slot
__global__ void __launch_bounds__(kBlock)
knn_heads_kernel(const float4* __restrict__ ref, const uint64_t* __restrict__ key, const int32_t* __restrict__ val,
                 int n, KnnSlot* __restrict__ slot, uint32_t hmask, float4* __restrict__ rpts) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint64_t k = key[j] >> 6;
  rpts[j] = ref[val[j]];
  if (j > 0 && (key[j - 1] >> 6) == k) return;
  uint32_t s = mix64(k) & hmask;
  for (uint32_t t = 0; t <= hmask; ++t) {  // (<= n cells in >= 2 n slots: ends at an empty slot)
    const uint64_t prev = atomicCAS((unsigned long long*)&slot[s].key, (unsigned long long)kEmpty, (unsigned long long)k);
    if (prev == kEmpty) { slot[s].start = j; return; }
    s = (s + 1) & hmask;
  }
}
__global__ void __launch_bounds__(kBlock)
knn_tails_kernel(const uint64_t* __restrict__ key, int n, KnnSlot* __restrict__ slot, uint32_t hmask) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint64_t k = key[j] >> 6;
  if (j + 1 < n && (key[j + 1] >> 6) == k) return;
  uint32_t s = mix64(k) & hmask;
  for (uint32_t t = 0; t <= hmask; ++t) {
    const uint64_t have = slot[s].key;
    if (have == k) { slot[s].end = j + 1; return; }
    if (have == kEmpty) return;
    s = (s + 1) & hmask;
  }
}

// query runs: a run = the sorted queries of one block
__global__ void __launch_bounds__(kBlock)
knn_runflag_kernel(const float4* __restrict__ query, const uint64_t* __restrict__ key, const int32_t* __restrict__ val,
                   int n, int32_t* __restrict__ flag, float4* __restrict__ qpts) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (qpts) qpts[j] = query[val[j]];
  flag[j] = (j == 0 || (key[j - 1] >> 6) != (key[j] >> 6)) ? 1 : 0;
}
__global__ void __launch_bounds__(kBlock)
knn_runs_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ runid, int n,
                int32_t* __restrict__ run_start, int32_t* __restrict__ cnt) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (flag[j]) run_start[runid[j] - 1] = j;
  if (j == n - 1) {
    cnt[0] = runid[j];
    run_start[runid[j]] = n;
  }
}
__global__ void knn_reset_kernel(int32_t* __restrict__ cnt) {
  if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
}

struct KnnArgs {
  const float4* rpts;        // hashed cloud in block order
  const uint64_t* rkey;      // its sort keys (the low six bits: the cell inside the block)
  const int32_t* rval;       // original indices of rpts rows
  const float4* qpts;        // queries in cell order
  const uint64_t* qkey;
  const int32_t* qval;
  const KnnSlot* slot;
  const int32_t* run_start;
  int32_t* cnt;
  int32_t* hard;
  const GridDesc* grid;
  int32_t* nbr_idx;
  float* nbr_d;
  int32_t* deg;
  uint32_t hmask;
  int df, K, M;
};

__device__ __forceinline__ bool knn_probe(const KnnSlot* __restrict__ slot, uint32_t hmask, uint64_t key, int& st,
                                          int& cn) {
  uint32_t s = mix64(key) & hmask;
  for (uint32_t t = 0; t <= hmask; ++t) {
    const uint4 w = ((const uint4*)slot)[s];
    const uint64_t k = ((uint64_t)w.y << 32) | w.x;
    if (k == key) { st = (int)w.z; cn = (int)w.w - (int)w.z; return cn > 0; }
    if (k == kEmpty) return false;
    s = (s + 1) & hmask;
  }
  return false;
}

__device__ __forceinline__ float knn_bcast(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

__global__ void __launch_bounds__(64)
knn_search_kernel(KnnArgs a) {
  __shared__ unsigned long long s_u[64 * kKnnRow];  // pass A: the lanes' histograms; pass B: the lanes' candidate rows
  __shared__ int s_bstart[32];
  __shared__ int s_bpref[33];
  __shared__ int s_run;
  const int lane = threadIdx.x;
  const GridDesc g = *a.grid;
  const float r2 = g.radius2, h = g.h;
  const float margin = 1e-5f * r2 + 1e-6f;
  const int K = a.K;
  const int base_bin = (int)(__float_as_uint(r2) >> kKnnShift) - (kKnnBins - 1);
  unsigned* const hist = (unsigned*)s_u + lane * kKnnHistRow;
  unsigned long long* const row = s_u + lane * kKnnRow;
  for (;;) {
    __syncthreads();
    if (lane == 0) s_run = atomicAdd(&a.cnt[1], 1);
    __syncthreads();
    const int r = __builtin_amdgcn_readfirstlane(s_run);
    if (r >= __builtin_amdgcn_readfirstlane(a.cnt[0])) break;  // (uniform: every wave ends here)
    const int j0 = __builtin_amdgcn_readfirstlane(a.run_start[r]), j1 = __builtin_amdgcn_readfirstlane(a.run_start[r + 1]);
    const uint64_t key0v = a.qkey[j0];
    const uint64_t key0 = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(key0v >> 32)) << 32) |
                          (unsigned)__builtin_amdgcn_readfirstlane((int)(key0v & 0xFFFFFFFFULL));
    const int fc = (int)(key0 >> 51), BZ = (int)((key0 >> 36) & 0x7FFF), BY = (int)((key0 >> 21) & 0x7FFF),
              BX = (int)((key0 >> 6) & 0x7FFF);
    const int X0 = 4 * BX, Y0 = 4 * BY, Z0 = 4 * BZ;
    const int tfc = fc + a.df;
    // the 27 blocks around the run's: lane t < 27 probes block (BX + t % 3 - 1, BY + (t / 3) % 3 - 1, BZ + t / 9 - 1)
    {
      int st = 0, cn = 0;
      if (lane < 27 && tfc >= 0 && tfc < kFrameCells) {
        const int bx = BX + lane % 3 - 1, by = BY + (lane / 3) % 3 - 1, bz = BZ + lane / 9 - 1;
        if (bx >= 0 && by >= 0 && bz >= 0 && bx < (kCoordMax >> 2) && by < (kCoordMax >> 2) && bz < (kCoordMax >> 2))
          knn_probe(a.slot, a.hmask, knn_block_key(tfc, bx, by, bz), st, cn);
      }
      int v = cn;
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        const int tt = __shfl_up(v, o, 64);
        if (lane >= o) v += tt;
      }
      if (lane < 32) { s_bstart[lane] = st; s_bpref[lane + 1] = v; }
      if (lane == 0) s_bpref[0] = 0;
    }
    __syncthreads();
    const int T = __builtin_amdgcn_readfirstlane(s_bpref[27]);
    for (int qb = j0; qb < j1; qb += 64) {
      const int j = qb + lane;
      const bool act = j < j1;
      const float4 p = act ? a.qpts[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      const int qi = act ? a.qval[j] : -1;
      const float fx = (p.y - g.lo[0]) / h, fy = (p.z - g.lo[1]) / h, fz = (p.w - g.lo[2]) / h;
      __syncthreads();  // (the rows of the last batch have left)
#pragma unroll 1
      for (int i = 0; i < kKnnHistRow; ++i) hist[i] = 0u;
      int total = 0;    // this query's in-radius candidates
      float tau = r2;
      int nrow = 0;     // candidates in this query's row (pass B)
      int maxpos = -1;  // a full row: where its largest entry is
      // cells that can still matter: the box around the balls of the queries that are not done (every cell, at first)
      int bx0 = 0, bx1 = kCoordMax, by0 = 0, by1 = kCoordMax, bz0 = 0, bz1 = kCoordMax;
      // One pass over the points of the 27 blocks. Pass A (pb = false), ring s: the rows of the cells at Chebyshev
      // distance s from the block (ring 1: <= 1) inside the box above go into the histograms. Pass B: the rows of the
      // cells up to ring s, those with dist2 <= tau into the lanes' rows.
      auto stream = [&](int s, bool pb) {
        for (int f0 = 0; f0 < T; f0 += 64) {
          const int f = f0 + lane;
          const bool has = f < T;
          int lo = 0, hi = 27;  // the block of flat index f: s_bpref[lo] <= f < s_bpref[lo + 1]
#pragma unroll
          for (int it = 0; it < 5; ++it) {
            const int mid = (lo + hi) >> 1;
            if (s_bpref[mid] <= f) lo = mid; else hi = mid;
          }
          float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
          int ci = -1;
          bool sel = false;
          if (has) {
            const int src = s_bstart[lo] + (f - s_bpref[lo]);
            const int kb = (int)(a.rkey[src] & 63ULL);
            const int x = 4 * (BX + lo % 3 - 1) + (kb & 3), y = 4 * (BY + (lo / 3) % 3 - 1) + ((kb >> 2) & 3),
                      z = 4 * (BZ + lo / 9 - 1) + (kb >> 4);
            const int ring = max(max(max(X0 - x, x - X0 - 3), max(Y0 - y, y - Y0 - 3)), max(max(Z0 - z, z - Z0 - 3), 0));
            sel = pb ? ring <= s
                     : ((s == 1 ? ring <= 1 : ring == s) && x >= bx0 && x <= bx1 && y >= by0 && y <= by1 && z >= bz0 && z <= bz1);
            if (sel) {
              c = a.rpts[src];
              if (pb) ci = a.rval[src];
            }
          }
          unsigned long long m = __ballot(sel);
          if (!pb) {
            while (m) {
              const int jj = __builtin_ctzll(m);
              m &= m - 1ULL;
              const float d0 = knn_bcast(c.x, jj) - p.x, d1 = knn_bcast(c.y, jj) - p.y, d2 = knn_bcast(c.z, jj) - p.z,
                          d3 = knn_bcast(c.w, jj) - p.w;
              float dist2 = 0.f;  // torch_hash_kernel.cu:364-368: dims 0..3 in order, starting from 0
              dist2 = dist2 + d0 * d0;
              dist2 = dist2 + d1 * d1;
              dist2 = dist2 + d2 * d2;
              dist2 = dist2 + d3 * d3;
              if (act && dist2 <= r2) {
                const int b = max((int)(__float_as_uint(dist2) >> kKnnShift) - base_bin, 0);
                atomicAdd(&hist[b >> 1], 1u << ((b & 1) << 4));
                ++total;
              }
            }
          } else {
            while (m) {
              const int jj = __builtin_ctzll(m);
              m &= m - 1ULL;
              const float d0 = knn_bcast(c.x, jj) - p.x, d1 = knn_bcast(c.y, jj) - p.y, d2 = knn_bcast(c.z, jj) - p.z,
                          d3 = knn_bcast(c.w, jj) - p.w;
              float dist2 = 0.f;
              dist2 = dist2 + d0 * d0;
              dist2 = dist2 + d1 * d1;
              dist2 = dist2 + d2 * d2;
              dist2 = dist2 + d3 * d3;
              const int cj = __builtin_amdgcn_readlane(ci, jj);
              if (act && dist2 <= tau) {
                const unsigned long long word = ((unsigned long long)__float_as_uint(dist2) << 32) | (unsigned)cj;
                if (nrow < kKnnCap) {
                  row[nrow] = word;
                  ++nrow;
                } else {
                  // a full row (more than kKnnCap - k candidates share the k-th one's bin: rare) keeps its kKnnCap
                  // smallest: the newcomer replaces the largest, if it is smaller
                  if (maxpos < 0) {
                    maxpos = 0;
                    for (int q = 1; q < kKnnCap; ++q) if (row[q] > row[maxpos]) maxpos = q;
                  }
                  if (word < row[maxpos]) {
                    row[maxpos] = word;
                    maxpos = 0;
                    for (int q = 1; q < kKnnCap; ++q) if (row[q] > row[maxpos]) maxpos = q;
                  }
                }
              }
            }
          }
        }
      };
      // ---- pass A: ring by ring until every query of the batch has its k-th neighbour inside the box
      int s_last = 0;
      for (int s = 1; s <= kKnnRings && T > 0; ++s) {
        s_last = s;
        stream(s, false);
        // the k-th distance is at most tau (the upper edge of the bin in which the count reaches k); a point not yet
        // seen lies outside the box, at least `e` away, or outside the ball this query had after the ring before
        bool done = !act;
        if (act && total >= K) {
          int cum = 0, bf = -1;
#pragma unroll 1
          for (int i = 0; i < kKnnBins / 2; ++i) {
            const unsigned w = hist[i];
            cum += (int)(w & 0xFFFFu);
            if (bf < 0 && cum >= K) bf = 2 * i;
            cum += (int)(w >> 16);
            if (bf < 0 && cum >= K) bf = 2 * i + 1;
          }
          const float edge = __uint_as_float(((unsigned)(bf + base_bin + 1) << kKnnShift) - 1u);
          tau = fminf(edge, r2);
          const float ex = fminf(fx - (float)(X0 - s), (float)(X0 + 4 + s) - fx);
          const float ey = fminf(fy - (float)(Y0 - s), (float)(Y0 + 4 + s) - fy);
          const float ez = fminf(fz - (float)(Z0 - s), (float)(Z0 + 4 + s) - fz);
          const float e = fmaxf(fminf(fminf(ex, ey), ez), 0.f) * h * 0.9999f;
          done = tau <= e * e - margin;
        }
        if (__ballot(!done) == 0ULL) break;
        if (s < kKnnRings) {
          // the next ring's cells matter only where the ball of a query that is not done reaches (tau only shrinks)
          const float rho = sqrtf(tau + margin) / h * 1.0001f + 1e-3f;
          int lx0 = done ? INT_MAX : (int)floorf(fx - rho), lx1 = done ? INT_MIN : (int)floorf(fx + rho);
          int ly0 = done ? INT_MAX : (int)floorf(fy - rho), ly1 = done ? INT_MIN : (int)floorf(fy + rho);
          int lz0 = done ? INT_MAX : (int)floorf(fz - rho), lz1 = done ? INT_MIN : (int)floorf(fz + rho);
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            lx0 = min(lx0, __shfl_xor(lx0, o, 64)); lx1 = max(lx1, __shfl_xor(lx1, o, 64));
            ly0 = min(ly0, __shfl_xor(ly0, o, 64)); ly1 = max(ly1, __shfl_xor(ly1, o, 64));
            lz0 = min(lz0, __shfl_xor(lz0, o, 64)); lz1 = max(lz1, __shfl_xor(lz1, o, 64));
          }
          bx0 = __builtin_amdgcn_readfirstlane(lx0); bx1 = __builtin_amdgcn_readfirstlane(lx1);
          by0 = __builtin_amdgcn_readfirstlane(ly0); by1 = __builtin_amdgcn_readfirstlane(ly1);
          bz0 = __builtin_amdgcn_readfirstlane(lz0); bz1 = __builtin_amdgcn_readfirstlane(lz1);
        }
      }
      // ---- pass B: the candidates up to tau, into the lanes' rows
      __syncthreads();  // (the histograms have been read)
      if (T > 0) stream(s_last, true);
      __syncthreads();
      const bool is_hard = act && total > kKnnCountLimit;  // (the 16-bit counters may have wrapped)
      const int n_mine = is_hard ? 0 : nrow;
      // ascending (dist2, index): insertion sort of the lane's own row
      for (int i = 1; i < n_mine; ++i) {
        const unsigned long long w = row[i];
        int q = i - 1;
        while (q >= 0) {
          const unsigned long long v = row[q];
          if (!(w < v)) break;
          row[q + 1] = v;
          --q;
        }
        row[q + 1] = w;
      }
      const int dq = is_hard ? -1 : min(n_mine, K);
      if (is_hard) a.hard[atomicAdd(&a.cnt[2], 1)] = j;
      else if (act) a.deg[qi] = dq;
#ifdef PCS_KNN_STATS
      if (act) {
        if (maxpos >= 0) atomicAdd(&a.cnt[5], 1);
        atomicAdd(&a.cnt[6], s_last);
        atomicAdd(&a.cnt[7], nrow);
        atomicAdd(&a.cnt[4], total);
      }
#endif
      __syncthreads();
      // the rows leave query by query: lanes 0..31 the indices, lanes 32..63 the distances, 128 bytes each
      const int nact = min(64, j1 - qb);
      const int e = lane & 31;
      for (int q = 0; q < nact; ++q) {
        const int oi = __builtin_amdgcn_readlane(qi, q);
        const int dd = __builtin_amdgcn_readlane(dq, q);
        if (e < dd) {
          const unsigned long long w = s_u[q * kKnnRow + e];
          if (lane < 32) a.nbr_idx[(int64_t)oi * K + e] = (int32_t)(unsigned)(w & 0xFFFFFFFFULL);
          else a.nbr_d[(int64_t)oi * K + e] = __uint_as_float((unsigned)(w >> 32));
        }
      }
    }
  }
}

// the queries the counters could not take: one thread each over the points of its 27 blocks, top k by insertion into
// LDS columns (the form of radius_search_kernel, pcs_radius_graph.hip), on the sorted layout
__global__ void __launch_bounds__(64)
knn_hard_kernel(KnnArgs a) {
  __shared__ unsigned long long sk[kKnnMaxK * 64];
  const int t = threadIdx.x;
  const int nh = a.cnt[2];
  const GridDesc g = *a.grid;
  const int K = a.K;
  for (int w = blockIdx.x * 64 + t; w < nh; w += gridDim.x * 64) {
    const int j = a.hard[w];
    const float4 p = a.qpts[j];
    const int qi = a.qval[j];
    const uint64_t key = a.qkey[j];
    const int tfc = (int)(key >> 51) + a.df, BZ = (int)((key >> 36) & 0x7FFF), BY = (int)((key >> 21) & 0x7FFF),
              BX = (int)((key >> 6) & 0x7FFF);
    int cnt = 0;
    if (tfc >= 0 && tfc < kFrameCells) {
      for (int b = 0; b < 27; ++b) {
        const int bx = BX + b % 3 - 1, by = BY + (b / 3) % 3 - 1, bz = BZ + b / 9 - 1;
        if (bx < 0 || by < 0 || bz < 0 || bx >= (kCoordMax >> 2) || by >= (kCoordMax >> 2) || bz >= (kCoordMax >> 2)) continue;
        int st = 0, cn = 0;
        if (!knn_probe(a.slot, a.hmask, knn_block_key(tfc, bx, by, bz), st, cn)) continue;
        for (int q = 0; q < cn; ++q) {
          const float4 c = a.rpts[st + q];
          const float d0 = c.x - p.x, d1 = c.y - p.y, d2 = c.z - p.z, d3 = c.w - p.w;
          float dist2 = 0.f;
          dist2 = dist2 + d0 * d0;
          dist2 = dist2 + d1 * d1;
          dist2 = dist2 + d2 * d2;
          dist2 = dist2 + d3 * d3;
          if (!(dist2 <= g.radius2)) continue;
          const unsigned long long word = ((unsigned long long)__float_as_uint(dist2) << 32) | (unsigned)a.rval[st + q];
          int pos = cnt;
          while (pos > 0) {
            const unsigned long long kw = sk[(pos - 1) * 64 + t];
            if (!(word < kw)) break;
            if (pos < K) sk[pos * 64 + t] = kw;
            --pos;
          }
          if (pos < K) {
            sk[pos * 64 + t] = word;
            if (cnt < K) ++cnt;
          }
        }
      }
    }
    a.deg[qi] = cnt;
    for (int e = 0; e < cnt; ++e) {
      const unsigned long long wv = sk[e * 64 + t];
      a.nbr_idx[(int64_t)qi * K + e] = (int32_t)(unsigned)(wv & 0xFFFFFFFFULL);
      a.nbr_d[(int64_t)qi * K + e] = __uint_as_float((unsigned)(wv >> 32));
    }
  }
}

// test hook (pcs_knn_force_hard): every query through the one-thread search
static int g_knn_force_hard = 0;
__global__ void __launch_bounds__(kBlock) knn_all_hard_kernel(int32_t* __restrict__ hard, int32_t* __restrict__ cnt, int M) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < M) hard[j] = j;
  if (j == 0) cnt[2] = M;
}

int knn_tiled_search(const float4* ref, int64_t N, const float4* query, int64_t M, const GridDesc* grid, int df,
                     int K, const KnnPlan& p, int32_t* err, int32_t* nbr_idx, float* nbr_d, int32_t* deg,
                     hipStream_t s) {
  const bool same = (const void*)ref == (const void*)query && N == M;
  const int nbN = (int)((N + kBlock - 1) / kBlock), nbM = (int)((M + kBlock - 1) / kBlock);
  knn_reset_kernel<<<1, 64, 0, s>>>(p.cnt);
  knn_key_kernel<<<nbN, kBlock, 0, s>>>(ref, (int)N, grid, p.rkey[0], p.rval[0], err);
  int rc = sort_pairs_u64(p.tmp, p.tmp_bytes, p.rkey[0], p.rkey[1], p.rval[0], p.rval[1], (size_t)N, 0, 63, s);
  if (rc) return rc;
  PCS_CHECK_HIP(hipMemsetAsync(p.slot, 0xFF, (size_t)p.H * sizeof(KnnSlot), s));
  knn_heads_kernel<<<nbN, kBlock, 0, s>>>(ref, p.rkey[1], p.rval[1], (int)N, p.slot, p.H - 1, p.rpts);
  knn_tails_kernel<<<nbN, kBlock, 0, s>>>(p.rkey[1], (int)N, p.slot, p.H - 1);
  const uint64_t* qkey = p.rkey[1];
  const int32_t* qval = p.rval[1];
  const float4* qpts = p.rpts;
  if (!same) {
    knn_key_kernel<<<nbM, kBlock, 0, s>>>(query, (int)M, grid, p.qkey[0], p.qval[0], err);
    rc = sort_pairs_u64(p.tmp, p.tmp_bytes, p.qkey[0], p.qkey[1], p.qval[0], p.qval[1], (size_t)M, 0, 63, s);
    if (rc) return rc;
    qkey = p.qkey[1];
    qval = p.qval[1];
    qpts = p.qpts;
  }
  knn_runflag_kernel<<<nbM, kBlock, 0, s>>>(query, qkey, qval, (int)M, p.runflag, same ? nullptr : p.qpts);
  rc = inclusive_scan_i32(p.tmp, p.tmp_bytes, p.runflag, p.runid, (size_t)M, s);
  if (rc) return rc;
  knn_runs_kernel<<<nbM, kBlock, 0, s>>>(p.runflag, p.runid, (int)M, p.run_start, p.cnt);
  KnnArgs a;
  a.rpts = p.rpts;
  a.rkey = p.rkey[1];
  a.rval = p.rval[1];
  a.qpts = qpts;
  a.qkey = qkey;
  a.qval = qval;
  a.slot = p.slot;
  a.run_start = p.run_start;
  a.cnt = p.cnt;
  a.hard = p.hard;
  a.grid = grid;
  a.nbr_idx = nbr_idx;
  a.nbr_d = nbr_d;
  a.deg = deg;
  a.hmask = p.H - 1;
  a.df = df;
  a.K = K;
  a.M = (int)M;
  if (g_knn_force_hard) {
    knn_all_hard_kernel<<<nbM, kBlock, 0, s>>>(p.hard, p.cnt, (int)M);
  } else {
    // a queue of runs: as many waves as the chip holds at this kernel's LDS footprint, at most one per 16 queries
    int64_t waves = (M + 15) / 16;
    if (waves > 256 * 6) waves = 256 * 6;
    knn_search_kernel<<<(int)waves, 64, 0, s>>>(a);
  }
  {
    int64_t blocks = (M + 63) / 64;
    if (blocks > 2048) blocks = 2048;
    knn_hard_kernel<<<(int)blocks, 64, 0, s>>>(a);
  }
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

}  // namespace pcs

extern "C" {
// test hook: 1 = every query of the tiled search takes the one-thread path (cross-check of the two)
int pcs_knn_force_hard(int on) {
  pcs::g_knn_force_hard = on ? 1 : 0;
  return PCS_OK;
}
}
