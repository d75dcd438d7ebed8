// Two-level cell grid for the 1-NN searches of the fused ICP (pcs_icp.hip), the sorted 1-NN radius
// graph (pcs_radius_graph.hip) and the tracker's harvest.
//
// FINE cells of size h live in an open-addressing table; every fine slot owns a fixed run of
// kCellCap rows of the point array (a candidate cell is one coalescable run of float4 rows, no
// pointer chasing), so the whole build is ONE kernel: claim the slot, take a position with an
// atomic counter, store the point. Cells that fill up spill into a small overflow list that every
// query scans (voxel-subsampled clouds put a handful of points in a cell: the list is normally
// empty). A COARSE table of 4x4x4-fine-cell blocks keeps a 64-bit occupancy mask per block. With
// h = (spatial radius)/4 a query's whole search ball lies inside its 3x3x3 coarse neighbourhood, so
// a query tests 27 masks held in registers and touches memory only for occupied fine cells.
// Order inside a cell is arbitrary (atomic counter); the search result does not depend on it
// because ties resolve by (dist2, original index).
#pragma once
#include "pcs_grid.hpp"

#include <climits>

namespace pcs {

constexpr int kCellCap = 16;

// A table slot is one 16-byte word so that a probe fetches the key and its payload together (one
// dependent load instead of two): fine = {key, point count}, coarse = {key, 64-bit occupancy mask}.
struct FineSlot {
  uint64_t key;
  int32_t count;  // points inserted (may exceed kCellCap: the rest spilled)
  int32_t pad;
};
struct CoarseSlot {
  uint64_t key;
  uint64_t mask;  // occupancy of the 64 fine cells of the block
};

struct Grid2 {
  FineSlot* fslot;    // [Hf]
  CoarseSlot* cslot;  // [Hc]
  float4* spts;       // [Hf * kCellCap] points, slot-major
  int32_t* sidx;      // [Hf * kCellCap] original index of spts rows
  int32_t* ovf;       // [1] number of spilled points
  float4* ovf_pts;    // [N]
  int32_t* ovf_idx;   // [N]
  float* fr;          // [2*kFrameCells] frame-value range per frame cell
  uint32_t Hf, Hc;
  uint32_t ovf_cap;   // rows of ovf_pts / ovf_idx
  size_t slot_words;  // 16-byte words of fslot + cslot (contiguous), for the reset loop
};

static inline uint32_t g2_pow2(uint64_t x) { uint32_t h = 64; while (h < x) h <<= 1; return h; }

static inline void plan_grid2(Arena& a, int64_t N, Grid2& g) {
  if (N < 1) N = 1;
  // fine slots: the power of two above 2 N for N points. A cell per point is the worst case, so a build fills the
  // table to half at most; voxel-sampled clouds put ~3 points in an occupied cell: a sixth. Every slot owns kCellCap
  // rows (320 bytes with their indices), which is where the memory of a grid goes: the power of two above 1.5 N was
  // measured too (round 4): a sequence's tracker workspace 1.57 -> 1.08 GiB, its ICP chunks 3 % and a batch 1 % longer
  // (longer probe walks) -- not taken. Every walk of a table is bounded by its length, whatever fills it.
  g.Hf = g2_pow2(2 * (uint64_t)N);
  g.Hc = g2_pow2(2 * (uint64_t)N);
  g.fslot = a.take<FineSlot>(g.Hf);    // Hf * 16 bytes is a multiple of the arena alignment, so the
  g.cslot = a.take<CoarseSlot>(g.Hc);  // two tables are contiguous
  g.slot_words = (size_t)g.Hf + g.Hc;
  g.ovf = a.take<int32_t>(4);
  g.spts = a.take<float4>((size_t)g.Hf * kCellCap);
  g.sidx = a.take<int32_t>((size_t)g.Hf * kCellCap);
  g.ovf_pts = a.take<float4>(N);
  g.ovf_idx = a.take<int32_t>(N);
  g.ovf_cap = (uint32_t)N;
  g.fr = a.take<float>(2 * kFrameCells);
}

// claim / find: `base` is an array of 16-byte slots whose first 8 bytes are the key
template <typename Slot>
__device__ __forceinline__ int g2_claim(Slot* __restrict__ base, uint32_t hmask, uint64_t key) {
  uint32_t slot = mix64(key) & hmask;
  for (uint32_t t = 0; t <= hmask; ++t) {  // (once around at most: -1 = every slot holds another key)
    uint64_t prev = atomicCAS((unsigned long long*)&base[slot].key, (unsigned long long)kEmpty,
                              (unsigned long long)key);
    if (prev == kEmpty || prev == key) return (int)slot;
    slot = (slot + 1) & hmask;
  }
  return -1;
}

// the same with a bounded walk: -1 when kClaimProbes slots in a row belong to other keys (a table that fills up
// beyond its design load: the caller then parks the point in the overflow list, which every query scans)
constexpr int kClaimProbes = 256;
template <typename Slot>
__device__ __forceinline__ int g2_claim_bounded(Slot* __restrict__ base, uint32_t hmask, uint64_t key) {
  uint32_t slot = mix64(key) & hmask;
  for (int t = 0; t < kClaimProbes; ++t) {
    uint64_t prev = atomicCAS((unsigned long long*)&base[slot].key, (unsigned long long)kEmpty,
                              (unsigned long long)key);
    if (prev == kEmpty || prev == key) return (int)slot;
    slot = (slot + 1) & hmask;
  }
  return -1;
}

// one 16-byte load per probe: returns the slot index (or -1) and the payload words in *lo / *hi
template <typename Slot>
__device__ __forceinline__ int g2_find(const Slot* __restrict__ base, uint32_t hmask, uint64_t key,
                                       uint32_t* lo, uint32_t* hi) {
  uint32_t slot = mix64(key) & hmask;
  for (uint32_t t = 0; t <= hmask; ++t) {  // (once around at most: a table without an empty slot ends the walk too)
    uint4 w = ((const uint4*)base)[slot];
    uint64_t k = ((uint64_t)w.y << 32) | w.x;
    if (k == key) { *lo = w.z; *hi = w.w; return (int)slot; }
    if (k == kEmpty) return -1;
    slot = (slot + 1) & hmask;
  }
  return -1;
}

// grid-stride reset of one grid (keys <- empty, counters / masks / spill count <- 0, frame ranges):
// callable from any kernel whose grid is done with the table
__device__ __forceinline__ void g2_clear(const Grid2& g, int64_t tid, int64_t nthreads) {
  const uint4 empty = make_uint4(~0u, ~0u, 0u, 0u);
  uint4* wf = (uint4*)g.fslot;
  for (int64_t q = tid; q < (int64_t)g.Hf; q += nthreads) wf[q] = empty;
  uint4* wc = (uint4*)g.cslot;
  for (int64_t q = tid; q < (int64_t)g.Hc; q += nthreads) wc[q] = empty;
  for (int64_t q = tid; q < kFrameCells; q += nthreads) {
    g.fr[q] = __builtin_inff();
    g.fr[kFrameCells + q] = -__builtin_inff();
  }
  if (tid == 0) g.ovf[0] = 0;
}

// The same tables used at the size that `n` points need (n known on the device only): the arrays
// are allocated for a capacity, hashed and reset at pow2(2 n). Every kernel that touches the grid
// between two resets must form the view from the same n.
__device__ __forceinline__ Grid2 g2_view(const Grid2& alloc, int n) {
  Grid2 g = alloc;
  uint32_t h = 64;
  uint64_t want = (uint64_t)(n < 1 ? 1 : n);
  want *= 2;
  while (h < want && h < alloc.Hf) h <<= 1;
  g.Hf = h;
  g.Hc = h < alloc.Hc ? h : alloc.Hc;
  g.slot_words = (size_t)g.Hf + g.Hc;
  return g;
}

static __global__ void __launch_bounds__(kBlock) g2_clear_kernel(Grid2 g) {
  g2_clear(g, blockIdx.x * (int64_t)blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// Build: one thread per point. Gating: the kernels of ICP iteration k do nothing once k > *stop
// (the stop rule has fired).
// loc (optional): where point i went -- row of spts (>= 0) or ~(row of the overflow list) (< 0)
__device__ __forceinline__ void g2_insert_point(const GridDesc& d, const Grid2& g, bool in, int i,
                                                float4 p, int* __restrict__ err, int32_t* __restrict__ loc = nullptr) {
  int fc = frame_cell(d, p.x);
  int x = (int)floorf((p.y - d.lo[0]) / d.h);
  int y = (int)floorf((p.z - d.lo[1]) / d.h);
  int z = (int)floorf((p.w - d.lo[2]) / d.h);
  bool ok = in && !(x < 0 || y < 0 || z < 0 || x >= kCoordMax || y >= kCoordMax || z >= kCoordMax ||
                    fc < 0 || fc >= kFrameCells);
  if (in && !ok) {
    atomicExch(err, 1);
    if (loc) loc[i] = INT_MIN;  // (nowhere: g2_move_point leaves such a point alone)
  }
  // frame-value range per frame cell: a cloud is usually ONE frame, so every point would hit
  // the same two words -- reduce across the wave first when the wave agrees on the frame cell
  {
    unsigned long long act = __ballot(ok);
    if (act) {
      int leader = __ffsll((long long)act) - 1;
      int lfc = __shfl(fc, leader, 64);
      bool uniform = __ballot(ok && fc != lfc) == 0;
      if (uniform) {
        float mn = wave_min(ok ? p.x : __builtin_inff());
        float mx = wave_max(ok ? p.x : -__builtin_inff());
        if ((threadIdx.x & 63) == leader) {
          atomic_min_f32(&g.fr[lfc], mn);
          atomic_max_f32(&g.fr[kFrameCells + lfc], mx);
        }
      } else if (ok) {
        atomic_min_f32(&g.fr[fc], p.x);
        atomic_max_f32(&g.fr[kFrameCells + fc], p.x);
      }
    }
  }
  if (!ok) return;
  // the fine and the coarse claim do not depend on each other: their first probes go out together
  // (two atomic round trips for the whole insert instead of four), collisions take the slow loops
  const uint64_t fkey = pack_cell(fc, x, y, z), ckey = pack_cell(fc, x >> 2, y >> 2, z >> 2);
  const uint32_t fmask = g.Hf - 1, cmask = g.Hc - 1;
  uint32_t fs = mix64(fkey) & fmask, cs = mix64(ckey) & cmask;
  uint64_t pf = atomicCAS((unsigned long long*)&g.fslot[fs].key, (unsigned long long)kEmpty,
                          (unsigned long long)fkey);
  uint64_t pc = atomicCAS((unsigned long long*)&g.cslot[cs].key, (unsigned long long)kEmpty,
                          (unsigned long long)ckey);
  if (!(pf == kEmpty || pf == fkey)) {
    const int f2 = g2_claim(g.fslot, fmask, fkey);
    if (f2 < 0) { atomicExch(err, 1); if (loc) loc[i] = INT_MIN; return; }  // (cannot happen: <= N cells in > 1.5 N slots)
    fs = (uint32_t)f2;
  }
  if (!(pc == kEmpty || pc == ckey)) {
    const int c2 = g2_claim(g.cslot, cmask, ckey);
    if (c2 < 0) { atomicExch(err, 1); if (loc) loc[i] = INT_MIN; return; }
    cs = (uint32_t)c2;
  }
  int bit = (x & 3) | ((y & 3) << 2) | ((z & 3) << 4);
  int pos = atomicAdd(&g.fslot[fs].count, 1);
  atomicOr((unsigned long long*)&g.cslot[cs].mask, 1ULL << bit);
  // (destination chosen in registers: written as two branches the stores were merged through a
  // pointer table in scratch memory)
  size_t at = (size_t)fs * kCellCap + (size_t)(pos < kCellCap ? pos : 0);
  float4* dp = g.spts;
  int32_t* di = g.sidx;
  if (pos >= kCellCap) {
    at = (size_t)atomicAdd(g.ovf, 1);
    dp = g.ovf_pts;
    di = g.ovf_idx;
  }
  dp[at] = p;
  di[at] = i;
  if (loc) loc[i] = pos >= kCellCap ? ~(int32_t)at : (int32_t)at;
}

// A grid whose points MOVE a little between two searches (the moving cloud of an ICP iteration): a point that stays
// in its fine cell -- nearly all of them once the loop converges -- only has its stored coordinates replaced: one
// 16-byte store, no atomics, no reset of the tables (a rebuild is five atomics and two scattered stores per point).
// A point that changes cell leaves a TOMBSTONE in its row (coordinates no query can reach, index -1: every scan
// treats it as a far point) and is appended to its new cell like a fresh insert, with bounded claims; rows are never
// reused, so every row has one writer per launch and the order of arrival still changes nothing. Cells and tables
// only grow: the caller counts the cell changes and rebuilds the grid from scratch when they add up (icp_solve_phase).
// loc[i]: as g2_insert_point wrote it. Returns true when the point changed its cell.
constexpr float kTombCoord = 1.0e18f;
__device__ __forceinline__ bool g2_move_point(const GridDesc& d, const Grid2& g, bool in, int i, float4 po, float4 pn,
                                              int32_t* __restrict__ loc, int* __restrict__ err) {
  if (!in) return false;
  const int fc = frame_cell(d, pn.x);
  const int ox = (int)floorf((po.y - d.lo[0]) / d.h), oy = (int)floorf((po.z - d.lo[1]) / d.h),
            oz = (int)floorf((po.w - d.lo[2]) / d.h);
  const int x = (int)floorf((pn.y - d.lo[0]) / d.h), y = (int)floorf((pn.z - d.lo[1]) / d.h),
            z = (int)floorf((pn.w - d.lo[2]) / d.h);
  const int32_t at0 = loc[i];
  // (a point that never got a row -- out of the grid's range, the call has failed -- or a word that is no location)
  if (at0 >= 0 ? (uint64_t)at0 >= (uint64_t)g.Hf * kCellCap : (uint32_t)~at0 >= g.ovf_cap) return false;
  if (x == ox && y == oy && z == oz) {
    if (at0 >= 0) g.spts[at0] = pn;
    else g.ovf_pts[~at0] = pn;
    return false;
  }
  const float4 tomb = make_float4(po.x, kTombCoord, kTombCoord, kTombCoord);
  if (at0 >= 0) {
    g.spts[at0] = tomb;
    g.sidx[at0] = -1;
  } else {
    g.ovf_pts[~at0] = tomb;
    g.ovf_idx[~at0] = -1;
  }
  if (x < 0 || y < 0 || z < 0 || x >= kCoordMax || y >= kCoordMax || z >= kCoordMax || fc < 0 || fc >= kFrameCells) {
    atomicExch(err, 1);
    return true;
  }
  const uint64_t fkey = pack_cell(fc, x, y, z), ckey = pack_cell(fc, x >> 2, y >> 2, z >> 2);
  const int fs = g2_claim_bounded(g.fslot, g.Hf - 1, fkey);
  const int cs = fs >= 0 ? g2_claim_bounded(g.cslot, g.Hc - 1, ckey) : -1;
  int pos = kCellCap;
  if (cs >= 0) {
    pos = atomicAdd(&g.fslot[fs].count, 1);
    atomicOr((unsigned long long*)&g.cslot[cs].mask, 1ULL << ((x & 3) | ((y & 3) << 2) | ((z & 3) << 4)));
  }
  if (pos < kCellCap) {
    const size_t at = (size_t)fs * kCellCap + (size_t)pos;
    g.spts[at] = pn;
    g.sidx[at] = i;
    loc[i] = (int32_t)at;
  } else {
    const int at = atomicAdd(g.ovf, 1);
    if ((uint32_t)at >= g.ovf_cap) {  // (cannot happen while the caller rebuilds as soon as the list grows)
      atomicExch(err, 1);
      return true;
    }
    g.ovf_pts[at] = pn;
    g.ovf_idx[at] = i;
    loc[i] = ~(int32_t)at;
  }
  return true;
}

static __global__ void __launch_bounds__(kBlock)
g2_insert_kernel(const float4* __restrict__ pts, int N, const GridDesc* __restrict__ gp, Grid2 g,
                 int k, const int* __restrict__ stop, int* __restrict__ err) {
  if (stop && k > *stop) return;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  GridDesc d = *gp;
  bool in = i < N;
  float4 p = in ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  g2_insert_point(d, g, in, i, p, err);
}

// bits of coarse block (X,Y,Z) whose fine cells lie within Chebyshev distance s of cell (cx,cy,cz)
__device__ __forceinline__ uint64_t g2_box_mask(int X, int Y, int Z, int cx, int cy, int cz, int s) {
  int x0 = max(cx - s - 4 * X, 0), x1 = min(cx + s - 4 * X, 3);
  int y0 = max(cy - s - 4 * Y, 0), y1 = min(cy + s - 4 * Y, 3);
  int z0 = max(cz - s - 4 * Z, 0), z1 = min(cz + s - 4 * Z, 3);
  if (x0 > x1 || y0 > y1 || z0 > z1) return 0;
  unsigned xb = ((1u << (x1 - x0 + 1)) - 1u) << x0;
  unsigned yb = ((1u << (y1 - y0 + 1)) - 1u) << y0;
  unsigned zb = ((1u << (z1 - z0 + 1)) - 1u) << z0;
  unsigned ys = (yb & 1u) | ((yb & 2u) << 3) | ((yb & 4u) << 6) | ((yb & 8u) << 9);  // bits 0,4,8,12
  unsigned row16 = xb * ys;
  uint64_t zs = (uint64_t)(zb & 1u) | ((uint64_t)(zb & 2u) << 15) | ((uint64_t)(zb & 4u) << 30) |
                ((uint64_t)(zb & 8u) << 45);                                         // bits 0,16,32,48
  return (uint64_t)row16 * zs;
}

// ---- verified cache of the fused ICP: how far things may move before a search result can change.
// Distances are SPATIAL and squared on input: a2 = nearest, b2 = lower bound of every other point.
// A full search compares fp32 sums of four squares whose relative rounding error is below 2.4e-7,
// so two candidates are reliably ordered only when their squared distances differ by more than
// E = 1e-6 x (radius^2): the budget D solves (sqrt(b2) - D)^2 - (sqrt(a2) + D)^2 > E, less a safety.
__device__ __forceinline__ float icp_cache_budget(float a2, float b2, float radius2_4d) {
  const float E = 1e-6f * radius2_4d + 1e-9f;
  const float s1 = sqrtf(fmaxf(a2, 0.f) + E), s2 = sqrtf(fmaxf(b2 - E, 0.f));
  const float bud = 0.5f * (s2 - s1) - E / (2.f * (s1 + s2) + 1e-12f);
  return bud * 0.999999f - 2e-5f;
}
// no partner: every point is at least sqrt(lb2) away, the spatial search radius is sqrt(rs2)
__device__ __forceinline__ float icp_cache_budget_none(float lb2, float rs2, float radius2_4d) {
  const float E = 1e-6f * radius2_4d + 1e-9f;
  const float bud = sqrtf(fmaxf(lb2 - E, 0.f)) - sqrtf(fmaxf(rs2, 0.f) + E);
  return bud * 0.999999f - 2e-5f;
}

// ---- second level of the verified cache (round 5). A search also hands over the (up to three) nearest points OTHER
// than the partner that it met (`others`), and L: a distance below which no point outside {partner, others} lies,
// safely rounded. The caller stores others and c = L - budget; while the budget runs down by what has moved, L - moved
// = budget + c (negative budgets included). A query whose budget has run out then measures its four cached points
// where they are NOW, exactly, and keeps the nearest of them as long as that one is nearer than budget + c: everything
// else is at least that far away. Exact -- the same statement as icp_cache_budget, one level up -- and it needs four
// 16-byte gathers instead of a search.
struct NnAux {
  int o1, o2, o3;  // original indices of the other cached points, -1: none
  float c;         // L - budget at the time of the statement; -inf: no statement
};
__device__ __forceinline__ float icp_cache_bound(float lb2, float radius2_4d) {
  const float E = 1e-6f * radius2_4d + 1e-9f;
  return sqrtf(fmaxf(lb2 - E, 0.f)) * 0.999999f - 2e-5f;
}

// Cooperative form: the 32 lanes of a half-wave serve ONE query (two queries per wave).
//   1. lanes 0..26 fetch the occupancy masks of the 3x3x3 coarse blocks around the query;
//   2. the 27 fine cells at Chebyshev distance <= 1 are dealt to 27 lanes (one round of memory
//      latency); most queries have their partner there and stop: everything else is >= h away;
//   3. otherwise every lane walks the OCCUPIED cells of its own block (set bits of its mask), each
//      pruned by its per-axis distance bound against the best found so far -- empty space costs
//      nothing, cells outside the ball cost a few ALU ops and no memory access.
// Pruning is conservative only: the result is the exact nearest point within the radius under the
// canonical order (dist2, original index), whatever the traversal order. Every lane of the wave
// must call this (shuffles); the lanes of a half pass the same p / df, `valid` = this half has a
// query. The result (original index or -1) is returned in every lane of the half.
//
// BUD = true (the fused ICP's verified cache): the search also says for how long its answer stays
// exact. The hashed cloud then has ONE frame value, so the frame term of the 4-D distance is the
// same constant for every candidate and "nearest" is decided by the spatial distance. Every pruning
// bound is widened by `mu` (spatial), so that all points within (nearest spatial distance + mu) --
// or, for a query without partner, within the spatial radius + mu -- have been looked at;
// *bud_out is then a distance: as long as neither the query nor any hashed point has moved further
// than that in total, a full search would return the same partner (or "none") again -- the fp32
// rounding of the 4-term distance sum is part of the bound (icp_cache_budget). The cell size must
// admit spatial radius + mu.
template <bool BUD = false>
__device__ __forceinline__ int g2_nearest_coop(const GridDesc& d, const Grid2& g, float4 p, int df,
                                               bool valid, int* __restrict__ overflow,
                                               float* __restrict__ best_dist2 = nullptr,
                                               float mu = 0.f, float* __restrict__ bud_out = nullptr,
                                               float seed_d2 = __builtin_inff(), NnAux* __restrict__ aux_out = nullptr,
                                               const bool ovf_by_cell = false) {
  // ovf_by_cell (static grids whose spilled points have been sorted by cell, nn1_spill_sort_kernel): a full cell's
  // slot holds (in `pad`) where its spilled points start in the overflow arrays, + 1 -- they are looked at with the
  // cell, and only then; otherwise every query scans the whole overflow list (a few hundred spilled points made that
  // scan 90 % of the extraction graph's search: 557 x 2 loads per query against ~100 for everything else)
  const int lane = threadIdx.x & 63;
  const int sub = lane & 31;
  const int half_base = lane & 32;
  int fc = frame_cell(d, p.x) + df;
  float fa = 1.f, fb = 0.f;
  if (valid && fc >= 0 && fc < kFrameCells) { fa = g.fr[fc]; fb = g.fr[kFrameCells + fc]; }
  valid = valid && fa <= fb;
  float dfl = fmaxf(fmaxf(fa - p.x, p.x - fb), 0.f);
  float dfsq = dfl * dfl;
  const float margin = 1e-5f * d.radius2 + 1e-6f;
  // spatial radius (squared) inside which everything is examined
  const float rs = sqrtf(fmaxf(d.radius2 - dfsq, 0.f));
  const float rinf2 = BUD ? (rs + mu) * (rs + mu) : d.radius2 - dfsq;
  float budget = rinf2 + margin;
  valid = valid && budget >= 0.f;
  const bool valid0 = valid;
  int rings = valid ? (int)ceilf(sqrtf(budget) / d.h) : 0;
  if (rings > 4) { if (sub == 0) atomicExch(overflow, 1); rings = 4; }
  int cx = (int)floorf((p.y - d.lo[0]) / d.h);
  int cy = (int)floorf((p.z - d.lo[1]) / d.h);
  int cz = (int)floorf((p.w - d.lo[2]) / d.h);
  int bx = cx >> 2, by = cy >> 2, bz = cz >> 2;
  // lanes 0..26 of the half fetch one coarse occupancy mask each
  uint64_t mymask = 0;
  int X = bx + sub % 3 - 1, Y = by + (sub / 3) % 3 - 1, Z = bz + sub / 9 - 1;
  if (valid && sub < 27) {
    bool reach = X >= 0 && Y >= 0 && Z >= 0;
    if (BUD && seed_d2 <= d.radius2) {
      // a point of the hashed cloud is known at seed_d2: nothing beyond the ball through it (+ mu) matters, and a
      // block that ball does not reach is not even asked for its mask (a ball of half the radius meets 8 of the 27)
      const float r = sqrtf(fmaxf(seed_d2 - dfsq, 0.f)) + mu;
      const float lim = (r * r + margin) * 1.00001f + 1e-6f;
      const float fx = (p.y - d.lo[0]) / d.h, fy = (p.z - d.lo[1]) / d.h, fz = (p.w - d.lo[2]) / d.h;
      const float ax = fmaxf(fmaxf((float)(4 * X) - fx, fx - (float)(4 * X + 4)), 0.f) * d.h;
      const float ay = fmaxf(fmaxf((float)(4 * Y) - fy, fy - (float)(4 * Y + 4)), 0.f) * d.h;
      const float az = fmaxf(fmaxf((float)(4 * Z) - fz, fz - (float)(4 * Z + 4)), 0.f) * d.h;
      reach = reach && !((ax * ax + ay * ay) + az * az > lim);
    }
    if (reach) {
      uint32_t lo = 0, hi = 0;
      int cs = g2_find(g.cslot, g.Hc - 1, pack_cell(fc, X, Y, Z), &lo, &hi);
      if (cs >= 0) mymask = ((uint64_t)hi << 32) | lo;
    }
  }
  unsigned long long occ = __ballot(mymask != 0);
  const bool any_block = ((occ >> half_base) & 0xFFFFFFFFULL) != 0;
  valid = valid && any_block;
  int cand = -1;
  float cd = 0.f;
  // BUD: the nearest point this lane has seen other than its best (i2 at d2v), and the smallest squared distance of
  // everything else it has seen (cd3)
  int i2 = -1;
  float d2v = __builtin_inff(), cd3 = __builtin_inff();
  auto widen = [&](float x2) {   // squared SPATIAL radius that must be examined around a best of x2 (4-D)
    if (!BUD) return x2 - dfsq;
    float r = sqrtf(fmaxf(x2 - dfsq, 0.f)) + mu;
    return r * r;
  };
  auto other = [&](int oi, float dist2) {  // a point that is not this lane's best
    if (oi == i2) return;                  // (met again)
    if (i2 < 0 || dist2 < d2v || (dist2 == d2v && oi < i2)) {
      if (i2 >= 0) cd3 = fminf(cd3, d2v);
      i2 = oi;
      d2v = dist2;
    } else {
      cd3 = fminf(cd3, dist2);
    }
  };

  auto consider = [&](float4 c, int oi) {
    float d0 = c.x - p.x, d1 = c.y - p.y, d2 = c.z - p.z, d3 = c.w - p.w;
    float dist2 = 0.f;
    dist2 = dist2 + d0 * d0;
    dist2 = dist2 + d1 * d1;
    dist2 = dist2 + d2 * d2;
    dist2 = dist2 + d3 * d3;
    if (BUD && oi < 0) return;  // (a tombstone of the moving grid: no point)
    if (dist2 <= d.radius2 && (cand < 0 || dist2 < cd || (dist2 == cd && oi < cand))) {
      if (BUD && cand >= 0) other(cand, cd);
      cand = oi;
      cd = dist2;
    } else if (BUD) {
      other(oi, dist2);
    }
  };
  // a cell's run is kCellCap contiguous rows: four rows and their indices are requested together
  // (one round of memory latency per four points instead of one per point)
  auto scan_cell = [&](int x, int y, int z) {
    uint32_t cnt_w = 0, pad_w = 0;
    int fs = g2_find(g.fslot, g.Hf - 1, pack_cell(fc, x, y, z), &cnt_w, &pad_w);
    if (fs < 0) return;
    int cnt = min((int)cnt_w, kCellCap);
    const float4* run = g.spts + (size_t)fs * kCellCap;
    const int4* idx = (const int4*)(g.sidx + (size_t)fs * kCellCap);
    for (int j = 0; j < cnt; j += 4) {
      float4 c0 = run[j], c1 = run[j + 1], c2 = run[j + 2], c3 = run[j + 3];
      int4 oi = idx[j >> 2];
      consider(c0, oi.x);
      if (j + 1 < cnt) consider(c1, oi.y);
      if (j + 2 < cnt) consider(c2, oi.z);
      if (j + 3 < cnt) consider(c3, oi.w);
    }
    if (ovf_by_cell && (int)cnt_w > kCellCap) {  // the cell's spilled points
      const int first = (int)pad_w - 1, extra = min((int)cnt_w - kCellCap, (int)g.ovf_cap - first);
      for (int j = 0; j < extra; ++j) consider(g.ovf_pts[first + j], g.ovf_idx[first + j]);
    }
  };
  // ---- spilled points (cells with more than kCellCap points; normally none): every lane of the
  //      half takes a stride of the list; same frame cell and radius test as the cell walk
  {
    // (valid0, not valid: a moved point whose cell could not be claimed lies here WITHOUT a bit in the coarse masks,
    //  g2_move_point; the length is clamped to the list's rows)
    int no = (valid0 && !ovf_by_cell) ? min(*g.ovf, (int)g.ovf_cap) : 0;
    for (int j = sub; j < no; j += 32) {
      float4 c = g.ovf_pts[j];
      if (frame_cell(d, c.x) != fc) continue;
      consider(c, g.ovf_idx[j]);
    }
  }
  // ---- near cells: Chebyshev distance <= 1, one cell per lane
  {
    bool cell_on = valid && sub < 27;
    int x = cx + sub % 3 - 1, y = cy + (sub / 3) % 3 - 1, z = cz + sub / 9 - 1;
    int q = cell_on ? ((x >> 2) - bx + 1) + 3 * ((y >> 2) - by + 1) + 9 * ((z >> 2) - bz + 1) : 0;
    unsigned lo32 = __shfl((unsigned)(mymask & 0xFFFFFFFFULL), half_base + q, 64);
    unsigned hi32 = __shfl((unsigned)(mymask >> 32), half_base + q, 64);
    uint64_t mk = ((uint64_t)hi32 << 32) | lo32;
    if (cell_on) {
      int bit = (x & 3) | ((y & 3) << 2) | ((z & 3) << 4);
      if ((mk >> bit) & 1ULL) scan_cell(x, y, z);
    }
  }
  // best of the half so far (needed to decide whether the far cells can matter)
  auto reduce_half = [&]() {
    if (__ballot(cand >= 0) == 0) return;  // nothing found anywhere in the wave
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      int oc = __shfl_xor(cand, o, 64);
      float od = __shfl_xor(cd, o, 64);
      bool take = oc >= 0 && (cand < 0 || od < cd || (od == cd && oc < cand));
      if (take) { cand = oc; cd = od; }
    }
  };
  // (BUD: the lanes keep their own second-best; the best is exchanged, so a lane may hold the half's
  //  best without having seen it -- its own former best then counts as one of its other points)
  {
    const int mine = cand;
    const float mine_d = cd;
    reduce_half();
    if (BUD && mine >= 0 && mine != cand) other(mine, mine_d);
  }
  // ---- far cells: everything at Chebyshev distance >= 2 is at least h away
  // BUD, seed_d2: the 4-D squared distance of a point KNOWN to be in the hashed cloud and inside the radius (the
  // query's partner of the last iteration, where it is now): the nearest point is at most that far, so nothing
  // beyond the ball through it (+ mu) can matter -- a third of the queries that come here have their partner
  // 0.3 .. 0.9 radii away, and the ball of the full radius holds ten times the cells of theirs
  const float seed_bound = (BUD && seed_d2 <= d.radius2) ? widen(seed_d2) + margin : __builtin_inff();
  float bound = fminf((cand >= 0 ? widen(cd) : rinf2) + margin, seed_bound);  // admissible |dxyz|^2 of a point that matters
  bool more = valid && rings >= 2 && !(d.h * d.h > bound);
  uint64_t rem = 0;
  if (more && sub < 27) rem = mymask & ~g2_box_mask(X, Y, Z, cx, cy, cz, 1);
  const float h2 = d.h * d.h;
  if (rem) {
    // keep only the cells of this block whose per-axis distance bound lies inside the ball that can
    // still hold a better point: the squared bound separates into per-axis terms (small integers),
    // so the 64 tests are 16 thresholds on 4 values. For a query with nothing nearby this removes
    // ~85 % of the 12^3 cells before the walk. (Slack >> the rounding of the threshold; the exact
    // per-cell test stays in the loop.)
    float ax2[4], ay2[4], az2[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float tx = (float)max(abs(4 * X + a - cx) - 1, 0), ty = (float)max(abs(4 * Y + a - cy) - 1, 0),
            tz = (float)max(abs(4 * Z + a - cz) - 1, 0);
      ax2[a] = tx * tx; ay2[a] = ty * ty; az2[a] = tz * tz;
    }
    float lim = (bound / h2) * 1.00001f + 1e-5f;
    uint64_t ball = 0;
#pragma unroll
    for (int zz = 0; zz < 4; ++zz) {
#pragma unroll
      for (int yy = 0; yy < 4; ++yy) {
        float t = lim - az2[zz] - ay2[yy];
        unsigned xb = (unsigned)(ax2[0] <= t) | ((unsigned)(ax2[1] <= t) << 1) |
                      ((unsigned)(ax2[2] <= t) << 2) | ((unsigned)(ax2[3] <= t) << 3);
        ball |= (uint64_t)xb << (4 * yy + 16 * zz);
      }
    }
    rem &= ball;
  }
  // the cells `cells` (bits of this lane's block) of the 27 blocks are dealt evenly to the 32 lanes of the half:
  // rank r of the concatenated set bits goes to lane r % 32, so the walk takes ceil(total / 32) rounds of
  // memory latency however unevenly the blocks are filled; prune: skip cells the current bound does not reach
  auto walk_cells = [&](uint64_t cells, bool prune) {
    int n_mine = __popcll(cells);
    int incl = n_mine;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      int t = __shfl_up(incl, o, 32);
      if (sub >= o) incl += t;
    }
    int total = __shfl(incl, half_base + 31, 64);
    int tot_o = __shfl_xor(total, 32, 64);
    int rounds = ((total > tot_o ? total : tot_o) + 31) >> 5;  // same for both halves of the wave
    unsigned rem_lo = (unsigned)(cells & 0xFFFFFFFFULL), rem_hi = (unsigned)(cells >> 32);
    for (int rd = 0; rd < rounds; ++rd) {
      int r = rd * 32 + sub;
      bool act = r < total;
      // owner lane q: first lane whose inclusive count exceeds r (binary search by shuffles)
      int lo = 0, hi = 31;
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        int mid = (lo + hi) >> 1;
        int v = __shfl(incl, half_base + mid, 64);
        if (v <= r) lo = mid + 1; else hi = mid;
      }
      int q = act ? lo : 0;
      int q_incl = __shfl(incl, half_base + q, 64);
      unsigned m_lo = __shfl(rem_lo, half_base + q, 64);
      unsigned m_hi = __shfl(rem_hi, half_base + q, 64);
      if (act) {
        int k = r - (q_incl - (__popc(m_lo) + __popc(m_hi)));  // rank within the owner's mask
        // k-th set bit of (m_hi:m_lo)
        unsigned w = m_lo;
        int pos = 0, c = __popc(m_lo);
        if (k >= c) { k -= c; w = m_hi; pos = 32; }
        c = __popc(w & 0xFFFFu);
        if (k >= c) { k -= c; w >>= 16; pos += 16; }
        w &= 0xFFFFu;
        c = __popc(w & 0xFFu);
        if (k >= c) { k -= c; w >>= 8; pos += 8; }
        w &= 0xFFu;
        c = __popc(w & 0xFu);
        if (k >= c) { k -= c; w >>= 4; pos += 4; }
        w &= 0xFu;
        c = __popc(w & 0x3u);
        if (k >= c) { k -= c; w >>= 2; pos += 2; }
        w &= 0x3u;
        if (k >= (int)(w & 1u)) pos += 1;
        int bit = pos;
        int QX = bx + q % 3 - 1, QY = by + (q / 3) % 3 - 1, QZ = bz + q / 9 - 1;
        int x = 4 * QX + (bit & 3), y = 4 * QY + ((bit >> 2) & 3), z = 4 * QZ + (bit >> 4);
        float ax = (float)max(abs(x - cx) - 1, 0), ay = (float)max(abs(y - cy) - 1, 0),
              az = (float)max(abs(z - cz) - 1, 0);
        float lb2 = ((ax * ax + ay * ay) + az * az) * h2;
        float bnd = fminf((cand >= 0 ? widen(cd) : rinf2) + margin, seed_bound);
        if (!prune || !(lb2 > bnd)) scan_cell(x, y, z);
      }
    }
    {
      const int mine = cand;
      const float mine_d = cd;
      reduce_half();
      if (BUD && mine >= 0 && mine != cand) other(mine, mine_d);
    }
  };
  if (__ballot(rem != 0)) walk_cells(rem, true);
  // BUD, a query that has found nothing within the radius: every remaining occupied cell of its 27 blocks as
  // well. It then knows the distance to the nearest point of the whole region and that everything else lies
  // beyond the region's boundary -- a budget that usually lasts for the rest of the registration, where the
  // ball alone says "nothing within mu" and is searched again as soon as something has moved that far.
  bool region_done = false;
  if (BUD) {
    uint64_t rest = 0;
    // (all of them again, also those of the ball: its walk may have skipped a cell by its distance bound,
    //  and a point met twice changes nothing while there is no partner)
    if (valid && cand < 0 && sub < 27) rest = mymask & ~g2_box_mask(X, Y, Z, cx, cy, cz, 1);
    if (__ballot(rest != 0)) walk_cells(rest, false);
    region_done = valid0 && cand < 0;
  }
  if (best_dist2) *best_dist2 = cand >= 0 ? cd : d.radius2;
  if (BUD) {
    // the three nearest of the lanes' other points (by (dist2, index)); what is left -- the lanes' entries not taken
    // and everything behind them -- bounds the points outside the cache
    int o[3] = {-1, -1, -1};
    float od0 = __builtin_inff();  // distance of the nearest other point (the runner-up, if any was taken)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      int mi = i2;
      float md = i2 >= 0 ? d2v : __builtin_inff();
#pragma unroll
      for (int s2 = 16; s2 > 0; s2 >>= 1) {
        const int oi = __shfl_xor(mi, s2, 64);
        const float odd = __shfl_xor(md, s2, 64);
        const bool take = oi >= 0 && (mi < 0 || odd < md || (odd == md && oi < mi));
        if (take) { mi = oi; md = odd; }
      }
      o[q] = mi;
      if (q == 0) od0 = md;
      if (mi >= 0 && i2 == mi) {  // taken: what this lane saw behind it is in cd3
        i2 = -1;
        d2v = __builtin_inff();
      }
    }
    float rest2 = fminf(cd3, i2 >= 0 ? d2v : __builtin_inff());
#pragma unroll
    for (int s2 = 16; s2 > 0; s2 >>= 1) rest2 = fminf(rest2, __shfl_xor(rest2, s2, 64));
    const float cd2 = fminf(od0, rest2);  // smallest squared distance seen other than the best's
    float bud = 0.f;
    NnAux ax = {-1, -1, -1, -__builtin_inff()};
    if (valid0 && fa == fb) {  // (several frame values in the cell: no statement, search again next time)
      if (cand >= 0) {
        // every point within spatial distance s1 + mu has been looked at: the runner-up is at least that far
        bud = icp_cache_budget(cd - dfsq, fminf(cd2 - dfsq, widen(cd)), d.radius2);
        ax.o1 = o[0]; ax.o2 = o[1]; ax.o3 = o[2];
        ax.c = icp_cache_bound(fminf(rest2 - dfsq, widen(cd)), d.radius2) - bud;
      } else if (!any_block || region_done) {
        // nothing within the radius; every point of the 3 x 3 x 3 blocks has been looked at (cd2; none at
        // all: +inf), the rest lies beyond their boundary
        float fx = (p.y - d.lo[0]) / d.h, fy = (p.z - d.lo[1]) / d.h, fz = (p.w - d.lo[2]) / d.h;
        float ex = fminf(fx - (float)(4 * (bx - 1)), (float)(4 * (bx + 2)) - fx);
        float ey = fminf(fy - (float)(4 * (by - 1)), (float)(4 * (by + 2)) - fy);
        float ez = fminf(fz - (float)(4 * (bz - 1)), (float)(4 * (bz + 2)) - fz);
        float e = fmaxf(fminf(fminf(ex, ey), ez), 0.f) * d.h * 0.9999f;
        bud = icp_cache_budget_none(fminf(cd2 - dfsq, e * e), d.radius2 - dfsq, d.radius2);
      } else {
        bud = icp_cache_budget_none(fminf(cd2 - dfsq, rinf2), d.radius2 - dfsq, d.radius2);
      }
    }
    if (bud_out) *bud_out = bud;
    if (aux_out) *aux_out = ax;
  }
  return cand;
}

// Can g2_nearest_seeded8 settle this query? (One lane; the same arithmetic as the search itself, so
// the answer is the search's own. dist2 = the candidate's 4-D squared distance as the search forms it.)
__device__ __forceinline__ bool g2_seed_admissible(const GridDesc& d, const Grid2& g, float4 p, int df,
                                                   float dist2, float mu) {
  int fc = frame_cell(d, p.x) + df;
  if (fc < 0 || fc >= kFrameCells) return false;
  const float fa = g.fr[fc], fb = g.fr[kFrameCells + fc];
  if (!(fa == fb) || !(dist2 <= d.radius2)) return false;
  const float dfl = fmaxf(fmaxf(fa - p.x, p.x - fb), 0.f);
  const float dfsq = dfl * dfl;
  const float margin = 1e-5f * d.radius2 + 1e-6f;
  const float rho = sqrtf(fmaxf(dist2 - dfsq, 0.f)) + mu;
  const float bound2 = rho * rho + margin;
  const float bound = sqrtf(bound2) * 1.000001f + 1e-6f;
  const float fx = (p.y - d.lo[0]) / d.h, fy = (p.z - d.lo[1]) / d.h, fz = (p.w - d.lo[2]) / d.h;
  const float br = bound / d.h;
  const int x0 = (int)floorf(fx - br), x1 = (int)floorf(fx + br);
  const int y0 = (int)floorf(fy - br), y1 = (int)floorf(fy + br);
  const int z0 = (int)floorf(fz - br), z1 = (int)floorf(fz + br);
  return x1 - x0 <= 2 && y1 - y0 <= 2 && z1 - z0 <= 2 && x0 >= 0 && y0 >= 0 && z0 >= 0;
}

// EIGHT lanes serve one query that comes with a candidate (in the ICP: its partner of the previous
// iteration, a real point of the hashed set): only the fine cells that intersect the ball through
// the candidate (widened by mu, see g2_nearest_coop<true>) can hold a point that matters, and once
// the loop has converged that is a handful of cells instead of the 27 + far walk of the cooperative
// search -- a few dozen bytes per query instead of a kilobyte, and eight queries per wave instead of
// two. The cells of the box around the ball (at most 3 x 3 x 3) are dealt to the eight lanes; cells
// the ball does not reach are never probed. Exact under the same canonical order (dist2, original
// index). The eight lanes of an aligned group pass the same arguments; all 64 lanes of the wave
// must call this (shuffles). Returns true when the search is complete (*res = the nearest point's
// index, possibly the seed's, *bud_out = how far things may move before the answer can change, in
// every lane of the group), false when the seed is not admissible or the box is larger (the caller
// then runs the cooperative search for this query).
__device__ __forceinline__ bool g2_nearest_seeded8(const GridDesc& d, const Grid2& g, float4 p, int df,
                                                   float4 seed_pt, int seed_idx, bool valid, float mu,
                                                   int* __restrict__ res, float* __restrict__ bud_out,
                                                   NnAux* __restrict__ aux_out = nullptr) {
  const int sub = threadIdx.x & 7;
  int fc = frame_cell(d, p.x) + df;
  bool ok = valid && seed_idx >= 0 && fc >= 0 && fc < kFrameCells;
  float fa = 1.f, fb = 0.f;
  if (ok) { fa = g.fr[fc]; fb = g.fr[kFrameCells + fc]; }
  ok = ok && fa <= fb;
  const float dfl = fmaxf(fmaxf(fa - p.x, p.x - fb), 0.f);
  const float dfsq = dfl * dfl;
  const float margin = 1e-5f * d.radius2 + 1e-6f;
  int cand = -1;
  float cd = 0.f;
  // the nearest point this lane has seen other than its best (i2 at d2v), and the smallest squared distance of
  // everything else it has seen (cd3): see g2_nearest_coop
  int i2 = -1;
  float d2v = __builtin_inff(), cd3 = __builtin_inff();
  auto other = [&](int oi, float dist2) {
    if (oi == i2) return;  // (met again: the seed in its cell)
    if (i2 < 0 || dist2 < d2v || (dist2 == d2v && oi < i2)) {
      if (i2 >= 0) cd3 = fminf(cd3, d2v);
      i2 = oi;
      d2v = dist2;
    } else {
      cd3 = fminf(cd3, dist2);
    }
  };
  auto consider = [&](float4 c, int oi) {
    float d0 = c.x - p.x, d1 = c.y - p.y, d2 = c.z - p.z, d3 = c.w - p.w;
    float dist2 = 0.f;
    dist2 = dist2 + d0 * d0;
    dist2 = dist2 + d1 * d1;
    dist2 = dist2 + d2 * d2;
    dist2 = dist2 + d3 * d3;
    if (oi < 0) return;  // (a tombstone of the moving grid: no point)
    if (dist2 <= d.radius2 && (cand < 0 || dist2 < cd || (dist2 == cd && oi < cand))) {
      if (cand >= 0) other(cand, cd);
      cand = oi;
      cd = dist2;
    } else if (oi != cand) {  // (the seed is met again in its cell)
      other(oi, dist2);
    }
  };
  if (ok) consider(seed_pt, seed_idx);
  ok = ok && cand >= 0 && fa == fb;  // else the partner has left the search radius
  const float rho = sqrtf(fmaxf(cd - dfsq, 0.f)) + mu;  // everything within (spatial) rho of the query is examined
  const float cover2 = rho * rho;
  const float bound2 = cover2 + margin;
  const float bound = ok ? sqrtf(bound2) * 1.000001f + 1e-6f : 0.f;
  const float fx = (p.y - d.lo[0]) / d.h, fy = (p.z - d.lo[1]) / d.h, fz = (p.w - d.lo[2]) / d.h;
  const float br = bound / d.h;
  const int x0 = (int)floorf(fx - br), x1 = (int)floorf(fx + br);
  const int y0 = (int)floorf(fy - br), y1 = (int)floorf(fy + br);
  const int z0 = (int)floorf(fz - br), z1 = (int)floorf(fz + br);
  const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
  ok = ok && nx <= 3 && ny <= 3 && nz <= 3 && x0 >= 0 && y0 >= 0 && z0 >= 0;
  const int ncell = ok ? nx * ny * nz : 0;
  const float h2 = d.h * d.h;
  const float lim = (bound2 / h2) * 1.00001f + 1e-5f;
  // spilled points (cells with more than kCellCap points; normally none)
  {
    const int no = ok ? min(*g.ovf, (int)g.ovf_cap) : 0;
    for (int j = sub; j < no; j += 8) {
      float4 c = g.ovf_pts[j];
      if (frame_cell(d, c.x) != fc) continue;
      consider(c, g.ovf_idx[j]);
    }
  }
  for (int ci = sub; ci < ncell; ci += 8) {
    const int ix = ci % nx, iy = (ci / nx) % ny, iz = ci / (nx * ny);
    const int x = x0 + ix, y = y0 + iy, z = z0 + iz;
    // squared distance (in cells) from the query to the cell's box; beyond the ball: skip
    const float ax = fmaxf(fmaxf((float)x - fx, fx - (float)(x + 1)), 0.f);
    const float ay = fmaxf(fmaxf((float)y - fy, fy - (float)(y + 1)), 0.f);
    const float az = fmaxf(fmaxf((float)z - fz, fz - (float)(z + 1)), 0.f);
    if ((ax * ax + ay * ay) + az * az > lim) continue;
    uint32_t cnt_w = 0, pad_w = 0;
    int fs = g2_find(g.fslot, g.Hf - 1, pack_cell(fc, x, y, z), &cnt_w, &pad_w);
    if (fs < 0) continue;
    const int cnt = min((int)cnt_w, kCellCap);
    const float4* run = g.spts + (size_t)fs * kCellCap;
    const int4* idx = (const int4*)(g.sidx + (size_t)fs * kCellCap);
    for (int j = 0; j < cnt; j += 4) {
      float4 c0 = run[j], c1 = run[j + 1], c2 = run[j + 2], c3 = run[j + 3];
      int4 oi = idx[j >> 2];
      consider(c0, oi.x);
      if (j + 1 < cnt) consider(c1, oi.y);
      if (j + 2 < cnt) consider(c2, oi.z);
      if (j + 3 < cnt) consider(c3, oi.w);
    }
  }
  // best of the eight lanes; a lane whose own best lost keeps it as one of its other points
  {
    const int mine = cand;
    const float mine_d = cd;
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      int oc = __shfl_xor(cand, o, 64);
      float od = __shfl_xor(cd, o, 64);
      bool take = oc >= 0 && (cand < 0 || od < cd || (od == cd && oc < cand));
      if (take) { cand = oc; cd = od; }
    }
    if (mine >= 0 && mine != cand) other(mine, mine_d);
  }
  // the three nearest other points of the group, and the bound of what is left (g2_nearest_coop)
  int ot[3] = {-1, -1, -1};
  float od0 = __builtin_inff();
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    int mi = i2;
    float md = i2 >= 0 ? d2v : __builtin_inff();
#pragma unroll
    for (int s2 = 4; s2 > 0; s2 >>= 1) {
      const int oi = __shfl_xor(mi, s2, 64);
      const float odd = __shfl_xor(md, s2, 64);
      const bool take = oi >= 0 && (mi < 0 || odd < md || (odd == md && oi < mi));
      if (take) { mi = oi; md = odd; }
    }
    ot[q] = mi;
    if (q == 0) od0 = md;
    if (mi >= 0 && i2 == mi) {
      i2 = -1;
      d2v = __builtin_inff();
    }
  }
  float rest2 = fminf(cd3, i2 >= 0 ? d2v : __builtin_inff());
#pragma unroll
  for (int s2 = 4; s2 > 0; s2 >>= 1) rest2 = fminf(rest2, __shfl_xor(rest2, s2, 64));
  const float cd2 = fminf(od0, rest2);
  *res = cand;
  float bud = 0.f;
  NnAux ax = {-1, -1, -1, -__builtin_inff()};
  if (ok && cand >= 0) {
    bud = icp_cache_budget(cd - dfsq, fminf(cd2 - dfsq, cover2), d.radius2);
    ax.o1 = ot[0]; ax.o2 = ot[1]; ax.o3 = ot[2];
    ax.c = icp_cache_bound(fminf(rest2 - dfsq, cover2), d.radius2) - bud;
  }
  *bud_out = bud;
  if (aux_out) *aux_out = ax;
  return ok;
}

}  // namespace pcs
