// Sequence tracker for gfx950: ClusterTracking.track_frame for EVERY tracking unit of a sequence,
// driven in lock-step by one host thread on one stream.
//
// Reference: pcdet/models/registration/preprocessors/cluster_tracking.py:430-787 (track_frame;
// SURVEY.md A.6) with its helpers sample_frame (:39-51), component_center / component_diameter /
// filter_components (:89-160), dist_compensate (:80-87) and smooth_velo (:162-199); the units are
// the (component key, anchor frame) pairs ClusterTracking.forward loops over (:853-884).
//
// In the reference every step of the tracking loop is ~450 small framework ops issued from the
// interpreter, one unit after the other. A unit is a chain of small, latency-bound launches, and the
// units of a sequence are independent (within a unit the backward pass must end before the forward
// pass starts: smooth_velo's decoupled weight decay touches the whole velocity tensor). So the
// units advance TOGETHER: every kernel below takes a by-value list of units (blockIdx.y = entry) and
// serves all units that are in the same phase, and the registrations of all units share the three
// launches of an ICP iteration (pcs_icp.hpp). Per "tick" the host enqueues, in this order,
//   finish of the registrations that stopped -> apply T -> (last level) component motion,
//   smooth_velo, stop rules, harvest -> direction / step set-up -> sampling of anchor and target
//   frame -> start of the next registrations -> one chunk of ICP iterations for every unit that is
//   registering -> state mirror into pinned memory,
// then synchronises once and reads a few words per unit. No kernel geometry depends on data the
// host has not seen: sizes that are born on the device (voxel counts) stay there, the kernels are
// grid-stride and the launch sizes are estimates from the previous step.
//
// Arithmetic follows the host-side restatement (pcdet/.../cluster_tracking.py of this package) op
// for op: fp32 where torch computes in fp32, f64 for the transforms, fixed-order segment sums, so
// both produce identical ids; a unit's results do not depend on which other units share its launches.
#include "pcs_icp.hpp"
#include "pcs_wgsort.hpp"

#include "pcs.h"

#include <algorithm>
#include <cstdlib>
#include <condition_variable>
#include <vector>
#include <mutex>
#include <thread>
#include <atomic>
#include <chrono>
#include <cstdio>

namespace pcs {

constexpr int kMaxLevels = 4;
// (kWg = 1024, the one-workgroup kernels' size: pcs_wgsort.hpp)

struct TrkState {
  int32_t active;     // any component still tracked (loop condition of the next step)
  int32_t err;
  int64_t ex_count;   // extracted points so far
  int64_t counts[4];  // anchor voxels, anchor non-stationary (Mn), target voxels (Ra), target non-stationary (Rn)
};

// one point set of a unit's sampling (anchor frame as moved so far / target frame)
struct SmpSet {
  uint64_t *k0, *k1;      // sort keys, ping-pong
  int32_t *v0, *v1;       // original indices, ping-pong
  int32_t *order, *seg;   // sorted original indices; voxel boundaries [v + 1]
  int64_t* d_num;         // voxels
  float4* sub;            // [v] voxel means
  int32_t* sub_comp;      // [v] voxel component (anchor only)
  int32_t* flag;          // [v] non-stationary
  float4* out;            // compacted non-stationary voxels
  int32_t* out_comp;      // their components (anchor only)
};

struct UnitDev {
  const float4* seq;
  const int64_t* seq_comp;
  const uint8_t* seq_stat;
  int64_t off_a, comp_base, ex_cap;
  int na, C, F, W, anchor, min_f;
  float4 *xyzf, *last;
  int64_t* comp64;
  int32_t* comp32;
  int32_t *order_c, *seg_c;
  int64_t *uniq_c, *d_numc;
  float *deg_f, *diam, *ctr, *centers, *diffs, *velo_pre, *ratio, *ratio_tmp, *red4, *red1, *dist;
  int64_t* deg_i;
  unsigned long long* sub_deg;
  uint8_t* stopped;
  double *Tl, *l1, *l1_tmp;
  SmpSet sa, sb;
  Grid2 hg;
  const GridDesc* hdesc;
  int32_t *hv_nbr, *hv_deg;
  TrkState* st;
  double* transforms;
  float *reg_errors, *edge_ratios, *velos;
  int64_t *cmin, *cmax;
  uint8_t *moving, *vcm;
  int64_t *ex_index, *ex_local, *ex_comp;
};

// per (unit, tick) arguments, 16 bytes so that kMaxBatch of them travel by value
struct StepArg {
  short next, col, last_col;
  signed char dir, lv;
  int nn;     // rows of the target frame
  int off_n;  // its first row in the sequence (sequences have < 2^31 rows)
};
struct Steps {
  StepArg a[kMaxBatch];
};
struct LevelSizes {
  float v[kMaxLevels][3];
};

// ------------------------------------------------------------------ block-wide helpers (1024 threads)
// (wg_excl_scan, wg_radix_sort_packed: pcs_wgsort.hpp)
// Stream compaction by one workgroup: the n items are cut into 16 contiguous wave slices; flag(i)
// says whether item i is kept, emit(i, pos) receives its rank among the kept ones (ascending i).
// Loads inside flag() of eight rounds are issued together. Returns the number kept (all threads).
template <class FlagFn, class EmitFn>
__device__ __forceinline__ int wg_compact(int n, FlagFn flag, EmitFn emit, int* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per_wave = ((n + kWg - 1) / kWg) * 64;
  const int begin = w * per_wave;
  const int end = begin + per_wave < n ? begin + per_wave : n;
  constexpr int U = 8;
  int cnt = 0;
  for (int i0 = begin; i0 < end; i0 += 64 * U) {
    bool f[U];
#pragma unroll
    for (int r = 0; r < U; ++r) {
      int i = i0 + r * 64 + lane;
      f[r] = i < end && flag(i);
    }
#pragma unroll
    for (int r = 0; r < U; ++r) cnt += __popcll(__ballot(f[r]));
  }
  __syncthreads();  // sh may still be read from a previous call
  if (lane == 0) sh[w] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int q = 0; q < kWg / 64; ++q) { int t = sh[q]; sh[q] = run; run += t; }
    sh[kWg / 64] = run;
  }
  __syncthreads();
  const int total = sh[kWg / 64];
  int run = sh[w];
  for (int i0 = begin; i0 < end; i0 += 64 * U) {
    bool f[U];
#pragma unroll
    for (int r = 0; r < U; ++r) {
      int i = i0 + r * 64 + lane;
      f[r] = i < end && flag(i);
    }
#pragma unroll
    for (int r = 0; r < U; ++r) {
      unsigned long long b = __ballot(f[r]);
      if (f[r]) emit(i0 + r * 64 + lane, run + __popcll(b & ((1ULL << lane) - 1ULL)));
      run += __popcll(b);
    }
  }
  return total;
}

// Stable LSD radix sort of n (key, value) pairs by key bits [0, nbits), 8 bits per pass, by ONE
// workgroup of 16 waves: wave w owns a contiguous slice of the input, so "digit-major, wave-minor,
// position within the slice" is the stable order. Returns 0 if the result is in (k0, v0), 1 if in
// (k1, v1). hist: LDS [16 * 256]; sh: LDS [kWg / 64 + 1].
__device__ __forceinline__ int wg_radix_sort(uint64_t* k0, int32_t* v0, uint64_t* k1, int32_t* v1, int n,
                                             int nbits, int* hist, int* sh) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int per_wave = ((n + kWg - 1) / kWg) * 64;
  const int begin = w * per_wave;
  const int end = begin + per_wave < n ? begin + per_wave : n;
  uint64_t *kin = k0, *kout = k1;
  int32_t *vin = v0, *vout = v1;
  int which = 0;
  constexpr int U = 8;  // rounds whose loads are issued together
  for (int b = 0; b < nbits; b += 8) {
    for (int q = tid; q < 16 * 256; q += kWg) hist[q] = 0;
    __syncthreads();
    for (int i0 = begin; i0 < end; i0 += 64 * U) {
      uint64_t kk[U];
#pragma unroll
      for (int r = 0; r < U; ++r) {
        int i = i0 + r * 64 + lane;
        kk[r] = i < end ? kin[i] : ~0ULL;
      }
#pragma unroll
      for (int r = 0; r < U; ++r) {
        int i = i0 + r * 64 + lane;
        if (i < end) atomicAdd(&hist[w * 256 + (int)((kk[r] >> b) & 255)], 1);
      }
    }
    __syncthreads();
    // exclusive scan in (digit, wave) order: thread t owns entries 4t .. 4t+3 of that order
    int c[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int e = 4 * tid + q;
      c[q] = hist[(e & 15) * 256 + (e >> 4)];
      sum += c[q];
    }
    int tot;
    int run = wg_excl_scan(sum, &tot, sh);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int e = 4 * tid + q;
      hist[(e & 15) * 256 + (e >> 4)] = run;
      run += c[q];
    }
    __syncthreads();
    for (int i0 = begin; i0 < end; i0 += 64 * U) {
      uint64_t kk[U];
      int32_t vv[U];
#pragma unroll
      for (int r = 0; r < U; ++r) {
        int i = i0 + r * 64 + lane;
        kk[r] = i < end ? kin[i] : ~0ULL;
        vv[r] = i < end ? vin[i] : 0;
      }
#pragma unroll
      for (int r = 0; r < U; ++r) {
        int i = i0 + r * 64 + lane;
        bool in = i < end;
        int d = (int)((kk[r] >> b) & 255);
        unsigned long long same = __ballot(in);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
          unsigned long long bm = __ballot((d >> bit) & 1);
          same &= ((d >> bit) & 1) ? bm : ~bm;
        }
        int rank = __popcll(same & ((1ULL << lane) - 1ULL));
        int cnt = __popcll(same);
        int base = 0;
        if (in) base = hist[w * 256 + d];
        if (in && rank == 0) hist[w * 256 + d] = base + cnt;  // after every lane of the group has read
        if (in) {
          kout[base + rank] = kk[r];
          vout[base + rank] = vv[r];
        }
      }
    }
    __syncthreads();
    uint64_t* tk = kin; kin = kout; kout = tk;
    int32_t* tv = vin; vin = vout; vout = tv;
    which ^= 1;
  }
  return which;
}

// ------------------------------------------------------------------ anchor set-up (per unit)
__global__ void __launch_bounds__(kBlock)
trk_load_anchor_kernel(const float4* __restrict__ src, const int64_t* __restrict__ comp_in, int64_t base,
                       int n, float4* __restrict__ xyzf, int64_t* __restrict__ comp64,
                       int32_t* __restrict__ comp32) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  xyzf[i] = src[i];
  int64_t c = comp_in[i] - base;
  comp64[i] = c;
  comp32[i] = (int32_t)c;
}

__global__ void trk_copy_i64_kernel(const int64_t* __restrict__ src, int64_t* __restrict__ dst) { *dst = *src; }

// per present component u (id c = uniq[u]): count, centre = fl32(sum) / fl32(count)
__global__ void __launch_bounds__(kBlock)
trk_comp_center_kernel(const int64_t* __restrict__ uniq, const int32_t* __restrict__ seg, int Cu,
                       const float* __restrict__ red4, float* __restrict__ deg_f,
                       int64_t* __restrict__ deg_i, float* __restrict__ ctr /*[C,4]*/) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= Cu) return;
  int64_t c = uniq[u];
  int n = seg[u + 1] - seg[u];
  float nf = (float)n;
  deg_f[c] = nf;
  deg_i[c] = n;
  ctr[c * 4 + 0] = red4[u * 4 + 0] / nf;
  ctr[c * 4 + 1] = red4[u * 4 + 1] / nf;
  ctr[c * 4 + 2] = red4[u * 4 + 2] / nf;
}

__global__ void __launch_bounds__(kBlock)
trk_point_dist_kernel(const float4* __restrict__ xyzf, const int32_t* __restrict__ comp, int n,
                      const float* __restrict__ ctr, float* __restrict__ dist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = xyzf[i];
  const float* c = ctr + (int64_t)comp[i] * 4;
  float dx = p.y - c[0], dy = p.z - c[1], dz = p.w - c[2];
  dist[i] = sqrtf((dx * dx + dy * dy) + dz * dz);
}

__global__ void __launch_bounds__(kBlock)
trk_comp_diam_kernel(const int64_t* __restrict__ uniq, int Cu, const float* __restrict__ red1,
                     float* __restrict__ diam) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < Cu) diam[uniq[u]] = red1[u] * 2.f;
}

// per component: masks, identity transforms, output defaults, centre of the anchor frame; the
// unit's `active` word says whether any component passed filter_components
__global__ void __launch_bounds__(kBlock)
trk_comp_init_kernel(int C, int F, int W, int frame_id, float max_diam, const float* __restrict__ deg_f,
                     const float* __restrict__ diam, const float* __restrict__ ctr,
                     uint8_t* __restrict__ vcm, double* __restrict__ transforms,
                     int64_t* __restrict__ cmin, int64_t* __restrict__ cmax,
                     float* __restrict__ centers, TrkState* __restrict__ st) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bool valid = deg_f[c] > 0.5f;
  if (max_diam > 0.f) valid = valid && (diam[c] < max_diam);
  vcm[c] = valid;
  if (valid) st->active = 1;
  cmin[c] = frame_id;
  cmax[c] = frame_id;
  for (int w = 0; w < W; ++w) {
    double* T = transforms + ((int64_t)c * W + w) * 16;
    for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.0 : 0.0;
  }
  float* cc = centers + ((int64_t)c * F + frame_id) * 3;
  cc[0] = ctr[c * 4 + 0];
  cc[1] = ctr[c * 4 + 1];
  cc[2] = ctr[c * 4 + 2];
}

// grid description of the harvest searches (1-NN of the target frame's points among the moved
// anchor points, radius NN_GRAPH.RADIUS): one for the whole sequence, padded so that tracked points
// stay inside the cell range
// one table slot <- a descriptor passed by value
__global__ void trk_store_icp_kernel(IcpProblem* __restrict__ slot, IcpProblem desc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *slot = desc;
}
// table slots bt.idx[e] <- staged[e] for all registrations that start in a tick, in ONE launch (`staged`
// is pinned host memory the kernel reads directly: a few hundred bytes per entry; it was one launch per
// unit, 10-30 launches at the head of a tick once a tick serves a hundred units)
__global__ void __launch_bounds__(64)
trk_store_icp_batch_kernel(IcpProblem* __restrict__ tab, const IcpProblem* __restrict__ staged, Batch bt) {
  const int e = blockIdx.x;
  if (e >= bt.n) return;
  const unsigned long long* src = (const unsigned long long*)(staged + e);
  unsigned long long* dst = (unsigned long long*)(tab + bt.idx[e]);
  static_assert(sizeof(IcpProblem) % 8 == 0, "descriptor copied in 8-byte words");
  for (int q = threadIdx.x; q < (int)(sizeof(IcpProblem) / 8); q += 64) dst[q] = src[q];
}

__global__ void trk_hdesc_kernel(const float* __restrict__ b8, float radius, float cell, float pad,
                                 GridDesc* __restrict__ g) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float vs0 = (float)(1 - 1e-3);
  g->vs0 = vs0;
  g->lo0 = 0.f - vs0 * 2.f;
  for (int d = 0; d < 3; ++d) g->lo[d] = b8[1 + d] - pad;
  g->h = cell;
  g->radius = radius;
  g->radius2 = radius * radius;
  g->rings = (int)ceilf(radius / cell);
}

// ------------------------------------------------------------------ batched kernels
// mirror of the units' state words into pinned host memory (the host then needs the stream
// synchronisation only)
__global__ void trk_publish_kernel(const UnitDev* __restrict__ U, Batch bt, TrkState* __restrict__ out) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < bt.n) out[e] = *U[bt.idx[e]].st;  // slot = position in the list
}

// the anchor's own points: kept when their component passed filter_components (one workgroup per unit)
__global__ void __launch_bounds__(kWg)
trk_anchor_emit_kernel(const UnitDev* __restrict__ U, Batch bt) {
  __shared__ int sh[kWg / 64 + 1];
  const UnitDev& u = U[bt.idx[blockIdx.x]];
  int tot = wg_compact(
      u.na, [&](int i) { return u.vcm[u.comp32[i]] != 0; },
      [&](int i, int pos) {
        if (pos < u.ex_cap) {
          u.ex_index[pos] = u.off_a + i;
          u.ex_local[pos] = i;
          u.ex_comp[pos] = u.comp32[i];
        } else {
          u.st->err = 2;
        }
      },
      sh);
  if (threadIdx.x == 0) u.st->ex_count = tot;
}

// start of a direction: nothing stopped but the components that failed filter_components, the
// anchor's points back where they were
__global__ void __launch_bounds__(kBlock)
trk_dir_init_kernel(const UnitDev* __restrict__ U, Batch bt) {
  const UnitDev& u = U[bt.idx[blockIdx.y]];
  const float4* anchor = u.seq + u.off_a;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < max(u.C, u.na); i += gridDim.x * blockDim.x) {
    if (i < u.C) {
      bool v = u.vcm[i];
      u.stopped[i] = !v;
      u.moving[i] = v;
    }
    if (i < u.na) {
      float4 p = anchor[i];
      u.xyzf[i] = p;
      u.last[i] = p;
    }
  }
}

// transforms[:, col] = transforms[:, col - dir]; constant-velocity prediction for the components
// still tracked (cluster_tracking.py:520-534)
__global__ void __launch_bounds__(kBlock)
trk_step_begin_kernel(const UnitDev* __restrict__ U, Batch bt, Steps sa) {
  const UnitDev& u = U[bt.idx[blockIdx.y]];
  const StepArg a = sa.a[blockIdx.y];
  const int F = u.F, W = u.W;
  const float fd = (float)a.dir;
  if (blockIdx.x == 0 && threadIdx.x == 0) u.st->active = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < max(u.C, u.na); i += gridDim.x * blockDim.x) {
    if (i < u.C) {
      double* dst = u.transforms + ((int64_t)i * W + a.col) * 16;
      const double* src = u.transforms + ((int64_t)i * W + (a.col - a.dir)) * 16;
      for (int k = 0; k < 16; ++k) dst[k] = src[k];
      if (a.last_col >= 0 && !u.stopped[i]) {
        const float* v = u.velos + ((int64_t)i * F + a.last_col) * 3;
        dst[3] += (double)(v[0] * fd);
        dst[7] += (double)(v[1] * fd);
        dst[11] += (double)(v[2] * fd);
      }
    }
    if (i < u.na && a.last_col >= 0) {
      int c = u.comp32[i];
      if (!u.stopped[c]) {
        const float* v = u.velos + ((int64_t)c * F + a.last_col) * 3;
        float4 p = u.xyzf[i];
        p.y = p.y + v[0] * fd;
        p.z = p.z + v[1] * fd;
        p.w = p.w + v[2] * fd;
        u.xyzf[i] = p;
      }
    }
  }
}

// ---- sample_frame, part 1: grid_cluster ids (same arithmetic as grid_ids_kernel: start / end =
// min / max of the cloud, frame axis widened by 0.5), sorted unique. ONE workgroup per (unit, set):
// bounds, keys, stable radix sort on exactly the bits the ids can have, voxel boundaries.
// blockIdx.x = set (0: the anchor's points as moved so far, 1: the target frame).
__global__ void __launch_bounds__(kWg)
smp_sort_kernel(const UnitDev* __restrict__ U, Batch bt, Steps sa, LevelSizes ls) {
  __shared__ int hist[16 * 256], hist2[16 * 256];
  __shared__ int sh[kWg / 64 + 1];
  __shared__ float smn[kWg / 64][4], smx[kWg / 64][4];
  __shared__ float bnd[8];
  const UnitDev& u = U[bt.idx[blockIdx.y]];
  const StepArg a = sa.a[blockIdx.y];
  const int set = blockIdx.x;
  const SmpSet& S = set ? u.sb : u.sa;
  const float4* __restrict__ pts = set ? u.seq + a.off_n : (const float4*)u.xyzf;
  const int n = set ? a.nn : u.na;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (set == 0)
    for (int c = tid; c < u.C; c += kWg) u.sub_deg[c] = 0ULL;
  if (n < 1) {
    if (tid == 0) { *S.d_num = 0; S.seg[0] = 0; }
    return;
  }
  // bounds (eight rows requested together: the loop is a chain of memory round trips otherwise)
  {
    float mn[4] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
    float mx[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i0 = tid; i0 < n; i0 += kWg * 8) {
      float4 q[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int i = i0 + r * kWg;
        q[r] = pts[i < n ? i : i0];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float4 p = q[r];
        mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
        mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
        mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
        mn[3] = fminf(mn[3], p.w); mx[3] = fmaxf(mx[3], p.w);
      }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      float x = wave_min(mn[d]), y = wave_max(mx[d]);
      if (lane == 0) { smn[w][d] = x; smx[w][d] = y; }
    }
    __syncthreads();
    if (tid < 4) {
      float x = smn[0][tid], y = smx[0][tid];
      for (int k = 1; k < kWg / 64; ++k) { x = fminf(x, smn[k][tid]); y = fmaxf(y, smx[k][tid]); }
      bnd[tid] = x;
      bnd[4 + tid] = y;
    }
    __syncthreads();
  }
  float st[4], en[4], sz[4] = {1.f, ls.v[a.lv][0], ls.v[a.lv][1], ls.v[a.lv][2]};
#pragma unroll
  for (int d = 0; d < 4; ++d) { st[d] = bnd[d]; en[d] = bnd[4 + d]; }
  st[0] = st[0] - 0.5f;
  en[0] = en[0] + 0.5f;
  int64_t stride[5];
  stride[0] = 1;
#pragma unroll
  for (int d = 0; d < 4; ++d) stride[d + 1] = stride[d] * ((int64_t)((en[d] - st[d]) / sz[d]) + 1);
  int nbits = 1;
  while (nbits < 62 && (1LL << nbits) < stride[4]) ++nbits;
  int ib = 1;
  while ((1 << ib) < n) ++ib;
  const bool packed = nbits + ib <= 64;  // (always, for clouds of this size: ~35 + 17 bits)
  const int per_wave = ((n + kWg - 1) / kWg) * 64;
  if (packed) {
    for (int q = tid; q < 16 * 256; q += kWg) hist[q] = 0;
    __syncthreads();
  }
  for (int i0 = tid; i0 < n; i0 += kWg * 8) {
    float4 q[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      int i = i0 + r * kWg;
      q[r] = pts[i < n ? i : i0];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      int i = i0 + r * kWg;
      if (i >= n) continue;
      float4 p = q[r];
      int64_t id = (int64_t)((p.x - st[0]) / sz[0]) * stride[0] + (int64_t)((p.y - st[1]) / sz[1]) * stride[1] +
                   (int64_t)((p.z - st[2]) / sz[2]) * stride[2] + (int64_t)((p.w - st[3]) / sz[3]) * stride[3];
      if (packed) {
        S.k0[i] = ((uint64_t)id << ib) | (uint64_t)i;
        atomicAdd(&hist[(i / per_wave) * 256 + (int)(id & 255)], 1);
      } else {
        S.k0[i] = (uint64_t)id;
        S.v0[i] = i;
      }
    }
  }
  __syncthreads();
  int tot;
  if (packed) {
    int which = wg_radix_sort_packed(S.k0, S.k1, n, ib, nbits, hist, hist2, sh);
    const uint64_t* __restrict__ sw = which ? S.k1 : S.k0;
    // voxel boundaries + the sorted order in its final place
    tot = wg_compact(
        n, [&](int i) { return i == 0 || (sw[i] >> ib) != (sw[i - 1] >> ib); },
        [&](int i, int pos) { S.seg[pos] = i; }, sh);
    const uint64_t imask = (1ULL << ib) - 1ULL;
    for (int i0 = tid; i0 < n; i0 += kWg * 8) {
      uint64_t vv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int i = i0 + r * kWg;
        vv[r] = sw[i < n ? i : i0];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int i = i0 + r * kWg;
        if (i < n) S.order[i] = (int32_t)(vv[r] & imask);
      }
    }
  } else {
    int which = wg_radix_sort(S.k0, S.v0, S.k1, S.v1, n, nbits, hist, sh);
    const uint64_t* __restrict__ sk = which ? S.k1 : S.k0;
    const int32_t* __restrict__ sv = which ? S.v1 : S.v0;
    tot = wg_compact(
        n, [&](int i) { return i == 0 || sk[i] != sk[i - 1]; }, [&](int i, int pos) { S.seg[pos] = i; }, sh);
    for (int i0 = tid; i0 < n; i0 += kWg * 8) {
      int32_t vv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int i = i0 + r * kWg;
        vv[r] = sv[i < n ? i : i0];
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int i = i0 + r * kWg;
        if (i < n) S.order[i] = vv[r];
      }
    }
  }
  if (tid == 0) { S.seg[tot] = n; *S.d_num = tot; }
}

// ---- sample_frame, part 2: per voxel the serial mean of fxyz, the stationary vote and (anchor) the
// median component and the per-component voxel count. blockIdx.z = set.
__global__ void __launch_bounds__(kBlock)
smp_emit_kernel(const UnitDev* __restrict__ U, Batch bt, Steps sa) {
  const UnitDev& u = U[bt.idx[blockIdx.y]];
  const StepArg a = sa.a[blockIdx.y];
  const int set = blockIdx.z;
  const SmpSet& S = set ? u.sb : u.sa;
  const int v = (int)*S.d_num;
  const int nround = (v + kBlock - 1) / kBlock;
  if ((int)blockIdx.x >= nround) return;
  const float4* __restrict__ pts = set ? u.seq + a.off_n : (const float4*)u.xyzf;
  const uint8_t* __restrict__ stat = u.seq_stat + (set ? a.off_n : u.off_a);
  const int64_t* __restrict__ comp64 = u.comp64;
  const int32_t* __restrict__ order = S.order;
  const int32_t* __restrict__ seg = S.seg;
  for (int r = blockIdx.x; r < nround; r += gridDim.x) {  // whole waves stay together (ballots below)
    const int s = r * kBlock + threadIdx.x;
    int64_t comp_of = -1;  // anchor voxels: the voxel's (median) component
    if (s < v) {
      const int bb = seg[s], e = seg[s + 1];
      // a coarse voxel holds dozens of rows and every row is two dependent loads (order -> point):
      // eight rows are requested together and then added in their original order (same sums, an
      // eighth of the round trips); the anchor's component ids ride along
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, stv = 0.f;
      int64_t first = 0;
      bool same = true;
      for (int p0 = bb; p0 < e; p0 += 8) {
        int o[8];
        float4 q[8];
        float sv[8];
        int64_t cc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (p0 + j < e) ? order[p0 + j] : -1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          sv[j] = 0.f;
          cc[j] = 0;
          if (o[j] >= 0) {
            q[j] = pts[o[j]];
            sv[j] = stat[o[j]] ? 1.f : 0.f;
            if (!set) cc[j] = comp64[o[j]];
          }
        }
        if (p0 == bb) first = cc[0];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (o[j] >= 0) {
            a0 = a0 + q[j].x; a1 = a1 + q[j].y; a2 = a2 + q[j].z; a3 = a3 + q[j].w;
            stv = stv + sv[j];
            same = same && (cc[j] == first);
          }
        }
      }
      float cnt = (float)(e - bb);
      S.sub[s] = make_float4(a0 / cnt, a1 / cnt, a2 / cnt, a3 / cnt);
      S.flag[s] = !((stv / cnt) > 0.5f);
      if (!set) {
        int k = (e - bb) / 2;
        int64_t res = first;
        if (!same) {
          for (int p = bb; p < e; ++p) {
            int64_t x = comp64[order[p]];
            int lt = 0, eq = 0;
            for (int q2 = bb; q2 < e; ++q2) {
              int64_t y = comp64[order[q2]];
              lt += (y < x);
              eq += (y == x);
            }
            if (lt <= k && k < lt + eq) { res = x; break; }
          }
        }
        S.sub_comp[s] = (int32_t)res;
        comp_of = res;
      }
    }
    // per-component voxel counts: neighbouring voxels mostly share a component and a large component
    // owns thousands of voxels, so one atomic per voxel queues up on a few addresses (~12 ns each);
    // the lanes of a wave that agree on the component send one add between them
    if (!set) {
      const int lane = threadIdx.x & 63;
      unsigned long long todo = __ballot(comp_of >= 0);
      while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        long long lc = __shfl((long long)comp_of, leader, 64);
        unsigned long long grp = __ballot(comp_of == lc);
        if (lane == leader) atomicAdd(&u.sub_deg[lc], (unsigned long long)__popcll(grp));
        todo &= ~grp;
      }
    }
  }
}

// ---- sample_frame, part 3: the non-stationary voxels, compacted in voxel order, and the counts the
// registration reads. One workgroup per (unit, set); blockIdx.x = set.
__global__ void __launch_bounds__(kWg)
smp_compact_kernel(const UnitDev* __restrict__ U, Batch bt) {
  __shared__ int sh[kWg / 64 + 1];
  const UnitDev& u = U[bt.idx[blockIdx.y]];
  const int set = blockIdx.x;
  const SmpSet& S = set ? u.sb : u.sa;
  const int v = (int)*S.d_num;
  int tot = wg_compact(
      v, [&](int s) { return S.flag[s] != 0; },
      [&](int s, int pos) {
        S.out[pos] = S.sub[s];
        if (!set) S.out_comp[pos] = S.sub_comp[s];
      },
      sh);
  if (threadIdx.x == 0) {
    u.st->counts[2 * set] = v;
    u.st->counts[2 * set + 1] = tot;
  }
}

// frame.fxyz[:, 1:] = (R @ xyz.double()).float() + t.float(); transforms[:, col] = T @ transforms[:, col]
__global__ void __launch_bounds__(kBlock)
trk_apply_T_kernel(const UnitDev* __restrict__ U, Batch bt, Steps sa) {
  const UnitDev& u = U[bt.idx[blockIdx.y]];
  const StepArg a = sa.a[blockIdx.y];
  const double* __restrict__ T = u.Tl;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < max(u.C, u.na); i += gridDim.x * blockDim.x) {
    if (i < u.na) {
      const double* M = T + (int64_t)u.comp32[i] * 16;
      float4 p = u.xyzf[i];
      double x = (double)p.y, y = (double)p.z, z = (double)p.w;
      float rx = (float)((M[0] * x + M[1] * y) + M[2] * z);
      float ry = (float)((M[4] * x + M[5] * y) + M[6] * z);
      float rz = (float)((M[8] * x + M[9] * y) + M[10] * z);
      p.y = rx + (float)M[3];
      p.z = ry + (float)M[7];
      p.w = rz + (float)M[11];
      u.xyzf[i] = p;
    }
    if (i < u.C) {
      const double* A = T + (int64_t)i * 16;
      double* B = u.transforms + ((int64_t)i * u.W + a.col) * 16;
      double R[16];
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
          double acc = A[r * 4 + 0] * B[0 * 4 + c];
          acc = acc + A[r * 4 + 1] * B[1 * 4 + c];
          acc = acc + A[r * 4 + 2] * B[2 * 4 + c];
          acc = acc + A[r * 4 + 3] * B[3 * 4 + c];
          R[r * 4 + c] = acc;
        }
      for (int k = 0; k < 16; ++k) B[k] = R[k];
    }
  }
}

// per present component (one wave each): centre of the moved points, mean point velocity
// (xyz - last_xyz) * dir with z zeroed, centre difference (cluster_tracking.py:629-648). The sums are
// the ordered wave sums of pcs_segment_reduce_f32 (lane l adds rows l, l + 64, ... in order, then a
// 64-lane butterfly), so they equal the op-by-op path bit for bit.
__global__ void __launch_bounds__(kBlock)
trk_motion_kernel(const UnitDev* __restrict__ U, Batch bt, Steps sa) {
  const UnitDev& u = U[bt.idx[blockIdx.y]];
  const StepArg a = sa.a[blockIdx.y];
  const int Cu = (int)*u.d_numc;
  const int lane = threadIdx.x & 63;
  const int F = u.F;
  const float fd = (float)a.dir;
  for (int seg = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; seg < Cu; seg += (gridDim.x * blockDim.x) >> 6) {
    const int b = u.seg_c[seg], e = u.seg_c[seg + 1];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, v0 = 0.f, v1 = 0.f, v2 = 0.f;
    for (int p = b + lane; p < e; p += 64) {
      int i = u.order_c[p];
      float4 x = u.xyzf[i], q = u.last[i];
      s0 = s0 + x.y; s1 = s1 + x.z; s2 = s2 + x.w;
      v0 = v0 + (x.y - q.y) * fd; v1 = v1 + (x.z - q.z) * fd; v2 = v2 + (x.w - q.w) * fd;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s0 = s0 + __shfl_xor(s0, o, 64); s1 = s1 + __shfl_xor(s1, o, 64); s2 = s2 + __shfl_xor(s2, o, 64);
      v0 = v0 + __shfl_xor(v0, o, 64); v1 = v1 + __shfl_xor(v1, o, 64); v2 = v2 + __shfl_xor(v2, o, 64);
    }
    if (lane == 0) {
      const int64_t c = u.uniq_c[seg];
      const float nf = (float)(e - b);
      float* cn = u.centers + ((int64_t)c * F + a.next) * 3;
      const float* cp = u.centers + ((int64_t)c * F + (a.next - a.dir)) * 3;
      float* vn = u.velos + ((int64_t)c * F + a.next) * 3;
      float* dn = u.diffs + ((int64_t)c * F + a.next) * 3;
      const float sum[3] = {s0, s1, s2};
      const float mean[3] = {v0 / nf, v1 / nf, v2 / nf};
      for (int k = 0; k < 3; ++k) {
        float ctr = sum[k] / nf;
        cn[k] = ctr;
        dn[k] = (ctr - cp[k]) * fd;
        float v = k < 2 ? mean[k] : 0.f;
        vn[k] = v;
        u.velo_pre[c * 3 + k] = v;
      }
    }
  }
}

__device__ __forceinline__ float norm3(float a, float b, float c) {
  return sqrtf((a * a + b * b) + c * c);
}

__device__ __forceinline__ float dist_compensate(int64_t deg) {
  // cluster_tracking.py:80-87
  if (deg < 0) return 0.f;
  if (deg < 10) return 1.f;
  if (deg < 40) return 0.5f;
  if (deg < 100) return 0.3f;
  if (deg < 200) return 0.2f;
  if (deg < 400) return 0.1f;
  return 0.f;
}

// after smooth_velo: move the points by the smoothing correction, then the stop rules
// (cluster_tracking.py:650-700) and the per-step records; the harvest grid is reset on the way
__global__ void __launch_bounds__(kBlock)
trk_after_smooth_kernel(const UnitDev* __restrict__ U, Batch bt, Steps sa, int min_move_frame,
                        float reg_coeff, float angle_thr) {
  const UnitDev& u = U[bt.idx[blockIdx.y]];
  const StepArg a = sa.a[blockIdx.y];
  const int F = u.F, W = u.W, next = a.next, dir = a.dir, frame_id = u.anchor, last_col = a.last_col;
  const float fd = (float)dir;
  g2_clear(u.hg, blockIdx.x * (int64_t)blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < max(u.C, u.na); i += gridDim.x * blockDim.x) {
    if (i < u.na) {
      int c = u.comp32[i];
      const float* vn = u.velos + ((int64_t)c * F + next) * 3;
      const float* vp = u.velo_pre + (int64_t)c * 3;
      float4 p = u.xyzf[i];
      p.y = p.y + (vn[0] - vp[0]) * fd;
      p.z = p.z + (vn[1] - vp[1]) * fd;
      p.w = p.w + (vn[2] - vp[2]) * fd;
      u.xyzf[i] = p;
      u.last[i] = p;
    }
    if (i < u.C) {
      int c = i;
      float* vn = u.velos + ((int64_t)c * F + next) * 3;
      const float* vp = u.velo_pre + (int64_t)c * 3;
      double* T = u.transforms + ((int64_t)c * W + a.col) * 16;
      T[3] += (double)((vn[0] - vp[0]) * fd);
      T[7] += (double)((vn[1] - vp[1]) * fd);
      T[11] += (double)((vn[2] - vp[2]) * fd);
      bool stop = u.stopped[c];
      float dm = u.diam[c];
      float thr = (reg_coeff * dm) * (1.f + dist_compensate(u.deg_i[c]));
      stop = stop || (u.l1[c] > (double)thr);
      stop = stop || (u.ratio[c] < 0.5f);
      if ((next - frame_id) * dir == min_move_frame) {
        const float* cn = u.centers + ((int64_t)c * F + next) * 3;
        const float* c0 = u.centers + ((int64_t)c * F + frame_id) * 3;
        float moved = norm3(cn[0] - c0[0], cn[1] - c0[1], cn[2] - c0[2]);
        u.moving[c] = u.moving[c] && (moved > 0.08f * dm);
      }
      float v0 = vn[0], v1 = vn[1], v2 = vn[2];
      if (last_col >= 0) {
        const float* lv = u.velos + ((int64_t)c * F + last_col) * 3;
        float dev = norm3(v0 - lv[0], v1 - lv[1], v2 - lv[2]);
        stop = stop || (dev > 0.24f * dm);
        const float* pv = u.velos + ((int64_t)c * F + (next - dir)) * 3;
        float nrm = norm3(v0, v1, v2) * norm3(pv[0], pv[1], pv[2]);
        nrm = fmaxf(nrm, 1e-6f);
        float cosv = ((v0 * pv[0] + v1 * pv[1]) + v2 * pv[2]) / nrm;
        cosv = fminf(fmaxf(cosv, -1.f), 1.f);
        float angle = acosf(cosv) / 3.14159265358979323846f * 180.0f;
        float nxy = sqrtf(v0 * v0 + v1 * v1);
        stop = stop || ((angle > angle_thr) && (nxy > 0.01f));
      }
      if (next == frame_id - 1) {
        float* vf = u.velos + ((int64_t)c * F + frame_id) * 3;
        vf[0] = v0; vf[1] = v1; vf[2] = v2;
      }
      u.stopped[c] = stop;
      if (!stop) {
        if (dir == -1) u.cmin[c] = next; else u.cmax[c] = next;
        u.st->active = 1;
      }
      u.reg_errors[(int64_t)c * F + next] = (float)u.l1[c];
      u.edge_ratios[(int64_t)c * F + next] = u.ratio[c];
    }
  }
}

// ---- harvest (cluster_tracking.py:702-741): the target frame's points whose nearest moved anchor
// point (NN_GRAPH: 1-NN within the radius, same frame) belongs to a component still tracked
__global__ void __launch_bounds__(kBlock)
hv_insert_kernel(const UnitDev* __restrict__ U, Batch bt, Steps sa) {
  const UnitDev& u = U[bt.idx[blockIdx.y]];
  const StepArg a = sa.a[blockIdx.y];
  if (a.nn < 1) return;
  const GridDesc d = *u.hdesc;
  const int nround = (u.na + kBlock - 1) / kBlock;
  for (int r = blockIdx.x; r < nround; r += gridDim.x) {
    int i = r * kBlock + threadIdx.x;
    bool in = i < u.na;
    float4 p = in ? u.xyzf[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    p.x = (float)a.next;  // frame.fxyz[:, 0] = next_frame_id for the duration of the query
    g2_insert_point(d, u.hg, in, i, p, &u.st->err);
  }
}

__global__ void __launch_bounds__(kBlock)
hv_search_kernel(const UnitDev* __restrict__ U, Batch bt, Steps sa) {
  const UnitDev& u = U[bt.idx[blockIdx.y]];
  const StepArg a = sa.a[blockIdx.y];
  const int M = a.nn;
  const GridDesc d = *u.hdesc;
  const float4* __restrict__ qry = u.seq + a.off_n;
  const int nround = (M + kBlock / 32 - 1) / (kBlock / 32);
  for (int r = blockIdx.x; r < nround; r += gridDim.x) {
    int qid = r * (kBlock / 32) + (threadIdx.x >> 5);
    bool valid = qid < M;
    float4 p = valid ? qry[qid] : make_float4(0.f, 0.f, 0.f, 0.f);
    int rr = g2_nearest_coop(d, u.hg, p, 0, valid, &u.st->err);
    if (valid && (threadIdx.x & 31) == 0) {
      // kept when the nearest moved anchor point belongs to a component that is still tracked
      int c = rr >= 0 ? u.comp32[rr] : -1;
      u.hv_deg[qid] = (c >= 0 && !u.stopped[c]) ? 1 : 0;
      u.hv_nbr[qid] = c;
    }
  }
}

// kept rows appended in point order: (index into the sequence, index within the frame, component).
// One workgroup per unit.
__global__ void __launch_bounds__(kWg)
hv_compact_kernel(const UnitDev* __restrict__ U, Batch bt, Steps sa) {
  __shared__ int sh[kWg / 64 + 1];
  const UnitDev& u = U[bt.idx[blockIdx.x]];
  const StepArg a = sa.a[blockIdx.x];
  const int m = a.nn;
  if (m < 1) return;
  const int64_t base = u.st->ex_count;
  int tot = wg_compact(
      m, [&](int q) { return u.hv_deg[q] != 0; },
      [&](int q, int pos) {
        int64_t o = base + pos;
        if (o < u.ex_cap) {
          u.ex_index[o] = a.off_n + q;
          u.ex_local[o] = q;
          u.ex_comp[o] = u.hv_nbr[q];
        } else {
          u.st->err = 2;
        }
      },
      sh);
  __syncthreads();  // every thread has read ex_count
  if (threadIdx.x == 0) u.st->ex_count = base + tot;
}

int smooth_velo_run_batch(int n, float* const* comp_velos, const float* const* comp_center_diffs,
                          const int64_t* C, const int* F, const int* frame_lo, const int* frame_hi,
                          float weight0, float weight, int num_itr, float stopping, void* const* ws,
                          const size_t* ws_bytes, hipStream_t s);

// ------------------------------------------------------------------ host side
struct UnitPlan {
  UnitDev d;
  void *ws_unique, *ws_icp, *ws_sv;
  size_t unique_bytes, icp_bytes, sv_bytes;
  size_t zero_begin, zero_end;
};

static void plan_set(Arena& a, int64_t n, bool anchor, SmpSet& S) {
  S.k0 = a.take<uint64_t>(n);
  S.k1 = a.take<uint64_t>(n);
  S.v0 = a.take<int32_t>(n);
  S.v1 = a.take<int32_t>(n);
  S.order = a.take<int32_t>(n);
  S.seg = a.take<int32_t>(n + 1);
  S.d_num = a.take<int64_t>(1);
  S.sub = a.take<float4>(n);
  S.sub_comp = anchor ? a.take<int32_t>(n) : nullptr;
  S.flag = a.take<int32_t>(n);
  S.out = a.take<float4>(n);
  S.out_comp = anchor ? a.take<int32_t>(n) : nullptr;
}

static bool plan_unit(Arena& a, int64_t na, int64_t nm, int64_t C, int F, UnitPlan& p) {
  if (na < 1) na = 1;
  if (nm < 1) nm = 1;
  UnitDev& d = p.d;
  d.st = a.take<TrkState>(1);
  d.xyzf = a.take<float4>(na + 1);  // + 1: the d = 4 reductions read one float past the last row
  d.last = a.take<float4>(na);
  d.comp64 = a.take<int64_t>(na);
  d.comp32 = a.take<int32_t>(na);
  d.dist = a.take<float>(na);
  d.order_c = a.take<int32_t>(na);
  d.seg_c = a.take<int32_t>(na + 1);
  d.uniq_c = a.take<int64_t>(na);
  d.d_numc = a.take<int64_t>(1);
  p.zero_begin = a.off;
  d.deg_f = a.take<float>(C);
  d.diam = a.take<float>(C);
  d.ctr = a.take<float>(C * 4);
  d.centers = a.take<float>(C * (int64_t)F * 3);
  d.diffs = a.take<float>(C * (int64_t)F * 3);
  d.velo_pre = a.take<float>(C * 3);
  d.ratio = a.take<float>(C);
  d.ratio_tmp = a.take<float>(C);
  d.deg_i = a.take<int64_t>(C);
  d.sub_deg = a.take<unsigned long long>(C);
  d.stopped = a.take<uint8_t>(C);
  d.l1 = a.take<double>(C);
  d.l1_tmp = a.take<double>(C);
  p.zero_end = a.off;
  d.Tl = a.take<double>(C * 16);
  d.red4 = a.take<float>(na * 4);
  d.red1 = a.take<float>(na);
  plan_set(a, na, true, d.sa);
  plan_set(a, nm, false, d.sb);
  plan_grid2(a, na, d.hg);
  d.hv_nbr = a.take<int32_t>(nm);
  d.hv_deg = a.take<int32_t>(nm);
  p.unique_bytes = pcs_unique_i64_workspace_bytes(na);
  p.ws_unique = a.take<char>(p.unique_bytes);
  p.icp_bytes = pcs_register_to_next_frame_workspace_bytes(na, nm, nm, C);
  p.ws_icp = a.take<char>(p.icp_bytes);
  p.sv_bytes = pcs_smooth_velo_workspace_bytes(C, F);
  p.ws_sv = a.take<char>(p.sv_bytes);
  return a.ok();
}

// what the units of one sequence share
struct SeqPlan {
  float* seq_b8;
  GridDesc* hdesc;
};

static bool plan_seq(Arena& a, SeqPlan& e) {
  e.seq_b8 = a.take<float>(8);
  e.hdesc = a.take<GridDesc>(1);
  return a.ok();
}

constexpr int kMaxCallUnits = 32;  // units one sequence may hand over (their descriptors are staged in one pinned page)

enum Phase { PH_FREE, PH_RESERVED, PH_START, PH_ICP, PH_POSTWAIT, PH_FINALWAIT, PH_DONE };

// one blocking call into the engine (lives on the caller's stack)
struct Call {
  std::mutex mu;
  std::condition_variable cv;
  int remaining = 0;  // units handed to the engine that have not finished
  int rc = PCS_OK;
};

struct HostUnit {
  PcsTrackUnit* io;
  Call* call;
  const int64_t* frame_offsets;  // the caller's host array (the caller blocks until the unit is done)
  UnitPlan p;
  int na, nm, C, F, W, anchor, min_f, max_f;
  int64_t off_a;
  int phase, dir, next, last_col, lv;
  int loop_before;  // iterations run before the last chunk (profile accounting)
  int fail;       // error found while collecting a tick (PCS_ERR_*)
  bool any_valid;
  int Cu;
  int64_t steps, reg_calls, tick0;
  int seen[kMaxLevels][3];  // {Mn, Ra, Rn} of the last sampling at each level (launch-size estimates)
  IcpHint hint;
  TrkState host;
  IcpState icp;
};

// EXPERIMENT (PCS_TICK_PROFILE=1): where the driving thread's time goes -- enqueueing a tick's launches,
// waiting for the tick's event; printed when the process ends
struct TickProfile {
  std::atomic<long long> enq_ns{0}, wait_ns{0}, ticks{0}, units{0}, iter_units{0};
  bool on = getenv("PCS_TICK_PROFILE") != nullptr;
  ~TickProfile() {
    if (on && ticks.load())
      fprintf(stderr, "[tick profile] ticks %lld: per tick enqueue %.1f us, wait %.1f us, units %.1f (iterating %.1f)\n",
              ticks.load(), enq_ns.load() / 1e3 / ticks.load(), wait_ns.load() / 1e3 / ticks.load(),
              (double)units.load() / ticks.load(), (double)iter_units.load() / ticks.load());
  }
};
static TickProfile g_tick_profile;
static inline long long now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline int nblk(int64_t n) { return (int)(((n < 1 ? 1 : n) + kBlock - 1) / kBlock); }

static inline int pow2_words(int64_t n) {
  int64_t h = 64;
  while (h < 2 * n) h <<= 1;
  return (int)h;
}

static bool same_params(const PcsTrackParams& a, const PcsTrackParams& b) {
  if (a.num_levels != b.num_levels || a.max_iter != b.max_iter || a.min_move_frame != b.min_move_frame ||
      a.smooth_num_itr != b.smooth_num_itr || a.angle_regularizer != b.angle_regularizer ||
      a.nn_radius != b.nn_radius || a.reg_error_coeff != b.reg_error_coeff ||
      a.angle_threshold != b.angle_threshold || a.max_diameter != b.max_diameter ||
      a.smooth_weight0 != b.smooth_weight0 || a.smooth_weight != b.smooth_weight ||
      a.smooth_stopping != b.smooth_stopping)
    return false;
  for (int l = 0; l < a.num_levels && l < PCS_TRACK_MAX_LEVELS; ++l) {
    if (a.radius[l] != b.radius[l] || a.stopping_delta[l] != b.stopping_delta[l]) return false;
    for (int k = 0; k < 3; ++k)
      if (a.voxel_size[l][k] != b.voxel_size[l][k]) return false;
  }
  return true;
}

// The lock-step engine of a device. Tracking units are independent chains of small, latency-bound
// launches whose duration hardly depends on their width, so ALL units that are being tracked on the
// device -- whichever call, host thread or sequence they come from -- advance together: one driver
// thread, one stream, one table of unit / ICP descriptors, and every launch of a tick serves every
// unit that is in that phase. A call prepares its units on its own stream (workspace layout, anchor
// set-up), hands them over and blocks until they are done; units join and leave at tick boundaries
// ("continuous batching"), so several sequences in flight give launches several sequences wide
// without any of them waiting for the others. A unit's results never depend on what shares its
// launches (integer moments, exact neighbours, fixed-order sums).
class Engine {
 public:
  // A device has a few engines (pcs_track_configure, default 3; measured 1 / 2 / 3 / 4 with early creation: 1747 /
  // 1537 / 1494 / 1638 ms per batch of 12: while one chain is in its single-workgroup phases -- sampling sort,
  // smooth_velo, compactions -- the others' iteration launches fill the chip; the runtime has four hardware
  // queues); a call's units all join the engine that has the fewest units at that moment.
  static constexpr int kMaxEngines = 8;
  static std::atomic<int>& cfg_engines() { static std::atomic<int> v{3}; return v; }
  static std::atomic<unsigned>& cfg_chunk_us() { static std::atomic<unsigned> v{3000u}; return v; }
  static int count() { return cfg_engines().load(std::memory_order_relaxed); }
  static Engine* get(int device) {
    static std::mutex mu;
    static Engine* inst[64][kMaxEngines] = {};
    const int n_engines = count();
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    Engine* best = nullptr;
    for (int g = 0; g < n_engines; ++g) {
      if (!inst[device][g]) {
        Engine* e = new Engine(device);  // never destroyed: the driver thread outlives main
        if (!e->start()) return nullptr;
        inst[device][g] = e;
      }
      if (!best || inst[device][g]->load() < best->load()) best = inst[device][g];
    }
    return best;
  }
  int load() {
    std::lock_guard<std::mutex> lk(mu_);
    return users_;
  }

  // n free slots for units with these parameters; blocks while the table is full or units with other
  // parameters are in flight
  int reserve(const PcsTrackParams& prm, int n, int* slots) {
    if (n < 1 || n > kMaxBatch) return PCS_ERR_ARG;
    std::unique_lock<std::mutex> lk(mu_);
    cv_slots_.wait(lk, [&] { return broken_ || (free_ >= n && (users_ == 0 || same_params(prm, prm_))); });
    if (broken_) return PCS_ERR_HIP;
    if (users_ == 0) prm_ = prm;
    int got = 0;
    for (int i = 0; i < kMaxBatch && got < n; ++i)
      if (hu_[i].phase == PH_FREE) {
        hu_[i].phase = PH_RESERVED;
        slots[got++] = i;
      }
    free_ -= n;
    users_ += n;
    return PCS_OK;
  }
  void release(const int* slots, int n) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (int e = 0; e < n; ++e) hu_[slots[e]].phase = PH_FREE;
      free_ += n;
      users_ -= n;
    }
    cv_slots_.notify_all();
  }
  void submit(Call* call, const int* slots, int n) {
    {
      std::lock_guard<std::mutex> lk(call->mu);
      call->remaining += n;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (int e = 0; e < n; ++e) {
        hu_[slots[e]].call = call;
        pending_[n_pending_++] = slots[e];
      }
    }
    cv_work_.notify_all();
  }
  HostUnit& unit(int slot) { return hu_[slot]; }
  UnitDev* d_units() { return d_units_; }
  IcpProblem* d_icp() { return d_icp_; }

 private:
  explicit Engine(int device) : device_(device) {
    for (int i = 0; i < kMaxBatch; ++i) hu_[i].phase = PH_FREE;
  }
  bool start() {
    if (hipSetDevice(device_) != hipSuccess) return false;
    // the descriptor tables and the pinned mirrors live as long as the process (the one allocation of
    // this library that is not the caller's workspace: the tables are shared by all calls)
    if (hipMalloc((void**)&d_units_, kMaxBatch * sizeof(UnitDev)) != hipSuccess) return false;
    if (hipMalloc((void**)&d_icp_, kMaxBatch * sizeof(IcpProblem)) != hipSuccess) return false;
    char* pg = nullptr;
    const size_t pin_bytes = kMaxBatch * (2 * sizeof(IcpState) + sizeof(TrkState) + 2 * sizeof(IcpProblem));
    if (hipHostMalloc((void**)&pg, pin_bytes, hipHostMallocDefault) != hipSuccess) return false;
    pin_icp_ = (IcpState*)pg;
    pin_fin_ = pin_icp_ + kMaxBatch;
    pin_trk_ = (TrkState*)(pin_fin_ + kMaxBatch);
    pin_desc_ = (IcpProblem*)(pin_trk_ + kMaxBatch);  // [2][kMaxBatch] descriptor staging, by tick parity
    std::thread([this] { run(); }).detach();
    std::unique_lock<std::mutex> lk(mu_);
    cv_slots_.wait(lk, [&] { return ready_ || broken_; });
    return ready_;
  }

  void finish_unit(int i, int rc) {  // driver thread; the unit's kernels have all completed
    HostUnit& h = hu_[i];
    Call* call = h.call;
    {
      std::lock_guard<std::mutex> lk(mu_);
      h.phase = PH_FREE;
      ++free_;
      --users_;
    }
    cv_slots_.notify_all();
    bool last;
    {
      std::lock_guard<std::mutex> lk(call->mu);
      if (rc != PCS_OK && call->rc == PCS_OK) call->rc = rc;
      last = --call->remaining == 0;
      if (last) call->cv.notify_all();  // (under the lock: the Call may vanish as soon as it is released)
    }
  }

  int enqueue_tick();
  void collect_tick();
  void run();

  const int device_;
  std::mutex mu_;
  std::condition_variable cv_work_, cv_slots_;
  bool ready_ = false, broken_ = false;
  int free_ = kMaxBatch, users_ = 0;
  int pending_[kMaxBatch], n_pending_ = 0;
  PcsTrackParams prm_;
  HostUnit hu_[kMaxBatch];
  UnitDev* d_units_ = nullptr;
  IcpProblem* d_icp_ = nullptr;
  // driver thread only
  hipStream_t st_ = nullptr;
  hipEvent_t ev_ = nullptr;
  IcpState *pin_icp_ = nullptr, *pin_fin_ = nullptr;
  TrkState* pin_trk_ = nullptr;
  IcpProblem* pin_desc_ = nullptr;
  int live_[kMaxBatch], n_live_ = 0;
  Batch b_all_, b_iter_, b_fin_;
  int64_t ticks_ = 0;
};

}  // namespace pcs

using namespace pcs;

#define TRK_RC(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc) return _rc;        \
  } while (0)

static_assert(kPinnedStageOff + kMaxCallUnits * sizeof(UnitDev) <= kPinnedPageBytes,
              "descriptor staging does not fit the pinned page");
static_assert(kMaxCallUnits * sizeof(TrkState) <= kPinnedStageOff - kPinnedTrackerOff, "tracker state mirror");
static_assert(sizeof(Batch) + sizeof(Steps) + sizeof(LevelSizes) + 64 <= 4096,
              "the by-value unit lists must fit the kernel-argument segment");

static int64_t unit_frame_max(const int64_t* frame_offsets, int anchor, int min_f, int max_f) {
  int64_t nm = 1;
  for (int f = min_f; f <= max_f; ++f) {
    int64_t c = frame_offsets[f + 1] - frame_offsets[f];
    if (f != anchor && c > nm) nm = c;
  }
  return nm;
}

// Decide every live unit's next phase from the state words read last and enqueue the next tick.
// Returns PCS_OK or an engine-fatal error (a failed launch); errors of single units fail their call only.
int Engine::enqueue_tick() {
  const PcsTrackParams* prm = &prm_;
  const int L = prm->num_levels;
  hipStream_t gs = st_;
  UnitDev* const d_units = d_units_;
  IcpProblem* const d_icp = d_icp_;
  HostUnit* const hu = hu_;
  auto in_range = [](const HostUnit& h, int f) { return f >= h.min_f && f <= h.max_f; };
  auto step_arg = [&](const HostUnit& h) {
    StepArg a;
    a.dir = (signed char)h.dir;
    a.next = (short)h.next;
    a.col = (short)(h.next - h.min_f);
    a.last_col = (short)h.last_col;
    a.lv = (signed char)h.lv;
    a.off_n = in_range(h, h.next) ? (int)h.frame_offsets[h.next] : 0;
    a.nn = in_range(h, h.next) ? (int)(h.frame_offsets[h.next + 1] - h.frame_offsets[h.next]) : 0;
    return a;
  };
  // ---- 0. units whose last tick raised an error take their whole call with them
  {
    bool any = false;
    for (int e = 0; e < n_live_; ++e) {
      HostUnit& h = hu[live_[e]];
      int rc = h.fail;
      if (!rc && h.phase == PH_ICP && h.icp.err) rc = PCS_ERR_RANGE;
      if (!rc && (h.phase == PH_POSTWAIT || h.phase == PH_FINALWAIT) && h.host.err) rc = PCS_ERR_RANGE;
      if (rc) {
        std::lock_guard<std::mutex> lk(h.call->mu);
        if (h.call->rc == PCS_OK) h.call->rc = rc;
        any = true;
      }
    }
    if (any) {
      int keep = 0;
      for (int e = 0; e < n_live_; ++e) {
        const int i = live_[e];
        int rc;
        {
          std::lock_guard<std::mutex> lk(hu[i].call->mu);
          rc = hu[i].call->rc;
        }
        if (rc != PCS_OK) finish_unit(i, rc);
        else live_[keep++] = i;
      }
      n_live_ = keep;
    }
  }
  ++ticks_;
  Batch b_fin, b_post, b_dir, b_step, b_smp, b_begin, b_iter;
  Steps s_fin, s_post, s_step, s_smp;
  b_fin.n = b_post.n = b_dir.n = b_step.n = b_smp.n = b_begin.n = b_iter.n = 0;
  static thread_local IcpHint h_fin[kMaxBatch], h_begin[kMaxBatch], h_iter[kMaxBatch];
  BatchInts kb;
  auto start_level = [&](int i) {  // sampling + start of the registration at hu[i].lv
    HostUnit& h = hu[i];
    s_smp.a[b_smp.n] = step_arg(h);
    b_smp.idx[b_smp.n++] = i;
    b_begin.idx[b_begin.n++] = i;
    h.loop_before = 0;
    h.icp.iters = 0;
    h.icp.loop_iters = 0;
    h.icp.stop_iter = INT_MAX;
    h.icp.err = 0;
    h.phase = PH_ICP;
  };
  auto start_step = [&](int i) {
    HostUnit& h = hu[i];
    h.lv = 0;
    ++h.steps;
    s_step.a[b_step.n] = step_arg(h);
    b_step.idx[b_step.n++] = i;
    start_level(i);
  };
  auto start_dir = [&](int i, int dir) -> bool {  // true if the direction has a first step
    HostUnit& h = hu[i];
    h.dir = dir;
    h.next = h.anchor + dir;
    h.last_col = (dir == 1 && h.anchor > 0) ? h.anchor : -1;
    return in_range(h, h.next) && h.any_valid;
  };
  int done[kMaxBatch], n_done = 0;
  for (int e = 0; e < n_live_; ++e) {
    const int i = live_[e];
    HostUnit& h = hu[i];
    switch (h.phase) {
      case PH_START: {
        // a backward pass without a step leaves nothing behind that the forward set-up does not redo
        bool go = start_dir(i, -1);
        if (!go) go = start_dir(i, 1);
        b_dir.idx[b_dir.n++] = i;
        if (go) start_step(i);
        else h.phase = PH_FINALWAIT;
        break;
      }
      case PH_ICP: {
        if (h.icp.stop_iter != INT_MAX || h.icp.iters >= prm->max_iter) {
          icp_profile_phases(h.icp.phase_ticks);
          if (g_tick_profile.on)  // profiling hook: one line per registration (sizes, iterations, phase microseconds)
            fprintf(stderr, "[reg] slot %d lv %d C %d Mn %lld Ra %lld Rn %lld iters %d loop %d us %.0f %.0f %.0f %.0f\n", i,
                    h.lv, h.C, (long long)h.host.counts[1], (long long)h.host.counts[2], (long long)h.host.counts[3],
                    h.icp.iters, h.icp.loop_iters, h.icp.phase_ticks[0] * 0.01, h.icp.phase_ticks[1] * 0.01,
                    h.icp.phase_ticks[2] * 0.01, h.icp.phase_ticks[3] * 0.01);
          s_fin.a[b_fin.n] = step_arg(h);
          h_fin[b_fin.n] = h.hint;
          b_fin.idx[b_fin.n++] = i;
          ++h.reg_calls;
          ++h.lv;
          if (h.lv < L) {
            start_level(i);
          } else {
            h.lv = L - 1;
            s_post.a[b_post.n] = step_arg(h);
            b_post.idx[b_post.n++] = i;
            h.phase = PH_POSTWAIT;
          }
        }
        break;
      }
      case PH_POSTWAIT: {
        h.last_col = h.next;
        h.next += h.dir;
        if (in_range(h, h.next) && h.host.active) {
          start_step(i);
        } else if (h.dir == -1) {
          bool go = start_dir(i, 1);
          b_dir.idx[b_dir.n++] = i;
          if (go) start_step(i);
          else h.phase = PH_FINALWAIT;
        } else {
          h.phase = PH_FINALWAIT;
        }
        break;
      }
      case PH_FINALWAIT: {
        h.io->info[0] = h.host.ex_count;
        h.io->info[1] = h.reg_calls;
        h.io->info[2] = h.steps;
        h.io->info[3] = ticks_ - h.tick0;
        h.phase = PH_DONE;
        done[n_done++] = i;
        break;
      }
      default:
        break;
    }
  }
  if (n_done) {
    int keep = 0;
    for (int e = 0; e < n_live_; ++e)
      if (hu[live_[e]].phase != PH_DONE) live_[keep++] = live_[e];
    n_live_ = keep;
    for (int e = 0; e < n_done; ++e) finish_unit(done[e], PCS_OK);
  }
  b_fin_ = b_fin;
  b_iter_.n = 0;
  b_all_.n = n_live_;
  for (int e = 0; e < n_live_; ++e) b_all_.idx[e] = live_[e];
  if (n_live_ == 0) return PCS_OK;
  // ---- 1. registrations that stopped: l1 error, coverage, outputs; then frame <- T frame
  if (b_fin.n) {
    TRK_RC(icp_enqueue_finish(d_icp, b_fin, h_fin, pin_fin_, gs));
    int g = 1;
    for (int e = 0; e < b_fin.n; ++e) g = max(g, nblk(max(hu[b_fin.idx[e]].na, hu[b_fin.idx[e]].C)));
    trk_apply_T_kernel<<<dim3(g, b_fin.n), kBlock, 0, gs>>>(d_units, b_fin, s_fin);
  }
  // ---- 2. after the last level: component motion, velocity smoothing, stop rules, harvest
  if (b_post.n) {
    int gc = 1, gp = 1, gq = 1;
    static thread_local float* sv_v[kMaxBatch];
    static thread_local const float* sv_d[kMaxBatch];
    static thread_local int64_t sv_C[kMaxBatch];
    static thread_local int sv_F[kMaxBatch], sv_lo[kMaxBatch], sv_hi[kMaxBatch];
    static thread_local void* sv_ws[kMaxBatch];
    static thread_local size_t sv_wsb[kMaxBatch];
    int nsv = 0;
    for (int e = 0; e < b_post.n; ++e) {
      HostUnit& h = hu[b_post.idx[e]];
      gc = max(gc, nblk((int64_t)h.Cu * 64));
      gp = max(gp, nblk(max(h.na, h.C)));
      gq = max(gq, nblk((int64_t)s_post.a[e].nn * 32));
      int a0 = h.anchor + h.dir, b0 = h.next;
      if (a0 != b0) {
        sv_v[nsv] = h.p.d.velos;
        sv_d[nsv] = h.p.d.diffs;
        sv_C[nsv] = h.C;
        sv_F[nsv] = h.F;
        sv_lo[nsv] = a0 < b0 ? a0 : b0;
        sv_hi[nsv] = a0 < b0 ? b0 : a0;
        sv_ws[nsv] = h.p.ws_sv;
        sv_wsb[nsv] = h.p.sv_bytes;
        ++nsv;
      }
    }
    if (gc > 2048) gc = 2048;
    if (gq > 8192) gq = 8192;
    trk_motion_kernel<<<dim3(gc, b_post.n), kBlock, 0, gs>>>(d_units, b_post, s_post);
    TRK_RC(smooth_velo_run_batch(nsv, sv_v, sv_d, sv_C, sv_F, sv_lo, sv_hi, prm->smooth_weight0,
                                 prm->smooth_weight, prm->smooth_num_itr, prm->smooth_stopping, sv_ws,
                                 sv_wsb, gs));
    trk_after_smooth_kernel<<<dim3(gp, b_post.n), kBlock, 0, gs>>>(d_units, b_post, s_post,
                                                                   prm->min_move_frame, prm->reg_error_coeff,
                                                                   prm->angle_threshold);
    hv_insert_kernel<<<dim3(gp, b_post.n), kBlock, 0, gs>>>(d_units, b_post, s_post);
    hv_search_kernel<<<dim3(gq, b_post.n), kBlock, 0, gs>>>(d_units, b_post, s_post);
    hv_compact_kernel<<<b_post.n, kWg, 0, gs>>>(d_units, b_post, s_post);
  }
  // ---- 3. direction / step set-up
  if (b_dir.n) {
    int g = 1;
    for (int e = 0; e < b_dir.n; ++e) g = max(g, nblk(max(hu[b_dir.idx[e]].na, hu[b_dir.idx[e]].C)));
    trk_dir_init_kernel<<<dim3(g, b_dir.n), kBlock, 0, gs>>>(d_units, b_dir);
  }
  if (b_step.n) {
    int g = 1;
    for (int e = 0; e < b_step.n; ++e) g = max(g, nblk(max(hu[b_step.idx[e]].na, hu[b_step.idx[e]].C)));
    trk_step_begin_kernel<<<dim3(g, b_step.n), kBlock, 0, gs>>>(d_units, b_step, s_step);
  }
  // ---- 4. sampling of the anchor (as moved so far) and of the target frame
  if (b_smp.n) {
    LevelSizes ls;
    for (int l = 0; l < kMaxLevels; ++l)
      for (int k = 0; k < 3; ++k) ls.v[l][k] = l < L ? prm->voxel_size[l][k] : 1.f;
    int gv = 1;
    for (int e = 0; e < b_smp.n; ++e) {
      HostUnit& h = hu[b_smp.idx[e]];
      int est = h.seen[h.lv][1] > 0 ? (int)(h.seen[h.lv][1] * 1.25) : max(h.na, h.nm) / 2;
      gv = max(gv, nblk(est));
    }
    smp_sort_kernel<<<dim3(2, b_smp.n), kWg, 0, gs>>>(d_units, b_smp, s_smp, ls);
    smp_emit_kernel<<<dim3(gv, b_smp.n, 2), kBlock, 0, gs>>>(d_units, b_smp, s_smp);
    smp_compact_kernel<<<dim3(2, b_smp.n), kWg, 0, gs>>>(d_units, b_smp);
  }
  // ---- 5. registrations that start: descriptor, reset, grids
  if (b_begin.n) {
    for (int e = 0; e < b_begin.n; ++e) {
      const int i = b_begin.idx[e];
      HostUnit& h = hu[i];
      UnitDev& d = h.p.d;
      IcpArgs ia;
      ia.mov = (float*)d.sa.out;
      ia.comp = d.sa.out_comp;
      ia.ref = (const float*)d.sb.out;
      ia.all = (const float*)d.sb.sub;
      ia.comp_deg = (const int64_t*)d.sub_deg;
      ia.cnt3 = &d.st->counts[1];
      ia.Mn = h.na;
      ia.Rn = h.nm;
      ia.Ra = h.nm;
      ia.C = h.C;
      ia.radius = prm->radius[h.lv];
      ia.frame_offset = h.next - h.anchor;
      ia.reg = prm->angle_regularizer;
      ia.delta = prm->stopping_delta[h.lv];
      ia.max_iter = prm->max_iter;
      ia.cell_size = 0.f;
      ia.T = d.Tl;
      ia.l1 = (h.lv == L - 1) ? d.l1 : d.l1_tmp;
      ia.ratio = (h.lv == 0) ? d.ratio : d.ratio_tmp;
      // (cluster_tracking.py keeps the error of the last level and the edge ratio of the first: the others are not computed)
      ia.want_l1 = h.lv == L - 1;
      ia.want_ratio = h.lv == 0;
      // the descriptors of a tick go up together: staged in pinned memory (one half per tick parity: the
      // previous tick's launch has completed before this half is written again), read by one launch
      TRK_RC(icp_describe(ia, h.p.ws_icp, h.p.icp_bytes, pin_desc_ + (size_t)(ticks_ & 1) * kMaxBatch + e));
      // launch sizes: what this level had at the previous step, else a guess
      const int* sn = h.seen[h.lv];
      h.hint.Mn = sn[0] >= 0 ? (int)(sn[0] * 1.25) + 256 : h.na / 4 + 256;
      h.hint.Ra = sn[1] >= 0 ? (int)(sn[1] * 1.25) + 256 : h.nm / 2 + 256;
      h.hint.Rn = sn[2] >= 0 ? (int)(sn[2] * 1.25) + 256 : h.nm / 4 + 256;
      h.hint.C = h.C;
      h.hint.clear_words = 4 * (size_t)pow2_words(h.hint.Mn) + 2 * (size_t)pow2_words(h.hint.Rn) +
                           2 * (size_t)pow2_words(h.hint.Ra);
      h_begin[e] = h.hint;
    }
    trk_store_icp_batch_kernel<<<b_begin.n, 64, 0, gs>>>(d_icp, pin_desc_ + (size_t)(ticks_ & 1) * kMaxBatch, b_begin);
    TRK_RC(icp_enqueue_begin(d_icp, b_begin, h_begin, gs));
  }
  // ---- 6. one chunk of iterations for every unit that is registering
  for (int e = 0; e < n_live_; ++e) {
    const int i = live_[e];
    HostUnit& h = hu[i];
    if (h.phase != PH_ICP) continue;
    kb.v[b_iter.n] = h.icp.iters;  // 0: the registration begins (its first iteration is a wide launch)
    h.loop_before = h.icp.iters;
    h_iter[b_iter.n] = h.hint;
    b_iter.idx[b_iter.n++] = i;
  }
  // (default schedule: 8 iterations as wide launches over all registering units; a resident schedule: every
  //  unit iterates on a workgroup of its own until its stop rule fires or the chunk's time budget has passed)
  if (b_iter.n)
    TRK_RC(icp_enqueue_chunk(d_icp, b_iter, kb, prm->max_iter, cfg_chunk_us().load(std::memory_order_relaxed), h_iter,
                             pin_icp_, gs, 0));
  b_iter_ = b_iter;
  if (g_tick_profile.on) { g_tick_profile.units += n_live_; g_tick_profile.iter_units += b_iter.n; }
  // ---- 7. state mirror; the event tells the host when the tick's words can be read
  trk_publish_kernel<<<(b_all_.n + 63) / 64, 64, 0, gs>>>(d_units, b_all_, pin_trk_);
  PCS_LAUNCH_CHECK();
  PCS_CHECK_HIP(hipEventRecord(ev_, gs));
  return PCS_OK;
}

// the tick has completed: take its state words
void Engine::collect_tick() {
  HostUnit* const hu = hu_;
  for (int e = 0; e < b_fin_.n; ++e)
    if (pin_fin_[e].err) hu[b_fin_.idx[e]].fail = PCS_ERR_RANGE;  // the coverage query can still overflow a grid
  for (int e = 0; e < b_all_.n; ++e) {
    HostUnit& h = hu[b_all_.idx[e]];
    h.host = pin_trk_[e];
    if (h.phase == PH_ICP) {
      h.seen[h.lv][0] = (int)h.host.counts[1];
      h.seen[h.lv][1] = (int)h.host.counts[2];
      h.seen[h.lv][2] = (int)h.host.counts[3];
    }
  }
  if (b_iter_.n) {
    double unit_iters = 0.0, bytes = 0.0, bytes_8d = 0.0, many_iters = 0.0, many_8d = 0.0;
    for (int e = 0; e < b_iter_.n; ++e) {
      HostUnit& h = hu[b_iter_.idx[e]];
      h.icp = pin_icp_[e];
      const double Mn = (double)h.host.counts[1], Rn = (double)h.host.counts[3], C = (double)h.C;
      const double it = (double)(h.icp.iters - h.loop_before);
      unit_iters += it;
      // per unit and iteration: neighbours (36 per query) + moments (40 per query, 272 per component) + solve and
      // apply (68 per moving point, 416 per component)
      bytes += it * (76.0 * (Mn + Rn) + 272.0 * C + 68.0 * Mn + 416.0 * C);
      // two graphs 2 (16 N + 16 M) + 2 x 16 E, the P / Q gathers 2 x 24 E, covariance and transform
      // 2 x 72 C -- with E = N + M (one edge per query at most) and C the components that have a moving point
      const double Cl = (double)h.icp.live_c;
      const double b8 = it * (112.0 * (Mn + Rn) + 144.0 * Cl);
      bytes_8d += b8;
      if (C > 5000.0) { many_iters += it; many_8d += b8; }  // (by the key's component IDs: the 0.25 m key numbers ~13 k)
    }
    icp_profile_account(unit_iters, bytes, 0, bytes_8d, many_iters, many_8d);
  }
}

void Engine::run() {
  // Default-priority stream: a high-priority one (its launches are one long dependent chain) measured 1.72
  // against 1.75 s per batch of 12 in round 3, BUT priority streams get hardware queues of their own, and with
  // 4 normal + 4 high-priority queues the process-wide freezes of DESIGN.md section 4 "The freeze" came back
  // (10-60 s steps). Default priority keeps the process on HIP's four queues.
  bool ok = hipSetDevice(device_) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&st_, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&ev_, hipEventDisableTiming) == hipSuccess;
  // (a first command now: whatever the runtime binds to a stream at its first use -- its hardware queue -- is
  //  bound in the order the engines are created, not in the order their first units arrive)
  ok = ok && hipMemsetAsync(d_units_, 0, sizeof(int), st_) == hipSuccess && hipStreamSynchronize(st_) == hipSuccess;
  {
    std::lock_guard<std::mutex> lk(mu_);
    ready_ = ok;
    broken_ = !ok;
  }
  cv_slots_.notify_all();
  if (!ok) return;
  const bool tp = g_tick_profile.on;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_work_.wait(lk, [&] { return n_pending_ > 0 || n_live_ > 0; });
      for (int e = 0; e < n_pending_; ++e) {  // units join at tick boundaries
        HostUnit& h = hu_[pending_[e]];
        h.phase = PH_START;
        h.tick0 = ticks_;
        live_[n_live_++] = pending_[e];
      }
      n_pending_ = 0;
    }
    long long t0 = tp ? now_ns() : 0;
    // The first ticks of a process take their launches one engine at a time: every kernel's FIRST launch loads
    // it (lazily), and rocprofv3's tool library has crashed three times in exactly that path when several
    // threads took first launches at once. ~0.2 ms of host time per tick for the first half second.
    static std::mutex cold_mu;
    static std::atomic<int> cold_ticks{0};
    const bool cold = cold_ticks.load(std::memory_order_relaxed) < 1024;
    if (cold) cold_mu.lock();
    int rc = enqueue_tick();
    if (cold) {
      cold_ticks.fetch_add(1, std::memory_order_relaxed);
      cold_mu.unlock();
    }
    if (tp) { g_tick_profile.enq_ns += now_ns() - t0; g_tick_profile.ticks += 1; t0 = now_ns(); }
    if (rc == PCS_OK && n_live_ > 0) {
      const long long w0 = now_ns();
      if (hipEventSynchronize(ev_) != hipSuccess) rc = PCS_ERR_HIP;
      const long long w1 = now_ns();
      if (w1 - w0 > 2000000000LL)  // diagnostic: a tick that took seconds
        fprintf(stderr, "[pcs] tracker tick %lld completed after %.1f s (%d units, %d iterating, %d finishing)\n",
                (long long)ticks_, (w1 - w0) / 1e9, n_live_, b_iter_.n, b_fin_.n);
      if (tp) g_tick_profile.wait_ns += w1 - t0;
      if (rc == PCS_OK) collect_tick();
    }
    if (rc != PCS_OK) {
      // a launch or the event wait failed: nothing of this engine can be trusted any more
      (void)hipStreamSynchronize(st_);
      for (int e = 0; e < n_live_; ++e) finish_unit(live_[e], rc);
      n_live_ = 0;
      fprintf(stderr, "[pcs] tracker engine of device %d: HIP failure, calls in flight fail with %d\n", device_, rc);
    }
  }
}

// Workspace layout, descriptors and anchor set-up of the units of ONE sequence on the caller's stream;
// the units are then in the engine's table (slots[]) and ready to be submitted.
static int prepare_sequence(Engine* eng, const PcsTrackParams* prm, const float* seq_fxyz,
                            const int64_t* frame_offsets, int num_frames, PcsTrackUnit* units, int n_units,
                            Arena& a, void* ws, int* slots, hipStream_t s) {
  char* page = (char*)thread_pinned_page();
  if (!page) return PCS_ERR_HIP;
  TrkState* pin_trk = (TrkState*)(page + kPinnedTrackerOff);
  UnitDev* stage_units = (UnitDev*)(page + kPinnedStageOff);
  SeqPlan sq;
  if (!plan_seq(a, sq)) return PCS_ERR_WORKSPACE;
  const int64_t n_seq = frame_offsets[num_frames];
  if (n_seq >= (1LL << 31) || num_frames >= 32767) return PCS_ERR_RANGE;
  for (int i = 0; i < n_units; ++i) {
    PcsTrackUnit& io = units[i];
    HostUnit& h = eng->unit(slots[i]);
    if (!io.seq_component || !io.seq_stationary || !io.ex_index || !io.ex_local || !io.ex_component ||
        !io.transforms || !io.reg_errors || !io.comp_edge_ratios || !io.comp_velos ||
        !io.comp_min_frame_id || !io.comp_max_frame_id || !io.moving || !io.valid_comp)
      return PCS_ERR_ARG;
    if (io.C < 1 || io.anchor_frame < 0 || io.anchor_frame >= num_frames || io.min_frame_id < 0 ||
        io.max_frame_id >= num_frames || io.min_frame_id > io.anchor_frame ||
        io.max_frame_id < io.anchor_frame)
      return PCS_ERR_ARG;
    h.io = &io;
    h.frame_offsets = frame_offsets;
    h.anchor = io.anchor_frame;
    h.min_f = io.min_frame_id;
    h.max_f = io.max_frame_id;
    h.F = h.max_f + 1;
    h.W = h.max_f - h.min_f + 1;
    h.C = (int)io.C;
    h.off_a = frame_offsets[h.anchor];
    int64_t na = frame_offsets[h.anchor + 1] - h.off_a;
    if (na < 1) return PCS_ERR_ARG;
    for (int f = h.min_f; f <= h.max_f; ++f)
      if (frame_offsets[f + 1] < frame_offsets[f]) return PCS_ERR_ARG;
    int64_t nm = unit_frame_max(frame_offsets, h.anchor, h.min_f, h.max_f);
    if (na >= (1LL << 30) || nm >= (1LL << 30) || io.C >= (1LL << 30)) return PCS_ERR_RANGE;
    h.na = (int)na;
    h.nm = (int)nm;
    if (!plan_unit(a, na, nm, io.C, h.F, h.p)) return PCS_ERR_WORKSPACE;
    UnitDev& d = h.p.d;
    d.seq = (const float4*)seq_fxyz;
    d.seq_comp = io.seq_component;
    d.seq_stat = io.seq_stationary;
    d.off_a = h.off_a;
    d.comp_base = io.comp_base;
    d.ex_cap = io.ex_cap;
    d.na = h.na;
    d.C = h.C;
    d.F = h.F;
    d.W = h.W;
    d.anchor = h.anchor;
    d.min_f = h.min_f;
    d.hdesc = sq.hdesc;
    d.transforms = io.transforms;
    d.reg_errors = io.reg_errors;
    d.edge_ratios = io.comp_edge_ratios;
    d.velos = io.comp_velos;
    d.cmin = io.comp_min_frame_id;
    d.cmax = io.comp_max_frame_id;
    d.moving = io.moving;
    d.vcm = io.valid_comp;
    d.ex_index = io.ex_index;
    d.ex_local = io.ex_local;
    d.ex_comp = io.ex_component;
    h.steps = h.reg_calls = 0;
    h.loop_before = 0;
    h.fail = 0;
    h.icp.err = 0;
    h.host.err = 0;
    for (int l = 0; l < kMaxLevels; ++l) h.seen[l][0] = h.seen[l][1] = h.seen[l][2] = -1;
    io.info[0] = io.info[1] = io.info[2] = io.info[3] = 0;
    stage_units[i] = d;
    PCS_CHECK_HIP(hipMemcpyAsync(eng->d_units() + slots[i], stage_units + i, sizeof(UnitDev),
                                 hipMemcpyHostToDevice, s));
  }
  UnitDev* const d_units = eng->d_units();

  // ---- set-up, pass 1: the anchors' components (sorted unique), everything zeroed
  const float nn_radius = prm->nn_radius;
  {
    const double nn_r = (double)nn_radius;
    const float nn_cell = (float)(sqrt(nn_r * nn_r + 3e-5 * nn_r * nn_r + 3e-6) / 4 * 1.0005);
    TRK_RC(pcs_minmax4_f32(seq_fxyz, n_seq, sq.seq_b8, s));
    trk_hdesc_kernel<<<1, 64, 0, s>>>(sq.seq_b8, nn_radius, nn_cell, 64.f + 4.f * nn_radius, sq.hdesc);
  }
  for (int i = 0; i < n_units; ++i) {
    HostUnit& h = eng->unit(slots[i]);
    UnitDev& d = h.p.d;
    PcsTrackUnit& io = *h.io;
    PCS_CHECK_HIP(hipMemsetAsync(d.st, 0, sizeof(TrkState), s));
    PCS_CHECK_HIP(hipMemsetAsync((char*)ws + h.p.zero_begin, 0, h.p.zero_end - h.p.zero_begin, s));
    PCS_CHECK_HIP(hipMemsetAsync(d.xyzf + h.na, 0, sizeof(float4), s));
    PCS_CHECK_HIP(hipMemsetAsync(io.comp_velos, 0, (size_t)h.C * h.F * 3 * sizeof(float), s));
    PCS_CHECK_HIP(hipMemsetAsync(io.reg_errors, 0, (size_t)h.C * h.F * sizeof(float), s));
    PCS_CHECK_HIP(hipMemsetAsync(io.comp_edge_ratios, 0, (size_t)h.C * h.F * sizeof(float), s));
    trk_load_anchor_kernel<<<nblk(h.na), kBlock, 0, s>>>(d.seq + h.off_a, io.seq_component + h.off_a,
                                                         io.comp_base, h.na, d.xyzf, d.comp64, d.comp32);
    TRK_RC(pcs_unique_i64(d.comp64, h.na, 64, d.uniq_c, nullptr, d.order_c, d.seg_c, d.d_numc,
                          h.p.ws_unique, h.p.unique_bytes, s));
    g2_clear_kernel<<<grid_for((int64_t)d.hg.slot_words), kBlock, 0, s>>>(d.hg);
    // the number of present components comes back through the state mirror (counts[0])
    trk_copy_i64_kernel<<<1, 1, 0, s>>>(d.d_numc, &d.st->counts[0]);
  }
  Batch all;
  all.n = n_units;
  for (int i = 0; i < n_units; ++i) all.idx[i] = slots[i];
  trk_publish_kernel<<<1, 64, 0, s>>>(d_units, all, pin_trk);
  PCS_LAUNCH_CHECK();
  PCS_CHECK_HIP(hipStreamSynchronize(s));
  // ---- set-up, pass 2: component size, centre, diameter and validity; the anchors' own points
  for (int i = 0; i < n_units; ++i) {
    HostUnit& h = eng->unit(slots[i]);
    UnitDev& d = h.p.d;
    h.Cu = (int)pin_trk[i].counts[0];
    if (h.Cu < 1 || h.Cu > h.C) return PCS_ERR_ARG;
    const int gU = nblk(h.Cu);
    const float* xyz1 = (const float*)d.xyzf + 1;  // rows (x, y, z, next row's frame) with stride 4
    TRK_RC(pcs_segment_reduce_f32(xyz1, 4, d.order_c, d.seg_c, h.Cu, 2, d.red4, s));
    trk_comp_center_kernel<<<gU, kBlock, 0, s>>>(d.uniq_c, d.seg_c, h.Cu, d.red4, d.deg_f, d.deg_i, d.ctr);
    trk_point_dist_kernel<<<nblk(h.na), kBlock, 0, s>>>(d.xyzf, d.comp32, h.na, d.ctr, d.dist);
    TRK_RC(pcs_segment_minmax_f32(d.dist, 1, d.order_c, d.seg_c, h.Cu, 1, d.red1, s));
    trk_comp_diam_kernel<<<gU, kBlock, 0, s>>>(d.uniq_c, h.Cu, d.red1, d.diam);
    trk_comp_init_kernel<<<nblk(h.C), kBlock, 0, s>>>(h.C, h.F, h.W, h.anchor, prm->max_diameter, d.deg_f,
                                                      d.diam, d.ctr, d.vcm, d.transforms, d.cmin, d.cmax,
                                                      d.centers, d.st);
  }
  trk_anchor_emit_kernel<<<n_units, kWg, 0, s>>>(d_units, all);
  trk_publish_kernel<<<1, 64, 0, s>>>(d_units, all, pin_trk);
  PCS_LAUNCH_CHECK();
  PCS_CHECK_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < n_units; ++i) {
    HostUnit& h = eng->unit(slots[i]);
    h.host = pin_trk[i];
    if (h.host.err) return PCS_ERR_RANGE;
    h.any_valid = h.host.active != 0;
  }
  return PCS_OK;
}

struct SeqIn {
  const float* seq_fxyz;
  const int64_t* frame_offsets;
  int num_frames;
  PcsTrackUnit* units;
  int n_units;
};

// every sequence is prepared on the caller's stream and handed to the device's engine as soon as it
// is ready (its units run while the next sequence is prepared); returns when all units are done
static int run_sequences(const PcsTrackParams* prm, const SeqIn* seqs, int n_seqs, void* ws, size_t ws_bytes,
                         hipStream_t s) {
  if (!prm || !seqs || !ws || n_seqs < 1) return PCS_ERR_ARG;
  if (prm->num_levels < 1 || prm->num_levels > kMaxLevels || prm->max_iter < 0) return PCS_ERR_ARG;
  for (int q = 0; q < n_seqs; ++q) {
    if (!seqs[q].seq_fxyz || !seqs[q].frame_offsets || !seqs[q].units) return PCS_ERR_ARG;
    if (seqs[q].n_units < 1 || seqs[q].n_units > kMaxCallUnits || seqs[q].num_frames < 1) return PCS_ERR_ARG;
  }
  int device = 0;
  PCS_CHECK_HIP(hipGetDevice(&device));
  Call call;
  Arena a(ws, ws_bytes);
  int rc = PCS_OK;
  const int n_eng = Engine::count();
  for (int q = 0; q < n_seqs && rc == PCS_OK; ++q) {
    // a sequence's units are dealt to the device's engines in contiguous shares (each share to the engine
    // that is least loaded at that moment): one sequence alone then runs as several chains side by side
    // instead of one, and a batch loads the engines evenly
    const int shares = seqs[q].n_units < n_eng ? seqs[q].n_units : n_eng;
    int first = 0;
    for (int g = 0; g < shares && rc == PCS_OK; ++g) {
      const int n = (seqs[q].n_units - first + (shares - g) - 1) / (shares - g);
      Engine* eng = Engine::get(device);
      if (!eng) { rc = PCS_ERR_HIP; break; }
      int slots[kMaxCallUnits];
      rc = eng->reserve(*prm, n, slots);
      if (rc != PCS_OK) break;
      rc = prepare_sequence(eng, prm, seqs[q].seq_fxyz, seqs[q].frame_offsets, seqs[q].num_frames,
                            seqs[q].units + first, n, a, ws, slots, s);
      if (rc != PCS_OK) {
        (void)hipStreamSynchronize(s);  // nothing of the set-up may still run on the slots / the workspace
        eng->release(slots, n);
        break;
      }
      eng->submit(&call, slots, n);
      first += n;
    }
  }
  {
    std::unique_lock<std::mutex> lk(call.mu);
    call.cv.wait(lk, [&] { return call.remaining == 0; });
    if (rc == PCS_OK) rc = call.rc;
  }
  return rc;
}

static size_t sequence_workspace_bytes(Arena& a, const int64_t* frame_offsets, int num_frames,
                                       const PcsTrackUnit* units, int n_units) {
  for (int g = 0; g < Engine::kMaxEngines; ++g) {  // (one per share of the sequence's units, see run_sequences)
    SeqPlan sq;
    plan_seq(a, sq);
  }
  for (int i = 0; i < n_units; ++i) {
    const PcsTrackUnit& io = units[i];
    if (io.anchor_frame < 0 || io.anchor_frame >= num_frames || io.min_frame_id < 0 ||
        io.max_frame_id >= num_frames || io.min_frame_id > io.max_frame_id)
      return 0;
    int64_t na = frame_offsets[io.anchor_frame + 1] - frame_offsets[io.anchor_frame];
    int64_t nm = unit_frame_max(frame_offsets, io.anchor_frame, io.min_frame_id, io.max_frame_id);
    UnitPlan p;
    plan_unit(a, na, nm, io.C < 1 ? 1 : io.C, io.max_frame_id + 1, p);
  }
  return a.off;
}

extern "C" {

// Create the device's lock-step engines (driver threads, streams) NOW rather than at the first tracking
// call. The HIP runtime maps streams to a few hardware queues in the order they are created: engines that
// come into being between the streams of a dozen host threads may end up sharing one queue, and two chains
// on one queue run one after the other (measured: 1.75 s per batch instead of 1.5 s). A caller that is about
// to create streams of its own calls this first. `device` < 0: the current device.
// engines per device (1..8; takes effect for devices whose engines do not exist yet) and the time budget of a
// chunk of the unit-resident ICP schedules in microseconds (0: none; see pcs_icp_set_schedule); values < 0 keep
// the current setting
int pcs_track_configure(int engines, int chunk_us) {
  if (engines > Engine::kMaxEngines) return PCS_ERR_ARG;
  if (engines >= 1) Engine::cfg_engines().store(engines, std::memory_order_relaxed);
  if (chunk_us >= 0) Engine::cfg_chunk_us().store((unsigned)chunk_us, std::memory_order_relaxed);
  return PCS_OK;
}

int pcs_track_engines_init(int device) {
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return PCS_ERR_HIP;
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) return PCS_ERR_HIP;
  if (cur != device && hipSetDevice(device) != hipSuccess) return PCS_ERR_HIP;
  Engine* e = Engine::get(device);
  if (cur != device) (void)hipSetDevice(cur);
  return e ? PCS_OK : PCS_ERR_HIP;
}

size_t pcs_track_sequence_workspace_bytes(const int64_t* frame_offsets, int num_frames,
                                          const PcsTrackUnit* units, int n_units) {
  if (!frame_offsets || !units || n_units < 1 || num_frames < 1) return 0;
  Arena a(nullptr, 0);
  return sequence_workspace_bytes(a, frame_offsets, num_frames, units, n_units);
}

int pcs_track_sequence(const PcsTrackParams* prm, const float* seq_fxyz, const int64_t* frame_offsets,
                       int num_frames, PcsTrackUnit* units, int n_units, void* ws, size_t ws_bytes,
                       hipStream_t s) {
  SeqIn in = {seq_fxyz, frame_offsets, num_frames, units, n_units};
  return run_sequences(prm, &in, 1, ws, ws_bytes, s);
}

size_t pcs_track_batch_workspace_bytes(int n_seqs, const int64_t* const* frame_offsets, const int* num_frames,
                                       const PcsTrackUnit* const* units, const int* n_units) {
  if (n_seqs < 1 || !frame_offsets || !num_frames || !units || !n_units) return 0;
  Arena a(nullptr, 0);
  for (int q = 0; q < n_seqs; ++q) {
    if (!frame_offsets[q] || !units[q] || n_units[q] < 1 || num_frames[q] < 1) return 0;
    if (!sequence_workspace_bytes(a, frame_offsets[q], num_frames[q], units[q], n_units[q])) return 0;
  }
  return a.off;
}

int pcs_track_batch(const PcsTrackParams* prm, int n_seqs, const float* const* seq_fxyz,
                    const int64_t* const* frame_offsets, const int* num_frames, PcsTrackUnit* const* units,
                    const int* n_units, void* ws, size_t ws_bytes, hipStream_t s) {
  if (n_seqs < 1 || n_seqs > 1024 || !seq_fxyz || !frame_offsets || !num_frames || !units || !n_units)
    return PCS_ERR_ARG;
  std::vector<SeqIn> in((size_t)n_seqs);
  for (int q = 0; q < n_seqs; ++q) in[q] = {seq_fxyz[q], frame_offsets[q], num_frames[q], units[q], n_units[q]};
  return run_sequences(prm, in.data(), n_seqs, ws, ws_bytes, s);
}

size_t pcs_track_frame_workspace_bytes(int64_t n_anchor, int64_t n_frame_max, int64_t C,
                                       int num_frames) {
  Arena a(nullptr, 0);
  SeqPlan sq;
  plan_seq(a, sq);
  UnitPlan p;
  plan_unit(a, n_anchor, n_frame_max, C < 1 ? 1 : C, num_frames < 1 ? 1 : num_frames, p);
  return a.off;
}

int pcs_track_frame(const PcsTrackParams* prm, const float* seq_fxyz, const int64_t* seq_component,
                    const uint8_t* seq_stationary, const int64_t* frame_offsets, int num_frames,
                    int anchor_frame, int min_frame_id, int max_frame_id, int64_t comp_base,
                    int64_t C, int64_t* ex_index, int64_t* ex_local, int64_t* ex_component,
                    int64_t ex_cap, double* transforms, float* reg_errors, float* comp_edge_ratios,
                    float* comp_velos, int64_t* comp_min_frame_id, int64_t* comp_max_frame_id,
                    uint8_t* moving, uint8_t* valid_comp, int64_t* info, void* ws, size_t ws_bytes,
                    hipStream_t s) {
  if (!info) return PCS_ERR_ARG;
  PcsTrackUnit u;
  u.seq_component = seq_component;
  u.seq_stationary = seq_stationary;
  u.anchor_frame = anchor_frame;
  u.min_frame_id = min_frame_id;
  u.max_frame_id = max_frame_id;
  u.comp_base = comp_base;
  u.C = C;
  u.ex_index = ex_index;
  u.ex_local = ex_local;
  u.ex_component = ex_component;
  u.ex_cap = ex_cap;
  u.transforms = transforms;
  u.reg_errors = reg_errors;
  u.comp_edge_ratios = comp_edge_ratios;
  u.comp_velos = comp_velos;
  u.comp_min_frame_id = comp_min_frame_id;
  u.comp_max_frame_id = comp_max_frame_id;
  u.moving = moving;
  u.valid_comp = valid_comp;
  SeqIn in = {seq_fxyz, frame_offsets, num_frames, &u, 1};
  int rc = run_sequences(prm, &in, 1, ws, ws_bytes, s);
  for (int k = 0; k < 4; ++k) info[k] = u.info[k];
  return rc;
}

}  // extern "C"
