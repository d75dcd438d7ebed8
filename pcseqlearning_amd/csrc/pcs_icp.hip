// Fused per-component ICP ("TLS solve") for gfx950.
//
// Reference: register_to_next_frame, pcdet/models/registration/preprocessors/
// registration_utils.py:83-206. Per iteration the reference runs two radius-graph
// builds (4 kernels + scan + 2 host syncs + 4 allocations each), two argsorts, a handful of
// scatter/segment kernels, a batched cuSOLVER 3x3 SVD and a blocking loss.item().
//
// Here one call owns the whole loop; state stays in the caller's workspace. Iteration k is THREE
// launches (the concurrent tracking units share a dispatch rate of a few hundred thousand kernels
// per second, so launches are the currency):
//   nn         1-NN both directions, 32 lanes per query (two-level grid, pcs_grid2.hpp); workgroup
//              0 first settles the stop rule of iteration k-1
//   moments    ONE pass over the matched pairs: n, sum p, sum q, sum p(x)q, sum |p-q|^2 per
//              component (raw moments in f64; the centred covariance and the loss follow
//              algebraically, so the reference's second pass over the edges is not needed)
//   apply      every moving point solves its own component's regularised Kabsch step from the
//              completed moments (cov + reg*T_rot -> rotation by Newton's polar iteration, Jacobi SVD
//              when det <= 0; R = V diag(1,1,det(VU^T)) U^T, t = c_ref - R c_mov, T <- T_i T) -- a
//              3x3 solve per point is cheaper than another launch --, moves (f64 -> fp32 as the
//              reference stores them), the component's first point stores T and the loss term; the
//              moved points go straight into the grid of iteration k+1 (double-buffered) and the grid
//              just searched is reset
// The three kernels take an ARRAY of problem descriptors (blockIdx.y = problem): a stand-alone call is
// an array of one, concurrent calls can share launches (IcpBatcher below). Every kernel of iteration
// k returns immediately when k > state.stop_iter, so the host enqueues iterations in chunks and
// reads one word per chunk (through pinned memory) instead of one loss per iteration.
// Per-component sums: a small LDS table keyed by component per workgroup, then one f64 hardware
// atomic per (workgroup, component, accumulator).
#include "pcs_grid2.hpp"

#include <hip/hip_ext.h>

#include <chrono>
#include <climits>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace pcs {

__global__ void noop_kernel() {}


struct IcpState {
  double last_error;
  double loss;
  int countdown;
  int stop_iter;  // iterations > stop_iter do not run (INT_MAX until the stop rule fires)
  int iters;      // iterations solved so far
  int err;
  int decided;    // last iteration whose stop rule has been evaluated
  int pad;
};

constexpr int kProfIters = 96;  // iterations per call whose icp_nn_kernel is timed when profiling
constexpr int kAcc = 20;  // n, sum p (3), sum q (3), sum p(x)q (9), sum |p-q|^2, sum dist, sum clamp, pad

struct IcpProblem;

struct IcpBufs {
  GridDesc* grid;
  float *b_ref, *b_mov, *b_all;
  IcpState* state;
  Grid2 gr, ga;      // non-stationary reference, all reference (static)
  Grid2 gm[2];       // moving points: iteration k searches gm[k & 1], its apply fills gm[(k+1) & 1]
  int32_t *nn_f, *nn_b;  // [2, Mn] / [2, Rn]: buffer k & 1 holds iteration k's neighbours
  float* prev;   // [Mn,4] positions before the last applied update
  double* acc;   // [2, C, kAcc] raw moments, buffer k & 1 for iteration k
  double* Tb;    // [2, C, 16] accumulated transforms, buffer k & 1 after iteration k
  double* lossp; // [ceil(Mn / kBlock)] per-workgroup sums of the component losses, last applied iteration
  int32_t* first; // [C] smallest moving-point index of the component (INT_MAX: none)

  int32_t* cnt;  // [C] matched moving points in the final coverage query
};

// One ICP problem as the iteration kernels see it. The kernels take an ARRAY of these and
// blockIdx.y selects the problem, so the iterations of several calls (the concurrent tracking units)
// can share launches; a stand-alone call is an array of one.
struct IcpProblem {
  float4* mov;
  float4* prev;
  const float4* ref;
  const int32_t* comp;
  const int32_t* first;
  int32_t *nn_f, *nn_b;
  double *acc, *Tb, *lossp;
  IcpState* st;
  const GridDesc* grid;
  Grid2 gr, gm[2];
  int64_t C;
  double reg, delta;
  int Mn, Rn, df, nparts;
  int k0;    // iteration number of chunk position 0
  int n_it;  // chunk positions this problem takes part in
  int probe_stage;  // 0 normal; timing probe (tools/icp_stages.py): the search returns after stage 1..3
};

static bool plan_icp(Arena& a, int64_t Mn, int64_t Rn, int64_t Ra, int64_t C, IcpBufs& b) {
  if (Mn < 1) Mn = 1;
  if (Rn < 1) Rn = 1;
  if (Ra < 1) Ra = 1;
  if (C < 1) C = 1;
  b.grid = a.take<GridDesc>(1);
  b.b_ref = a.take<float>(24);  // ref, mov, all: 8 floats each, contiguous
  b.b_mov = b.b_ref ? b.b_ref + 8 : nullptr;
  b.b_all = b.b_ref ? b.b_ref + 16 : nullptr;
  b.state = a.take<IcpState>(1);
  plan_grid2(a, Rn, b.gr);
  plan_grid2(a, Mn, b.gm[0]);
  plan_grid2(a, Mn, b.gm[1]);
  plan_grid2(a, Ra, b.ga);
  b.nn_f = a.take<int32_t>(2 * Mn);
  b.nn_b = a.take<int32_t>(2 * Rn);
  b.prev = a.take<float>(Mn * 4);
  b.acc = a.take<double>(2 * C * kAcc);
  b.Tb = a.take<double>(2 * C * 16);
  b.lossp = a.take<double>((Mn + kBlock - 1) / kBlock + 1);
  b.first = a.take<int32_t>(C);

  b.cnt = a.take<int32_t>(C);
  return a.ok();
}

// grid origin padded so that points moved by the registration stay inside the 17-bit cell range
__global__ void icp_setup_kernel(const float* __restrict__ b_ref, const float* __restrict__ b_mov,
                                 const float* __restrict__ b_all, float radius, float cell,
                                 float pad, GridDesc* __restrict__ g, IcpState* __restrict__ st) {
  if (threadIdx.x != 0) return;
  float mn[4];
  for (int d = 0; d < 4; ++d) mn[d] = fminf(fminf(b_ref[d], b_mov[d]), b_all[d]);
  float vs0 = (float)(1 - 1e-3);
  g->vs0 = vs0;
  g->lo0 = mn[0] - vs0 * 2.f;
  for (int d = 0; d < 3; ++d) g->lo[d] = mn[1 + d] - pad;
  g->h = cell;
  g->radius = radius;
  g->radius2 = radius * radius;
  g->rings = (int)ceilf(radius / cell);
  st->last_error = 1e10;
  st->loss = 0.0;
  st->countdown = 3;
  st->stop_iter = INT_MAX;
  st->iters = 0;
  st->err = 0;
  st->decided = 0;
}

// start of a call: bounds <- empty, T <- identity, accumulators and counters <- 0, the three grids
// <- empty (one launch instead of a dozen memsets)
__global__ void __launch_bounds__(kBlock)
icp_clear_kernel(float* __restrict__ bounds24, double* __restrict__ Tb, double* __restrict__ acc,
                 int32_t* __restrict__ first, int32_t* __restrict__ cnt, int64_t C, Grid2 gr, Grid2 gm0,
                 Grid2 gm1, Grid2 ga) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (i < 24) bounds24[i] = ((i & 7) < 4) ? 1e30f : -1e30f;
  for (int64_t q = i; q < 2 * C * 16; q += stride) {
    int e = (int)(q & 15);
    Tb[q] = (e == 0 || e == 5 || e == 10 || e == 15) ? 1.0 : 0.0;
  }
  for (int64_t q = i; q < 2 * C * kAcc; q += stride) acc[q] = 0.0;
  for (int64_t q = i; q < C; q += stride) { cnt[q] = 0; first[q] = INT_MAX; }
  g2_clear(gr, i, stride);
  g2_clear(gm0, i, stride);
  g2_clear(gm1, i, stride);
  g2_clear(ga, i, stride);
}

// the three grids of a call in one launch: blockIdx.y = 0 non-stationary reference, 1 all reference,
// 2 the moving points as they arrive (the grid of iteration 1)
__global__ void __launch_bounds__(kBlock)
icp_insert3_kernel(const float4* __restrict__ ref, int Rn, const float4* __restrict__ all, int Ra,
                   const float4* __restrict__ mov, int Mn, const GridDesc* __restrict__ gp, Grid2 gr,
                   Grid2 ga, Grid2 gm1, IcpState* __restrict__ st) {
  const float4* pts = blockIdx.y == 0 ? ref : (blockIdx.y == 1 ? all : mov);
  const int n = blockIdx.y == 0 ? Rn : (blockIdx.y == 1 ? Ra : Mn);
  if ((int64_t)blockIdx.x * kBlock >= n) return;
  const Grid2& g = blockIdx.y == 0 ? gr : (blockIdx.y == 1 ? ga : gm1);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  GridDesc d = *gp;
  bool in = i < n;
  float4 p = in ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  g2_insert_point(d, g, in, i, p, &st->err);
}

// min / max of the three clouds in one launch: blockIdx.y selects the cloud
__global__ void __launch_bounds__(kBlock)
icp_bounds_kernel(const float4* __restrict__ ref, int Rn, const float4* __restrict__ mov, int Mn,
                  const float4* __restrict__ all, int Ra, float* __restrict__ bounds24,
                  const int32_t* __restrict__ comp, int32_t* __restrict__ first) {
  const float4* pts = blockIdx.y == 0 ? ref : (blockIdx.y == 1 ? mov : all);
  int n = blockIdx.y == 0 ? Rn : (blockIdx.y == 1 ? Mn : Ra);
  float* out8 = bounds24 + 8 * blockIdx.y;
  float mn[4] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
  float mx[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 p = pts[i];
    if (blockIdx.y == 1) atomicMin(&first[comp[i]], i);  // the thread that will own the component's T
    mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
    mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
    mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
    mn[3] = fminf(mn[3], p.w); mx[3] = fmaxf(mx[3], p.w);
  }
  __shared__ float smn[kBlock / 64][4], smx[kBlock / 64][4];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    float a = wave_min(mn[d]), b = wave_max(mx[d]);
    if (lane == 0) { smn[w][d] = a; smx[w][d] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 4 && n > 0) {
    int d = threadIdx.x;
    float a = smn[0][d], b = smx[0][d];
    for (int k = 1; k < kBlock / 64; ++k) { a = fminf(a, smn[k][d]); b = fmaxf(b, smx[k][d]); }
    atomic_min_f32(&out8[d], a);
    atomic_max_f32(&out8[4 + d], b);
  }
}

// loss-delta / 3-strike stop rule of iteration k_done (registration_utils.py:186-193), evaluated by
// ONE workgroup from the per-workgroup loss partials of the apply launch (fixed reduction tree). It may lower stop_iter to
// k_done, which still admits every kernel of iteration k_done.
__device__ __forceinline__ void icp_decide(IcpState* __restrict__ st, const double* __restrict__ lossp,
                                           int nparts, int k_done, double delta) {
  __shared__ double red[kBlock / 64];
  if (k_done < 1 || st->decided >= k_done || st->stop_iter != INT_MAX) return;  // uniform per workgroup
  double part = 0.0;
  for (int q = threadIdx.x; q < nparts; q += kBlock) part += lossp[q];
  part = wave_sum_f64(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    double loss = 0.0;
    for (int q = 0; q < kBlock / 64; ++q) loss += red[q];
    st->loss = loss;
    if (st->last_error - loss < delta) st->countdown -= 1;
    else st->countdown = 3;
    if (st->countdown <= 0) st->stop_iter = k_done;
    else st->last_error = loss;
    st->decided = k_done;
  }
}

// end of a chunk: the stop rule of every problem's last iteration (the others were settled by the
// next search launch), and the state mirrored where the host reads it (`out` may be pinned host
// memory: the host then needs no copy, only the stream synchronisation)
__global__ void __launch_bounds__(kBlock)
icp_decide_kernel(const IcpProblem* __restrict__ P, int kbase, IcpState* __restrict__ out) {
  const IcpProblem& pr = P[blockIdx.y];
  icp_decide(pr.st, pr.lossp, pr.nparts, pr.k0 + kbase + pr.n_it - 1, pr.delta);
  __syncthreads();
  if (out && threadIdx.x == 0) out[blockIdx.y] = *pr.st;
}
__global__ void __launch_bounds__(kBlock)
icp_decide_kernel_v(const IcpProblem pr, int kbase, IcpState* __restrict__ out) {
  icp_decide(pr.st, pr.lossp, pr.nparts, pr.k0 + kbase + pr.n_it - 1, pr.delta);
  __syncthreads();
  if (out && threadIdx.x == 0) out[0] = *pr.st;
}

// one f64 atomic per (wave, component, accumulator): lanes sharing a component are summed by
// 64-lane butterflies first. All 64 lanes must call this.
template <int NV>
__device__ __forceinline__ void grouped_atomic_add(bool active, int key, const double (&v)[NV],
                                                   double* __restrict__ base, int stride) {
  int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(active);
  while (todo) {
    int leader = __ffsll((long long)todo) - 1;
    int lkey = __shfl(key, leader, 64);
    bool mine = active && key == lkey;
    unsigned long long grp = __ballot(mine);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double x = wave_sum_f64(mine ? v[i] : 0.0);
      if (lane == leader) atomicAdd(&base[(int64_t)lkey * stride + i], x);
    }
    todo &= ~grp;
  }
}

// 32 lanes per query. Queries [0,Mn): forward (moving -> reference, df = +frame_offset);
// queries [Mn, Mn+Rn): backward (reference -> moving, df = -frame_offset).
__device__ __forceinline__ void icp_nn_kernel_body(const IcpProblem& pr, int j, int kbase) {
  if (j >= pr.n_it) return;
  const int k = pr.k0 + kbase + j;
  IcpState* st = pr.st;
  if (k > st->stop_iter) return;
  const int Mn = pr.Mn, Rn = pr.Rn;
  if (blockIdx.x != 0 && (int64_t)blockIdx.x * (kBlock / 32) >= (int64_t)Mn + Rn) return;
  // iteration k, first launch: workgroup 0 settles the stop rule of iteration k-1 (its loss partials
  // were written by the previous launch). The other workgroups may run before the stop is known:
  // they only write this iteration's neighbour buffers, which nothing reads after a stop.
  if (blockIdx.x == 0) icp_decide(st, pr.lossp, pr.nparts, k - 1, pr.delta);
  GridDesc g = *pr.grid;
  int qid = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  int sub = threadIdx.x & 31;
  bool fwd = qid < Mn;
  bool valid = qid < Mn + Rn;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) p = fwd ? pr.mov[qid] : pr.ref[qid - Mn];
  // a wave may straddle the forward/backward boundary: its two halves then search different grids
  bool any_fwd = __ballot(valid && fwd) != 0;
  bool any_bwd = __ballot(valid && !fwd) != 0;
  int res = -1;
  if (pr.probe_stage == 1) {  // timing probe: launch + head loads only (stages 11.. belong to apply)
    if (valid && sub == 0 && p.x == 1.2345e30f) pr.nn_f[0] = 0;
    return;
  }
  if (any_fwd) {
    int r = g2_nearest_coop(g, pr.gr, p, pr.df, valid && fwd, &st->err, nullptr, pr.probe_stage);
    if (fwd) res = r;
  }
  if (any_bwd) {
    int r = g2_nearest_coop(g, pr.gm[k & 1], p, -pr.df, valid && !fwd, &st->err, nullptr, pr.probe_stage);
    if (!fwd) res = r;
  }
  if (valid && sub == 0) {
    if (fwd) pr.nn_f[(size_t)(k & 1) * (Mn > 0 ? Mn : 1) + qid] = res;
    else pr.nn_b[(size_t)(k & 1) * (Rn > 0 ? Rn : 1) + (qid - Mn)] = res;
  }
}

// array form (blockIdx.y = problem) and by-value form for a single problem: the descriptor then
// sits in the kernel-argument segment, one dependent global load less at the head of the launch
__global__ void __launch_bounds__(kBlock)
icp_nn_kernel(const IcpProblem* __restrict__ P, int j, int kbase) {
  icp_nn_kernel_body(P[blockIdx.y], j, kbase);
}
__global__ void __launch_bounds__(kBlock)
icp_nn_kernel_v(const IcpProblem pr, int j, int kbase) {
  icp_nn_kernel_body(pr, j, kbase);
}

// raw moments of the matched pairs per component, one thread per potential edge. Each workgroup
// first accumulates into a small LDS table keyed by component (ds_add_f64; neighbouring edges
// mostly share a component), then flushes one global f64 atomic per (component, accumulator).
constexpr int kMomSlots = 64;
__device__ __forceinline__ void icp_moments_kernel_body(const IcpProblem& pr, int j, int kbase) {
  if (j >= pr.n_it) return;
  const int k = pr.k0 + kbase + j;
  if (k > pr.st->stop_iter) return;
  const int Mn = pr.Mn, Rn = pr.Rn;
  if ((int64_t)blockIdx.x * kBlock >= (int64_t)Mn + Rn) return;
  const float4* __restrict__ mov = pr.mov;
  const float4* __restrict__ ref = pr.ref;
  const int32_t* __restrict__ comp = pr.comp;
  const int32_t* __restrict__ nn_f = pr.nn_f + (size_t)(k & 1) * (Mn > 0 ? Mn : 1);
  const int32_t* __restrict__ nn_b = pr.nn_b + (size_t)(k & 1) * (Rn > 0 ? Rn : 1);
  double* __restrict__ acc = pr.acc + (size_t)(k & 1) * pr.C * kAcc;
  __shared__ int skey[kMomSlots];
  __shared__ double sacc[kMomSlots][17];
  for (int q = threadIdx.x; q < kMomSlots; q += kBlock) skey[q] = -1;
  for (int q = threadIdx.x; q < kMomSlots * 17; q += kBlock) (&sacc[0][0])[q] = 0.0;
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool active = false;
  int c = 0, m = -1, r = -1;
  if (i < Mn) { m = i; r = nn_f[i]; active = r >= 0; }
  else if (i < Mn + Rn) { r = i - Mn; m = nn_b[r]; active = m >= 0; }
  if (active) {
    float4 pf = mov[m], qf = ref[r];
    c = comp[m];
    double p[3] = {pf.y, pf.z, pf.w}, q[3] = {qf.y, qf.z, qf.w};
    double v[17];
    v[0] = 1.0;
    double d2 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      v[1 + a] = p[a];
      v[4 + a] = q[a];
#pragma unroll
      for (int b = 0; b < 3; ++b) v[7 + a * 3 + b] = p[a] * q[b];
      double e = p[a] - q[a];
      d2 += e * e;
    }
    v[16] = d2;
    // claim / find the component's LDS slot (linear probing); fall back to global atomics if full
    int slot = (int)(((unsigned)c * 2654435761u) >> 26) & (kMomSlots - 1);
    int found = -1;
    for (int t = 0; t < kMomSlots; ++t) {
      int old = atomicCAS(&skey[slot], -1, c);
      if (old == -1 || old == c) { found = slot; break; }
      slot = (slot + 1) & (kMomSlots - 1);
    }
    if (found >= 0) {
#pragma unroll
      for (int q2 = 0; q2 < 17; ++q2) atomicAdd(&sacc[found][q2], v[q2]);
    } else {
#pragma unroll
      for (int q2 = 0; q2 < 17; ++q2) atomicAdd(&acc[(int64_t)c * kAcc + q2], v[q2]);
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < kMomSlots * 17; q += kBlock) {
    int sl = q / 17, a = q - sl * 17;
    int key = skey[sl];
    if (key >= 0) atomicAdd(&acc[(int64_t)key * kAcc + a], sacc[sl][a]);
  }
}

// array form (blockIdx.y = problem) and by-value form for a single problem: the descriptor then
// sits in the kernel-argument segment, one dependent global load less at the head of the launch
__global__ void __launch_bounds__(kBlock)
icp_moments_kernel(const IcpProblem* __restrict__ P, int j, int kbase) {
  icp_moments_kernel_body(P[blockIdx.y], j, kbase);
}
__global__ void __launch_bounds__(kBlock)
icp_moments_kernel_v(const IcpProblem pr, int j, int kbase) {
  icp_moments_kernel_body(pr, j, kbase);
}

// centre of the matched points: fp32 segment mean, then promoted (registration_utils.py:150-151)
__device__ __forceinline__ void centres(const double* __restrict__ a, double* cm, double* cr) {
  double n = a[0];
  if (n > 0.5) {
    for (int d = 0; d < 3; ++d) {
      cm[d] = (double)(float)(a[1 + d] / n);
      cr[d] = (double)(float)(a[4 + d] / n);
    }
  } else {
    for (int d = 0; d < 3; ++d) cm[d] = cr[d] = 0.0;
  }
}

// l1 passes on the positions saved before the last update. MODE 0: sum of dist; MODE 1: sum of
// dist clamped to mean +- trunc (truncated_robust_mean, registration_utils.py:44-58).
template <int MODE>
__global__ void __launch_bounds__(kBlock)
icp_l1_kernel(const float4* __restrict__ mov, int Mn, const float4* __restrict__ ref, int Rn,
              const int32_t* __restrict__ comp, const int32_t* __restrict__ nn_f,
              const int32_t* __restrict__ nn_b, double* __restrict__ acc, double trunc) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool active = false;
  int c = 0, m = -1, r = -1;
  if (i < Mn) { m = i; r = nn_f[i]; active = r >= 0; }
  else if (i < Mn + Rn) { r = i - Mn; m = nn_b[r]; active = m >= 0; }
  double v[1] = {0.0};
  if (active) {
    c = comp[m];
    float4 p = mov[m], q = ref[r];
    const double* a = acc + (int64_t)c * kAcc;
    double cm[3], cr[3];
    centres(a, cm, cr);
    double e0 = ((double)p.y - cm[0]) - ((double)q.y - cr[0]);
    double e1 = ((double)p.z - cm[1]) - ((double)q.z - cr[1]);
    double e2 = ((double)p.w - cm[2]) - ((double)q.w - cr[2]);
    double dist = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
    if (MODE == 0) {
      v[0] = dist;
    } else {
      double mean = a[17] / a[0];
      v[0] = fmin(fmax(dist, mean - trunc), mean + trunc);
    }
  }
  grouped_atomic_add<1>(active, c, v, acc + (MODE == 0 ? 17 : 18), kAcc);
}

// ---- 3x3 SVD by one-sided Jacobi (f64), singular values descending ----
__device__ inline void svd3(const double M[9], double U[9], double S[3], double V[9]) {
  double A[9];
  for (int i = 0; i < 9; ++i) { A[i] = M[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < 3; ++i) {
          al += A[i * 3 + p] * A[i * 3 + p];
          be += A[i * 3 + q] * A[i * 3 + q];
          ga += A[i * 3 + p] * A[i * 3 + q];
        }
        double lim = 1e-15 * sqrt(al * be);
        if (fabs(ga) <= lim || ga == 0.0) continue;
        off = fmax(off, fabs(ga) / (sqrt(al * be) + 1e-300));
        double zeta = (be - al) / (2.0 * ga);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          double ap = A[i * 3 + p], aq = A[i * 3 + q];
          A[i * 3 + p] = c * ap - s * aq;
          A[i * 3 + q] = s * ap + c * aq;
          double vp = V[i * 3 + p], vq = V[i * 3 + q];
          V[i * 3 + p] = c * vp - s * vq;
          V[i * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  for (int j = 0; j < 3; ++j)
    S[j] = sqrt(A[0 * 3 + j] * A[0 * 3 + j] + A[1 * 3 + j] * A[1 * 3 + j] + A[2 * 3 + j] * A[2 * 3 + j]);
  int ord[3] = {0, 1, 2};
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (S[ord[b]] > S[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
  double A2[9], V2[9], S2[3];
  for (int j = 0; j < 3; ++j) {
    S2[j] = S[ord[j]];
    for (int i = 0; i < 3; ++i) { A2[i * 3 + j] = A[i * 3 + ord[j]]; V2[i * 3 + j] = V[i * 3 + ord[j]]; }
  }
  for (int j = 0; j < 3; ++j) {
    S[j] = S2[j];
    for (int i = 0; i < 3; ++i) {
      V[i * 3 + j] = V2[i * 3 + j];
      U[i * 3 + j] = S2[j] > 1e-300 ? A2[i * 3 + j] / S2[j] : 0.0;
    }
  }
  // complete U when a singular value vanished (rank-deficient M): orthonormal completion
  if (S[2] <= 1e-300) {
    if (S[1] <= 1e-300) {
      if (S[0] <= 1e-300) { for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
      double u0[3] = {U[0], U[3], U[6]};
      int k = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
      double e[3] = {0, 0, 0}; e[k] = 1.0;
      double d = e[0] * u0[0] + e[1] * u0[1] + e[2] * u0[2];
      double w[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
      double nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      U[1] = w[0] / nw; U[4] = w[1] / nw; U[7] = w[2] / nw;
    }
    double a[3] = {U[0], U[3], U[6]}, b[3] = {U[1], U[4], U[7]};
    U[2] = a[1] * b[2] - a[2] * b[1];
    U[5] = a[2] * b[0] - a[0] * b[2];
    U[8] = a[0] * b[1] - a[1] * b[0];
  }
}

__device__ inline double det3(const double A[9]) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) +
         A[2] * (A[3] * A[7] - A[4] * A[6]);
}

// regularised Kabsch step of one component from its raw moments `a` and accumulated transform Tc:
// rotation R, translation t, updated transform Tn = [R t] Tc and the component's loss term
__device__ inline void icp_solve_component(const double* __restrict__ a, const double* __restrict__ Tc,
                                           double reg, double R[9], double t[3], double Tn[16],
                                           double* ls_out) {
  double n = a[0];
  double cm[3], cr[3];
  centres(a, cm, cr);
  double M[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      // sum P(x)Q = sum p q^T - cm sum q^T - sum p cr^T + n cm cr^T;  robust_mean divides by n
      double cov = a[7 + i * 3 + j] - cm[i] * a[4 + j] - a[1 + i] * cr[j] + n * cm[i] * cr[j];
      if (n > 0.5) cov = cov / n; else cov = 0.0;
      M[i * 3 + j] = cov + Tc[i * 4 + j] * reg;
    }
  // R = V diag(1, 1, det(V U^T)) U^T for M = U S V^T. With det M > 0 (always, unless the matches
  // outweigh the regulariser reg * T_rot) that is the transpose of the orthogonal polar factor of M,
  // which Newton's iteration X <- (X + X^-T) / 2 reaches in a handful of steps with one division
  // each; the Jacobi SVD (dozens of dependent f64 sqrt / div) is kept for the rest.
  bool have_R = false;
  {
    double fro = 0.0;
    for (int i = 0; i < 9; ++i) fro += M[i] * M[i];
    double dM = det3(M);
    if (dM > 0.0 && fro > 0.0) {
      double sc = 1.7320508075688772 / sqrt(fro);  // singular values to ~1
      double X[9];
      for (int i = 0; i < 9; ++i) X[i] = M[i] * sc;
      bool conv = false;
      for (int itn = 0; itn < 12 && !conv; ++itn) {
        double Cf[9];  // cofactor matrix: X^-T = Cf / det X
        Cf[0] = X[4] * X[8] - X[5] * X[7];
        Cf[1] = X[5] * X[6] - X[3] * X[8];
        Cf[2] = X[3] * X[7] - X[4] * X[6];
        Cf[3] = X[2] * X[7] - X[1] * X[8];
        Cf[4] = X[0] * X[8] - X[2] * X[6];
        Cf[5] = X[1] * X[6] - X[0] * X[7];
        Cf[6] = X[1] * X[5] - X[2] * X[4];
        Cf[7] = X[2] * X[3] - X[0] * X[5];
        Cf[8] = X[0] * X[4] - X[1] * X[3];
        double dx = X[0] * Cf[0] + X[1] * Cf[1] + X[2] * Cf[2];
        if (!(dx > 1e-300)) break;
        double inv = 1.0 / dx, delta = 0.0;
        for (int i = 0; i < 9; ++i) {
          double xn = 0.5 * (X[i] + Cf[i] * inv);
          delta = fmax(delta, fabs(xn - X[i]));
          X[i] = xn;
        }
        conv = delta < 1e-15;
      }
      if (conv) {
        // one more step from a converged iterate changes nothing above 1e-30: X is orthogonal
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) R[i * 3 + j] = X[j * 3 + i];
        have_R = true;
      }
    }
  }
  if (!have_R) {
    double U[9], S[3], V[9];
    svd3(M, U, S, V);
    double VUt[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        VUt[i * 3 + j] = V[i * 3 + 0] * U[j * 3 + 0] + V[i * 3 + 1] * U[j * 3 + 1] + V[i * 3 + 2] * U[j * 3 + 2];
    double sg = det3(VUt);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        R[i * 3 + j] = V[i * 3 + 0] * U[j * 3 + 0] + V[i * 3 + 1] * U[j * 3 + 1] + sg * V[i * 3 + 2] * U[j * 3 + 2];
  }
  for (int i = 0; i < 3; ++i)
    t[i] = cr[i] - (cm[0] * R[i * 3 + 0] + cm[1] * R[i * 3 + 1] + cm[2] * R[i * 3 + 2]);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j)
      Tn[i * 4 + j] = R[i * 3 + 0] * Tc[0 * 4 + j] + R[i * 3 + 1] * Tc[1 * 4 + j] +
                      R[i * 3 + 2] * Tc[2 * 4 + j] + t[i] * Tc[3 * 4 + j];
  for (int j = 0; j < 4; ++j) Tn[12 + j] = Tc[12 + j];
  // sum |P - Q|^2 = sum |p-q|^2 - 2 d . sum(p-q) + n |d|^2,  d = cm - cr
  double ls = 0.0;
  if (n > 0.5) {
    ls = a[16];
    for (int q = 0; q < 3; ++q) {
      double d = cm[q] - cr[q];
      ls += -2.0 * d * (a[1 + q] - a[4 + q]) + n * d * d;
    }
  }
  *ls_out = ls;
}

// iteration k, last launch: every moving point solves its own component's step from the completed
// moments (redundantly across the points of a component: a 3x3 SVD is cheaper than another launch)
// and moves; the component's first point also stores the updated transform (read buffer k-1, write
// buffer k) and the loss term. The previous positions are kept for the l1 passes. The moved points go
// straight into the grid of the next iteration (the moving grid is double-buffered: no separate
// build launch), and the grid this iteration searched is reset.
__device__ __forceinline__ void icp_apply_kernel_body(const IcpProblem& pr, int j, int kbase) {
  if (j >= pr.n_it) return;
  const int k = pr.k0 + kbase + j;
  IcpState* st = pr.st;
  if (k > st->stop_iter) return;
  if ((int)blockIdx.x >= pr.nparts) return;
  const int Mn = pr.Mn;
  const int64_t C = pr.C;
  const double reg = pr.reg;
  float4* __restrict__ mov = pr.mov;
  float4* __restrict__ prev = pr.prev;
  const int32_t* __restrict__ comp = pr.comp;
  const int32_t* __restrict__ first = pr.first;
  double* __restrict__ acc2 = pr.acc;
  double* __restrict__ Tb = pr.Tb;
  double* __restrict__ lossp = pr.lossp;
  const GridDesc* __restrict__ gp = pr.grid;
  const Grid2& gm_done = pr.gm[k & 1];
  const Grid2& gm_next = pr.gm[(k + 1) & 1];
  __shared__ double red[kBlock / 64];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  // the searches of this iteration are done with gm_done: reset it (it is filled again by the apply
  // of iteration k+1); gm_next was reset by the apply of iteration k-1 and takes the moved points
  g2_clear(gm_done, i, (int64_t)pr.nparts * blockDim.x);
  if (i == 0) st->iters = k;

  double ls_mine = 0.0;
  float4 moved = make_float4(0.f, 0.f, 0.f, 0.f);
  const int probe = pr.probe_stage;  // timing probe, stages 11..13 cut this kernel short
  if (probe == 11) {  // launch + grid reset + head loads
    if (i < Mn && comp[i] == -12345) lossp[0] = 0.0;
    return;
  }
  if (i < Mn) {
    int c = comp[i];
    const double* a = acc2 + ((size_t)(k & 1) * C + c) * kAcc;
    const double* Tc = Tb + ((size_t)((k - 1) & 1) * C + c) * 16;
    double R[9], t[3], Tn[16], ls;
    icp_solve_component(a, Tc, reg, R, t, Tn, &ls);
    if (probe == 12) {  // + moments / transform loads + the solve
      if (R[0] == -12345.0) lossp[0] = ls;
      return;
    }
    float4 p = mov[i];
    prev[i] = p;
    double x = p.y, y = p.z, z = p.w;
    float nx = (float)(x * R[0] + y * R[1] + z * R[2] + t[0]);
    float ny = (float)(x * R[3] + y * R[4] + z * R[5] + t[1]);
    float nz = (float)(x * R[6] + y * R[7] + z * R[8] + t[2]);
    moved = make_float4(p.x, nx, ny, nz);
    mov[i] = moved;
    if (first[c] == i) {
      // the component's owner: transform, loss term, and the other moment buffer zeroed for the
      // next iteration (nothing reads it during this launch)
      double* To = Tb + ((size_t)(k & 1) * C + c) * 16;
      for (int q = 0; q < 16; ++q) To[q] = Tn[q];
      double* an = acc2 + ((size_t)((k + 1) & 1) * C + c) * kAcc;
      for (int q = 0; q < kAcc; ++q) an[q] = 0.0;
      ls_mine = ls;
    }
  }
  if (probe == 13) return;  // + move and stores, without the grid insert and the loss reduction
  {
    GridDesc d = *gp;
    g2_insert_point(d, gm_next, i < Mn, i, moved, &st->err);
  }

  // loss partial of this workgroup (fixed order: lanes, then waves)
  ls_mine = wave_sum_f64(ls_mine);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ls_mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int q = 0; q < kBlock / 64; ++q) tot += red[q];
    lossp[blockIdx.x] = tot;
  }
}

// array form (blockIdx.y = problem) and by-value form for a single problem: the descriptor then
// sits in the kernel-argument segment, one dependent global load less at the head of the launch
__global__ void __launch_bounds__(kBlock)
icp_apply_kernel(const IcpProblem* __restrict__ P, int j, int kbase) {
  icp_apply_kernel_body(P[blockIdx.y], j, kbase);
}
__global__ void __launch_bounds__(kBlock)
icp_apply_kernel_v(const IcpProblem pr, int j, int kbase) {
  icp_apply_kernel_body(pr, j, kbase);
}

// coverage: moving points (final positions) with a reference point (stationary included) in range
__global__ void __launch_bounds__(kBlock)
icp_coverage_kernel(const float4* __restrict__ mov, int Mn, const int32_t* __restrict__ comp,
                    const GridDesc* __restrict__ gp, int df, Grid2 ga, IcpState* __restrict__ st,
                    int32_t* __restrict__ cnt) {
  GridDesc g = *gp;
  int qid = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  bool valid = qid < Mn;
  float4 p = valid ? mov[qid] : make_float4(0.f, 0.f, 0.f, 0.f);
  int r = g2_nearest_coop(g, ga, p, df, valid, &st->err);
  if (valid && (threadIdx.x & 31) == 0 && r >= 0) atomicAdd(&cnt[comp[qid]], 1);
}

__global__ void __launch_bounds__(kBlock)
icp_finish_kernel(int64_t C, const double* __restrict__ acc, const int32_t* __restrict__ cnt,
                  const int64_t* __restrict__ comp_deg, int have_l1, const double* __restrict__ Tfin,
                  double* __restrict__ T, double* __restrict__ l1, float* __restrict__ ratio,
                  const IcpState* __restrict__ st, IcpState* __restrict__ out) {
  if (out && blockIdx.x == 0 && threadIdx.x == 0) *out = *st;  // err word for the host (pinned)
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < C;
       c += (int64_t)gridDim.x * blockDim.x) {
    double n = acc[c * kAcc];
    l1[c] = (have_l1 && n > 0.5) ? acc[c * kAcc + 18] / n : 0.0;
    // component_edge_count / (component_deg.float() + 1e-6)  (registration_utils.py:197-199)
    ratio[c] = (float)cnt[c] / ((float)comp_deg[c] + 1e-6f);
    for (int q = 0; q < 16; ++q) T[c * 16 + q] = Tfin[c * 16 + q];
  }
}

// ---- optional HIP-event timing of the dominant kernel (icp_nn_kernel), for bench.py ----
struct NnProfile {
  double ms = 0.0;
  int64_t launches = 0;
  double bytes = 0.0;
};
static bool g_prof_on = false;
static NnProfile g_prof_total;
static std::mutex g_prof_mu;
struct NnEvents {
  hipEvent_t ev[2 * 96];
  bool have = false;
};
static thread_local NnEvents t_ev;

// launches of one chunk (n_max iteration positions): an array of problems already on the device, or
// (host_prob != nullptr) ONE problem passed by value
static int icp_launch_chunk(const IcpProblem* d_probs, const IcpProblem* host_prob, int nprob, int n_max,
                            int kbase, int max_nb_q, int max_nb_e, int max_nb_m, IcpState* out,
                            hipStream_t s, hipEvent_t* ev) {
  for (int j = 0; j < n_max; ++j) {
    if (host_prob) {
      // profiling: the events ride on the dispatch itself (hipExtLaunchKernelGGL), so their
      // difference is the kernel's own begin-to-end time, as rocprofv3 reports it. The search kernel
      // by default; PCS_PROF_KERNEL=moments|apply moves them (tools/icp_stages.py)
      static const int target = [] {
        const char* e = getenv("PCS_PROF_KERNEL");
        return !e ? 0 : (e[0] == 'm' ? 1 : (e[0] == 'a' ? 2 : 0));
      }();
      if (ev && target == 0) {
        hipExtLaunchKernelGGL(icp_nn_kernel_v, dim3(max_nb_q), dim3(kBlock), 0, s, ev[2 * j], ev[2 * j + 1],
                              0, *host_prob, j, kbase);
      } else {
        icp_nn_kernel_v<<<max_nb_q, kBlock, 0, s>>>(*host_prob, j, kbase);
      }
      if (ev && target == 1) {
        hipExtLaunchKernelGGL(icp_moments_kernel_v, dim3(max_nb_e), dim3(kBlock), 0, s, ev[2 * j],
                              ev[2 * j + 1], 0, *host_prob, j, kbase);
      } else {
        icp_moments_kernel_v<<<max_nb_e, kBlock, 0, s>>>(*host_prob, j, kbase);
      }
      if (ev && target == 2) {
        hipExtLaunchKernelGGL(icp_apply_kernel_v, dim3(max_nb_m), dim3(kBlock), 0, s, ev[2 * j],
                              ev[2 * j + 1], 0, *host_prob, j, kbase);
      } else {
        icp_apply_kernel_v<<<max_nb_m, kBlock, 0, s>>>(*host_prob, j, kbase);
      }
    } else {
      if (ev) {
        hipExtLaunchKernelGGL(icp_nn_kernel, dim3(max_nb_q, nprob), dim3(kBlock), 0, s, ev[2 * j],
                              ev[2 * j + 1], 0, d_probs, j, kbase);
      } else {
        icp_nn_kernel<<<dim3(max_nb_q, nprob), kBlock, 0, s>>>(d_probs, j, kbase);
      }
      icp_moments_kernel<<<dim3(max_nb_e, nprob), kBlock, 0, s>>>(d_probs, j, kbase);
      icp_apply_kernel<<<dim3(max_nb_m, nprob), kBlock, 0, s>>>(d_probs, j, kbase);
    }
  }
  // the stop rule of the chunk's last iteration (the others were settled by the next search)
  if (host_prob) icp_decide_kernel_v<<<1, kBlock, 0, s>>>(*host_prob, kbase, out);
  else icp_decide_kernel<<<dim3(1, nprob), kBlock, 0, s>>>(d_probs, kbase, out);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// ---- shared launches for concurrent calls ---------------------------------------------------
// Kernels of different streams overlap only 2-3 deep on this hardware, while the iteration kernels
// of one problem leave most of the device idle. Threads that registered as participants (the
// tracking units, pcs_tracker.hip) therefore hand their chunks to ONE launcher thread: whatever is
// pending when the launcher is free goes out as one array of problems (blockIdx.y = problem), so
// nine chains of ~50 us iterations become one chain of wider kernels. A caller synchronises its own
// stream before submitting and continues on it after the launcher has synchronised the shared
// stream, so no cross-stream ordering is needed. While a batch runs (~0.4 ms) later requests queue
// up, which is what keeps the batches full without any waiting window.
constexpr int kMaxBatch = 64;
constexpr int kMaxGroups = 16;
static double g_batch_stats[4] = {0, 0, 0, 0};  // batches, problems, busy seconds, idle seconds
struct IcpRequest {
  IcpProblem prob;
  int nb_q, nb_e, nb_m;
  double bytes_per_launch;
  IcpState result;
  int rc = 0;
  bool done = false;
};

class IcpBatcher {
 public:
  // one launcher (thread + stream) per group id: callers that move in lock-step -- the component
  // keys of one anchor frame -- share a group, independent groups run side by side
  static IcpBatcher& get(int group) {
    static std::mutex gmu;
    static IcpBatcher* inst[kMaxGroups] = {};  // never destroyed: the launcher threads outlive main
    std::lock_guard<std::mutex> lk(gmu);
    int g = group < 0 ? 0 : group % kMaxGroups;
    if (!inst[g]) inst[g] = new IcpBatcher();
    return *inst[g];
  }
  void add_participant() {
    std::lock_guard<std::mutex> lk(mu_);
    ++participants_;
  }
  void remove_participant() {
    std::lock_guard<std::mutex> lk(mu_);
    --participants_;
  }
  // a participant is between the start of an ICP call and the end of its iteration loop: it WILL
  // submit a chunk shortly, so the launcher holds the next batch until all of these have
  void enter_icp() {
    std::lock_guard<std::mutex> lk(mu_);
    ++in_icp_;
  }
  void leave_icp() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      --in_icp_;
    }
    cv_work_.notify_one();
  }
  int run(IcpRequest* req) {
    std::unique_lock<std::mutex> lk(mu_);
    if (!started_) {
      if (hipGetDevice(&device_) != hipSuccess) return PCS_ERR_HIP;
      started_ = true;
      std::thread(&IcpBatcher::loop, this).detach();
    }
    req->done = false;
    pending_.push_back(req);
    cv_work_.notify_one();
    cv_done_.wait(lk, [&] { return req->done; });
    return req->rc;
  }

 private:
  void loop() {
    bool ok = hipSetDevice(device_) == hipSuccess &&
              hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(&d_probs_, kMaxBatch * sizeof(IcpProblem)) == hipSuccess &&
              hipHostMalloc(&h_probs_, kMaxBatch * sizeof(IcpProblem)) == hipSuccess &&
              hipHostMalloc(&h_state_, kMaxBatch * sizeof(IcpState)) == hipSuccess;
    hipEvent_t ev[2 * 16];
    bool have_ev = false;
    std::vector<IcpRequest*> batch;
    auto t_free = std::chrono::steady_clock::now();
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        // go when every participant that is inside an ICP call has submitted its chunk (those that
        // were in the previous batch need a few microseconds to come back); 2 ms safety net
        for (;;) {
          bool ready = cv_work_.wait_for(lk, std::chrono::milliseconds(2), [&] {
            return !pending_.empty() && (int)pending_.size() >= in_icp_;
          });
          if (ready || !pending_.empty()) break;
        }
        batch.clear();
        while (!pending_.empty() && (int)batch.size() < kMaxBatch) {
          batch.push_back(pending_.front());
          pending_.erase(pending_.begin());
        }
      }
      auto t_start = std::chrono::steady_clock::now();
      int rc = ok ? PCS_OK : PCS_ERR_HIP;
      int n = (int)batch.size(), n_max = 0, q = 1, e = 1, m = 1;
      double bytes = 0.0;
      if (rc == PCS_OK) {
        for (int i = 0; i < n; ++i) {
          h_probs_[i] = batch[i]->prob;
          if (batch[i]->prob.n_it > n_max) n_max = batch[i]->prob.n_it;
          if (batch[i]->nb_q > q) q = batch[i]->nb_q;
          if (batch[i]->nb_e > e) e = batch[i]->nb_e;
          if (batch[i]->nb_m > m) m = batch[i]->nb_m;
          bytes += batch[i]->bytes_per_launch;
        }
        bool prof = g_prof_on && n_max <= 16;
        if (prof && !have_ev) {
          for (int i = 0; i < 32; ++i) prof = prof && hipEventCreate(&ev[i]) == hipSuccess;
          have_ev = prof;
        }
        if (hipMemcpyAsync(d_probs_, h_probs_, (size_t)n * sizeof(IcpProblem), hipMemcpyHostToDevice,
                           stream_) != hipSuccess)
          rc = PCS_ERR_HIP;
        if (rc == PCS_OK)
          rc = icp_launch_chunk(d_probs_, nullptr, n, n_max, 0, q, e, m, h_state_, stream_, prof ? ev : nullptr);
        if (rc == PCS_OK && hipStreamSynchronize(stream_) != hipSuccess) rc = PCS_ERR_HIP;
        if (rc == PCS_OK && prof) {
          double ms_sum = 0.0;
          for (int j = 0; j < n_max; ++j) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[2 * j], ev[2 * j + 1]) == hipSuccess) ms_sum += ms;
          }
          std::lock_guard<std::mutex> lk(g_prof_mu);
          g_prof_total.ms += ms_sum;
          g_prof_total.launches += n_max;
          g_prof_total.bytes += bytes * n_max;
        }
      }
      {
        std::lock_guard<std::mutex> lk(mu_);
        auto t_end = std::chrono::steady_clock::now();
        g_batch_stats[0] += 1;
        g_batch_stats[1] += n;
        g_batch_stats[2] += std::chrono::duration<double>(t_end - t_start).count();
        g_batch_stats[3] += std::chrono::duration<double>(t_start - t_free).count();
        t_free = t_end;
        for (int i = 0; i < n; ++i) {
          if (rc == PCS_OK) batch[i]->result = h_state_[i];
          batch[i]->rc = rc;
          batch[i]->done = true;
        }
      }
      cv_done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  std::vector<IcpRequest*> pending_;
  int participants_ = 0;
  int in_icp_ = 0;
  bool started_ = false;
  int device_ = 0;
  hipStream_t stream_ = nullptr;
  IcpProblem* d_probs_ = nullptr;
  IcpProblem* h_probs_ = nullptr;
  IcpState* h_state_ = nullptr;
};

// set by pcs_icp_share_launches(): the calling thread's ICP calls go through the shared launcher
static thread_local bool t_share_launches = false;
static thread_local int t_share_group = 0;
static thread_local double t_share_wait_s = 0.0;  // time this thread spent inside IcpBatcher::run
static thread_local double t_icp_s = 0.0;         // time this thread spent inside the ICP entry point

}  // namespace pcs

using namespace pcs;

extern "C" {

int pcs_minmax4_f32(const float* pts, int64_t n, float* out8, hipStream_t s);

// bench.py support: time icp_nn_kernel with HIP events on the launch stream (any number of host
// threads / streams; totals are accumulated under a mutex).
__global__ void busy_kernel(int ticks) {  // ~ticks x 10 ns on the 100 MHz wall clock
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}
int pcs_profile_busy_loop(int64_t n, int ticks, int blocks, hipStream_t s) {
  for (int64_t i = 0; i < n; ++i) busy_kernel<<<blocks, 64, 0, s>>>(ticks);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

__global__ void touch_kernel(float4* buf, int mode, int n, int flip);
// the same chain of dependent kernels as one captured graph launched n / nodes times: does the graph
// executor shorten the device-side gap between dependent kernels? (tools/launch_scaling.py, mode graphN)
int pcs_profile_graph_loop(int64_t n, int nodes, int ticks, int blocks, hipStream_t s) {
  if (nodes < 1) nodes = 1;
  float4* tbuf = nullptr;
  if (ticks < 0) {  // ticks < 0: the nodes are touch kernels (4096 threads load one row each)
    PCS_CHECK_HIP(hipMalloc(&tbuf, 4096 * sizeof(float4)));
    PCS_CHECK_HIP(hipMemsetAsync(tbuf, 0, 4096 * sizeof(float4), s));
    PCS_CHECK_HIP(hipStreamSynchronize(s));
  }
  if (nodes == 1 && ticks < 0) {  // no graph: the same chain as plain stream launches
    for (int64_t i = 0; i < n; ++i) touch_kernel<<<16, 256, 0, s>>>(tbuf, 0, 4096, 0);
    PCS_CHECK_HIP(hipStreamSynchronize(s));
    hipFree(tbuf);
    return PCS_OK;
  }
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  PCS_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nodes; ++i) {
    if (ticks > 0) busy_kernel<<<blocks, 64, 0, s>>>(ticks);
    else if (ticks < 0) touch_kernel<<<16, 256, 0, s>>>(tbuf, 0, 4096, 0);
    else pcs::noop_kernel<<<1, 64, 0, s>>>();
  }
  PCS_CHECK_HIP(hipStreamEndCapture(s, &graph));
  PCS_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for (int64_t i = 0; i < n; i += nodes) PCS_CHECK_HIP(hipGraphLaunch(exec, s));
  PCS_CHECK_HIP(hipStreamSynchronize(s));
  hipGraphExecDestroy(exec);
  hipGraphDestroy(graph);
  if (tbuf) hipFree(tbuf);
  return PCS_OK;
}

// a stream whose kernels run on the compute units of `mask` only (hipExtStreamCreateWithCUMask):
// probe for confining each tracking unit to its own XCDs (tools/icp_concurrency.py --cu-mask)
int pcs_stream_create_cu_mask(const uint32_t* mask, int words, void** out) {
  if (!mask || !out || words < 1) return PCS_ERR_ARG;
  hipStream_t s = nullptr;
  PCS_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
  *out = (void*)s;
  return PCS_OK;
}

// What does a kernel that merely touches memory cost, begin to end, in a dependent chain? mode 0: every
// thread loads one 16-B row and exits; 1: stores one; 2: kernels alternate store / load of the same
// buffer; 3: loads a row, then a second one whose address depends on the first (two round trips).
// Returns the mean begin-to-end time of one launch in microseconds (events on the dispatch).
__global__ void touch_kernel(float4* buf, int mode, int n, int flip) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 1 || (mode == 2 && !flip)) {
    buf[i] = make_float4((float)i, 0.f, 0.f, (float)flip);
    return;
  }
  float4 v = buf[i];
  if (mode == 3) {
    int j = ((int)v.x + i * 7919) % n;
    if (j < 0) j = -j;
    v = buf[j];
  }
  if (v.y == 123.456f) buf[i].z = 1.f;  // never true: keeps the loads alive
}
double pcs_profile_touch_loop(int64_t n_launch, int mode, int rows, hipStream_t s) {
  float4* buf = nullptr;
  if (rows < 256) rows = 256;
  if (hipMalloc(&buf, (size_t)rows * sizeof(float4)) != hipSuccess) return -1.0;
  hipMemsetAsync(buf, 0, (size_t)rows * sizeof(float4), s);
  const int reps = 64;
  hipEvent_t ev[2 * reps];
  for (int i = 0; i < 2 * reps; ++i) hipEventCreate(&ev[i]);
  int blocks = (rows + 255) / 256;
  double total = 0.0;
  int64_t timed = 0;
  for (int64_t done = 0; done < n_launch; done += reps) {
    for (int r = 0; r < reps; ++r)
      hipExtLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(256), 0, s, ev[2 * r], ev[2 * r + 1], 0, buf, mode,
                            rows, r & 1);
    hipStreamSynchronize(s);
    for (int r = 0; r < reps; ++r) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]) == hipSuccess) { total += ms; ++timed; }
    }
  }
  for (int i = 0; i < 2 * reps; ++i) hipEventDestroy(ev[i]);
  hipFree(buf);
  return timed ? total * 1e3 / (double)timed : -1.0;
}

int pcs_profile_launch_loop(int64_t n, hipStream_t s) {
  for (int64_t i = 0; i < n; ++i) pcs::noop_kernel<<<1, 64, 0, s>>>();
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// n x {8 no-op launches, one small device->host read, stream sync}; mode 0: pageable destination
// (hipMemcpyAsync), mode 1: pinned destination, mode 2: pinned word written by the kernel itself
__global__ void probe_write_kernel(int* dst, int v) { *dst = v; }
int pcs_profile_sync_loop(int64_t n, int mode, hipStream_t s) {
  int* dev = nullptr;
  int* pinned = nullptr;
  PCS_CHECK_HIP(hipMalloc(&dev, 64));
  PCS_CHECK_HIP(hipHostMalloc(&pinned, 64));
  int host[16];
  for (int64_t i = 0; i < n; ++i) {
    for (int j = 0; j < 8; ++j) pcs::noop_kernel<<<1, 64, 0, s>>>();
    if (mode == 0) {
      PCS_CHECK_HIP(hipMemcpyAsync(host, dev, 56, hipMemcpyDeviceToHost, s));
    } else if (mode == 1) {
      PCS_CHECK_HIP(hipMemcpyAsync(pinned, dev, 56, hipMemcpyDeviceToHost, s));
    } else {
      probe_write_kernel<<<1, 1, 0, s>>>(pinned, (int)i);
    }
    PCS_CHECK_HIP(hipStreamSynchronize(s));
  }
  hipFree(dev);
  hipHostFree(pinned);
  return PCS_OK;
}

// the calling thread's pcs_register_to_next_frame calls use (1) / stop using (0) the shared launcher
int pcs_icp_share_launches(int enable) {
  // enable = 0: off; enable = g >= 1: share launches with the other threads of group g
  if (t_share_launches) IcpBatcher::get(t_share_group).remove_participant();
  t_share_launches = enable != 0;
  t_share_group = enable;
  if (t_share_launches) IcpBatcher::get(t_share_group).add_participant();
  return PCS_OK;
}

// diagnostic: {batches, problems, seconds launching + waiting for batches, seconds idle} of the
// shared launcher since the last call (reset on read)
// diagnostic: seconds the calling thread has spent waiting in the shared launcher / inside
// pcs_register_to_next_frame since the last call (reset on read)
int pcs_profile_thread_icp_read(double* out2) {
  if (!out2) return PCS_ERR_ARG;
  out2[0] = t_share_wait_s;
  out2[1] = t_icp_s;
  t_share_wait_s = 0.0;
  t_icp_s = 0.0;
  return PCS_OK;
}

int pcs_profile_batcher_read(double* out4) {
  if (!out4) return PCS_ERR_ARG;
  for (int i = 0; i < 4; ++i) { out4[i] = g_batch_stats[i]; g_batch_stats[i] = 0; }
  return PCS_OK;
}

int pcs_profile_icp_nn(int enable) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = enable != 0;
  g_prof_total = NnProfile();
  return PCS_OK;
}

// out3 (host): {milliseconds, launches, algorithmic bytes} accumulated since the last enable.
int pcs_profile_icp_nn_read(double* out3) {
  if (!out3) return PCS_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  out3[0] = g_prof_total.ms;
  out3[1] = (double)g_prof_total.launches;
  out3[2] = g_prof_total.bytes;
  return PCS_OK;
}

size_t pcs_register_to_next_frame_workspace_bytes(int64_t Mn, int64_t Rn, int64_t Ra, int64_t C) {
  Arena a(nullptr, 0);
  IcpBufs b;
  plan_icp(a, Mn, Rn, Ra, C, b);
  return a.off;
}

// BLOCKING (the one entry point that synchronises the stream: the reference reads the loss on the
// host every iteration, this reads one word per `chunk` iterations).
//   mov_fxyz [Mn,4]  non-stationary moving points, updated in place
//   mov_comp [Mn]    int32 component of each moving point, in [0, C)
//   ref_fxyz [Rn,4]  non-stationary reference points; ref_all [Ra,4] all reference points
//   comp_deg [C]     int64 number of moving points (stationary included) per component
//   T [C,16] f64 out; l1_err [C] f64 out (last iteration); edge_ratio [C] f32 out
//   info[4] host out: iterations run, stop reason (1 = countdown, 0 = max_iter), 0, 0
int pcs_register_to_next_frame(float* mov_fxyz, const int32_t* mov_comp, int64_t Mn,
                               const float* ref_fxyz, int64_t Rn, const float* ref_all, int64_t Ra,
                               const int64_t* comp_deg, int64_t C, double radius, int frame_offset,
                               double angle_regularizer, int max_iter, double stopping_delta,
                               float cell_size, int chunk, double* T, double* l1_err,
                               float* edge_ratio, int32_t* info, void* ws, size_t ws_bytes,
                               hipStream_t s) {
  if (!T || !l1_err || !edge_ratio || !ws || !info || C < 1 || Mn < 0 || Rn < 0 || Ra < 0 ||
      !comp_deg || !(radius > 0) || max_iter < 0 || (Mn > 0 && (!mov_fxyz || !mov_comp)) ||
      (Rn > 0 && !ref_fxyz) || (Ra > 0 && !ref_all))
    return PCS_ERR_ARG;
  if (Mn + Rn >= (1LL << 26)) return PCS_ERR_RANGE;
  struct CallTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~CallTimer() { t_icp_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  } call_timer;
  Arena a(ws, ws_bytes);
  IcpBufs b;
  if (!plan_icp(a, Mn, Rn, Ra, C, b)) return PCS_ERR_WORKSPACE;
  // graph.radius = (radius^2 + frame_offset^2) ** 0.5, handed to the kernel as fp32
  float rad = (float)sqrt(radius * radius + (double)frame_offset * (double)frame_offset);
  // fine cell = (largest admissible spatial distance) / 4, so that the 3x3x3 coarse blocks cover
  // the search ball: |dxyz|^2 <= rad^2 - df^2 + margin with df = frame_offset for integer frame
  // values (the kernel flags PCS_ERR_RANGE if the data violate that). cell_size may enlarge it.
  double rad2 = (double)rad * (double)rad;
  double budget = rad2 - (double)frame_offset * (double)frame_offset + (2e-5 * rad2 + 2e-6);
  if (budget < radius * radius) budget = radius * radius;
  float cell = (float)(sqrt(budget) * 0.25 * 1.0005);
  if (cell_size > cell) cell = cell_size;
  float pad = 4.f * (float)radius + 16.f;
  if (chunk < 1) chunk = 8;

  // one launch resets everything the call owns, one finds the three bounding boxes
  {
    int64_t words = (int64_t)(b.gr.slot_words + b.ga.slot_words + 2 * b.gm[0].slot_words) + C * kAcc;
    icp_clear_kernel<<<grid_for(words), kBlock, 0, s>>>(b.b_ref, b.Tb, b.acc, b.first, b.cnt, C, b.gr,
                                                        b.gm[0], b.gm[1], b.ga);
    int64_t nmax = Mn > Rn ? Mn : Rn;
    if (Ra > nmax) nmax = Ra;
    int gb = (int)((nmax + kBlock - 1) / kBlock);
    if (gb < 1) gb = 1;
    if (gb > 256) gb = 256;
    icp_bounds_kernel<<<dim3(gb, 3), kBlock, 0, s>>>((const float4*)ref_fxyz, (int)Rn,
                                                     (const float4*)mov_fxyz, (int)Mn,
                                                     (const float4*)ref_all, (int)Ra, b.b_ref, mov_comp,
                                                     b.first);
  }
  int nb_m = (int)((Mn + kBlock - 1) / kBlock);
  if (nb_m < 1) nb_m = 1;
  int nb_e = (int)((Mn + Rn + kBlock - 1) / kBlock);
  if (nb_e < 1) nb_e = 1;
  int nb_q = (int)(((Mn + Rn) * 32 + kBlock - 1) / kBlock);  // 32 lanes per query
  if (nb_q < 1) nb_q = 1;

  // this call's problem descriptor
  IcpRequest req;
  {
    IcpProblem& pr = req.prob;
    pr.mov = (float4*)mov_fxyz;
    pr.prev = (float4*)b.prev;
    pr.ref = (const float4*)ref_fxyz;
    pr.comp = mov_comp;
    pr.first = b.first;
    pr.nn_f = b.nn_f;
    pr.nn_b = b.nn_b;
    pr.acc = b.acc;
    pr.Tb = b.Tb;
    pr.lossp = b.lossp;
    pr.st = b.state;
    pr.grid = b.grid;
    pr.gr = b.gr;
    pr.gm[0] = b.gm[0];
    pr.gm[1] = b.gm[1];
    pr.C = C;
    pr.reg = angle_regularizer;
    pr.delta = stopping_delta;
    pr.Mn = (int)Mn;
    pr.Rn = (int)Rn;
    pr.df = frame_offset;
    pr.nparts = nb_m;
    pr.k0 = 1;
    pr.n_it = 0;
    {
      static const int probe = [] { const char* e = getenv("PCS_NN_PROBE_STAGE"); return e ? atoi(e) : 0; }();
      pr.probe_stage = probe;
    }
    req.nb_q = nb_q;
    req.nb_e = nb_e;
    req.nb_m = nb_m;
    // per launch: read every query point and every hashed point (16 B rows), write one int32
    // neighbour per query (16 N + 16 M per graph, two graphs per iteration)
    req.bytes_per_launch = 2.0 * 16.0 * (double)(Mn + Rn) + 4.0 * (double)(Mn + Rn);
  }
  const bool shared = t_share_launches;
  // grid description, then the three grids in one launch
  icp_setup_kernel<<<1, 64, 0, s>>>(b.b_ref, b.b_mov, b.b_all, rad, cell, pad, b.grid, b.state);
  {
    int64_t nmax = Mn > Rn ? Mn : Rn;
    if (Ra > nmax) nmax = Ra;
    if (nmax > 0)
      icp_insert3_kernel<<<dim3((unsigned)((nmax + kBlock - 1) / kBlock), 3), kBlock, 0, s>>>(
          (const float4*)ref_fxyz, (int)Rn, (const float4*)ref_all, (int)Ra, (const float4*)mov_fxyz,
          (int)Mn, b.grid, b.gr, b.ga, b.gm[1], b.state);
  }
  // status words come back through pinned memory written by the kernels (no device-to-host copy)
  IcpState* pin = (IcpState*)thread_pinned_page();
  struct IcpPhase {  // tells the shared launcher that this thread is about to submit chunks
    bool on;
    explicit IcpPhase(bool o) : on(o) { if (on) IcpBatcher::get(t_share_group).enter_icp(); }
    void end() { if (on) { IcpBatcher::get(t_share_group).leave_icp(); on = false; } }
    ~IcpPhase() { end(); }
  } phase(shared);
  bool prof = g_prof_on && !shared;
  if (prof && !t_ev.have) {
    for (int i = 0; i < 2 * 96; ++i) PCS_CHECK_HIP(hipEventCreate(&t_ev.ev[i]));
    t_ev.have = true;
  }
  IcpState host;
  host.stop_iter = INT_MAX;
  host.iters = 0;
  host.err = 0;
  int it = 0;
  while (it < max_iter) {
    int n_it = (max_iter - it < chunk) ? (max_iter - it) : chunk;
    if (shared) {
      // everything this call enqueued so far must be complete before the shared stream touches it
      PCS_CHECK_HIP(hipStreamSynchronize(s));
      req.prob.k0 = it + 1;
      req.prob.n_it = n_it;
      auto tw = std::chrono::steady_clock::now();
      int rc2 = IcpBatcher::get(t_share_group).run(&req);
      t_share_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
      if (rc2) return rc2;
      host = req.result;
    } else {
      // the descriptor travels by value (k0 = 1; the chunk's first iteration is passed as kbase)
      req.prob.n_it = n_it;
      bool pk = prof && it + n_it <= kProfIters;
      int rc2 = icp_launch_chunk(nullptr, &req.prob, 1, n_it, it, nb_q, nb_e, nb_m, pin, s,
                                 pk ? &t_ev.ev[2 * it] : nullptr);
      if (rc2) return rc2;
      if (!pin) PCS_CHECK_HIP(hipMemcpyAsync(&host, b.state, sizeof(IcpState), hipMemcpyDeviceToHost, s));
      PCS_CHECK_HIP(hipStreamSynchronize(s));
      if (pin) host = *pin;
    }
    it += n_it;
    if (host.err) return PCS_ERR_RANGE;
    if (host.stop_iter != INT_MAX) break;
  }
  phase.end();
  if (prof) {
    int n_done = host.iters < kProfIters ? host.iters : kProfIters;
    double ms_sum = 0.0;
    for (int k = 0; k < n_done; ++k) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, t_ev.ev[2 * k], t_ev.ev[2 * k + 1]) == hipSuccess) ms_sum += ms;
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_total.ms += ms_sum;
    g_prof_total.launches += n_done;
    g_prof_total.bytes += (double)n_done * req.bytes_per_launch;
  }
  info[0] = host.iters;
  info[1] = host.stop_iter != INT_MAX ? 1 : 0;
  info[2] = info[3] = 0;
  double* acc_fin = b.acc + (size_t)(host.iters & 1) * C * kAcc;  // moments of the last iteration
  const int32_t* nnf_fin = b.nn_f + (size_t)(host.iters & 1) * (Mn > 0 ? Mn : 1);
  const int32_t* nnb_fin = b.nn_b + (size_t)(host.iters & 1) * (Rn > 0 ? Rn : 1);
  const double* T_fin = b.Tb + (size_t)(host.iters & 1) * C * 16;
  if (host.iters > 0) {
    // l1 error of the last iteration: positions before its update are in `prev`
    icp_l1_kernel<0><<<nb_e, kBlock, 0, s>>>((const float4*)b.prev, (int)Mn, (const float4*)ref_fxyz,
                                             (int)Rn, mov_comp, nnf_fin, nnb_fin, acc_fin, 0.3);
    icp_l1_kernel<1><<<nb_e, kBlock, 0, s>>>((const float4*)b.prev, (int)Mn, (const float4*)ref_fxyz,
                                             (int)Rn, mov_comp, nnf_fin, nnb_fin, acc_fin, 0.3);
  }
  if (Mn > 0 && Ra > 0)
    icp_coverage_kernel<<<(int)((Mn * 32 + kBlock - 1) / kBlock), kBlock, 0, s>>>(
        (const float4*)mov_fxyz, (int)Mn, mov_comp, b.grid, frame_offset, b.ga, b.state, b.cnt);
  // NB: the coverage kernel may still raise st->err; it runs before this launch on the same stream
  icp_finish_kernel<<<grid_for(C), kBlock, 0, s>>>(C, acc_fin, b.cnt, comp_deg, host.iters > 0, T_fin, T,
                                                   l1_err, edge_ratio, b.state, pin);
  PCS_LAUNCH_CHECK();
  if (!pin) PCS_CHECK_HIP(hipMemcpyAsync(&host, b.state, sizeof(IcpState), hipMemcpyDeviceToHost, s));
  PCS_CHECK_HIP(hipStreamSynchronize(s));
  if (pin) host = *pin;
  if (host.err) return PCS_ERR_RANGE;
  return PCS_OK;
}

}  // extern "C"
