// Fused per-component ICP ("TLS solve") for gfx950, batched.
//
// Reference: register_to_next_frame, pcdet/models/registration/preprocessors/
// registration_utils.py:83-206. Per iteration the reference runs two radius-graph
// builds (4 kernels + scan + 2 host syncs + 4 allocations each), two argsorts, a handful of
// scatter/segment kernels, a batched cuSOLVER 3x3 SVD and a blocking loss.item().
//
// Here a call is a descriptor in a device table (pcs_icp.hpp) and iteration k is THREE launches that
// serve every problem of the batch (blockIdx.y = batch entry):
//   nn         1-NN both directions, 32 lanes per query (two-level grid, pcs_grid2.hpp); workgroup
//              0 of each entry first settles the stop rule of iteration k-1
//   moments    ONE pass over the matched pairs: n, sum p, sum q, sum p(x)q, sum |p-q|^2 per
//              component as 64-bit fixed-point integers (LDS table per workgroup, then one integer
//              atomic per (workgroup, component, accumulator)): order-free, hence bit-reproducible;
//              the centred covariance and the loss follow algebraically, so the reference's second
//              pass over the edges is not needed
//   apply      every moving point solves its own component's regularised Kabsch step from the
//              completed moments (cov + reg*T_rot -> rotation by Newton's polar iteration, Jacobi SVD
//              when det <= 0; R = V diag(1,1,det(VU^T)) U^T, t = c_ref - R c_mov, T <- T_i T) -- a
//              3x3 solve per point is cheaper than another launch --, moves (f64 -> fp32 as the
//              reference stores them), the component's first point stores T and the loss term; the
//              moved points go straight into the grid of iteration k+1 (double-buffered) and the grid
//              just searched is reset
// Every kernel of iteration k returns immediately when k > state.stop_iter, so the host enqueues
// iterations in chunks and reads one word per chunk (through pinned memory) instead of one loss per
// iteration. All kernels are grid-stride: the launch geometry never changes a result.
#include "pcs_icp.hpp"

#include <hip/hip_ext.h>

#include <mutex>

namespace pcs {

bool plan_icp(Arena& a, int64_t Mn, int64_t Rn, int64_t Ra, int64_t C, IcpBufs& b) {
  if (Mn < 1) Mn = 1;
  if (Rn < 1) Rn = 1;
  if (Ra < 1) Ra = 1;
  if (C < 1) C = 1;
  b.geom = a.take<IcpGeom>(1);
  b.bounds = a.take<float>(24);  // ref, mov, all: 8 floats each
  b.state = a.take<IcpState>(1);
  plan_grid2(a, Rn, b.gr);
  plan_grid2(a, Mn, b.gm[0]);
  plan_grid2(a, Mn, b.gm[1]);
  plan_grid2(a, Ra, b.ga);
  b.nn_f = a.take<int32_t>(2 * Mn);
  b.nn_b = a.take<int32_t>(2 * Rn);
  b.bud_f = a.take<float>(2 * Mn);
  b.bud_b = a.take<float>(2 * Rn);
  b.wl = a.take<int32_t>(Mn + Rn);
  b.wls = a.take<int32_t>(Mn + Rn);
  b.sstep = a.take<unsigned int>(2 * kStepH);
  b.prev = a.take<float>(Mn * 4);
  b.acc = a.take<long long>(2 * C * kAcc);
  b.Tb = a.take<double>(2 * C * 16);
  b.Rt = a.take<double>(C * 12);
  b.lossfx = a.take<long long>(4 * kLossParts);  // [2][kLossParts] fine sums, then [2][kLossParts] coarse check sums
  b.corg = a.take<float4>(C);
  b.first = a.take<int32_t>(C);
  b.cov = a.take<int32_t>(C);
  return a.ok();
}

int icp_describe(const IcpArgs& a, void* ws, size_t ws_bytes, IcpProblem* out) {
  if (!out || !ws || !a.T || !a.l1 || !a.ratio || a.C < 1 || a.Mn < 0 || a.Rn < 0 || a.Ra < 0 ||
      !a.comp_deg || !(a.radius > 0) || a.max_iter < 0 || (a.Mn > 0 && (!a.mov || !a.comp)) ||
      (a.Rn > 0 && !a.ref) || (a.Ra > 0 && !a.all))
    return PCS_ERR_ARG;
  if (a.Mn + a.Rn >= (1LL << 26)) return PCS_ERR_RANGE;
  Arena ar(ws, ws_bytes);
  IcpBufs b;
  if (!plan_icp(ar, a.Mn, a.Rn, a.Ra, a.C, b)) return PCS_ERR_WORKSPACE;
  IcpProblem& pr = *out;
  // graph.radius = (radius^2 + frame_offset^2) ** 0.5, handed to the kernel as fp32
  const double fo = (double)a.frame_offset;
  pr.rad = (float)sqrt(a.radius * a.radius + fo * fo);
  // fine cell = (largest admissible spatial distance) / 4, so that the 3x3x3 coarse blocks cover
  // the search ball: |dxyz|^2 <= rad^2 - df^2 + margin with df = frame_offset for integer frame
  // values (the kernel flags PCS_ERR_RANGE if the data violate that). cell_size may enlarge it.
  // every search is widened by mu (1/16 of the spatial radius) so that it can say how long its
  // answer stays valid; the cells must admit the widened ball
  pr.mu = (float)(a.radius / 16.0);
  double rad2 = (double)pr.rad * (double)pr.rad;
  double budget = (a.radius + pr.mu) * (a.radius + pr.mu) + (2e-5 * rad2 + 2e-6);
  if (budget < a.radius * a.radius) budget = a.radius * a.radius;
  pr.cell = (float)(sqrt(budget) * 0.25 * 1.0005);
  if (a.cell_size > pr.cell) pr.cell = a.cell_size;
  pr.pad = 4.f * (float)a.radius + 16.f;
  pr.mov = (float4*)a.mov;
  pr.prev = (float4*)b.prev;
  pr.ref = (const float4*)a.ref;
  pr.all = (const float4*)a.all;
  pr.comp = a.comp;
  pr.comp_deg = a.comp_deg;
  pr.cnt3 = a.cnt3;
  pr.first = b.first;
  pr.nn_f = b.nn_f;
  pr.nn_b = b.nn_b;
  pr.bud_f = b.bud_f;
  pr.bud_b = b.bud_b;
  pr.wl = b.wl;
  pr.wls = b.wls;
  pr.sstep = b.sstep;
  pr.acc = b.acc;
  pr.Tb = b.Tb;
  pr.Rt = b.Rt;
  pr.lossfx = b.lossfx;
  pr.corg = b.corg;
  pr.cov = b.cov;
  pr.st = b.state;
  pr.geom = b.geom;
  pr.bounds = b.bounds;
  pr.gr = b.gr;
  pr.ga = b.ga;
  pr.gm[0] = b.gm[0];
  pr.gm[1] = b.gm[1];
  pr.T_out = a.T;
  pr.l1_out = a.l1;
  pr.ratio_out = a.ratio;
  pr.C = a.C;
  pr.reg = a.reg;
  pr.delta = a.delta;
  pr.Mn = (int)a.Mn;
  pr.Rn = (int)a.Rn;
  pr.Ra = (int)a.Ra;
  pr.capM = (int)(a.Mn > 0 ? a.Mn : 1);
  pr.capR = (int)(a.Rn > 0 ? a.Rn : 1);
  pr.df = a.frame_offset;
  pr.max_iter = a.max_iter;
  return PCS_OK;
}

__device__ __forceinline__ long long to_fx(double v) { return __double2ll_rn(v * kFx); }
#ifdef PCS_WG_TRACE
__device__ unsigned long long* g_wgt = nullptr;
__device__ unsigned int g_wgt_n = 0;
constexpr unsigned kWgtCap = 6u << 20;
#define WGT_BEGIN() const unsigned long long wgt0 = __builtin_amdgcn_s_memrealtime()
#define WGT_MARK(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define WGT_END(kid, kk, m1, m2)                                                                       \
  do {                                                                                                 \
    __syncthreads();                                                                                   \
    if (threadIdx.x == 0 && g_wgt) {                                                                   \
      unsigned i_ = atomicAdd(&g_wgt_n, 1u);                                                           \
      if (i_ < kWgtCap) {                                                                              \
        g_wgt[5 * (size_t)i_] = ((unsigned long long)(kid) << 56) | ((unsigned long long)(kk) << 40) | \
                                ((unsigned long long)blockIdx.y << 32) | ((unsigned long long)blockIdx.z << 28) | blockIdx.x; \
        g_wgt[5 * (size_t)i_ + 1] = wgt0;                                                              \
        g_wgt[5 * (size_t)i_ + 2] = __builtin_amdgcn_s_memrealtime();                                  \
        g_wgt[5 * (size_t)i_ + 3] = (m1);                                                              \
        g_wgt[5 * (size_t)i_ + 4] = (m2);                                                              \
      }                                                                                                \
    }                                                                                                  \
  } while (0)
#else
#define WGT_BEGIN()
#define WGT_MARK(var)
#define WGT_END(kid, kk, m1, m2)
#endif

// start of a call: bounds <- empty, T <- identity, accumulators and counters <- 0, state <- initial,
// the four grids <- empty (one launch instead of a dozen memsets)
__global__ void __launch_bounds__(kBlock)
icp_clear_kernel(const IcpProblem* __restrict__ tab, Batch bt) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  const int64_t C = pr.C;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (i < 24) pr.bounds[i] = ((i & 7) < 4) ? 1e30f : -1e30f;
  if (i == 0) {
    IcpState* st = pr.st;
    st->last_error = 1e10;
    st->loss = 0.0;
    st->countdown = 3;
    st->stop_iter = INT_MAX;
    st->iters = 0;
    st->err = 0;
    st->decided = 0;
    st->wl_n[0] = st->wl_n[1] = 0;
    st->ws_n[0] = st->ws_n[1] = 0;
    st->dmax[0] = st->dmax[1] = 0u;
  }
  for (int64_t q = i; q < 2 * C * 16; q += stride) {
    int e = (int)(q & 15);
    pr.Tb[q] = (e == 0 || e == 5 || e == 10 || e == 15) ? 1.0 : 0.0;
  }
  for (int64_t q = i; q < 2 * C * kAcc; q += stride) pr.acc[q] = 0;
  for (int64_t q = i; q < 2 * kStepH; q += stride) pr.sstep[q] = 0u;
  for (int64_t q = i; q < C; q += stride) { pr.cov[q] = 0; pr.first[q] = INT_MAX; }
  if (i < 4 * kLossParts) pr.lossfx[i] = 0;
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  g2_clear(g2_view(pr.gr, Rn), i, stride);
  g2_clear(g2_view(pr.gm[0], Mn), i, stride);
  g2_clear(g2_view(pr.gm[1], Mn), i, stride);
  g2_clear(g2_view(pr.ga, Ra), i, stride);
}

// min / max of the three clouds: blockIdx.z selects the cloud; also the owner (smallest index) of
// every component among the moving points
__global__ void __launch_bounds__(kBlock)
icp_bounds_kernel(const IcpProblem* __restrict__ tab, Batch bt) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  const int which = blockIdx.z;
  const float4* pts = which == 0 ? pr.ref : (which == 1 ? (const float4*)pr.mov : pr.all);
  const int n = which == 0 ? Rn : (which == 1 ? Mn : Ra);
  if ((int64_t)blockIdx.x * kBlock >= n) return;
  float* out8 = pr.bounds + 8 * which;
  float mn[4] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
  float mx[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 p = pts[i];
    if (which == 1) atomicMin(&pr.first[pr.comp[i]], i);  // the thread that will own the component's T
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
  if (threadIdx.x < 4) {
    int d = threadIdx.x;
    float a = smn[0][d], b = smx[0][d];
    for (int k = 1; k < kBlock / 64; ++k) { a = fminf(a, smn[k][d]); b = fmaxf(b, smx[k][d]); }
    atomic_min_f32(&out8[d], a);
    atomic_max_f32(&out8[4 + d], b);
  }
}

// grid origin padded so that points moved by the registration stay inside the 17-bit cell range;
// origin of every component's fixed-point moments = its first moving point as it arrives (any fixed
// point near the component would do: the moments are translation-invariant where they are used, the
// origin only keeps the integers small)
__global__ void __launch_bounds__(kBlock)
icp_setup_kernel(const IcpProblem* __restrict__ tab, Batch bt) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < pr.C; c += (int64_t)gridDim.x * blockDim.x) {
    int f = pr.first[c];
    pr.corg[c] = f != INT_MAX ? pr.mov[f] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const float* b_ref = pr.bounds;
  const float* b_mov = pr.bounds + 8;
  const float* b_all = pr.bounds + 16;
  GridDesc* g = &pr.geom->g;
  float mn[4];
  for (int d = 0; d < 4; ++d) mn[d] = fminf(fminf(b_ref[d], b_mov[d]), b_all[d]);
  float vs0 = (float)(1 - 1e-3);
  g->vs0 = vs0;
  g->lo0 = mn[0] - vs0 * 2.f;
  for (int d = 0; d < 3; ++d) g->lo[d] = mn[1 + d] - pr.pad;
  g->h = pr.cell;
  g->radius = pr.rad;
  g->radius2 = pr.rad * pr.rad;
  g->rings = (int)ceilf(pr.rad / pr.cell);
}

// the three grids of a call in one launch: blockIdx.z = 0 non-stationary reference, 1 all reference,
// 2 the moving points as they arrive (the grid of iteration 1)
__global__ void __launch_bounds__(kBlock)
icp_insert3_kernel(const IcpProblem* __restrict__ tab, Batch bt) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  const int which = blockIdx.z;
  const float4* pts = which == 0 ? pr.ref : (which == 1 ? pr.all : (const float4*)pr.mov);
  const int n = which == 0 ? Rn : (which == 1 ? Ra : Mn);
  if ((int64_t)blockIdx.x * kBlock >= n) return;
  const Grid2 g = g2_view(which == 0 ? pr.gr : (which == 1 ? pr.ga : pr.gm[1]), n);
  GridDesc d = pr.geom->g;
  const int nround = (n + kBlock - 1) / kBlock;
  for (int r = blockIdx.x; r < nround; r += gridDim.x) {  // whole workgroups stay together (shuffles)
    int i = r * kBlock + threadIdx.x;
    bool in = i < n;
    float4 p = in ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    g2_insert_point(d, g, in, i, p, &pr.st->err);
  }
}

// loss-delta / 3-strike stop rule of iteration k_done (registration_utils.py:186-193), evaluated by
// ONE wave from the loss of the apply launch: the components' terms arrive as 64-bit fixed-point
// partial sums (kLossParts addresses; integer adds commute, so the total does not depend on any
// launch geometry or arrival order). Summing the [C] terms here instead -- ten dependent rounds of
// loads for the keys with ten thousand components -- made every budget-test launch of those keys
// 9 us longer than the others' (20.6 against 11.5 us). It may lower stop_iter to k_done, which
// still admits every kernel of iteration k_done.
__device__ __forceinline__ void icp_decide(const IcpProblem& pr, int k_done) {
  IcpState* st = pr.st;
  if (k_done < 1 || st->decided >= k_done || st->stop_iter != INT_MAX) return;  // uniform per workgroup
  if (threadIdx.x >= 64) return;
  long long* part = pr.lossfx + (size_t)(k_done & 1) * kLossParts;
  long long* chk = pr.lossfx + (size_t)(2 + (k_done & 1)) * kLossParts;
  long long v = 0, vc = 0;
  if (threadIdx.x < kLossParts) {
    v = part[threadIdx.x];
    vc = chk[threadIdx.x];
    part[threadIdx.x] = 0;  // (the solve launch of iteration k_done + 2 adds here next)
    chk[threadIdx.x] = 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    v += __shfl_xor(v, o, 64);
    vc += __shfl_xor(vc, o, 64);
  }
  if (threadIdx.x == 0) {
    // every component's term is within the fixed-point range, their SUM over many large components need
    // not be (2^27 in total at 36 fractional bits): the same terms also add up at 10 fractional bits
    // (range 2^53), and a fine sum that wrapped disagrees with the coarse one by ~2^28
    if (fabs((double)v * kFxInv - (double)vc * (1.0 / 1024.0)) > 1.0 + (double)pr.C * (1.0 / 1024.0)) st->err = 1;
    const double loss = (double)v * kFxInv;
    st->loss = loss;
    if (st->last_error - loss < pr.delta) st->countdown -= 1;
    else st->countdown = 3;
    if (st->countdown <= 0) st->stop_iter = k_done;
    else st->last_error = loss;
    st->decided = k_done;
  }
}

// end of a chunk: the stop rule of every entry's last iteration (the others were settled by the
// next search launch), and the state mirrored where the host reads it (`out` may be pinned host
// memory: the host then needs no copy, only the stream synchronisation)
__global__ void __launch_bounds__(kBlock)
icp_decide_kernel(const IcpProblem* __restrict__ tab, Batch bt, BatchInts kbase, int n_it,
                  IcpState* __restrict__ out) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.x]];
  int k_last = kbase.v[blockIdx.x] + n_it;
  if (k_last > pr.max_iter) k_last = pr.max_iter;
  if (k_last > pr.st->iters) k_last = pr.st->iters;  // iterations that did not run have no loss
  icp_decide(pr, k_last);
  __syncthreads();
  if (out && threadIdx.x == 0) out[blockIdx.x] = *pr.st;
}

// ---- raw moments of the matched pairs per component (n, sum p, sum q, sum p(x)q, sum |p-q|^2),
// added by whichever kernel settles a query's neighbour: the budget test for the cached ones, the
// search kernel for the rest -- no launch of their own. Each workgroup first accumulates into a small
// LDS table keyed by component (64-bit integer LDS adds; neighbouring queries mostly share a
// component), then flushes one global integer atomic per (component, accumulator). Integer sums:
// the result does not depend on any arrival order.
constexpr int kMomSlots = 64;
constexpr int kMomSlotsSearch = 16;  // a search item settles 8 or 32 pairs: a small table keeps its reset and drain short
struct MomTable {
  int* skey;                      // LDS [nslots]
  unsigned long long* sacc;       // LDS [nslots * 17]
  unsigned long long* acc;        // global [C, kAcc] of the iteration
  int C;                          // components (keys outside [0, C) are never flushed)
  int nslots;                     // power of two <= kMomSlots
};
__device__ __forceinline__ void mom_init(const MomTable& t) {  // every thread of the workgroup; then a barrier
  for (int q = threadIdx.x; q < t.nslots; q += kBlock) t.skey[q] = -1;
  for (int q = threadIdx.x; q < t.nslots * 17; q += kBlock) t.sacc[q] = 0ULL;
}
// the pair (moving point pf of component c, reference point qf)
// the 17 fixed-point terms of one pair relative to the component's origin
__device__ __forceinline__ void mom_terms(const IcpProblem& pr, float4 pf, float4 qf, int c, double lim2,
                                          long long v[17]) {
  const float4 og = pr.corg[c];
  double p[3] = {(double)pf.y - (double)og.y, (double)pf.z - (double)og.z, (double)pf.w - (double)og.w};
  double q[3] = {(double)qf.y - (double)og.y, (double)qf.z - (double)og.z, (double)qf.w - (double)og.w};
  if (p[0] * p[0] + p[1] * p[1] + p[2] * p[2] > lim2 || q[0] * q[0] + q[1] * q[1] + q[2] * q[2] > lim2)
    pr.st->err = 1;  // a component too large for the fixed-point range
  v[0] = 1;
  double d2 = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    v[1 + a] = to_fx(p[a]);
    v[4 + a] = to_fx(q[a]);
#pragma unroll
    for (int b = 0; b < 3; ++b) v[7 + a * 3 + b] = to_fx(p[a] * q[b]);
    double e = p[a] - q[a];
    d2 += e * e;
  }
  v[16] = to_fx(d2);
}
// 17 sums of component c into the workgroup's table
__device__ __forceinline__ void mom_put(const MomTable& t, int c, const long long v[17]) {
  // claim / find the component's LDS slot (linear probing); fall back to global atomics if full
  int slot = (int)(((unsigned)c * 2654435761u) >> 26) & (t.nslots - 1);
  int found = -1;
  for (int i = 0; i < t.nslots; ++i) {
    int old = atomicCAS(&t.skey[slot], -1, c);
    if (old == -1 || old == c) { found = slot; break; }
    slot = (slot + 1) & (t.nslots - 1);
  }
  if (found >= 0) {
#pragma unroll
    for (int q2 = 0; q2 < 17; ++q2) atomicAdd(&t.sacc[found * 17 + q2], (unsigned long long)v[q2]);
  } else {
#pragma unroll
    for (int q2 = 0; q2 < 17; ++q2) atomicAdd(&t.acc[(int64_t)c * kAcc + q2], (unsigned long long)v[q2]);
  }
}
__device__ __forceinline__ void mom_add(const IcpProblem& pr, const MomTable& t, float4 pf, float4 qf, int c,
                                        double lim2) {
  long long v[17];
  mom_terms(pr, pf, qf, c, lim2, v);
  mom_put(t, c, v);
}
// Register-lean form: the component's slot first, then every term is formed and added at once (no array
// of 17 64-bit terms kept alive across shuffles). Lanes of one component meet at the same 17 LDS words;
// the adds are fire-and-forget.
__device__ __forceinline__ void mom_add_stream(const IcpProblem& pr, const MomTable& t, float4 pf, float4 qf, int c,
                                               double lim2) {
  int slot = (int)(((unsigned)c * 2654435761u) >> 26) & (t.nslots - 1);
  int found = -1;
  for (int i = 0; i < t.nslots; ++i) {
    int old = atomicCAS(&t.skey[slot], -1, c);
    if (old == -1 || old == c) { found = slot; break; }
    slot = (slot + 1) & (t.nslots - 1);
  }
  unsigned long long* base = found >= 0 ? t.sacc + found * 17 : t.acc + (int64_t)c * kAcc;
  const float4 og = pr.corg[c];
  const double p0 = (double)pf.y - (double)og.y, p1 = (double)pf.z - (double)og.z, p2 = (double)pf.w - (double)og.w;
  const double q0 = (double)qf.y - (double)og.y, q1 = (double)qf.z - (double)og.z, q2 = (double)qf.w - (double)og.w;
  if (p0 * p0 + p1 * p1 + p2 * p2 > lim2 || q0 * q0 + q1 * q1 + q2 * q2 > lim2) pr.st->err = 1;
  atomicAdd(base + 0, 1ULL);
  atomicAdd(base + 1, (unsigned long long)to_fx(p0));
  atomicAdd(base + 2, (unsigned long long)to_fx(p1));
  atomicAdd(base + 3, (unsigned long long)to_fx(p2));
  atomicAdd(base + 4, (unsigned long long)to_fx(q0));
  atomicAdd(base + 5, (unsigned long long)to_fx(q1));
  atomicAdd(base + 6, (unsigned long long)to_fx(q2));
  atomicAdd(base + 7, (unsigned long long)to_fx(p0 * q0));
  atomicAdd(base + 8, (unsigned long long)to_fx(p0 * q1));
  atomicAdd(base + 9, (unsigned long long)to_fx(p0 * q2));
  atomicAdd(base + 10, (unsigned long long)to_fx(p1 * q0));
  atomicAdd(base + 11, (unsigned long long)to_fx(p1 * q1));
  atomicAdd(base + 12, (unsigned long long)to_fx(p1 * q2));
  atomicAdd(base + 13, (unsigned long long)to_fx(p2 * q0));
  atomicAdd(base + 14, (unsigned long long)to_fx(p2 * q1));
  atomicAdd(base + 15, (unsigned long long)to_fx(p2 * q2));
  const double e0 = p0 - q0, e1 = p1 - q1, e2 = p2 - q2;
  atomicAdd(base + 16, (unsigned long long)to_fx(e0 * e0 + e1 * e1 + e2 * e2));
}
// The same for a whole wave (every lane calls; `active` lanes hold a pair). Neighbouring queries
// mostly belong to one component: when all active lanes agree on it their terms are summed by
// shuffles first and ONE lane adds the 17 totals (otherwise 64 lanes queue up on the same 17 words).
__device__ __forceinline__ void mom_add_wave(const IcpProblem& pr, const MomTable& t, bool active, float4 pf,
                                             float4 qf, int c, double lim2) {
  const unsigned long long am = __ballot(active);
  if (am == 0) return;
  const int leader = __ffsll((long long)am) - 1;
  const int c0 = __shfl(c, leader, 64);
  const bool uniform = __ballot(active && c != c0) == 0 && __popcll(am) >= 8;
  if (!uniform) {
    // Several components in the wave (the keys with thousands of small clusters: always). The points
    // come in voxel order, so a component's points sit next to each other row by row: the RUNS of equal
    // component add up by shuffles first and only a run's first lane goes to the table -- 64 lanes with
    // 17 LDS adds each on a handful of slots queued up eight deep (the budget-test launch of those keys
    // took 20.6 us against 11.5 us for the others).
    const int lane = threadIdx.x & 63;
    const int c_prev = __shfl_up(c, 1, 64);
    const bool a_prev = ((am >> (lane > 0 ? lane - 1 : 0)) & 1ULL) != 0 && lane > 0;
    const bool head = active && (!a_prev || c_prev != c);
    const unsigned long long hm = __ballot(head);
    const int rid = active ? __popcll(hm & ((2ULL << lane) - 1ULL)) : -1 - lane;  // run number (inactive: unique)
    long long v[17];
#pragma unroll
    for (int q = 0; q < 17; ++q) v[q] = 0;
    if (active) mom_terms(pr, pf, qf, c, lim2, v);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int rid_o = __shfl_down(rid, o, 64);
      const bool take = lane + o < 64 && rid_o == rid;
#pragma unroll
      for (int q = 0; q < 17; ++q) {
        const long long tq = __shfl_down(v[q], o, 64);
        if (take) v[q] += tq;
      }
    }
    if (head) mom_put(t, c, v);
    return;
  }
  const float4 og = pr.corg[c0];
  long long v[17];
#pragma unroll
  for (int q = 0; q < 17; ++q) v[q] = 0;
  if (active) {
    double p[3] = {(double)pf.y - (double)og.y, (double)pf.z - (double)og.z, (double)pf.w - (double)og.w};
    double q[3] = {(double)qf.y - (double)og.y, (double)qf.z - (double)og.z, (double)qf.w - (double)og.w};
    if (p[0] * p[0] + p[1] * p[1] + p[2] * p[2] > lim2 || q[0] * q[0] + q[1] * q[1] + q[2] * q[2] > lim2)
      pr.st->err = 1;  // a component too large for the fixed-point range
    v[0] = 1;
    double d2 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      v[1 + a] = to_fx(p[a]);
      v[4 + a] = to_fx(q[a]);
#pragma unroll
      for (int b = 0; b < 3; ++b) v[7 + a * 3 + b] = to_fx(p[a] * q[b]);
      double e = p[a] - q[a];
      d2 += e * e;
    }
    v[16] = to_fx(d2);
  }
#pragma unroll
  for (int q = 0; q < 17; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o, 64);
  }
  // lanes 0..16 add one total each
  const int lane = threadIdx.x & 63;
  if (lane < 17) {
    long long mine = v[0];
#pragma unroll
    for (int q = 1; q < 17; ++q)
      if (lane == q) mine = v[q];
    atomicAdd(&t.acc[(int64_t)c0 * kAcc + lane], (unsigned long long)mine);
  }
}

__device__ __forceinline__ void mom_flush(const MomTable& t) {  // after a barrier; every thread
  for (int q = threadIdx.x; q < t.nslots * 17; q += kBlock) {
    int sl = q / 17, a = q - sl * 17;
    int key = t.skey[sl];
    if (key >= 0 && key < t.C) atomicAdd(&t.acc[(int64_t)key * kAcc + a], t.sacc[q]);
  }
}

// XCD-aware placement (flag bit 2, PCS_ICP_XCD): workgroups are dealt round-robin over the eight XCDs in
// dispatch order (x fastest), each XCD with its own L2. The (x, entry) grid is re-read so that the
// workgroups that share an XCD (equal linear id mod 8) serve a CONTIGUOUS eighth of the (entry, x) space:
// an entry's clouds, grids and neighbour buffers are then fetched into one or two L2s instead of eight.
// A bijection of the grid whatever its size (the last n mod 8 workgroups keep their place); speed only.
struct WgXY { int x, y; };
__device__ __forceinline__ WgXY xcd_wg(bool on) {
  WgXY w = {(int)blockIdx.x, (int)blockIdx.y};
  if (!on) return w;
  const unsigned gx = gridDim.x, n = gx * gridDim.y, cpx = n >> 3, lin = blockIdx.y * gx + blockIdx.x;
  if (lin >= (cpx << 3)) return w;
  const unsigned swz = (lin & 7u) * cpx + (lin >> 3);
  w.y = (int)(swz / gx);
  w.x = (int)(swz - (unsigned)w.y * gx);
  return w;
}

// Queries [0,Mn): forward (moving -> reference, df = +frame_offset); queries [Mn, Mn+Rn): backward
// (reference -> moving, df = -frame_offset). The neighbours are a VERIFIED CACHE: every search also
// returns how far things may move before its answer can change (half the gap to the runner-up, or
// the distance by which the nearest point misses the radius). From iteration 2 on a query first
// subtracts what has moved since -- its own last step (forward) or the largest step of any moving
// point (backward) -- and keeps its neighbour without touching the grid while the budget lasts
// (98 % of the searches of a converging loop return the partner they returned before).
//   icp_nn_kernel         one LANE per query: the budget test; the queries that must search again
//                         are compacted into two work lists (one atomic per wave and list)
//   icp_nn_search_kernel  blockIdx.z & 1 = 0: eight lanes per query around the old partner
//                         (g2_nearest_seeded8) for the list of queries whose partner is within a
//                         cell or so; 1: the cooperative search (32 lanes per query) for the rest --
//                         and for every query in iteration 1. blockIdx.z >> 1 = the direction: a
//                         workgroup serves forward OR backward queries (one grid, one copy of the
//                         search in the code: the two-direction form spilled ~400 scalar registers).
// Exact: the same neighbours as a full search, always.
template <bool STREAM>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8)))
icp_nn_kernel(const IcpProblem* __restrict__ tab, Batch bt, BatchInts kbase, int j, int lds_steps) {
  WGT_BEGIN();
  const WgXY wg = xcd_wg((lds_steps & 4) != 0);
  const IcpProblem& pr = tab[bt.idx[wg.y]];
  const int k = kbase.v[wg.y] + 1 + j;
  IcpState* st = pr.st;
  // The gate is read ONCE per workgroup. Workgroup 0 lowers stop_iter during this very launch (below),
  // and the waves of a workgroup do not start together: with a per-thread gate some waves of a
  // workgroup could leave while the others go on with a moment table in LDS that is only partly
  // initialised -- whatever another stream's kernel left there then served as component ids in
  // mom_flush (an out-of-range atomic: the memory fault seen once in ~50 sequences with two or more
  // lock-step groups in flight, never with one).
  __shared__ int s_go;
  if (threadIdx.x == 0) s_go = (k > pr.max_iter || k > st->stop_iter) ? 0 : 1;
  __syncthreads();
  if (!s_go) return;
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  const int nq = Mn + Rn;
  const int nround = (nq + kBlock - 1) / kBlock;
  if (wg.x != 0 && wg.x >= nround) return;
  // iteration k, first launch: workgroup 0 settles the stop rule of iteration k-1 (its loss terms
  // were written by the previous launch). The other workgroups may run before the stop is known:
  // they only write this iteration's neighbour buffers, which nothing reads after a stop.
  if (wg.x == 0) icp_decide(pr, k - 1);
  WGT_MARK(wg_m1);
  if (k == 1) {  // no neighbours yet: every query is searched for
    if (lds_steps & 2) return;  // (icp_first_kernel has dealt them to the two kinds of search)
    for (int q = wg.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) pr.wl[q] = q;
    if (wg.x == 0 && threadIdx.x == 0) { st->wl_n[0] = Mn; st->wl_n[1] = Rn; }
    return;
  }
  __shared__ int skey[kMomSlots];
  __shared__ unsigned long long sacc[kMomSlots * 17];
  const MomTable mt = {skey, sacc, (unsigned long long*)pr.acc + (size_t)(k & 1) * pr.C * kAcc, (int)pr.C, kMomSlots};
  const double lim2 = icp_limit_sq(nq);
  mom_init(mt);
  __syncthreads();
  GridDesc g = pr.geom->g;
  const int lane = threadIdx.x & 63;
  int32_t* nn_f = pr.nn_f + (size_t)(k & 1) * pr.capM;
  int32_t* nn_b = pr.nn_b + (size_t)(k & 1) * pr.capR;
  float* bud_f = pr.bud_f + (size_t)(k & 1) * pr.capM;
  float* bud_b = pr.bud_b + (size_t)(k & 1) * pr.capR;
  const int32_t* __restrict__ seed_f = pr.nn_f + (size_t)((k - 1) & 1) * pr.capM;
  const int32_t* __restrict__ seed_b = pr.nn_b + (size_t)((k - 1) & 1) * pr.capR;
  const float* __restrict__ sbud_f = pr.bud_f + (size_t)((k - 1) & 1) * pr.capM;
  const float* __restrict__ sbud_b = pr.bud_b + (size_t)((k - 1) & 1) * pr.capR;
  const Grid2 g_ref = g2_view(pr.gr, Rn), g_mov = g2_view(pr.gm[k & 1], Mn);
  // steps of the moving points in iteration k-1, by the place they moved into
  const unsigned int* __restrict__ sstep = pr.sstep + (size_t)((k - 1) & 1) * kStepH;
  // A backward query looks up eight hashed slots of the 16 KB step table: eight scattered requests where
  // the rest of the query has three. A workgroup that serves backward queries (its last round reaches past
  // Mn) copies the table into LDS once -- coalesced, L2-resident: every workgroup of the unit reads it.
  __shared__ unsigned int s_step[kStepH];
  {
    const int r_last = wg.x + ((nround - 1 - wg.x) / (int)gridDim.x) * (int)gridDim.x;
    if ((lds_steps & 1) && k > 1 && (r_last + 1) * kBlock > Mn) {
      for (int q = threadIdx.x; q < kStepH / 4; q += kBlock) ((uint4*)s_step)[q] = ((const uint4*)sstep)[q];
      __syncthreads();
    }
  }
  for (int r = wg.x; r < nround; r += gridDim.x) {  // whole waves stay together (ballots below)
    const int qid = r * kBlock + threadIdx.x;
    const bool valid = qid < nq;
    const bool fwd = qid < Mn;
    bool to_seeded = false, to_coop = false;
    bool m_act = false;  // a settled pair whose moments are due: (moving point, reference point, component)
    float4 m_p = make_float4(0.f, 0.f, 0.f, 0.f), m_q = m_p;
    int m_c = 0;
    if (valid) {
      float4 p, sp = make_float4(0.f, 0.f, 0.f, 0.f);
      int s;
      float bud, disp;
      if (fwd) {
        p = pr.mov[qid];
        float4 o = pr.prev[qid];
        s = seed_f[qid];
        bud = sbud_f[qid];
        float dx = p.y - o.y, dy = p.z - o.z, dz = p.w - o.w;
        disp = sqrtf((dx * dx + dy * dy) + dz * dz);
      } else {
        p = pr.ref[qid - Mn];
        s = seed_b[qid - Mn];
        bud = sbud_b[qid - Mn];
        // a moving point that can matter for this query lies in the 3 x 3 x 3 blocks of 4^3 cells
        // around it -- 12 cells per axis, i.e. at most 2 x 2 x 2 blocks of 8^3 cells: the largest
        // step recorded there bounds how far any such point has moved since iteration k-1
        const int cx = (int)floorf((p.y - g.lo[0]) / g.h), cy = (int)floorf((p.z - g.lo[1]) / g.h),
                  cz = (int)floorf((p.w - g.lo[2]) / g.h);
        const int sx = (4 * ((cx >> 2) - 1)) >> 3, sy = (4 * ((cy >> 2) - 1)) >> 3, sz = (4 * ((cz >> 2) - 1)) >> 3;
        unsigned int m = 0u;
        if (lds_steps & 1) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
            m = max(m, s_step[icp_step_slot(sx + (q & 1), sy + ((q >> 1) & 1), sz + (q >> 2))]);
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q)
            m = max(m, sstep[icp_step_slot(sx + (q & 1), sy + ((q >> 1) & 1), sz + (q >> 2))]);
        }
        disp = __uint_as_float(m);
      }
      bud = bud - disp * 1.000001f - 1e-7f;
      float dist2 = __builtin_inff();
      if (s >= 0) {
        sp = fwd ? pr.ref[s] : pr.mov[s];
        float d0 = sp.x - p.x, d1 = sp.y - p.y, d2 = sp.z - p.z, d3 = sp.w - p.w;
        dist2 = 0.f;
        dist2 = dist2 + d0 * d0;
        dist2 = dist2 + d1 * d1;
        dist2 = dist2 + d2 * d2;
        dist2 = dist2 + d3 * d3;
      }
      if (bud > 1e-5f) {
        // still exact: nothing has moved far enough to change the answer
        int res = -1;
        if (s >= 0) {
          if (dist2 <= g.radius2) res = s;
          else bud = 0.f;  // the partner left the radius (so did everything else): search next time
        }
        if (fwd) { nn_f[qid] = res; bud_f[qid] = bud; }
        else { nn_b[qid - Mn] = res; bud_b[qid - Mn] = bud; }
        if (res >= 0) {  // the pair's moments (forward: this moving point; backward: the partner)
          m_act = true;
          m_p = fwd ? p : sp;
          m_q = fwd ? sp : p;
          m_c = pr.comp[fwd ? qid : s];
        }
      } else if (s >= 0 && g2_seed_admissible(g, fwd ? g_ref : g_mov, p, fwd ? pr.df : -pr.df, dist2, pr.mu)) {
        to_seeded = true;
      } else {
        to_coop = true;
      }
    }
    if (STREAM) {
      if (m_act) mom_add_stream(pr, mt, m_p, m_q, m_c, lim2);
    } else {
      mom_add_wave(pr, mt, m_act, m_p, m_q, m_c, lim2);
    }
    // the work lists, by direction (a wave is all forward or all backward, except the one at Mn)
    const unsigned long long mf = __ballot(fwd);
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
      const unsigned long long md = dir ? ~mf : mf;
      const unsigned long long ms = __ballot(to_seeded) & md, mc = __ballot(to_coop) & md;
      const int off = dir ? Mn : 0;
      if (ms) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&st->ws_n[dir], __popcll(ms));
        base = __shfl(base, 0, 64);
        if (to_seeded && fwd == (dir == 0)) pr.wls[off + base + __popcll(ms & ((1ULL << lane) - 1ULL))] = qid;
      }
      if (mc) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&st->wl_n[dir], __popcll(mc));
        base = __shfl(base, 0, 64);
        if (to_coop && fwd == (dir == 0)) pr.wl[off + base + __popcll(mc & ((1ULL << lane) - 1ULL))] = qid;
      }
    }
  }
  __syncthreads();
  WGT_MARK(wg_m2);
  mom_flush(mt);
  WGT_END(1, k, wg_m1, wg_m2);
}

// Iteration 1 of a registration: no partner yet. Half of all cooperative searches are these -- 32 lanes
// walking 27 coarse blocks per query. A point of the OTHER cloud that lies in the query's own fine cell is a
// seed as good as an old partner: where the search's own admission test accepts it, the query goes to the
// eight-lane search around it (exact: the same result as the cooperative search), the seed's index parked in
// the neighbour buffer that an iteration 0 would have written; the others take the cooperative search.
// A launch of its own (entries not in their first iteration leave at once) so that the budget test keeps
// its registers.
__global__ void __launch_bounds__(kBlock)
icp_first_kernel(const IcpProblem* __restrict__ tab, Batch bt, BatchInts kbase, int j, int flags) {
  const WgXY wg = xcd_wg((flags & 4) != 0);
  const IcpProblem& pr = tab[bt.idx[wg.y]];
  const int k = kbase.v[wg.y] + 1 + j;
  IcpState* st = pr.st;
  if (k != 1 || k > pr.max_iter || k > st->stop_iter) return;
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  const int nq = Mn + Rn;
  const int nround = (nq + kBlock - 1) / kBlock;
  const GridDesc g = pr.geom->g;
  const int lane = threadIdx.x & 63;
  const Grid2 g_ref = g2_view(pr.gr, Rn), g_mov = g2_view(pr.gm[1], Mn);
  for (int r = wg.x; r < nround; r += gridDim.x) {  // whole waves stay together (ballots below)
    const int qid = r * kBlock + threadIdx.x;
    const bool valid = qid < nq;
    const bool fwd = qid < Mn;
    bool to_seeded = false;
    if (valid) {
      const float4 p = fwd ? pr.mov[qid] : pr.ref[qid - Mn];
      Grid2 gh = g_ref;  // (a copy selected field by field: a reference to either would live in scratch memory)
      if (!fwd) gh = g_mov;
      const int dfq = fwd ? pr.df : -pr.df;
      const int fc = frame_cell(g, p.x) + dfq;
      const int cx = (int)floorf((p.y - g.lo[0]) / g.h), cy = (int)floorf((p.z - g.lo[1]) / g.h),
                cz = (int)floorf((p.w - g.lo[2]) / g.h);
      if (fc >= 0 && fc < kFrameCells && cx >= 0 && cy >= 0 && cz >= 0) {
        uint32_t cnt_w = 0, pad_w = 0;
        const int fs = g2_find(gh.fslot, gh.Hf - 1, pack_cell(fc, cx, cy, cz), &cnt_w, &pad_w);
        if (fs >= 0 && cnt_w > 0) {
          const float4 sp = gh.spts[(size_t)fs * kCellCap];
          const int sidx = gh.sidx[(size_t)fs * kCellCap];
          const float d0 = sp.x - p.x, d1 = sp.y - p.y, d2 = sp.z - p.z, d3 = sp.w - p.w;
          float dist2 = 0.f;
          dist2 = dist2 + d0 * d0;
          dist2 = dist2 + d1 * d1;
          dist2 = dist2 + d2 * d2;
          dist2 = dist2 + d3 * d3;
          if (g2_seed_admissible(g, gh, p, dfq, dist2, pr.mu)) {
            to_seeded = true;
            if (fwd) pr.nn_f[qid] = sidx;  // buffer (k - 1) & 1 = 0: where the searches read their seeds
            else pr.nn_b[qid - Mn] = sidx;
          }
        }
      }
    }
    const bool to_coop = valid && !to_seeded;
    const unsigned long long mf = __ballot(fwd);
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
      const unsigned long long md = dir ? ~mf : mf;
      const unsigned long long ms = __ballot(to_seeded) & md, mc = __ballot(to_coop) & md;
      const int off = dir ? Mn : 0;
      if (ms) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&st->ws_n[dir], __popcll(ms));
        base = __shfl(base, 0, 64);
        if (to_seeded && fwd == (dir == 0)) pr.wls[off + base + __popcll(ms & ((1ULL << lane) - 1ULL))] = qid;
      }
      if (mc) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&st->wl_n[dir], __popcll(mc));
        base = __shfl(base, 0, 64);
        if (to_coop && fwd == (dir == 0)) pr.wl[off + base + __popcll(mc & ((1ULL << lane) - 1ULL))] = qid;
      }
    }
  }
}

// The searches of one iteration for ALL entries of the batch, as a queue: a work item is one round of
// one (entry, direction, kind) work list -- 8 cooperative or 32 seeded queries -- and the workgroups
// take the items w = blockIdx.x, blockIdx.x + gridDim.x, ... of the concatenated lists (cooperative
// lists first: they are the long ones). Every workgroup first forms the same table of rounds from the
// list lengths on the device (4 words per entry). The form this replaces was a (rounds of the largest
// list) x entries x 4 grid whose ~10 k workgroups mostly found nothing to do: dealing them out took
// longer than the searches -- the workgroups with work started up to 26 us apart (58 at the 90th
// percentile, 187 us at the 99th, the first iteration of a registration, whose seeded slices are all
// empty), in a launch whose mean workgroup lasts 25 us (tools/wg_trace.py).
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8)))
icp_nn_search_kernel(const IcpProblem* __restrict__ tab, Batch bt, BatchInts kbase, int j, int flags) {
  WGT_BEGIN();
  static_assert(4 * kMaxBatch <= 2 * kBlock, "two lists per thread");
  __shared__ int s_pref[4 * kMaxBatch + 1];
  __shared__ int s_wsum[kBlock / 64];
  __shared__ int skey[kMomSlotsSearch];
  __shared__ unsigned long long sacc[kMomSlotsSearch * 17];
  const int nlist = 4 * bt.n;  // list t: t < 2n cooperative, else seeded; entry (t mod 2n) / 2, direction t & 1
  // thread x holds the rounds of lists 2x and 2x + 1 (nlist <= 2 * kBlock)
  int my_rounds[2] = {0, 0};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int t = 2 * (int)threadIdx.x + h;
    if (t < nlist) {
      const bool coop_t = t < 2 * bt.n;
      const int tt = coop_t ? t : t - 2 * bt.n;
      const int e = tt >> 1, dir_t = tt & 1;
      const IcpProblem& pe = tab[bt.idx[e]];
      const int ke = kbase.v[e] + 1 + j;
      const IcpState* se = pe.st;
      if (!(ke > pe.max_iter || ke > se->stop_iter)) {
        const int nw = coop_t ? se->wl_n[dir_t] : se->ws_n[dir_t];
        const int per = coop_t ? kBlock / 32 : kBlock / 8;
        my_rounds[h] = (nw + per - 1) / per;
      }
    }
  }
  {  // exclusive prefix of the rounds
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int mine = my_rounds[0] + my_rounds[1];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int before = 0;
    for (int q = 0; q < wv; ++q) before += s_wsum[q];
    const int ex = before + incl - mine;
    const int t0 = 2 * (int)threadIdx.x;
    if (t0 < nlist) s_pref[t0] = ex;
    if (t0 + 1 < nlist) s_pref[t0 + 1] = ex + my_rounds[0];
    if (t0 == nlist - 2) s_pref[nlist] = ex + mine;  // (nlist = 4 n is even)
    __syncthreads();
  }
  const int R = s_pref[nlist];
  int w_first = blockIdx.x, w_step = gridDim.x, w_end = R;
  if ((flags & 4) && (gridDim.x & 7u) == 0) {
    // (see xcd_wg) the workgroups that share an XCD take a contiguous eighth of the items: the lists are
    // concatenated entry by entry, so an XCD's L2 sees the grids of an eighth of the entries
    const int lab = blockIdx.x & 7, slot = blockIdx.x >> 3;
    w_first = (int)(((int64_t)R * lab) >> 3) + slot;
    w_end = (int)(((int64_t)R * (lab + 1)) >> 3);
    w_step = gridDim.x >> 3;
  }
  for (int w = w_first; w < w_end; w += w_step) {
    int lo = 0, hi = nlist - 1;  // the list of item w: the last t with s_pref[t] <= w
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_pref[mid] <= w) lo = mid; else hi = mid - 1;
    }
    const int r = w - s_pref[lo];
    const bool coop = lo < 2 * bt.n;
    const int tt = coop ? lo : lo - 2 * bt.n;
    const int e = tt >> 1, dir = tt & 1;  // 0: forward queries (moving -> reference grid), 1: backward
    const IcpProblem& pr = tab[bt.idx[e]];
    const int k = kbase.v[e] + 1 + j;
    IcpState* st = pr.st;
    const int nw = coop ? st->wl_n[dir] : st->ws_n[dir];
    int Mn, Rn, Ra;
    icp_sizes(pr, Mn, Rn, Ra);
    GridDesc g = pr.geom->g;
    const bool fwd = dir == 0;
    // everything below is uniform in the workgroup: one direction, one grid, one instance of the search
    int32_t* __restrict__ nn_o = fwd ? pr.nn_f + (size_t)(k & 1) * pr.capM : pr.nn_b + (size_t)(k & 1) * pr.capR;
    float* __restrict__ bud_o = fwd ? pr.bud_f + (size_t)(k & 1) * pr.capM : pr.bud_b + (size_t)(k & 1) * pr.capR;
    const Grid2 gs = fwd ? g2_view(pr.gr, Rn) : g2_view(pr.gm[k & 1], Mn);  // the hashed cloud
    const float4* __restrict__ qpts = fwd ? (const float4*)pr.mov : pr.ref;    // the queries' cloud
    const float4* __restrict__ hpts = fwd ? pr.ref : (const float4*)pr.mov;    // the hashed cloud's points
    const int qoff = fwd ? 0 : Mn;
    const int df = fwd ? pr.df : -pr.df;
    const MomTable mt = {skey, sacc, (unsigned long long*)pr.acc + (size_t)(k & 1) * pr.C * kAcc, (int)pr.C,
                         kMomSlotsSearch};
    const double lim2 = icp_limit_sq(Mn + Rn);
    mom_init(mt);
    __syncthreads();
    if (coop) {
      const int sub = threadIdx.x & 31;
      const int32_t* __restrict__ list = pr.wl + qoff;
      const int wq = r * (kBlock / 32) + (threadIdx.x >> 5);
      const bool valid = wq < nw;
      const int qi = valid ? list[wq] - qoff : 0;  // index in the queries' cloud
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) p = qpts[qi];
      float bud = 0.f;
      const int res = g2_nearest_coop<true>(g, gs, p, df, valid, &st->err, nullptr, pr.mu, &bud);
      if (valid && sub == 0) {
        nn_o[qi] = res;
        bud_o[qi] = bud;
        if (res >= 0) {  // (moving point, reference point, the moving point's component)
          if (fwd) mom_add_stream(pr, mt, p, hpts[res], pr.comp[qi], lim2);
          else mom_add_stream(pr, mt, hpts[res], p, pr.comp[res], lim2);
        }
      }
    } else {
      const int32_t* __restrict__ seed = fwd ? pr.nn_f + (size_t)((k - 1) & 1) * pr.capM
                                             : pr.nn_b + (size_t)((k - 1) & 1) * pr.capR;
      const int32_t* __restrict__ list = pr.wls + qoff;
      const int wq = r * (kBlock / 8) + (threadIdx.x >> 3);
      const bool valid = wq < nw;
      const int qi = valid ? list[wq] - qoff : 0;
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f), sp = p;
      int sd = -1;
      if (valid) {
        p = qpts[qi];
        sd = seed[qi];
        sp = hpts[sd];
      }
      int res = -1;
      float bud = 0.f;
      const bool ok = g2_nearest_seeded8(g, gs, p, df, sp, sd, valid, pr.mu, &res, &bud);
      if (valid && (threadIdx.x & 7) == 0) {
        if (!ok) st->err = 4;  // cannot happen: the admission test is the search's own arithmetic
        nn_o[qi] = res;
        bud_o[qi] = bud;
        if (res >= 0) {
          if (fwd) mom_add_stream(pr, mt, p, hpts[res], pr.comp[qi], lim2);
          else mom_add_stream(pr, mt, hpts[res], p, pr.comp[res], lim2);
        }
      }
    }
    __syncthreads();
    mom_flush(mt);
    __syncthreads();  // (the table is initialised again for the next item)
  }
#ifdef PCS_WG_TRACE
  if (w_first < w_end) WGT_END(2, 0, 0ULL, 0ULL);
#endif
}

// centre of the matched points: fp32 segment mean, then promoted (registration_utils.py:150-151);
// cs = the same centres relative to the moment origin
__device__ __forceinline__ void centres(const long long* __restrict__ a, const double* org,
                                        double* cm, double* cr, double* sp, double* sq) {
  double n = (double)a[0];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    sp[d] = (double)a[1 + d] * kFxInv;
    sq[d] = (double)a[4 + d] * kFxInv;
  }
  // (selects, not branches around indexed stores: the arrays stay in registers)
  const bool any = n > 0.5;
  const double nn = any ? n : 1.0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double m = (double)(float)(sp[d] / nn + org[d]);
    const double r = (double)(float)(sq[d] / nn + org[d]);
    cm[d] = any ? m : 0.0;
    cr[d] = any ? r : 0.0;
  }
}

// l1 passes on the positions saved before the last update. MODE 0: sum of dist; MODE 1: sum of
// dist clamped to mean +- trunc (truncated_robust_mean, registration_utils.py:44-58). Fixed-point
// integer sums, like the moments.
template <int MODE>
__global__ void __launch_bounds__(kBlock)
icp_l1_kernel(const IcpProblem* __restrict__ tab, Batch bt, double trunc) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  const int iters = pr.st->iters;
  if (iters < 1) return;
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  const int ne = Mn + Rn;
  const float4* __restrict__ mov = pr.prev;
  const float4* __restrict__ ref = pr.ref;
  const int32_t* __restrict__ comp = pr.comp;
  const int32_t* __restrict__ nn_f = pr.nn_f + (size_t)(iters & 1) * pr.capM;
  const int32_t* __restrict__ nn_b = pr.nn_b + (size_t)(iters & 1) * pr.capR;
  long long* __restrict__ acc = pr.acc + (size_t)(iters & 1) * pr.C * kAcc;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += gridDim.x * blockDim.x) {
    int m = -1, r = -1;
    bool active = false;
    if (i < Mn) { m = i; r = nn_f[i]; active = r >= 0; }
    else { r = i - Mn; m = nn_b[r]; active = m >= 0; }
    if (!active) continue;
    int c = comp[m];
    float4 p = mov[m], q = ref[r];
    const long long* a = acc + (int64_t)c * kAcc;
    const float4 og = pr.corg[c];
    const double org[3] = {(double)og.y, (double)og.z, (double)og.w};
    double cm[3], cr[3], sp[3], sq[3];
    centres(a, org, cm, cr, sp, sq);
    double e0 = ((double)p.y - cm[0]) - ((double)q.y - cr[0]);
    double e1 = ((double)p.z - cm[1]) - ((double)q.z - cr[1]);
    double e2 = ((double)p.w - cm[2]) - ((double)q.w - cr[2]);
    double dist = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
    double v;
    if (MODE == 0) {
      v = dist;
    } else {
      double mean = ((double)a[17] * kFxInv) / (double)a[0];
      v = fmin(fmax(dist, mean - trunc), mean + trunc);
    }
    atomicAdd((unsigned long long*)&acc[(int64_t)c * kAcc + (MODE == 0 ? 17 : 18)],
              (unsigned long long)to_fx(v));
  }
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
#pragma unroll
  for (int j = 0; j < 3; ++j)
    S[j] = sqrt(A[0 * 3 + j] * A[0 * 3 + j] + A[1 * 3 + j] * A[1 * 3 + j] + A[2 * 3 + j] * A[2 * 3 + j]);
  // columns by descending singular value: three compare-and-swaps of whole columns (no indexed
  // access: the arrays stay in registers, the kernel needs no scratch memory)
#define PCS_SVD_SWAP(a, b)                                             \
  if (S[b] > S[a]) {                                                   \
    double ts = S[a]; S[a] = S[b]; S[b] = ts;                          \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                    \
      double ta = A[i * 3 + a]; A[i * 3 + a] = A[i * 3 + b]; A[i * 3 + b] = ta; \
      double tv = V[i * 3 + a]; V[i * 3 + a] = V[i * 3 + b]; V[i * 3 + b] = tv; \
    }                                                                  \
  }
  PCS_SVD_SWAP(0, 1)
  PCS_SVD_SWAP(0, 2)
  PCS_SVD_SWAP(1, 2)
#undef PCS_SVD_SWAP
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int i = 0; i < 3; ++i) U[i * 3 + j] = S[j] > 1e-300 ? A[i * 3 + j] / S[j] : 0.0;
  }
  // complete U when a singular value vanished (rank-deficient M): orthonormal completion
  if (S[2] <= 1e-300) {
    if (S[1] <= 1e-300) {
      if (S[0] <= 1e-300) { for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
      double u0[3] = {U[0], U[3], U[6]};
      int k = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
      double e[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
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
__device__ inline void icp_solve_component(const long long* __restrict__ a, const double* org,
                                           const double* __restrict__ Tc, double reg, double R[9],
                                           double t[3], double Tn[16], double* ls_out) {
  double n = (double)a[0];
  double cm[3], cr[3], sp[3], sq[3];
  centres(a, org, cm, cr, sp, sq);
  double cms[3], crs[3];  // centres relative to the moment origin
  for (int d = 0; d < 3; ++d) { cms[d] = cm[d] - org[d]; crs[d] = cr[d] - org[d]; }
  double M[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      // sum P(x)Q = sum p q^T - cm sum q^T - sum p cr^T + n cm cr^T (coordinates relative to the
      // origin: the expression is translation-invariant);  robust_mean divides by n
      double spq = (double)a[7 + i * 3 + j] * kFxInv;
      double cov = spq - cms[i] * sq[j] - sp[i] * crs[j] + n * cms[i] * crs[j];
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
#ifdef PCS_WG_TRACE
  {
    const unsigned long long need = __ballot(!have_R);
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&g_g2cnt[11], 1ULL);
      if (need) { atomicAdd(&g_g2cnt[10], 1ULL); atomicAdd(&g_g2cnt[9], (unsigned long long)__popcll(need)); }
    }
  }
#endif
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
    ls = (double)a[16] * kFxInv;
    for (int q = 0; q < 3; ++q) {
      double d = cm[q] - cr[q];
      ls += -2.0 * d * (sp[q] - sq[q]) + n * d * d;
    }
  }
  *ls_out = ls;
}

// iteration k, third launch: one THREAD per component solves the regularised Kabsch step from the
// completed moments: rotation and translation of the step (Rt, read by every point of the component
// in the next launch), the updated transform (read buffer k-1, write buffer k), the loss term; the
// other moment buffer is zeroed for the next iteration. (Until round 3 every moving POINT solved its
// own component's step inside the apply launch -- one launch fewer, fine for a handful of units per
// launch, but ~50 redundant 3x3 solves per component and 182 registers per lane once a launch carries
// a hundred units.)
constexpr int kSolveBlock = 64;  // one wave per workgroup: 24 KB of LDS, six workgroups per CU beside the other kernels
template <bool STAGED>
__global__ void __launch_bounds__(kSolveBlock)
icp_solve_kernel(const IcpProblem* __restrict__ tab, Batch bt, BatchInts kbase, int j) {
  // The per-component records (20 moment words, 16 transform words in, 12 + 16 out) are read and written
  // by one thread each: 64 cache lines per wave and load instruction. With 256-thread workgroups that
  // made the launch as long as the budget test (38 us at 24 units); one-wave workgroups spread the same
  // gathers over four times as many workgroups. STAGED (off by default, see icp_enqueue_chunk): the
  // records pass through LDS -- those of a workgroup's 64 consecutive components are contiguous, so the
  // copies are coalesced.
  __shared__ long long s_acc[STAGED ? kSolveBlock * kAcc : 1];
  __shared__ double s_T[STAGED ? kSolveBlock * 16 : 1];    // transform in, then transform out
  __shared__ double s_rt[STAGED ? kSolveBlock * 12 : 1];
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  const int k = kbase.v[blockIdx.y] + 1 + j;
  IcpState* st = pr.st;
  if (k > pr.max_iter || k > st->stop_iter) return;  // (one wave per workgroup: uniform)
  const int64_t C = pr.C;
  const double reg = pr.reg;
  long long* __restrict__ acc_k = pr.acc + (size_t)(k & 1) * C * kAcc;
  long long* __restrict__ acc_n = pr.acc + (size_t)((k + 1) & 1) * C * kAcc;
  const double* __restrict__ T_in = pr.Tb + (size_t)((k - 1) & 1) * C * 16;
  double* __restrict__ T_out = pr.Tb + (size_t)(k & 1) * C * 16;
  long long loss_fx = 0;  // loss terms of the components this thread owns (fixed point)
  long long loss_ck = 0;  // the same at 10 fractional bits (overflow check of the sum, see icp_decide)
  const int nround = (int)((C + kSolveBlock - 1) / kSolveBlock);
  for (int r = blockIdx.x; r < nround; r += gridDim.x) {
    const int64_t c0 = (int64_t)r * kSolveBlock;
    const int nc = (int)(C - c0 < kSolveBlock ? C - c0 : kSolveBlock);
    const int64_t c = c0 + threadIdx.x;
    if (STAGED) {
      for (int q = threadIdx.x; q < nc * kAcc; q += kSolveBlock) {
        s_acc[q] = acc_k[c0 * kAcc + q];
        acc_n[c0 * kAcc + q] = 0;  // the other moment buffer, for the next iteration (nothing reads it now)
      }
      for (int q = threadIdx.x; q < nc * 16; q += kSolveBlock) s_T[q] = T_in[c0 * 16 + q];
      __syncthreads();
    }
    if ((int)threadIdx.x < nc) {
      const long long* a = STAGED ? s_acc + threadIdx.x * kAcc : acc_k + c * kAcc;
      const double* Tc = STAGED ? s_T + threadIdx.x * 16 : T_in + c * 16;
      double R[9], t[3], Tn[16];
      const bool has = pr.first[c] != INT_MAX;  // components with a moving point
      if (has) {
        const float4 og = pr.corg[c];
        const double org[3] = {(double)og.y, (double)og.z, (double)og.w};
        double ls;
        icp_solve_component(a, org, Tc, reg, R, t, Tn, &ls);
        if (!(fabs(ls) < 6.0e7)) st->err = 1;  // (beyond the fixed-point range of a term)
        else {
          loss_fx += to_fx(ls);
          loss_ck += __double2ll_rn(ls * 1024.0);
        }
      } else {  // nothing moves: the transform stays what it is (the identity the reset wrote)
#pragma unroll
        for (int q = 0; q < 16; ++q) Tn[q] = Tc[q];
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = (q % 4 == 0) ? 1.0 : 0.0;
        t[0] = t[1] = t[2] = 0.0;
      }
      if (STAGED) {
#pragma unroll
        for (int q = 0; q < 16; ++q) s_T[threadIdx.x * 16 + q] = Tn[q];  // (a thread's own record: read above)
#pragma unroll
        for (int q = 0; q < 9; ++q) s_rt[threadIdx.x * 12 + q] = R[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) s_rt[threadIdx.x * 12 + 9 + q] = t[q];
      } else if (has) {
        double* rt = pr.Rt + c * 12;
#pragma unroll
        for (int q = 0; q < 9; ++q) rt[q] = R[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) rt[9 + q] = t[q];
        for (int q = 0; q < 16; ++q) T_out[c * 16 + q] = Tn[q];
        for (int q = 0; q < kAcc; ++q) acc_n[c * kAcc + q] = 0;
      }
    }
    if (STAGED) {
      __syncthreads();
      for (int q = threadIdx.x; q < nc * 16; q += kSolveBlock) T_out[c0 * 16 + q] = s_T[q];
      for (int q = threadIdx.x; q < nc * 12; q += kSolveBlock) pr.Rt[c0 * 12 + q] = s_rt[q];
      __syncthreads();  // (the buffers are filled again by the next round)
    }
  }
  // the iteration's loss: one integer add per wave that owns a component, spread over kLossParts words
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    loss_fx += __shfl_xor(loss_fx, o, 64);
    loss_ck += __shfl_xor(loss_ck, o, 64);
  }
  if ((threadIdx.x & 63) == 0 && (loss_fx != 0 || loss_ck != 0)) {
    atomicAdd((unsigned long long*)&pr.lossfx[(size_t)(k & 1) * kLossParts + (blockIdx.x & (kLossParts - 1))],
              (unsigned long long)loss_fx);
    atomicAdd((unsigned long long*)&pr.lossfx[(size_t)(2 + (k & 1)) * kLossParts + (blockIdx.x & (kLossParts - 1))],
              (unsigned long long)loss_ck);
  }
}

// iteration k, last launch: every moving point moves by its component's step. The previous positions are
// kept for the l1 passes. The moved points go straight into the grid of the next iteration (the moving
// grid is double-buffered: no separate build launch), and the grid this iteration searched is reset.
__global__ void __launch_bounds__(kBlock)
icp_apply_kernel(const IcpProblem* __restrict__ tab, Batch bt, BatchInts kbase, int j, int flags) {
  WGT_BEGIN();
  const WgXY wg = xcd_wg((flags & 4) != 0);
  const IcpProblem& pr = tab[bt.idx[wg.y]];
  const int k = kbase.v[wg.y] + 1 + j;
  IcpState* st = pr.st;
  if (k > pr.max_iter || k > st->stop_iter) return;
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  float4* __restrict__ mov = pr.mov;
  float4* __restrict__ prev = pr.prev;
  const int32_t* __restrict__ comp = pr.comp;
  const double* __restrict__ Rt = pr.Rt;
  const Grid2 gm_done = g2_view(pr.gm[k & 1], Mn);
  const Grid2 gm_next = g2_view(pr.gm[(k + 1) & 1], Mn);
  const int64_t tid = wg.x * (int64_t)blockDim.x + threadIdx.x;
  // the searches of this iteration are done with gm_done: reset it (it is filled again by the apply
  // of iteration k+1); gm_next was reset by the apply of iteration k-1 and takes the moved points
  g2_clear(gm_done, tid, (int64_t)gridDim.x * blockDim.x);
  if (tid == 0) {
    st->iters = k;
    st->wl_n[0] = st->wl_n[1] = 0;  // the work lists of iteration k have been served
    st->ws_n[0] = st->ws_n[1] = 0;
    st->dmax[(k + 1) & 1] = 0u;   // takes the steps of iteration k+1
  }
  {  // the step table of iteration k+1 (nothing reads it during this launch)
    unsigned int* nxt = pr.sstep + (size_t)((k + 1) & 1) * kStepH;
    for (int64_t q = tid; q < kStepH; q += (int64_t)gridDim.x * blockDim.x) nxt[q] = 0u;
  }
  unsigned int* __restrict__ sstep = pr.sstep + (size_t)(k & 1) * kStepH;
  WGT_MARK(wg_m1);
  float step = 0.f;  // largest step of this thread's points
  const GridDesc d = pr.geom->g;
  const int nround = (Mn + kBlock - 1) / kBlock;
#ifdef PCS_WG_TRACE
  int wgt_rounds = 0, wgt_owner = 0;
#endif
  for (int r = wg.x; r < nround; r += gridDim.x) {  // whole workgroups stay together (shuffles)
    const int i = r * kBlock + threadIdx.x;
    float4 moved = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef PCS_WG_TRACE
    ++wgt_rounds;
    if (i < Mn && pr.first[pr.comp[i]] == i) ++wgt_owner;
#endif
    if (i < Mn) {
      const double* __restrict__ R = Rt + (int64_t)comp[i] * 12;
      const double* __restrict__ t = R + 9;
      float4 p = mov[i];
      prev[i] = p;
      double x = p.y, y = p.z, z = p.w;
      float nx = (float)(x * R[0] + y * R[1] + z * R[2] + t[0]);
      float ny = (float)(x * R[3] + y * R[4] + z * R[5] + t[1]);
      float nz = (float)(x * R[6] + y * R[7] + z * R[8] + t[2]);
      moved = make_float4(p.x, nx, ny, nz);
      mov[i] = moved;
      {
        // the verified cache's budget unit: how far this point moved, filed under the place it moved to
        float dx = nx - p.y, dy = ny - p.z, dz = nz - p.w;
        float st1 = sqrtf((dx * dx + dy * dy) + dz * dz) * 1.000001f + 1e-7f;
        step = fmaxf(step, st1);
        const int fx = (int)floorf((nx - d.lo[0]) / d.h), fy = (int)floorf((ny - d.lo[1]) / d.h),
                  fz = (int)floorf((nz - d.lo[2]) / d.h);
        atomicMax(&sstep[icp_step_slot(fx >> 3, fy >> 3, fz >> 3)], __float_as_uint(st1));
      }
    }
    g2_insert_point(d, gm_next, i < Mn, i, moved, &st->err);
  }
  WGT_MARK(wg_m2);
  step = wave_max(step);
  if ((threadIdx.x & 63) == 0 && step > 0.f) atomicMax(&st->dmax[k & 1], __float_as_uint(step));
#ifdef PCS_WG_TRACE
  {
    __shared__ int s_own;
    if (threadIdx.x == 0) s_own = 0;
    __syncthreads();
    atomicAdd(&s_own, wgt_owner);
    __syncthreads();
    const unsigned long long info = ((unsigned long long)wgt_rounds << 32) | (unsigned)s_own;
    WGT_END(4, k, info, wg_m2);
  }
#endif
}

#ifdef PCS_WG_TRACE
extern "C" int pcs_wg_trace_set(unsigned long long* buf) {
  unsigned z = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wgt), &buf, sizeof(buf)) != hipSuccess) return -3;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wgt_n), &z, sizeof(z)) != hipSuccess) return -3;
  return 0;
}
extern "C" int pcs_g2_counters(unsigned long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_g2cnt), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
extern "C" unsigned pcs_wg_trace_count() {
  unsigned n = 0;
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wgt_n), sizeof(n));
  return n;
}
#endif

// coverage: moving points (final positions) with a reference point (stationary included) in range
__global__ void __launch_bounds__(kBlock)
icp_coverage_kernel(const IcpProblem* __restrict__ tab, Batch bt) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  if (Mn < 1 || Ra < 1) return;
  GridDesc g = pr.geom->g;
  const Grid2 g_all = g2_view(pr.ga, Ra);
  const int nround = (Mn + kBlock / 32 - 1) / (kBlock / 32);
  for (int r = blockIdx.x; r < nround; r += gridDim.x) {
    int qid = r * (kBlock / 32) + (threadIdx.x >> 5);
    bool valid = qid < Mn;
    float4 p = valid ? pr.mov[qid] : make_float4(0.f, 0.f, 0.f, 0.f);
    int rr = g2_nearest_coop(g, g_all, p, pr.df, valid, &pr.st->err);
    if (valid && (threadIdx.x & 31) == 0 && rr >= 0) atomicAdd(&pr.cov[pr.comp[qid]], 1);
  }
}

__global__ void __launch_bounds__(kBlock)
icp_finish_kernel(const IcpProblem* __restrict__ tab, Batch bt, IcpState* __restrict__ out) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  const IcpState* st = pr.st;
  if (out && blockIdx.x == 0 && threadIdx.x == 0) out[blockIdx.y] = *st;  // err word for the host (pinned)
  const int iters = st->iters;
  const int64_t C = pr.C;
  const long long* acc = pr.acc + (size_t)(iters & 1) * C * kAcc;  // moments of the last iteration
  const double* Tfin = pr.Tb + (size_t)(iters & 1) * C * 16;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < C;
       c += (int64_t)gridDim.x * blockDim.x) {
    double n = (double)acc[c * kAcc];
    pr.l1_out[c] = (iters > 0 && n > 0.5) ? ((double)acc[c * kAcc + 18] * kFxInv) / n : 0.0;
    // component_edge_count / (component_deg.float() + 1e-6)  (registration_utils.py:197-199)
    pr.ratio_out[c] = (float)pr.cov[c] / ((float)pr.comp_deg[c] + 1e-6f);
    for (int q = 0; q < 16; ++q) pr.T_out[c * 16 + q] = Tfin[c * 16 + q];
  }
}

// ---- optional HIP-event timing of the three iteration kernels, for bench.py's roofline ----
struct KernelProfile {
  double ms = 0.0;
  double launches = 0.0;
  double bytes = 0.0;
};
static bool g_prof_on = false;
static KernelProfile g_prof[3];
static std::mutex g_prof_mu;
constexpr int kProfChunk = 16;  // chunk limit of the stand-alone entry point
struct ProfEvents {
  hipEvent_t ev[6];  // one iteration of a chunk carries events
  bool have = false;
};
constexpr int kProfSlots = 8;  // chunks one thread may have in flight (one per lock-step group)
static thread_local ProfEvents t_evs[kProfSlots];
static thread_local int t_ev_useds[kProfSlots];  // iterations of the slot's last chunk that carried events

static inline int gx_for(int64_t items, int per_block, int cap) {
  int64_t g = (items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

int icp_enqueue_begin(const IcpProblem* tab, const Batch& bt, const IcpHint* hints, hipStream_t s) {
  if (bt.n < 1) return PCS_OK;
  int64_t words = 1, nmax = 1, Cmax = 1;
  for (int e = 0; e < bt.n; ++e) {
    if (hints[e].C > Cmax) Cmax = hints[e].C;
    int64_t w = (int64_t)hints[e].clear_words + hints[e].C * kAcc;
    if (w > words) words = w;
    int64_t m = hints[e].Mn > hints[e].Rn ? hints[e].Mn : hints[e].Rn;
    if (hints[e].Ra > m) m = hints[e].Ra;
    if (m > nmax) nmax = m;
  }
  // one launch resets everything the calls own, one finds the bounding boxes, one writes the grid
  // descriptions, one fills the three static grids
  icp_clear_kernel<<<dim3(gx_for(words, kBlock, 2048), bt.n), kBlock, 0, s>>>(tab, bt);
  icp_bounds_kernel<<<dim3(gx_for(nmax, kBlock, 256), bt.n, 3), kBlock, 0, s>>>(tab, bt);
  icp_setup_kernel<<<dim3(gx_for(Cmax, kBlock, 64), bt.n), kBlock, 0, s>>>(tab, bt);
  icp_insert3_kernel<<<dim3(gx_for(nmax, kBlock, 4096), bt.n, 3), kBlock, 0, s>>>(tab, bt);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

int icp_enqueue_chunk(const IcpProblem* tab, const Batch& bt, const BatchInts& kbase, int n_it,
                      const IcpHint* hints, IcpState* out, hipStream_t s, int prof_slot) {
  if (bt.n < 1 || n_it < 1) return PCS_OK;
  if (prof_slot < 0 || prof_slot >= kProfSlots) prof_slot = 0;
  ProfEvents& t_ev = t_evs[prof_slot];
  int& t_ev_used = t_ev_useds[prof_slot];
  int64_t q = 1, m = 1, qsum = 0, cmax = 1;
  for (int e = 0; e < bt.n; ++e) {
    int64_t qe = (int64_t)hints[e].Mn + hints[e].Rn;
    if (qe > q) q = qe;
    qsum += qe;
    if (hints[e].Mn > m) m = hints[e].Mn;
    if (hints[e].C > cmax) cmax = hints[e].C;
  }
  const int nb_s = gx_for(cmax, kSolveBlock, 1024);  // solve: one thread per component
  // PCS_ICP_MOM_STREAM: the budget test's moments as immediate LDS adds per lane (58 VGPRs, 7 waves per SIMD)
  // instead of the shuffle pre-reduction of runs (123 VGPRs, 4 waves per SIMD)
  // (measured A/B, batches of 12: 1.50 s either way -- the launch is not short of waves -- so the default
  // stays the shuffle form, whose workgroups send fewer LDS atomics)
  static const bool mom_stream = [] { const char* e = getenv("PCS_ICP_MOM_STREAM"); return e && atoi(e) != 0; }();
  // bit 0 (PCS_ICP_LDS_STEPS, default on): the step table in LDS for the backward queries; bit 1
  // (PCS_ICP_SEED_FIRST, default on): iteration 1 seeds its searches from the query's own fine cell
  static const int lds_steps = [] {
    const char* e = getenv("PCS_ICP_LDS_STEPS");
    const char* f = getenv("PCS_ICP_SEED_FIRST");
    const char* x = getenv("PCS_ICP_XCD");  // bit 2: XCD-aware placement of the iteration kernels (xcd_wg)
    return ((e && atoi(e) == 0) ? 0 : 1) | ((f && atoi(f) == 0) ? 0 : 2) | ((x && atoi(x) != 0) ? 4 : 0);
  }();
  // PCS_ICP_SOLVE_STAGED=1: the records through LDS (measured A/B on one box, batches of 12: 1.57 s staged
  // against 1.49 s direct -- the coalesced copies cost more than the 64-line gathers of a one-wave
  // workgroup, which overlap with other waves' arithmetic); default: direct
  static const bool solve_staged = [] { const char* e = getenv("PCS_ICP_SOLVE_STAGED"); return e && atoi(e) != 0; }();
  bool any_first = false;
  for (int e = 0; e < bt.n; ++e) any_first = any_first || kbase.v[e] == 0;
  // budget test: one lane per query. A workgroup resets and drains its moment table once whatever the
  // number of rounds it serves (a third of a one-round workgroup's time, profiles/r02_wg_trace.txt):
  // once the launch is several generations of workgroups wide, each takes 2-4 rounds
  const int64_t wg_total = (qsum + kBlock - 1) / kBlock;
  const int rounds_per_wg = wg_total > 7168 ? 4 : (wg_total > 5120 ? 3 : (wg_total > 3072 ? 2 : 1));
  const int nb_q = gx_for((q + rounds_per_wg - 1) / rounds_per_wg, kBlock, 1024);
  // the searches are a queue over the work lists of all entries (items of 8 cooperative / 32 seeded
  // queries): iteration 1 searches for every query, later ones for ~6 % + ~12 % of them
  const int nb_c1 = (int)(qsum / 8 < 64 ? 64 : (qsum / 8 > 2048 ? 2048 : qsum / 8)) & ~7;  // (multiples of 8: xcd_wg)
  const int nb_cs = (int)(qsum / 48 < 64 ? 64 : (qsum / 48 > 2048 ? 2048 : qsum / 48)) & ~7;
  const int nb_m = gx_for(m, kBlock, 4096);
  bool prof = g_prof_on;
  if (prof && !t_ev.have) {
    for (int i = 0; i < 6; ++i) PCS_CHECK_HIP(hipEventCreate(&t_ev.ev[i]));
    t_ev.have = true;
  }
  // ONE iteration of the chunk carries events (its position rotates from chunk to chunk, so that every
  // position -- the full search of iteration 1 included -- is sampled at its own rate): timing every
  // launch cost 7 % of the step (hipExtLaunchKernelGGL + four event records per iteration)
  static thread_local unsigned t_rot = 0;
  const int js = prof ? (int)(t_rot++ % (unsigned)n_it) : -1;
  t_ev_used = prof ? js + 1 : 0;
  for (int j = 0; j < n_it; ++j) {
    const int nb_c = (j == 0 && any_first) ? nb_c1 : nb_cs;
    if (j == 0 && any_first && (lds_steps & 2))
      icp_first_kernel<<<dim3(gx_for(q, kBlock, 1024), bt.n), kBlock, 0, s>>>(tab, bt, kbase, j, lds_steps);
    if (j == js) {
      // the events ride on the dispatch itself (hipExtLaunchKernelGGL), so their difference is the
      // kernel's own begin-to-end time, as rocprofv3 reports it
      hipEvent_t* e = &t_ev.ev[0];
      // (the search is two launches; the pair is timed as one kernel, first start to last stop)
      if (mom_stream)
        hipExtLaunchKernelGGL(icp_nn_kernel<true>, dim3(nb_q, bt.n), dim3(kBlock), 0, s, e[0], nullptr, 0, tab, bt, kbase, j, lds_steps);
      else
        hipExtLaunchKernelGGL(icp_nn_kernel<false>, dim3(nb_q, bt.n), dim3(kBlock), 0, s, e[0], nullptr, 0, tab, bt, kbase, j, lds_steps);
      hipExtLaunchKernelGGL(icp_nn_search_kernel, dim3(nb_c), dim3(kBlock), 0, s, nullptr, e[1], 0, tab,
                            bt, kbase, j, lds_steps);
      if (solve_staged)
        hipExtLaunchKernelGGL(icp_solve_kernel<true>, dim3(nb_s, bt.n), dim3(kSolveBlock), 0, s, e[4], nullptr, 0, tab,
                              bt, kbase, j);
      else
        hipExtLaunchKernelGGL(icp_solve_kernel<false>, dim3(nb_s, bt.n), dim3(kSolveBlock), 0, s, e[4], nullptr, 0, tab,
                              bt, kbase, j);
      hipExtLaunchKernelGGL(icp_apply_kernel, dim3(nb_m, bt.n), dim3(kBlock), 0, s, nullptr, e[5], 0, tab, bt,
                            kbase, j, lds_steps);
    } else {
      if (mom_stream) icp_nn_kernel<true><<<dim3(nb_q, bt.n), kBlock, 0, s>>>(tab, bt, kbase, j, lds_steps);
      else icp_nn_kernel<false><<<dim3(nb_q, bt.n), kBlock, 0, s>>>(tab, bt, kbase, j, lds_steps);
      icp_nn_search_kernel<<<nb_c, kBlock, 0, s>>>(tab, bt, kbase, j, lds_steps);
      if (solve_staged) icp_solve_kernel<true><<<dim3(nb_s, bt.n), kSolveBlock, 0, s>>>(tab, bt, kbase, j);
      else icp_solve_kernel<false><<<dim3(nb_s, bt.n), kSolveBlock, 0, s>>>(tab, bt, kbase, j);
      icp_apply_kernel<<<dim3(nb_m, bt.n), kBlock, 0, s>>>(tab, bt, kbase, j, lds_steps);
    }
  }
  // the stop rule of the chunk's last iteration (the others were settled by the next search)
  icp_decide_kernel<<<bt.n, kBlock, 0, s>>>(tab, bt, kbase, n_it, out);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// after the stream has been synchronised: add the chunk's launches to the profile. live[j] = number
// of entries that really ran iteration position j; bytes3[j*3 + {0,1,2}] = their algorithmic bytes in
// the search / moments / apply launch. Launches in which no entry was live are not counted.
void icp_profile_account(int n_it, const int* live, const double* bytes3, int prof_slot) {
  if (prof_slot < 0 || prof_slot >= kProfSlots) prof_slot = 0;
  ProfEvents& t_ev = t_evs[prof_slot];
  int& t_ev_used = t_ev_useds[prof_slot];
  if (!g_prof_on || t_ev_used < 1) return;
  const int j = t_ev_used - 1;  // the iteration of the chunk that carried the events
  t_ev_used = 0;
  if (j >= n_it || live[j] < 1) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int q = 0; q < 3; q += 2) {  // 0: the two neighbour kernels (moments included), 2: solve + apply
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t_ev.ev[2 * q], t_ev.ev[2 * q + 1]) != hipSuccess) continue;
    g_prof[q].ms += ms;
    g_prof[q].launches += 1.0;
    g_prof[q].bytes += bytes3[j * 3 + q];
  }
}

int icp_enqueue_finish(const IcpProblem* tab, const Batch& bt, const IcpHint* hints, IcpState* out,
                       hipStream_t s) {
  if (bt.n < 1) return PCS_OK;
  int64_t q = 1, m = 1, C = 1;
  for (int e = 0; e < bt.n; ++e) {
    int64_t qe = (int64_t)hints[e].Mn + hints[e].Rn;
    if (qe > q) q = qe;
    if (hints[e].Mn > m) m = hints[e].Mn;
    if (hints[e].C > C) C = hints[e].C;
  }
  // l1 error of the last iteration: positions before its update are in `prev`
  const int nb_e = gx_for(q, kBlock, 2048);
  icp_l1_kernel<0><<<dim3(nb_e, bt.n), kBlock, 0, s>>>(tab, bt, 0.3);
  icp_l1_kernel<1><<<dim3(nb_e, bt.n), kBlock, 0, s>>>(tab, bt, 0.3);
  icp_coverage_kernel<<<dim3(gx_for(m * 32, kBlock, 8192), bt.n), kBlock, 0, s>>>(tab, bt);
  // NB: the coverage kernel may still raise st->err; it runs before this launch on the same stream
  icp_finish_kernel<<<dim3(gx_for(C, kBlock, 256), bt.n), kBlock, 0, s>>>(tab, bt, out);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

}  // namespace pcs

using namespace pcs;

extern "C" {

// bench.py support: HIP-event timing of the three iteration kernels of the fused ICP on their launch
// stream. enable = 1 resets and starts accumulating.
int pcs_profile_icp(int enable) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = enable != 0;
  for (int q = 0; q < 3; ++q) g_prof[q] = KernelProfile();
  return PCS_OK;
}

// out9 (host): {milliseconds, launches, algorithmic bytes} of icp_nn_kernel, icp_moments_kernel,
// icp_apply_kernel accumulated since the last enable.
int pcs_profile_icp_read(double* out9) {
  if (!out9) return PCS_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int q = 0; q < 3; ++q) {
    out9[3 * q + 0] = g_prof[q].ms;
    out9[3 * q + 1] = g_prof[q].launches;
    out9[3 * q + 2] = g_prof[q].bytes;
  }
  return PCS_OK;
}

size_t pcs_register_to_next_frame_workspace_bytes(int64_t Mn, int64_t Rn, int64_t Ra, int64_t C) {
  Arena a(nullptr, 0);
  IcpBufs b;
  plan_icp(a, Mn, Rn, Ra, C, b);
  a.take<IcpProblem>(1);
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
  if (!info || !ws) return PCS_ERR_ARG;
  IcpArgs a;
  a.mov = mov_fxyz; a.comp = mov_comp; a.ref = ref_fxyz; a.all = ref_all; a.comp_deg = comp_deg;
  a.cnt3 = nullptr;
  a.Mn = Mn; a.Rn = Rn; a.Ra = Ra; a.C = C;
  a.radius = radius; a.frame_offset = frame_offset; a.reg = angle_regularizer; a.delta = stopping_delta;
  a.max_iter = max_iter; a.cell_size = cell_size;
  a.T = T; a.l1 = l1_err; a.ratio = edge_ratio;
  IcpProblem host_pr;
  int rc = icp_describe(a, ws, ws_bytes, &host_pr);
  if (rc) return rc;
  // the descriptor table (one slot) sits behind the call's buffers
  IcpProblem* tab = nullptr;
  {
    Arena ar(ws, ws_bytes);
    IcpBufs b;
    plan_icp(ar, Mn, Rn, Ra, C, b);
    tab = ar.take<IcpProblem>(1);
    if (!ar.ok() || !tab) return PCS_ERR_WORKSPACE;
  }
  // status words come back through pinned memory written by the kernels (no device-to-host copy);
  // the descriptor goes up from the pinned staging area (the call ends with a synchronisation, so
  // the next call of this thread cannot overwrite it early)
  char* page = (char*)thread_pinned_page();
  IcpState* pin = (IcpState*)page;
  if (page) {
    IcpProblem* stage = (IcpProblem*)(page + kPinnedStageOff);
    *stage = host_pr;
    PCS_CHECK_HIP(hipMemcpyAsync(tab, stage, sizeof(IcpProblem), hipMemcpyHostToDevice, s));
  } else {
    PCS_CHECK_HIP(hipMemcpyAsync(tab, &host_pr, sizeof(IcpProblem), hipMemcpyHostToDevice, s));
    PCS_CHECK_HIP(hipStreamSynchronize(s));
  }
  if (chunk < 1) chunk = 8;
  if (chunk > kProfChunk) chunk = kProfChunk;
  Batch bt;
  bt.n = 1;
  bt.idx[0] = 0;
  IcpHint hint;
  hint.Mn = (int)Mn; hint.Rn = (int)Rn; hint.Ra = (int)Ra; hint.C = C;
  hint.clear_words = host_pr.gr.slot_words + host_pr.ga.slot_words + 2 * host_pr.gm[0].slot_words;
  rc = icp_enqueue_begin(tab, bt, &hint, s);
  if (rc) return rc;
  IcpState host;
  host.stop_iter = INT_MAX;
  host.iters = 0;
  host.err = 0;
  int it = 0;
  const double by_nn = 76.0 * (double)(Mn + Rn) + 272.0 * (double)C, by_mom = 0.0,
               by_app = 68.0 * (double)Mn + 416.0 * (double)C;
  while (it < max_iter) {
    int n_it = (max_iter - it < chunk) ? (max_iter - it) : chunk;
    BatchInts kb;
    kb.v[0] = it;
    rc = icp_enqueue_chunk(tab, bt, kb, n_it, &hint, pin, s, 0);
    if (rc) return rc;
    if (!pin) PCS_CHECK_HIP(hipMemcpyAsync(&host, host_pr.st, sizeof(IcpState), hipMemcpyDeviceToHost, s));
    PCS_CHECK_HIP(hipStreamSynchronize(s));
    if (pin) host = *pin;
    {
      int live[kProfChunk];
      double b3[3 * kProfChunk];
      for (int j = 0; j < n_it; ++j) {
        live[j] = (it + 1 + j <= host.iters) ? 1 : 0;
        b3[3 * j] = by_nn; b3[3 * j + 1] = by_mom; b3[3 * j + 2] = by_app;
      }
      icp_profile_account(n_it, live, b3, 0);
    }
    it += n_it;
    if (host.err) return PCS_ERR_RANGE;
    if (host.stop_iter != INT_MAX) break;
  }
  info[0] = host.iters;
  info[1] = host.stop_iter != INT_MAX ? 1 : 0;
  info[2] = info[3] = 0;
  rc = icp_enqueue_finish(tab, bt, &hint, pin, s);
  if (rc) return rc;
  if (!pin) PCS_CHECK_HIP(hipMemcpyAsync(&host, host_pr.st, sizeof(IcpState), hipMemcpyDeviceToHost, s));
  PCS_CHECK_HIP(hipStreamSynchronize(s));
  if (pin) host = *pin;
  if (host.err) return PCS_ERR_RANGE;
  return PCS_OK;
}

}  // extern "C"
