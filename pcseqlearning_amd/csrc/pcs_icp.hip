// Fused per-component ICP ("TLS solve") for gfx950, batched.
//
// Reference: register_to_next_frame, pcdet/models/registration/preprocessors/
// registration_utils.py:83-206. Per iteration the reference runs two radius-graph
// builds (4 kernels + scan + 2 host syncs + 4 allocations each), two argsorts, a handful of
// scatter/segment kernels, a batched cuSOLVER 3x3 SVD and a blocking loss.item().
//
// Here a call is a descriptor in a device table (pcs_icp.hpp) and an iteration is four PHASES:
//   budget     one lane per query: the verified cache (see icp_budget_phase); settled pairs add their
//              raw moments, the rest is compacted into work lists
//   search     the listed queries: eight lanes around the old partner, or the cooperative 32-lane
//              search (two-level grid, pcs_grid2.hpp); their pairs' moments
//   solve      one thread per component: regularised Kabsch step from the completed moments (cov +
//              reg*T_rot -> rotation by Newton's polar iteration, Jacobi SVD when det <= 0;
//              R = V diag(1,1,det(VU^T)) U^T, t = c_ref - R c_mov, T <- T_i T), the loss term
//   apply      every moving point moves (f64 -> fp32 as the reference stores them) and enters the grid
//              of the next iteration (double-buffered); the grid just searched is reset
// The moments are 64-bit fixed-point integers (n, sum p, sum q, sum p(x)q, sum |p-q|^2 per component):
// order-free, hence bit-reproducible; the centred covariance and the loss follow algebraically, so the
// reference's second pass over the edges is not needed.
//
// Who runs the phases (round 4):
//   * the UNIT-RESIDENT LOOP (icp_loop_kernel): ONE 1024-thread workgroup owns a registration for a whole
//     chunk and runs its iterations back to back -- the phases are separated by workgroup barriers, the
//     stop rule is settled by the workgroup itself, nothing is launched and nothing crosses a CU between
//     iterations, the unit's state stays in the L2 of one XCD. Rounds 1-3 ran every phase as a launch
//     of its own over all units (four launches per iteration, ~7000 per sequence, each a chain of cold
//     dependent loads): the chip spent its time waiting for launches to start and end. Units of a batch
//     run side by side on CUs of their own instead of taking turns on the whole chip;
//   * wide launches (icp_first / icp_nn / icp_nn_search / icp_solve / icp_apply kernels, blockIdx.y =
//     entry) for the first iteration(s) of a registration, whose full searches want the whole chip.
// Both are the same device functions; every loop is a stride over (workgroup index, workgroup count):
// the launch geometry never changes a result.
#include "pcs_icp.hpp"

#include <hip/hip_ext.h>

#include <atomic>
#include <cstdio>
#include <mutex>

namespace pcs {

// pcs_icp_set_grid_mode: the moving grid rebuilt at every iteration (the form before the in-place update; the
// cross-check of tests/test_gpu_tracking.py) instead of when the solve phase finds it due
static std::atomic<int> g_rebuild_always{0};
// pcs_icp_set_widening: every search is widened by radius / g_mu_div (any value is exact; it trades the cells a
// search looks at against how long its statement lasts)
static std::atomic<double> g_mu_div{16.0};
static std::atomic<int> g_cache_levels{2};
static bool icp_rebuild_always() { return g_rebuild_always.load(std::memory_order_relaxed) != 0; }

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
  b.aux_f = a.take<NnAux>(Mn);
  b.aux_b = a.take<NnAux>(Rn);
  b.wl = a.take<int32_t>(Mn + Rn);
  b.wls = a.take<int32_t>(Mn + Rn);
  b.sstep = a.take<unsigned int>(2 * kStepH);
  b.mloc = a.take<int32_t>(Mn);
  b.rsrt = a.take<float4>(Rn);
  b.msrt = a.take<float4>(Mn);
  b.trow = a.take<int32_t>(kTileRows + 1);
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
  pr.mu = (float)(a.radius / g_mu_div.load(std::memory_order_relaxed));
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
  pr.aux_f = b.aux_f;
  pr.aux_b = b.aux_b;
  pr.wl = b.wl;
  pr.wls = b.wls;
  pr.sstep = b.sstep;
  pr.mloc = b.mloc;
  pr.rsrt = b.rsrt;
  pr.msrt = b.msrt;
  pr.trow = b.trow;
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
  pr.want = (a.want_l1 ? kIcpWantL1 : 0) | (a.want_ratio ? kIcpWantRatio : 0) | (icp_rebuild_always() ? kIcpRebuildAlways : 0) |
            (g_cache_levels.load(std::memory_order_relaxed) == 1 ? kIcpNoSecondLevel : 0);
  return PCS_OK;
}

// Fixed point with kFxBits fractional bits, round to nearest even. For |v| < 2^15 (every moment term: see
// icp_limit_sq) adding 1.5 * 2^52 leaves the rounded integer in the low mantissa bits: two instructions
// where the library's double -> long long conversion takes about twenty, the same result bit for bit
// (v * 2^36 is exact). The components' loss terms use the whole 63-bit range: to_fx_wide.
__device__ __forceinline__ long long to_fx(double v) {
  const double magic = 6755399441055744.0;  // 1.5 * 2^52
  return __double_as_longlong(v * kFx + magic) - __double_as_longlong(magic);
}
__device__ __forceinline__ long long to_fx_wide(double v) { return __double2ll_rn(v * kFx); }
// Sum of a 32-bit word over the 64 lanes, valid in lane 63: row shifts (an inclusive scan of each row of
// 16 lanes) and the two row broadcasts of GFX9's DPP -- six adds, no LDS crossbar, no address arithmetic.
__device__ __forceinline__ int wave_sum_i32_lane63(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}
// The same for a moment term (|v| < 2^51): its low 26 bits and the rest add up separately without carries
// (64 x 2^26 = 2^32, 64 x 2^25 = 2^31), so a 64-bit sum is two independent 32-bit sums.
__device__ __forceinline__ long long wave_sum_fx_lane63(long long v) {
  const int lo = (int)((unsigned long long)v & 0x3FFFFFFULL);
  const int hi = (int)(v >> 26);
  const unsigned slo = (unsigned)wave_sum_i32_lane63(lo);
  const int shi = wave_sum_i32_lane63(hi);
  return ((long long)shi << 26) + (long long)slo;
}

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
    st->loop_iters = 0;
    st->mode = 0;
    st->live_c = 0;
    st->gcur[0] = st->gcur[1] = 1;  // (icp_insert3_kernel fills gm[1])
    st->grebuild = 1;               // (a fresh grid: the first solve phase takes its overflow count as the baseline)
    st->gmoves = 0u;
    st->govf0 = 0;
    st->phase_ticks[0] = st->phase_ticks[1] = st->phase_ticks[2] = st->phase_ticks[3] = 0ULL;
  }
  for (int64_t q = i; q < 2 * C * 16; q += stride) {  // (component-minor: word e of every component side by side)
    int e = (int)((q / C) & 15);
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
  if (pr.want & kIcpWantRatio) g2_clear(g2_view(pr.ga, Ra), i, stride);
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
  int live = 0;
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < pr.C; c += (int64_t)gridDim.x * blockDim.x) {
    int f = pr.first[c];
    pr.corg[c] = f != INT_MAX ? pr.mov[f] : make_float4(0.f, 0.f, 0.f, 0.f);
    live += f != INT_MAX;
  }
  live = wave_sum_i32(live);
  if ((threadIdx.x & 63) == 0 && live) atomicAdd(&pr.st->live_c, live);
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
  if (which == 1 && !(pr.want & kIcpWantRatio)) return;  // (the grid of all reference points serves the coverage query only)
  const Grid2 g = g2_view(which == 0 ? pr.gr : (which == 1 ? pr.ga : pr.gm[1]), n);
  GridDesc d = pr.geom->g;
  const int nround = (n + kBlock - 1) / kBlock;
  for (int r = blockIdx.x; r < nround; r += gridDim.x) {  // whole workgroups stay together (shuffles)
    int i = r * kBlock + threadIdx.x;
    bool in = i < n;
    float4 p = in ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    g2_insert_point(d, g, in, i, p, &pr.st->err, which == 2 ? pr.mloc : nullptr);
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
    part[threadIdx.x] = 0;  // (the solve phase of iteration k_done + 2 adds here next)
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

// ---- raw moments of the matched pairs per component (n, sum p, sum q, sum p(x)q, sum |p-q|^2),
// added by whichever phase settles a query's neighbour: the budget test for the cached ones, the
// searches for the rest. Each workgroup first accumulates into an LDS table keyed by component
// (64-bit integer LDS adds; neighbouring queries mostly share a component), then flushes one global
// integer atomic per (component, accumulator). Integer sums: the result does not depend on any
// arrival order.
constexpr int kMomSlots = 128;        // wide budget-test workgroups (256 threads, a few rounds of queries)
constexpr int kMomSlotsSearch = 16;  // a wide search item settles 8 or 32 pairs: a small table keeps its reset and drain short
constexpr int kMomSlotsLoop = 512;   // the unit-resident loop: one table for ALL pairs of an iteration
constexpr int kHeavyBit = 1 << 30;   // tile path: flag of a work-list entry (query ids stay below 2^26)
constexpr int kMomProbes = 8;        // slots a component may be looked for in before its pair goes to global memory
// Layout of the per-component records. The moment accumulators are RECORD-major ([C, kAcc]): a workgroup's table
// flush and a search item's pairs add 17 consecutive words of one component -- three cache lines of atomics; in the
// component-minor form (17 lines per record) the same adds made a chunk of iterations 7 % longer under load
// (A/B on one box, round 4). The transforms (Tb) and steps (Rt) are component-minor ([16, C], [12, C]): they are read
// by one thread per component (solve) or through the L1 by every point of the component (apply).
__device__ __forceinline__ int64_t acc_ix(int q, int c, int64_t C) { return (int64_t)c * kAcc + q; }
__device__ __forceinline__ int64_t acc_stride(int64_t C) { return 1; }
__device__ __forceinline__ int64_t rt_ix(int q, int c, int64_t C) { return (int64_t)q * C + c; }
__device__ __forceinline__ int64_t tb_ix(int q, int c, int64_t C) { return (int64_t)q * C + c; }
struct MomTable {
  int* skey;                      // LDS [nslots]
  unsigned long long* sacc;       // LDS [nslots * 17]
  unsigned long long* acc;        // global [C, kAcc] of the iteration
  int C;                          // components (keys outside [0, C) are never flushed)
  int nslots;                     // power of two
};
__device__ __forceinline__ void mom_init(const MomTable& t) {  // every thread of the workgroup; then a barrier
  for (int q = threadIdx.x; q < t.nslots; q += blockDim.x) t.skey[q] = -1;
  for (int q = threadIdx.x; q < t.nslots * 17; q += blockDim.x) t.sacc[q] = 0ULL;
}
// the 17 fixed-point terms of one pair (moving point pf of component c, reference point qf) relative to
// the component's origin
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
// the component's LDS slot (linear probing), -1 if the table is full
__device__ __forceinline__ int mom_slot(const MomTable& t, int c) {
  int slot = (int)(((unsigned)c * 2654435761u) >> 20) & (t.nslots - 1);
  // (a bounded walk: a unit with more components than slots fills the table, and the rest of its pairs
  //  then goes to the global accumulators at once instead of probing every slot first)
  for (int i = 0; i < kMomProbes; ++i) {
    int old = atomicCAS(&t.skey[slot], -1, c);
    if (old == -1 || old == c) return slot;
    slot = (slot + 1) & (t.nslots - 1);
  }
  return -1;
}
// 17 sums of component c into the workgroup's table (global atomics if it is full)
__device__ __forceinline__ void mom_put(const MomTable& t, int c, const long long v[17]) {
  const int found = mom_slot(t, c);
  if (found >= 0) {
#pragma unroll
    for (int q2 = 0; q2 < 17; ++q2) atomicAdd(&t.sacc[found * 17 + q2], (unsigned long long)v[q2]);
  } else {
#pragma unroll
    for (int q2 = 0; q2 < 17; ++q2) atomicAdd(&t.acc[acc_ix(q2, c, t.C)], (unsigned long long)v[q2]);
  }
}
// Register-lean form for single lanes (the searches' result lanes): the slot first, then every term is
// formed and added at once (no array of 17 64-bit terms kept alive across shuffles).
__device__ __forceinline__ void mom_add_stream(const IcpProblem& pr, const MomTable& t, float4 pf, float4 qf, int c,
                                               double lim2) {
  const int found = mom_slot(t, c);
  unsigned long long* base = found >= 0 ? t.sacc + found * 17 : t.acc + acc_ix(0, c, t.C);
  const int64_t sd = found >= 0 ? 1 : acc_stride(t.C);  // (the global accumulators: one record per component)
  const float4 og = pr.corg[c];
  const double p0 = (double)pf.y - (double)og.y, p1 = (double)pf.z - (double)og.z, p2 = (double)pf.w - (double)og.w;
  const double q0 = (double)qf.y - (double)og.y, q1 = (double)qf.z - (double)og.z, q2 = (double)qf.w - (double)og.w;
  if (p0 * p0 + p1 * p1 + p2 * p2 > lim2 || q0 * q0 + q1 * q1 + q2 * q2 > lim2) pr.st->err = 1;
  atomicAdd(base + 0 * sd, 1ULL);
  atomicAdd(base + 1 * sd, (unsigned long long)to_fx(p0));
  atomicAdd(base + 2 * sd, (unsigned long long)to_fx(p1));
  atomicAdd(base + 3 * sd, (unsigned long long)to_fx(p2));
  atomicAdd(base + 4 * sd, (unsigned long long)to_fx(q0));
  atomicAdd(base + 5 * sd, (unsigned long long)to_fx(q1));
  atomicAdd(base + 6 * sd, (unsigned long long)to_fx(q2));
  atomicAdd(base + 7 * sd, (unsigned long long)to_fx(p0 * q0));
  atomicAdd(base + 8 * sd, (unsigned long long)to_fx(p0 * q1));
  atomicAdd(base + 9 * sd, (unsigned long long)to_fx(p0 * q2));
  atomicAdd(base + 10 * sd, (unsigned long long)to_fx(p1 * q0));
  atomicAdd(base + 11 * sd, (unsigned long long)to_fx(p1 * q1));
  atomicAdd(base + 12 * sd, (unsigned long long)to_fx(p1 * q2));
  atomicAdd(base + 13 * sd, (unsigned long long)to_fx(p2 * q0));
  atomicAdd(base + 14 * sd, (unsigned long long)to_fx(p2 * q1));
  atomicAdd(base + 15 * sd, (unsigned long long)to_fx(p2 * q2));
  const double e0 = p0 - q0, e1 = p1 - q1, e2 = p2 - q2;
  atomicAdd(base + 16 * sd, (unsigned long long)to_fx(e0 * e0 + e1 * e1 + e2 * e2));
}
// The same for a whole wave (every lane calls; `active` lanes hold a pair). Neighbouring queries mostly
// belong to one component: when all active lanes agree on it their terms are summed over the wave first
// (DPP, wave_sum_fx_lane63) and ONE lane adds the 17 totals -- otherwise 64 lanes queue up on the same 17
// words. Several components in the wave (the keys with thousands of small clusters: always): every lane
// adds its own terms; lanes of one component meet at the same words, a few deep.
__device__ __forceinline__ void mom_add_wave(const IcpProblem& pr, const MomTable& t, bool active, float4 pf,
                                             float4 qf, int c, double lim2) {
  const unsigned long long am = __ballot(active);
  if (am == 0) return;
  const int leader = __ffsll((long long)am) - 1;
  const int c0 = __shfl(c, leader, 64);
  const bool uniform = __ballot(active && c != c0) == 0 && __popcll(am) >= 8;
  long long v[17];
#pragma unroll
  for (int q = 0; q < 17; ++q) v[q] = 0;
  if (active) mom_terms(pr, pf, qf, c, lim2, v);
  if (!uniform) {
    if (active) mom_put(t, c, v);
    return;
  }
#pragma unroll
  for (int q = 0; q < 17; ++q) v[q] = wave_sum_fx_lane63(v[q]);
  if ((threadIdx.x & 63) == 63) mom_put(t, c0, v);
}

__device__ __forceinline__ void mom_flush(const MomTable& t) {  // after a barrier; every thread
  for (int q = threadIdx.x; q < t.nslots * 17; q += blockDim.x) {
    int sl = q / 17, a = q - sl * 17;
    int key = t.skey[sl];
    if (key >= 0 && key < t.C) {
      const unsigned long long v = t.sacc[q];
      if (v) atomicAdd(&t.acc[acc_ix(a, key, t.C)], v);
    }
  }
}

// diagnostics (built with -DPCS_ICP_STATS only): [0] queries tested, [1] budgets that had run out, [2] of those kept
// by the second level, [3] queries handed to a search
__device__ unsigned long long g_icp_stats[16];

// ---- second level of the verified cache (NnAux, pcs_grid2.hpp): query `qid` (forward: moving point qid against the
// reference cloud; backward: reference point qid - Mn against the moving cloud) whose budget `bud` (already less
// what moved in the last iteration) has run out. Everything outside {partner s, the three cached others} is at least
// bud + c away. The cached points are measured where they are now -- the 4-D squared distance exactly as a search
// forms it -- and the nearest of them by (dist2, index) is THE nearest point while it is nearer than that bound and
// inside the radius. Then: neighbour and budget of iteration k and the cache are rewritten (the old partner takes
// the new one's place among the others), *m_p / *m_q = the pair (moving point, reference point), true. Otherwise
// nothing is written: the caller sends the query to a search. iteration parity from `nn_o` / `bud_o`.
__device__ __forceinline__ bool icp_second_level_impl(const IcpProblem& pr, const GridDesc& g, bool fwd, int qi,
                                                      int32_t* __restrict__ nn_o, float* __restrict__ bud_o,
                                                      float4 p, int s, float4 sp, float dist2, float c0s, float bud,
                                                      float4& m_p, float4& m_q, int& winner) {
  NnAux* __restrict__ auxp = (fwd ? pr.aux_f : pr.aux_b) + qi;
  if (pr.want & kIcpNoSecondLevel) return false;  // (pcs_icp_set_cache_levels(1): the cross-check of the tests)
  const NnAux ax = *auxp;
  const float Lrem = bud + ax.c;  // nothing outside the cache is nearer than this (spatial distance)
  if (!(Lrem > 0.f)) return false;
  const float4* __restrict__ hpts = fwd ? pr.ref : (const float4*)pr.mov;
  int bi = s;
  float bd = dist2, second = __builtin_inff();
  float bx = sp.x, by = sp.y, bz = sp.z, bw = sp.w;  // the nearest cached point so far (scalars: see below)
  int slot = 0;  // which cached point won: 0 = the partner
  const int o[3] = {ax.o1, ax.o2, ax.o3};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const bool have = o[q] >= 0;
    const float4 c = hpts[have ? o[q] : s];
    const float d0 = c.x - p.x, d1 = c.y - p.y, d2 = c.z - p.z, d3 = c.w - p.w;
    float dd = 0.f;
    dd = dd + d0 * d0;
    dd = dd + d1 * d1;
    dd = dd + d2 * d2;
    dd = dd + d3 * d3;
    const bool better = have && (dd < bd || (dd == bd && o[q] < bi));
    second = have ? fminf(second, better ? bd : dd) : second;
    bd = better ? dd : bd;
    bi = better ? o[q] : bi;
    bx = better ? c.x : bx;
    by = better ? c.y : by;
    bz = better ? c.z : bz;
    bw = better ? c.w : bw;
    slot = better ? q + 1 : slot;
  }
  const float4 bp = make_float4(bx, by, bz, bw);
  const float E = 1e-6f * g.radius2 + 1e-9f;
  const float s1 = sqrtf(fmaxf(bd - c0s, 0.f) + E);
  if (!(bd <= g.radius2) || !(s1 + 2e-5f < Lrem)) return false;
  // how long the new partner stays the nearest: half the gap to the runner-up -- the second of the cached points, or
  // whatever lies outside the cache
  const float nb = icp_cache_budget(bd - c0s, fminf(second - c0s, Lrem * Lrem), g.radius2);
  nn_o[qi] = bi;
  bud_o[qi] = nb;
  NnAux nx = ax;
  if (slot == 1) nx.o1 = s;
  else if (slot == 2) nx.o2 = s;
  else if (slot == 3) nx.o3 = s;
  nx.c = Lrem - nb;
  *auxp = nx;
  m_p = fwd ? p : bp;
  m_q = fwd ? bp : p;
  winner = bi;
  return true;
}

// ---- phase 1: the budget test. Queries [0,Mn): forward (moving -> reference, df = +frame_offset);
// queries [Mn, Mn+Rn): backward (reference -> moving, df = -frame_offset). The neighbours are a VERIFIED
// CACHE: every search also returns how far things may move before its answer can change (half the gap
// to the runner-up, or the distance by which the nearest point misses the radius). From iteration 2 on
// a query first subtracts what has moved since -- its own last step (forward) or the largest step of any
// moving point near it (backward: the 8^3-cell blocks around it in the lossy step table, which can only
// over-estimate) -- and keeps its neighbour without touching a grid while the budget lasts (98 % of
// the searches of a converging loop return the partner they returned before). The rest is compacted into
// work lists, by direction: cnt4[0..1] queries for the cooperative search, cnt4[2..3] queries whose old
// partner is near enough to be searched around (g2_nearest_seeded8). Exact: the same neighbours as a
// full search, always. Workgroup wgx of nwg takes the rounds wgx, wgx + nwg, ... of BS queries; s_step
// is the step table of iteration k-1 (in LDS when the workgroup serves backward queries).
// What the budget test of one query needs from memory before it can decide anything: the query, its partner of the
// last iteration where it is now, its budget, the pair's component. Two levels of dependent loads (partner and
// component follow the partner's index) that the wide launch puts in flight BEFORE its workgroup stages the step table
// and passes its barrier (icp_nn_kernel): the chain of a settled query is then table | query -> partner -> origin
// instead of table -> query -> partner -> origin.
struct BudgetPre {
  float4 p, sp;
  int s, c_pair;
  float bud;  // forward queries: already less what the query has moved since the last iteration
};
__device__ __forceinline__ BudgetPre icp_budget_load(const IcpProblem& pr, int k, int Mn, int Rn, int qid) {
  BudgetPre q;
  q.p = q.sp = make_float4(0.f, 0.f, 0.f, 0.f);
  q.s = -1;
  q.c_pair = 0;
  q.bud = 0.f;
  if (qid >= Mn + Rn) return q;
  const bool fwd = qid < Mn;
  if (fwd) {
    q.p = pr.mov[qid];
    const float4 o = pr.prev[qid];
    q.s = (pr.nn_f + (size_t)((k - 1) & 1) * pr.capM)[qid];
    q.bud = (pr.bud_f + (size_t)((k - 1) & 1) * pr.capM)[qid];
    const float dx = q.p.y - o.y, dy = q.p.z - o.z, dz = q.p.w - o.w;
    const float disp = sqrtf((dx * dx + dy * dy) + dz * dz);
    q.bud = q.bud - disp * 1.000001f - 1e-7f;
  } else {
    q.p = pr.ref[qid - Mn];
    q.s = (pr.nn_b + (size_t)((k - 1) & 1) * pr.capR)[qid - Mn];
    q.bud = (pr.bud_b + (size_t)((k - 1) & 1) * pr.capR)[qid - Mn];
  }
  // (the pair's component is requested together with the partner, not after the distance test: one level less
  //  in the chain of dependent loads that a settled query is -- point / partner index, partner, origin)
  q.c_pair = (fwd || q.s >= 0) ? pr.comp[fwd ? qid : q.s] : 0;
  if (q.s >= 0) q.sp = fwd ? pr.ref[q.s] : pr.mov[q.s];
  return q;
}

template <int BS, bool SPLIT = true>
__device__ __forceinline__ void icp_budget_phase(const IcpProblem& pr, int k, int Mn, int Rn, int wgx, int nwg,
                                                 const GridDesc& g, const MomTable& mt,
                                                 const unsigned int* __restrict__ s_step, int* cnt4,
                                                 const BudgetPre* first_round = nullptr) {
  const int nq = Mn + Rn;
  const int nround = (nq + BS - 1) / BS;
  const double lim2 = icp_limit_sq(nq);
  const int lane = threadIdx.x & 63;
  int32_t* nn_f = pr.nn_f + (size_t)(k & 1) * pr.capM;
  int32_t* nn_b = pr.nn_b + (size_t)(k & 1) * pr.capR;
  float* bud_f = pr.bud_f + (size_t)(k & 1) * pr.capM;
  float* bud_b = pr.bud_b + (size_t)(k & 1) * pr.capR;
  const Grid2 g_ref = g2_view(pr.gr, Rn), g_mov = g2_view(pr.gm[pr.st->gcur[k & 1] & 1], Mn);
  for (int r = wgx; r < nround; r += nwg) {  // whole waves stay together (ballots below)
    const int qid = r * BS + threadIdx.x;
    const bool valid = qid < nq;
    const bool fwd = qid < Mn;
    bool to_seeded = false, to_coop = false;
    bool heavy = false;  // (SPLIT = false) a failed query without a partner inside the radius to start from
    bool stat_l1 = false, stat_l2 = false;  // (diagnostics)
    int win2 = -1;  // the cached point the second level chose
    bool m_act = false;  // a settled pair whose moments are due: (moving point, reference point, component)
    float4 m_p = make_float4(0.f, 0.f, 0.f, 0.f), m_q = m_p;
    int m_c = 0;
    // (the first round of a wide launch arrives loaded: see BudgetPre)
    const BudgetPre ld = (first_round && r == wgx) ? *first_round : icp_budget_load(pr, k, Mn, Rn, qid);
    if (valid) {
      const float4 p = ld.p, sp = ld.sp;
      const int s = ld.s;
      float bud = ld.bud;
      if (!fwd) {
        // a moving point that can matter for this query lies in the 3 x 3 x 3 blocks of 4^3 cells
        // around it -- 12 cells per axis, i.e. at most 2 x 2 x 2 blocks of 8^3 cells: the largest
        // step recorded there bounds how far any such point has moved since iteration k-1
        const int cx = (int)floorf((p.y - g.lo[0]) / g.h), cy = (int)floorf((p.z - g.lo[1]) / g.h),
                  cz = (int)floorf((p.w - g.lo[2]) / g.h);
        const int sx = (4 * ((cx >> 2) - 1)) >> 3, sy = (4 * ((cy >> 2) - 1)) >> 3, sz = (4 * ((cz >> 2) - 1)) >> 3;
        unsigned int m = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          m = max(m, s_step[icp_step_slot(sx + (q & 1), sy + ((q >> 1) & 1), sz + (q >> 2))]);
        bud = bud - __uint_as_float(m) * 1.000001f - 1e-7f;
      }
      float dist2 = __builtin_inff();
      float c0s = 0.f;  // the frame term of the distance
      const int c_pair = ld.c_pair;
      if (s >= 0) {
        float d0 = sp.x - p.x, d1 = sp.y - p.y, d2 = sp.z - p.z, d3 = sp.w - p.w;
        dist2 = 0.f;
        dist2 = dist2 + d0 * d0;
        dist2 = dist2 + d1 * d1;
        dist2 = dist2 + d2 * d2;
        dist2 = dist2 + d3 * d3;
        c0s = d0 * d0;
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
          m_c = c_pair;
        }
      } else if (s >= 0 && icp_second_level_impl(pr, g, fwd, fwd ? qid : qid - Mn, fwd ? nn_f : nn_b, fwd ? bud_f : bud_b, p, s, sp, dist2, c0s, bud, m_p, m_q, win2)) {
        // the budget has run out, but the nearest of the four cached points, measured where they are now, is still
        // nearer than anything outside the cache can be (icp_second_level wrote neighbour, budget and cache)
        m_act = true;
        stat_l1 = stat_l2 = true;
        m_c = fwd ? c_pair : pr.comp[win2];  // (backward: the component of the moving point that won)
      } else if (SPLIT && s >= 0 && g2_seed_admissible(g, fwd ? g_ref : g_mov, p, fwd ? pr.df : -pr.df, dist2, pr.mu)) {
        to_seeded = true;  // (the tile path has ONE kind of search: SPLIT = false)
        stat_l1 = true;
      } else {
        to_coop = true;
        stat_l1 = s >= 0;
        if (!SPLIT) {
          // a query whose old partner is so near that the ball through it (+ mu) stays inside the 27 cells
          // around the query is searched by one lane; the others take a wave each (tile_nearest_wave)
          const float r = sqrtf(fmaxf(dist2 - c0s, 0.f)) + pr.mu;
          heavy = !(s >= 0 && dist2 <= g.radius2 && g.h * g.h * 0.998f > r * r + (1e-5f * g.radius2 + 1e-6f));
        }
      }
    }
    mom_add_wave(pr, mt, m_act, m_p, m_q, m_c, lim2);
#ifdef PCS_ICP_STATS
    {
      const unsigned long long mv = __ballot(valid), ms_ = __ballot(to_seeded || to_coop), m2 = __ballot(stat_l2),
                               m1 = __ballot(stat_l1);
      const unsigned long long mb = ~__ballot(fwd);  // (the same four of the backward queries alone)
      if (lane == 0) {
        atomicAdd(&g_icp_stats[0], (unsigned long long)__popcll(mv));
        atomicAdd(&g_icp_stats[1], (unsigned long long)__popcll(m1));
        atomicAdd(&g_icp_stats[2], (unsigned long long)__popcll(m2));
        atomicAdd(&g_icp_stats[3], (unsigned long long)__popcll(ms_));
        atomicAdd(&g_icp_stats[4], (unsigned long long)__popcll(mv & mb));
        atomicAdd(&g_icp_stats[5], (unsigned long long)__popcll(m1 & mb));
        atomicAdd(&g_icp_stats[6], (unsigned long long)__popcll(m2 & mb));
        atomicAdd(&g_icp_stats[7], (unsigned long long)__popcll(ms_ & mb));
      }
    }
#endif
    // the work lists, by direction (a wave is all forward or all backward, except the one at Mn)
    const unsigned long long mf = __ballot(fwd);
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
      const unsigned long long md = dir ? ~mf : mf;
      const unsigned long long ms = __ballot(to_seeded) & md, mc = __ballot(to_coop) & md;
      const int off = dir ? Mn : 0;
      if (ms) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&cnt4[2 + dir], __popcll(ms));
        base = __shfl(base, 0, 64);
        if (to_seeded && fwd == (dir == 0)) pr.wls[off + base + __popcll(ms & ((1ULL << lane) - 1ULL))] = qid;
      }
      if (mc) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&cnt4[dir], __popcll(mc));
        base = __shfl(base, 0, 64);
        if (to_coop && fwd == (dir == 0))
          pr.wl[off + base + __popcll(mc & ((1ULL << lane) - 1ULL))] = qid | (heavy ? kHeavyBit : 0);
      }
    }
  }
}

// Iteration 1 of a registration: no partner yet. Half of all cooperative searches are these -- 32 lanes
// walking 27 coarse blocks per query. A point of the OTHER cloud that lies in the query's own fine cell is a
// seed as good as an old partner: where the search's own admission test accepts it, the query goes to the
// eight-lane search around it (exact: the same result as the cooperative search), the seed's index parked in
// the neighbour buffer that an iteration 0 would have written; the others take the cooperative search.
template <int BS>
__device__ __forceinline__ void icp_first_phase(const IcpProblem& pr, int Mn, int Rn, int wgx, int nwg,
                                                const GridDesc& g, int* cnt4) {
  const int nq = Mn + Rn;
  const int nround = (nq + BS - 1) / BS;
  const int lane = threadIdx.x & 63;
  const Grid2 g_ref = g2_view(pr.gr, Rn), g_mov = g2_view(pr.gm[1], Mn);
  for (int r = wgx; r < nround; r += nwg) {  // whole waves stay together (ballots below)
    const int qid = r * BS + threadIdx.x;
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
      int seed = -1;
      float seed_d2 = __builtin_inff();
      if (fc >= 0 && fc < kFrameCells && cx >= 0 && cy >= 0 && cz >= 0) {
        // the first point of the query's own cell; with an empty cell the nearest of the first points of the 26
        // cells around it. Any point of the other cloud inside the radius will do: it bounds the search (the
        // ball through it instead of the ball of the whole radius: most first-iteration searches were the latter)
        auto probe = [&](int x, int y, int z) {
          uint32_t cnt_w = 0, pad_w = 0;
          const int fs = g2_find(gh.fslot, gh.Hf - 1, pack_cell(fc, x, y, z), &cnt_w, &pad_w);
          if (fs < 0 || cnt_w == 0) return;
          // the cell's point with the smallest original index (the rows of a cell are in arrival order, which varies
          // from run to run: with row 0 as the seed the budgets and the work lists of a registration were not
          // reproducible, only its results)
          const int cnt = min((int)cnt_w, kCellCap);
          const int4* __restrict__ iv = (const int4*)(gh.sidx + (size_t)fs * kCellCap);
          int row = 0, best_i = INT_MAX;
          for (int j = 0; j < cnt; j += 4) {
            const int4 o = iv[j >> 2];
            if (o.x >= 0 && o.x < best_i) { best_i = o.x; row = j; }
            if (j + 1 < cnt && o.y >= 0 && o.y < best_i) { best_i = o.y; row = j + 1; }
            if (j + 2 < cnt && o.z >= 0 && o.z < best_i) { best_i = o.z; row = j + 2; }
            if (j + 3 < cnt && o.w >= 0 && o.w < best_i) { best_i = o.w; row = j + 3; }
          }
          if (best_i == INT_MAX) return;
          const float4 sp = gh.spts[(size_t)fs * kCellCap + row];
          const float d0 = sp.x - p.x, d1 = sp.y - p.y, d2 = sp.z - p.z, d3 = sp.w - p.w;
          float dist2 = 0.f;
          dist2 = dist2 + d0 * d0;
          dist2 = dist2 + d1 * d1;
          dist2 = dist2 + d2 * d2;
          dist2 = dist2 + d3 * d3;
          if (dist2 < seed_d2) {
            seed_d2 = dist2;
            seed = best_i;
          }
        };
        probe(cx, cy, cz);
        if (seed < 0) {
#pragma unroll 1
          for (int q = 0; q < 27; ++q) {
            if (q == 13) continue;
            const int x = cx + q % 3 - 1, y = cy + (q / 3) % 3 - 1, z = cz + q / 9 - 1;
            if (x >= 0 && y >= 0 && z >= 0) probe(x, y, z);
          }
        }
      }
      if (!(seed_d2 <= g.radius2)) seed = -1;  // (a seed must lie inside the radius)
      // buffer (k - 1) & 1 = 0: where BOTH searches read their seeds (-1: none)
      if (fwd) pr.nn_f[qid] = seed;
      else pr.nn_b[qid - Mn] = seed;
      to_seeded = seed >= 0 && g2_seed_admissible(g, gh, p, dfq, seed_d2, pr.mu);
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
        if (lane == 0) base = atomicAdd(&cnt4[2 + dir], __popcll(ms));
        base = __shfl(base, 0, 64);
        if (to_seeded && fwd == (dir == 0)) pr.wls[off + base + __popcll(ms & ((1ULL << lane) - 1ULL))] = qid;
      }
      if (mc) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&cnt4[dir], __popcll(mc));
        base = __shfl(base, 0, 64);
        if (to_coop && fwd == (dir == 0)) pr.wl[off + base + __popcll(mc & ((1ULL << lane) - 1ULL))] = qid;
      }
    }
  }
}

// ---- phase 2: one round of one work list -- BS / 32 cooperative or BS / 8 seeded queries of one
// direction (0: forward queries against the reference grid, 1: backward against the moving grid), so
// that everything is uniform in the workgroup: one grid, one instance of the search in the code (the
// two-direction form spilled ~400 scalar registers). nw = length of the list.
template <int BS>
__device__ __forceinline__ void icp_search_item(const IcpProblem& pr, IcpState* st, int k, int Mn, int Rn,
                                                const GridDesc& g, bool coop, int dir, int r, int nw,
                                                const MomTable& mt) {
  const bool fwd = dir == 0;
  int32_t* __restrict__ nn_o = fwd ? pr.nn_f + (size_t)(k & 1) * pr.capM : pr.nn_b + (size_t)(k & 1) * pr.capR;
  float* __restrict__ bud_o = fwd ? pr.bud_f + (size_t)(k & 1) * pr.capM : pr.bud_b + (size_t)(k & 1) * pr.capR;
  NnAux* __restrict__ aux_o = fwd ? pr.aux_f : pr.aux_b;
  const Grid2 gs = fwd ? g2_view(pr.gr, Rn) : g2_view(pr.gm[st->gcur[k & 1] & 1], Mn);  // the hashed cloud
  const float4* __restrict__ qpts = fwd ? (const float4*)pr.mov : pr.ref;    // the queries' cloud
  const float4* __restrict__ hpts = fwd ? pr.ref : (const float4*)pr.mov;    // the hashed cloud's points
  const int qoff = fwd ? 0 : Mn;
  const int df = fwd ? pr.df : -pr.df;
  const double lim2 = icp_limit_sq(Mn + Rn);
  if (coop) {
    const int sub = threadIdx.x & 31;
    const int32_t* __restrict__ list = pr.wl + qoff;
    const int wq = r * (BS / 32) + (threadIdx.x >> 5);
    const bool valid = wq < nw;
    const int qi = valid ? list[wq] - qoff : 0;  // index in the queries' cloud
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    float seed_d2 = __builtin_inff();
    if (valid) {
      p = qpts[qi];
      {  // the partner of the last iteration (iteration 1: the seed icp_first_phase found), where it is now: an
         // upper bound of the nearest distance
        const int32_t* __restrict__ seedc = fwd ? pr.nn_f + (size_t)((k - 1) & 1) * pr.capM
                                                : pr.nn_b + (size_t)((k - 1) & 1) * pr.capR;
        const int sd = seedc[qi];
        if (sd >= 0) {
          const float4 sp = hpts[sd];
          const float d0 = sp.x - p.x, d1 = sp.y - p.y, d2 = sp.z - p.z, d3 = sp.w - p.w;
          float dist2 = 0.f;
          dist2 = dist2 + d0 * d0;
          dist2 = dist2 + d1 * d1;
          dist2 = dist2 + d2 * d2;
          dist2 = dist2 + d3 * d3;
          seed_d2 = dist2;
        }
      }
    }
    float bud = 0.f;
    NnAux ax;
    const int res = g2_nearest_coop<true>(g, gs, p, df, valid, &st->err, nullptr, pr.mu, &bud, seed_d2, &ax);
#ifdef PCS_ICP_STATS
    if (valid && sub == 0) {
      atomicAdd(&g_icp_stats[8], 1ULL);
      if (k > 1) atomicAdd(&g_icp_stats[seed_d2 < __builtin_inff() ? 11 : 10], 1ULL);
      if (k > 1 && res < 0) atomicAdd(&g_icp_stats[12], 1ULL);
      if (k == 1) atomicAdd(&g_icp_stats[13], 1ULL);
      if (k == 1 && res < 0) atomicAdd(&g_icp_stats[14], 1ULL);
    }
#endif
    if (valid && sub == 0) {
      nn_o[qi] = res;
      bud_o[qi] = bud;
      aux_o[qi] = ax;
      if (res >= 0) {  // (moving point, reference point, the moving point's component)
        if (fwd) mom_add_stream(pr, mt, p, hpts[res], pr.comp[qi], lim2);
        else mom_add_stream(pr, mt, hpts[res], p, pr.comp[res], lim2);
      }
    }
  } else {
    const int32_t* __restrict__ seed = fwd ? pr.nn_f + (size_t)((k - 1) & 1) * pr.capM
                                           : pr.nn_b + (size_t)((k - 1) & 1) * pr.capR;
    const int32_t* __restrict__ list = pr.wls + qoff;
    const int wq = r * (BS / 8) + (threadIdx.x >> 3);
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
    NnAux ax;
    const bool ok = g2_nearest_seeded8(g, gs, p, df, sp, sd, valid, pr.mu, &res, &bud, &ax);
#ifdef PCS_ICP_STATS
    if (valid && (threadIdx.x & 7) == 0) atomicAdd(&g_icp_stats[9], 1ULL);
#endif
    if (valid && (threadIdx.x & 7) == 0) {
      if (!ok) st->err = 4;  // cannot happen: the admission test is the search's own arithmetic
      nn_o[qi] = res;
      bud_o[qi] = bud;
      aux_o[qi] = ax;
      if (res >= 0) {
        if (fwd) mom_add_stream(pr, mt, p, hpts[res], pr.comp[qi], lim2);
        else mom_add_stream(pr, mt, hpts[res], p, pr.comp[res], lim2);
      }
    }
  }
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
  if (iters < 1 || !(pr.want & kIcpWantL1)) return;
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
    long long a[7];  // n, sum p, sum q
#pragma unroll
    for (int q = 0; q < 7; ++q) a[q] = acc[acc_ix(q, c, (int)pr.C)];
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
      double mean = ((double)acc[acc_ix(17, c, (int)pr.C)] * kFxInv) / (double)a[0];
      v = fmin(fmax(dist, mean - trunc), mean + trunc);
    }
    atomicAdd((unsigned long long*)&acc[acc_ix(MODE == 0 ? 17 : 18, c, (int)pr.C)],
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

// ---- phase 3: one THREAD per component solves the regularised Kabsch step from the completed moments:
// rotation and translation of the step (Rt, read by every point of the component in the apply phase),
// the updated transform (read buffer k-1, write buffer k), the loss term; the other moment buffer is
// zeroed for the next iteration. The components' loss terms leave as one fixed-point integer add per
// wave, spread over kLossParts words (`part` picks the word): order-free.
template <int BS>
__device__ __forceinline__ void icp_solve_phase(const IcpProblem& pr, IcpState* st, int k, int wgx, int nwg, int part) {
  const int64_t C = pr.C;
  const double reg = pr.reg;
  long long* __restrict__ acc_k = pr.acc + (size_t)(k & 1) * C * kAcc;
  long long* __restrict__ acc_n = pr.acc + (size_t)((k + 1) & 1) * C * kAcc;
  const double* __restrict__ T_in = pr.Tb + (size_t)((k - 1) & 1) * C * 16;
  double* __restrict__ T_out = pr.Tb + (size_t)(k & 1) * C * 16;
  if (wgx == 0 && threadIdx.x == 0) {
    // the moving grid of the next iteration (nothing of this registration runs beside the solve phase): the apply
    // phase of this iteration moves the points in place (g2_move_point) unless the cell changes since the last
    // rebuild have added up to a quarter of the points -- tombstones and visited cells only accumulate -- or the
    // overflow list has grown (every query scans it): then it rebuilds into the other, empty grid
    int Mn, Rn, Ra;
    icp_sizes(pr, Mn, Rn, Ra);
    const int cur = st->gcur[k & 1] & 1;
    const int ovf = *pr.gm[cur].ovf;
    if (st->grebuild) st->govf0 = ovf;  // (the grid is fresh: built by the last apply phase, or by the set-up)
    const bool reb = (uint64_t)st->gmoves * 4u > (uint64_t)Mn || ovf > st->govf0 || (pr.want & kIcpRebuildAlways);
    st->grebuild = reb ? 1 : 0;
    if (reb) st->gmoves = 0u;
    st->gcur[(k + 1) & 1] = reb ? (cur ^ 1) : cur;
  }
  long long loss_fx = 0;  // loss terms of the components this thread owns (fixed point)
  long long loss_ck = 0;  // the same at 10 fractional bits (overflow check of the sum, see icp_decide)
  const int nround = (int)((C + BS - 1) / BS);
  for (int r = wgx; r < nround; r += nwg) {
    const int64_t c = (int64_t)r * BS + threadIdx.x;
    if (c < C && pr.first[c] != INT_MAX) {  // components with a moving point (the others keep the identity the reset wrote)
      // the transforms are component-minor (thread c reads word q next to its neighbours'); the moments one record
      // per component (see acc_ix)
      long long a[17];
      double Tc[16];
#pragma unroll
      for (int q = 0; q < 17; ++q) a[q] = acc_k[acc_ix(q, c, (int)C)];
#pragma unroll
      for (int q = 0; q < 16; ++q) Tc[q] = T_in[tb_ix(q, c, C)];
      double R[9], t[3], Tn[16];
      const float4 og = pr.corg[c];
      const double org[3] = {(double)og.y, (double)og.z, (double)og.w};
      double ls;
      icp_solve_component(a, org, Tc, reg, R, t, Tn, &ls);
      if (!(fabs(ls) < 6.0e7)) st->err = 1;  // (beyond the fixed-point range of a term)
      else {
        loss_fx += to_fx_wide(ls);
        loss_ck += __double2ll_rn(ls * 1024.0);
      }
      double* rt = pr.Rt;
#pragma unroll
      for (int q = 0; q < 9; ++q) rt[rt_ix(q, c, C)] = R[q];
#pragma unroll
      for (int q = 0; q < 3; ++q) rt[rt_ix(9 + q, c, C)] = t[q];
#pragma unroll
      for (int q = 0; q < 16; ++q) T_out[tb_ix(q, c, C)] = Tn[q];
#pragma unroll
      for (int q = 0; q < kAcc; ++q) acc_n[acc_ix(q, c, (int)C)] = 0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    loss_fx += __shfl_xor(loss_fx, o, 64);
    loss_ck += __shfl_xor(loss_ck, o, 64);
  }
  if ((threadIdx.x & 63) == 0 && (loss_fx != 0 || loss_ck != 0)) {
    atomicAdd((unsigned long long*)&pr.lossfx[(size_t)(k & 1) * kLossParts + (part & (kLossParts - 1))],
              (unsigned long long)loss_fx);
    atomicAdd((unsigned long long*)&pr.lossfx[(size_t)(2 + (k & 1)) * kLossParts + (part & (kLossParts - 1))],
              (unsigned long long)loss_ck);
  }
}

// ---- phase 4: every moving point moves by its component's step. The previous positions are kept for the
// l1 passes. The moved points go straight into the grid of the next iteration (the moving grid is
// double-buffered), and the grid this iteration searched is reset; the step table of iteration k takes
// every point's step under the place it moved to.
template <int BS>
__device__ __forceinline__ void icp_apply_phase(const IcpProblem& pr, IcpState* st, int k, int Mn, int wgx, int nwg,
                                                const GridDesc& d) {
  float4* __restrict__ mov = pr.mov;
  float4* __restrict__ prev = pr.prev;
  const int32_t* __restrict__ comp = pr.comp;
  const double* __restrict__ Rt = pr.Rt;
  const int64_t C = pr.C;
  // the moving grid (IcpState::gcur): in place, or -- decided by this iteration's solve phase -- rebuilt into the
  // other grid, which is empty, while the one the searches of this iteration used is reset
  const int cur = st->gcur[k & 1] & 1;
  const bool rebuild = st->grebuild != 0;
  const Grid2 gm_cur = g2_view(pr.gm[cur], Mn);
  const Grid2 gm_oth = g2_view(pr.gm[cur ^ 1], Mn);
  const int64_t tid = wgx * (int64_t)BS + threadIdx.x;
  const int64_t nthreads = (int64_t)nwg * BS;
  if (rebuild) g2_clear(gm_cur, tid, nthreads);
  if (tid == 0) {
    st->iters = k;
    st->wl_n[0] = st->wl_n[1] = 0;  // the work lists of iteration k have been served
    st->ws_n[0] = st->ws_n[1] = 0;
    st->dmax[(k + 1) & 1] = 0u;   // takes the steps of iteration k+1
  }
  {  // the step table of iteration k+1 (nothing reads it during this phase)
    unsigned int* nxt = pr.sstep + (size_t)((k + 1) & 1) * kStepH;
    for (int64_t q = tid; q < kStepH; q += nthreads) nxt[q] = 0u;
  }
  unsigned int* __restrict__ sstep = pr.sstep + (size_t)(k & 1) * kStepH;
  float step = 0.f;  // largest step of this thread's points
  int changed = 0;   // this thread's points that left their cell
  const int nround = (Mn + BS - 1) / BS;
  for (int r = wgx; r < nround; r += nwg) {  // whole workgroups stay together (shuffles)
    const int i = r * BS + threadIdx.x;
    float4 moved = make_float4(0.f, 0.f, 0.f, 0.f), before = moved;
    float st1 = 0.f;
    int sslot = -1;
    if (i < Mn) {
      const int ci = comp[i];
      double R[9], t[3];
#pragma unroll
      for (int q = 0; q < 9; ++q) R[q] = Rt[rt_ix(q, ci, C)];
#pragma unroll
      for (int q = 0; q < 3; ++q) t[q] = Rt[rt_ix(9 + q, ci, C)];
      float4 p = mov[i];
      prev[i] = p;
      before = p;
      double x = p.y, y = p.z, z = p.w;
      float nx = (float)(x * R[0] + y * R[1] + z * R[2] + t[0]);
      float ny = (float)(x * R[3] + y * R[4] + z * R[5] + t[1]);
      float nz = (float)(x * R[6] + y * R[7] + z * R[8] + t[2]);
      moved = make_float4(p.x, nx, ny, nz);
      mov[i] = moved;
      {
        // the verified cache's budget unit: how far this point moved, filed under the place it moved to
        float dx = nx - p.y, dy = ny - p.z, dz = nz - p.w;
        st1 = sqrtf((dx * dx + dy * dy) + dz * dz) * 1.000001f + 1e-7f;
        step = fmaxf(step, st1);
        const int fx = (int)floorf((nx - d.lo[0]) / d.h), fy = (int)floorf((ny - d.lo[1]) / d.h),
                  fz = (int)floorf((nz - d.lo[2]) / d.h);
        sslot = icp_step_slot(fx >> 3, fy >> 3, fz >> 3);
      }
    }
    {
      // neighbours in the array are neighbours in space (the clouds arrive in voxel order): a wave's points fall
      // into a few 8 x 8 x 8-cell blocks, and one lane per block files the largest step of its lanes (four rounds;
      // what is left goes lane by lane) -- one atomic per block and wave instead of one per point
      bool pending = sslot >= 0;
      unsigned long long rem = __ballot(pending);
      for (int round = 0; round < 4 && rem; ++round) {
        const int leader = __ffsll((long long)rem) - 1;
        const int ls = __shfl(sslot, leader, 64);
        const bool mine = pending && sslot == ls;
        const float m = wave_max(mine ? st1 : 0.f);
        if ((int)(threadIdx.x & 63) == leader) atomicMax(&sstep[ls], __float_as_uint(m));
        pending = pending && !mine;
        rem = __ballot(pending);
      }
      if (pending) atomicMax(&sstep[sslot], __float_as_uint(st1));
    }
    if (rebuild) g2_insert_point(d, gm_oth, i < Mn, i, moved, &st->err, pr.mloc);
    else if (g2_move_point(d, gm_cur, i < Mn, i, before, moved, pr.mloc, &st->err)) changed += 1;
  }
  step = wave_max(step);
  if ((threadIdx.x & 63) == 0 && step > 0.f) atomicMax(&st->dmax[k & 1], __float_as_uint(step));
  if (!rebuild) {
    changed = wave_sum_i32_lane63(changed);
    if ((threadIdx.x & 63) == 63 && changed > 0) atomicAdd(&st->gmoves, (unsigned int)changed);
  }
}

// ============================================================ wide launches (blockIdx.y = batch entry)
// Entry e runs iteration kbase[e] + 1 + j, unless that is past wide_limit (the iterations from there on belong
// to the unit-resident loop).
// The gate is read ONCE per workgroup: workgroup 0 of the budget-test launch lowers stop_iter during that
// very launch, and the waves of a workgroup do not start together -- with a per-thread gate some waves of
// a workgroup could leave while the others go on with a moment table in LDS that is only partly
// initialised (the memory fault of round 2: whatever another stream's kernel had left in LDS then served
// as component ids in mom_flush).
__global__ void __launch_bounds__(kBlock)
icp_first_kernel(const IcpProblem* __restrict__ tab, Batch bt, BatchInts kbase) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  IcpState* st = pr.st;
  if (kbase.v[blockIdx.y] != 0 || 1 > pr.max_iter || 1 > st->stop_iter) return;
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  if ((int64_t)blockIdx.x * kBlock >= (int64_t)Mn + Rn) return;
  const GridDesc g = pr.geom->g;
  icp_first_phase<kBlock>(pr, Mn, Rn, blockIdx.x, gridDim.x, g, &st->wl_n[0]);
}

__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8)))
icp_nn_kernel(const IcpProblem* __restrict__ tab, Batch bt, BatchInts kbase, int j, int wide_limit) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  IcpState* st = pr.st;
  const int k = kbase.v[blockIdx.y] + 1 + j;
  // (the sizes are read beside the gate, not after its barrier: one level of dependent loads less)
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  __shared__ int s_go;
  if (threadIdx.x == 0) s_go = (k == 1 || k > wide_limit || k > pr.max_iter || k > st->stop_iter) ? 0 : 1;  // (iteration 1: icp_first_kernel)
  __syncthreads();
  if (!s_go) return;
  const int nq = Mn + Rn;
  const int nround = (nq + kBlock - 1) / kBlock;
  if (blockIdx.x != 0 && (int)blockIdx.x >= nround) return;
  // the first round's queries, partners and components are requested now: they are in flight while the workgroup
  // settles the stop rule, clears its moment table and stages the step table
  const BudgetPre pre = icp_budget_load(pr, k, Mn, Rn, (int)blockIdx.x * kBlock + (int)threadIdx.x);
  // workgroup 0 settles the stop rule of iteration k-1 (its loss terms were written by the previous
  // launch). The other workgroups may run before the stop is known: they only write this iteration's
  // neighbour buffers, which nothing reads after a stop.
  if (blockIdx.x == 0) icp_decide(pr, k - 1);
  __shared__ int skey[kMomSlots];
  __shared__ unsigned long long sacc[kMomSlots * 17];
  __shared__ unsigned int s_step[kStepH];
  const MomTable mt = {skey, sacc, (unsigned long long*)pr.acc + (size_t)(k & 1) * pr.C * kAcc, (int)pr.C, kMomSlots};
  mom_init(mt);
  {
    // a workgroup that serves backward queries (its last round reaches past Mn) copies the 16 KB step table
    // into LDS once -- coalesced, L2-resident: eight hashed slots per query were eight scattered requests
    const unsigned int* __restrict__ sstep = pr.sstep + (size_t)((k - 1) & 1) * kStepH;
    const int r_last = blockIdx.x + ((nround - 1 - (int)blockIdx.x) / (int)gridDim.x) * (int)gridDim.x;
    if ((r_last + 1) * kBlock > Mn)
      for (int q = threadIdx.x; q < kStepH / 4; q += kBlock) ((uint4*)s_step)[q] = ((const uint4*)sstep)[q];
  }
  __syncthreads();
  const GridDesc g = pr.geom->g;
  icp_budget_phase<kBlock>(pr, k, Mn, Rn, blockIdx.x, gridDim.x, g, mt, s_step, &st->wl_n[0], &pre);
  __syncthreads();
  mom_flush(mt);
}

// The searches of one iteration for ALL entries of the batch, as a queue: a work item is one round of
// one (entry, direction, kind) work list -- 8 cooperative or 32 seeded queries -- and the workgroups
// take the items w = blockIdx.x, blockIdx.x + gridDim.x, ... of the concatenated lists (cooperative
// lists first: they are the long ones). Every workgroup first forms the same table of rounds from the
// list lengths on the device (4 words per entry).
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8)))
icp_nn_search_kernel(const IcpProblem* __restrict__ tab, Batch bt, BatchInts kbase, int j, int wide_limit) {
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
      const IcpState* se = pe.st;
      const int ke = kbase.v[e] + 1 + j;
      if (!(ke > wide_limit || ke > pe.max_iter || ke > se->stop_iter)) {
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
  for (int w = blockIdx.x; w < R; w += gridDim.x) {
    int lo = 0, hi = nlist - 1;  // the list of item w: the last t with s_pref[t] <= w
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_pref[mid] <= w) lo = mid; else hi = mid - 1;
    }
    const int r = w - s_pref[lo];
    const bool coop = lo < 2 * bt.n;
    const int tt = coop ? lo : lo - 2 * bt.n;
    const int e = tt >> 1, dir = tt & 1;
    const IcpProblem& pr = tab[bt.idx[e]];
    IcpState* st = pr.st;
    const int k = kbase.v[e] + 1 + j;
    const int nw = coop ? st->wl_n[dir] : st->ws_n[dir];
    int Mn, Rn, Ra;
    icp_sizes(pr, Mn, Rn, Ra);
    const GridDesc g = pr.geom->g;
    const MomTable mt = {skey, sacc, (unsigned long long*)pr.acc + (size_t)(k & 1) * pr.C * kAcc, (int)pr.C,
                         kMomSlotsSearch};
    mom_init(mt);
    __syncthreads();
    icp_search_item<kBlock>(pr, st, k, Mn, Rn, g, coop, dir, r, nw, mt);
    __syncthreads();
    mom_flush(mt);
    __syncthreads();  // (the table is initialised again for the next item)
  }
}

// The per-component records (20 moment words, 16 transform words in, 12 + 16 out) are read and written by
// one thread each: 64 cache lines per wave and load instruction. One-wave workgroups spread those gathers
// over four times as many workgroups as 256-thread ones (which made the launch as long as the budget test).
constexpr int kSolveBlock = 64;
__global__ void __launch_bounds__(kSolveBlock)
icp_solve_kernel(const IcpProblem* __restrict__ tab, Batch bt, BatchInts kbase, int j, int wide_limit) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  IcpState* st = pr.st;
  const int k = kbase.v[blockIdx.y] + 1 + j;
  if (k > wide_limit || k > pr.max_iter || k > st->stop_iter) return;  // (one wave per workgroup: uniform)
  icp_solve_phase<kSolveBlock>(pr, st, k, blockIdx.x, gridDim.x, blockIdx.x);
}

__global__ void __launch_bounds__(kBlock)
icp_apply_kernel(const IcpProblem* __restrict__ tab, Batch bt, BatchInts kbase, int j, int wide_limit) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  IcpState* st = pr.st;
  const int k = kbase.v[blockIdx.y] + 1 + j;
  if (k > wide_limit || k > pr.max_iter || k > st->stop_iter) return;
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  const GridDesc d = pr.geom->g;
  icp_apply_phase<kBlock>(pr, st, k, Mn, blockIdx.x, gridDim.x, d);
}

// ============================================================ the unit-resident loop
// ONE workgroup of 1024 threads per batch entry runs that entry's iterations st->iters + 1, + 2, ... back to
// back: budget test, searches, solve, stop rule, apply -- separated by workgroup barriers only. It leaves
// when the stop rule has fired, at max_iter, after n_it iterations, or when budget_ticks (100 MHz ticks of
// s_memrealtime; 0: no limit) have passed since it started; the entry's state is mirrored to out[entry]
// (may be pinned host memory). Everything the iterations of one entry exchange -- neighbour buffers, work
// lists, moments, transforms, the moving grid -- is written and read by this one workgroup, i.e. through
// one CU's L1 and one XCD's L2: no cross-CU visibility is needed between phases, a workgroup barrier
// (which waits for the wave's outstanding memory operations) orders them.
// The phases are separate (not inlined) functions: each gets the 128 registers of a 1024-thread workgroup
// for itself -- inlined into one body the compiler kept values of every phase alive across all of them
// (256 registers at 512 threads, and spills into scratch memory on top). Their LDS lives at namespace scope.
constexpr int kLoopBlock = 1024;
__shared__ int l_cnt[4];  // work-list lengths of the iteration: cooperative fwd / bwd, seeded fwd / bwd
__shared__ int l_key[kMomSlotsLoop];
__shared__ unsigned long long l_acc[kMomSlotsLoop * 17];
__shared__ unsigned int l_step[kStepH];

__device__ __forceinline__ MomTable loop_mom_table(const IcpProblem& pr, int k) {
  return MomTable{l_key, l_acc, (unsigned long long*)pr.acc + (size_t)(k & 1) * pr.C * kAcc, (int)pr.C, kMomSlotsLoop};
}
// start of an iteration: empty moment table, the step table of iteration k-1 in LDS
__device__ __noinline__ void loop_begin_phase(const IcpProblem* __restrict__ prp, int k) {
  const IcpProblem& pr = *prp;
  mom_init(loop_mom_table(pr, k));
  if (k > 1) {
    const unsigned int* __restrict__ sstep = pr.sstep + (size_t)((k - 1) & 1) * kStepH;
    for (int q = threadIdx.x; q < kStepH / 4; q += kLoopBlock) ((uint4*)l_step)[q] = ((const uint4*)sstep)[q];
  }
}
__device__ __noinline__ void loop_budget_phase(const IcpProblem* __restrict__ prp, int k, int Mn, int Rn) {
  const IcpProblem& pr = *prp;
  const GridDesc g = pr.geom->g;
  // (iteration 1: every query is dealt to one of the searches)
  if (k == 1) icp_first_phase<kLoopBlock>(pr, Mn, Rn, 0, 1, g, l_cnt);
  else icp_budget_phase<kLoopBlock>(pr, k, Mn, Rn, 0, 1, g, loop_mom_table(pr, k), l_step, l_cnt);
}
// the rounds of the four lists; the waves run free (no barrier between rounds)
__device__ __noinline__ void loop_search_phase(const IcpProblem* __restrict__ prp, int k, int Mn, int Rn) {
  const IcpProblem& pr = *prp;
  const GridDesc g = pr.geom->g;
  const MomTable mt = loop_mom_table(pr, k);
#pragma unroll 1
  for (int li = 0; li < 4; ++li) {
    const bool coop = li < 2;
    const int dir = li & 1;
    const int nw = l_cnt[li];
    const int per = coop ? kLoopBlock / 32 : kLoopBlock / 8;
    const int rounds = (nw + per - 1) / per;
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) icp_search_item<kLoopBlock>(pr, pr.st, k, Mn, Rn, g, coop, dir, r, nw, mt);
  }
}
__device__ __noinline__ void loop_flush_phase(const IcpProblem* __restrict__ prp, int k) {
  mom_flush(loop_mom_table(*prp, k));
}
__device__ __noinline__ void loop_solve_phase(const IcpProblem* __restrict__ prp, int k) {
  icp_solve_phase<kLoopBlock>(*prp, prp->st, k, 0, 1, threadIdx.x >> 6);
}
__device__ __noinline__ void loop_apply_phase(const IcpProblem* __restrict__ prp, int k, int Mn) {
  const IcpProblem& pr = *prp;
  const GridDesc g = pr.geom->g;
  icp_apply_phase<kLoopBlock>(pr, pr.st, k, Mn, 0, 1, g);
}

__global__ void __launch_bounds__(kLoopBlock)
icp_loop_kernel(const IcpProblem* __restrict__ tab, Batch bt, int n_it, int wide_limit, unsigned int budget_ticks,
                int all_modes, IcpState* __restrict__ out) {
  const IcpProblem* prp = tab + bt.idx[blockIdx.x];
  IcpState* st = prp->st;
  __shared__ int s_go;
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  int Mn, Rn, Ra;
  icp_sizes(*prp, Mn, Rn, Ra);
  int k = st->iters + 1;  // (uniform: the last launch's apply phase wrote it)
  const int k_end = k + n_it;
  // the stop rule of the last iteration a wide launch ran (nothing to do when this loop ran it)
  icp_decide(*prp, k - 1);
  unsigned long long ticks[4] = {0ULL, 0ULL, 0ULL, 0ULL};
  int done = 0;
  for (; k < k_end; ++k) {
    __syncthreads();  // (the decision above / below, the previous iteration's apply phase)
    unsigned long long t0 = 0ULL;
    if (threadIdx.x == 0) {
      t0 = __builtin_amdgcn_s_memrealtime();
      bool go = !(k <= wide_limit || k > prp->max_iter || k > st->stop_iter);  // (k <= wide_limit: a wide launch's)
      if (st->mode == 1 || (!all_modes && st->mode == 0)) go = false;  // a registration of the tile path: icp_tile_kernel's
      if (go && budget_ticks && done > 0 && t0 - t_begin > (unsigned long long)budget_ticks) go = false;
      s_go = go ? 1 : 0;
      l_cnt[0] = l_cnt[1] = l_cnt[2] = l_cnt[3] = 0;
    }
    loop_begin_phase(prp, k);
    __syncthreads();
    if (!s_go) break;
    loop_budget_phase(prp, k, Mn, Rn);
    __syncthreads();
    unsigned long long t1 = 0ULL;
    if (threadIdx.x == 0) { t1 = __builtin_amdgcn_s_memrealtime(); ticks[0] += t1 - t0; }
    loop_search_phase(prp, k, Mn, Rn);
    __syncthreads();
    loop_flush_phase(prp, k);
    __syncthreads();
    unsigned long long t2 = 0ULL;
    if (threadIdx.x == 0) { t2 = __builtin_amdgcn_s_memrealtime(); ticks[1] += t2 - t1; }
    // solve, then the stop rule of this iteration (it still admits the apply phase of iteration k)
    loop_solve_phase(prp, k);
    __syncthreads();
    icp_decide(*prp, k);
    unsigned long long t3 = 0ULL;
    if (threadIdx.x == 0) { t3 = __builtin_amdgcn_s_memrealtime(); ticks[2] += t3 - t2; }
    loop_apply_phase(prp, k, Mn);
    if (threadIdx.x == 0) ticks[3] += __builtin_amdgcn_s_memrealtime() - t3;
    ++done;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    st->loop_iters += done;
#pragma unroll
    for (int q = 0; q < 4; ++q) st->phase_ticks[q] += ticks[q];
    if (out) out[blockIdx.x] = *st;
  }
}

// ============================================================ the tile path of the unit-resident loop
// The searches of the cache path walk hash grids in global memory: every probe and every row of
// candidates is a scattered 16-byte request through the CU's L1 (~250 per searched query) and a chain of
// dependent loads -- one CU spent ~900 us per iteration of a 24 k-query unit in them. The tile path keeps
// the verified cache (the budget test, lean) and serves the queries that fail it from LDS:
//   * both clouds are kept SORTED BY BLOCK ROW (a block = 4 x 4 fine cells in x / y, all z; a row = the
//     blocks of one y range): the static reference cloud once per registration, the moving cloud at the
//     start of every iteration (a counting sort with the row counters in LDS). h = (spatial radius + mu) / 4,
//     so a query's widened search ball lies inside its own row and the two next to it;
//   * a BAND = as many consecutive rows as fit: their points and the rows on either side -- one contiguous,
//     coalesced range of the sorted cloud, at most kTileCap points -- are copied into LDS SORTED BY CELL
//     BUCKET (bucket = hash of the fine cell; a counting sort, so a bucket's points are one contiguous run;
//     cells that share a bucket share a run: a query then looks at a few points more, never at fewer)
//     beside direct-mapped 64-bit occupancy masks of the 4 x 4 x 4-cell blocks (OR of whatever shares the
//     slot);
//   * the failed queries, sorted by the band their row belongs to, then search with ONE LANE each, in LDS
//     only: the query's own cell, the neighbours whose distance bound admits a point that matters, and --
//     if the best so far (or the radius) still reaches further -- the occupied cells of the 27 blocks
//     around it. No hash keys, no probing loops, no pointer chasing, no global memory in the search.
// The search is exact (conservative pruning; the canonical order (dist2, original index) decides) and
// returns the same validity budget statement as the cooperative search (everything within the nearest
// distance + mu has been looked at), so the two paths produce the same neighbours bit for bit. There
// is no global grid of the moving points any more: apply only moves them. Preconditions (one frame value
// per cloud, at most kTileRows rows, no three rows with more than kTileCap points) are checked by the
// first launch; registrations that fail them take the cache path (icp_loop_kernel), decided on the
// device, no host round trip.
constexpr int kTileCap = 4096;      // candidates staged at once
constexpr int kTileBuckets = 8192;  // cell buckets (>= 2 per staged point)
constexpr int kTileBlocks = 2048;   // block occupancy masks
constexpr int kTileMomSlots = 64;
constexpr int kTileBands = 64;      // bands a cloud may be cut into (kTileCap points each: 256 k points)
__shared__ float4 t_pts[kTileCap];              // staged candidates by bucket: (original index bits, x, y, z)
__shared__ unsigned int t_head[kTileBuckets];   // bucket -> first staged point << 16 | points; scratch of the binning
__shared__ unsigned long long t_bmask[kTileBlocks];
__shared__ int t_rowr[kTileRows + 1];           // first sorted reference point of every block row
__shared__ int t_rowm[kTileRows + 1];           // the same for the moving points of this iteration
__shared__ unsigned int t_step[kStepH];         // the step table of iteration k-1 (budget test)
__shared__ int t_key[kTileMomSlots];
__shared__ unsigned long long t_acc[kTileMomSlots * 17];
__shared__ short t_band_lo[kTileBands + 1];     // band b = rows [lo[b], lo[b+1]) of the candidates' cloud
__shared__ int t_boff[2 * kTileBands + 1];      // the failed queries of band b: [off[2b], off[2b+2]) of the sorted list,
__shared__ int t_bcur[2 * kTileBands];          // those with a partner to start from first
__shared__ int t_wsum[kLoopBlock / 64 + 1];
__shared__ int t_cnt[4];
__shared__ int t_flag[4];

struct TileGeom {
  float lo0, lo1, lo2, inv_h, h;
  float radius2, margin, mu;
  int nrows;
};
__device__ __forceinline__ TileGeom tile_geom(const IcpProblem& pr) {
  const GridDesc g = pr.geom->g;
  TileGeom t;
  t.lo0 = g.lo[0]; t.lo1 = g.lo[1]; t.lo2 = g.lo[2];
  t.h = g.h;
  t.inv_h = 1.f / g.h;
  t.radius2 = g.radius2;
  t.margin = 1e-5f * g.radius2 + 1e-6f;
  t.mu = pr.mu;
  // block rows: the y extent of both clouds as they arrive, plus the pad the grid origin was given
  const float ymax = fmaxf(pr.bounds[6], pr.bounds[14]);
  const float cyf = floorf((ymax + pr.pad - g.lo[1]) * t.inv_h);
  t.nrows = cyf < 4.f * (float)kTileRows ? (((int)cyf) >> 2) + 2 : kTileRows + 1;
  return t;
}
__device__ __forceinline__ MomTable tile_mom_table(const IcpProblem& pr, int k) {
  return MomTable{t_key, t_acc, (unsigned long long*)pr.acc + (size_t)(k & 1) * pr.C * kAcc, (int)pr.C, kTileMomSlots};
}
// cells by multiplication with 1 / h (every cell of the tile path comes from here, so the buckets are
// consistent; the pruning bounds below carry the slack for its rounding)
__device__ __forceinline__ int tile_cell(float v, float lo, float inv_h) { return (int)floorf((v - lo) * inv_h); }
constexpr unsigned kTP1 = 73856093u, kTP2 = 19349663u, kTP3 = 83492791u;
__device__ __forceinline__ unsigned tile_mix(unsigned h) { return h ^ (h >> 15); }
__device__ __forceinline__ unsigned tile_bucket(int cx, int cy, int cz) {
  return tile_mix((unsigned)cx * kTP1 ^ (unsigned)cy * kTP2 ^ (unsigned)cz * kTP3) & (kTileBuckets - 1);
}
__device__ __forceinline__ int tile_row_of(const TileGeom& t, float4 p) {  // block row of a point, -1: outside
  const int cy = tile_cell(p.z, t.lo1, t.inv_h), cx = tile_cell(p.y, t.lo0, t.inv_h), cz = tile_cell(p.w, t.lo2, t.inv_h);
  const int r = cy >> 2;
  return (cx < 0 || cy < 0 || cz < 0 || r >= t.nrows || cx >= kCoordMax || cz >= kCoordMax) ? -1 : r;
}
// exclusive prefix of one int per thread over the workgroup (kLoopBlock threads); *total = the sum
__device__ __forceinline__ int tile_wg_scan(int mine, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  __syncthreads();  // (t_wsum may still be read from the last call)
  if (lane == 63) t_wsum[wv] = incl;
  __syncthreads();
  int before = 0, all = 0;
  for (int q = 0; q < kLoopBlock / 64; ++q) {
    const int v = t_wsum[q];
    if (q < wv) before += v;
    all += v;
  }
  *total = all;
  return before + incl - mine;
}

// counting sort of a cloud by block row: dst = (index bits, x, y, z) row by row, rows[r] = first point of
// row r (rows[r] = n from the last occupied row on, for every r <= kTileRows). Points of a row arrive in
// arbitrary order: nothing downstream depends on it. Points outside the rows raise *err.
__device__ __forceinline__ void tile_bin(const float4* __restrict__ src, int n, float4* __restrict__ dst, int* rows,
                                         const TileGeom& t, int* __restrict__ err) {
  static_assert(kTileRows == kLoopBlock, "one row counter per thread");
  int* hist = (int*)t_head;
  int* cur = hist + kTileRows;
  for (int q = threadIdx.x; q < 2 * kTileRows; q += kLoopBlock) hist[q] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += kLoopBlock) {
    int r = tile_row_of(t, src[i]);
    if (r < 0) { *err = 1; r = 0; }
    atomicAdd(&hist[r], 1);
  }
  __syncthreads();
  {
    int total;
    const int ex = tile_wg_scan(hist[threadIdx.x], &total);
    rows[threadIdx.x] = ex;
    cur[threadIdx.x] = ex;
    if (threadIdx.x == 0) rows[kTileRows] = n;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += kLoopBlock) {
    const float4 p = src[i];
    int r = tile_row_of(t, p);
    if (r < 0) r = 0;
    const int pos = atomicAdd(&cur[r], 1);
    if (pos < 0 || pos >= n) { atomicExch(err, 103); continue; }  // (cannot happen: the same rows as counted)
    dst[pos] = make_float4(__int_as_float(i), p.y, p.z, p.w);
  }
  __syncthreads();
}

// largest number of points in three consecutive rows (what a band of one row stages), per wave
__device__ __forceinline__ int tile_max_triple(const int* rows, int nrows) {
  int v = 0;
  for (int r = threadIdx.x; r < nrows; r += kLoopBlock) {
    const int a = rows[r > 0 ? r - 1 : 0], b = rows[r + 2 < nrows ? r + 2 : nrows];
    v = max(v, b - a);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}

// The bands of a cloud (candidates): greedily as many rows as fit kTileCap with the rows on either side.
// Every thread forms the same table; returns the number of bands (t_band_lo[nb] = nrows), -1 if three rows
// do not fit or there are more than kTileBands.
__device__ __forceinline__ int tile_plan_bands(const int* crow, int nrows) {
  int nb = 0, r0 = 0;
  while (r0 < nrows) {
    const int s0 = crow[r0 > 0 ? r0 - 1 : 0];
    int r1 = r0 + 1;
    if (crow[r1 + 1 < nrows ? r1 + 1 : nrows] - s0 > kTileCap) return -1;
    while (r1 < nrows && crow[r1 + 2 < nrows ? r1 + 2 : nrows] - s0 <= kTileCap) ++r1;
    if (nb >= kTileBands) return -1;
    if (threadIdx.x == 0) t_band_lo[nb] = (short)r0;
    ++nb;
    r0 = r1;
  }
  if (threadIdx.x == 0) t_band_lo[nb] = (short)nrows;
  return nb;
}

// a band's candidates: rows of the sorted cloud [s0, s0 + ns) into LDS, sorted by cell bucket, and the
// block masks. Three sweeps over the range (count, place; the points come from L2).
__device__ __forceinline__ void tile_stage(const float4* __restrict__ srt, int s0, int ns, const TileGeom& t) {
  for (int q = threadIdx.x; q < kTileBuckets; q += kLoopBlock) t_head[q] = 0u;
  for (int q = threadIdx.x; q < kTileBlocks; q += kLoopBlock) t_bmask[q] = 0ULL;
  __syncthreads();
  for (int j = threadIdx.x; j < ns; j += kLoopBlock) {
    const float4 p = srt[s0 + j];
    const int cx = tile_cell(p.y, t.lo0, t.inv_h), cy = tile_cell(p.z, t.lo1, t.inv_h), cz = tile_cell(p.w, t.lo2, t.inv_h);
    atomicAdd(&t_head[tile_bucket(cx, cy, cz)], 1u);
    const unsigned bs = tile_mix((unsigned)(cx >> 2) * kTP2 ^ (unsigned)(cy >> 2) * kTP3 ^ (unsigned)(cz >> 2) * kTP1) &
                        (kTileBlocks - 1);
    const int bit = (cx & 3) | ((cy & 3) << 2) | ((cz & 3) << 4);
    atomicOr(&t_bmask[bs], 1ULL << bit);
  }
  __syncthreads();
  {  // starts: thread x owns buckets x, x + 1024, ... (any disjoint layout of the runs will do)
    unsigned c[kTileBuckets / kLoopBlock];
    int mine = 0;
#pragma unroll
    for (int q = 0; q < kTileBuckets / kLoopBlock; ++q) {
      c[q] = t_head[threadIdx.x + q * kLoopBlock];
      mine += (int)c[q];
    }
    int total;
    int run = tile_wg_scan(mine, &total);
#pragma unroll
    for (int q = 0; q < kTileBuckets / kLoopBlock; ++q) {
      t_head[threadIdx.x + q * kLoopBlock] = (unsigned)run << 16;
      run += (int)c[q];
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < ns; j += kLoopBlock) {
    const float4 p = srt[s0 + j];
    const int cx = tile_cell(p.y, t.lo0, t.inv_h), cy = tile_cell(p.z, t.lo1, t.inv_h), cz = tile_cell(p.w, t.lo2, t.inv_h);
    const unsigned old = atomicAdd(&t_head[tile_bucket(cx, cy, cz)], 1u);
    const unsigned pos = (old >> 16) + (old & 0xFFFFu);
    if (pos < (unsigned)kTileCap) t_pts[pos] = p;
  }
  __syncthreads();
}

// The nearest staged point of (px, py, pz) within the radius under the canonical order (dist2, original
// index), and for how long that answer stays exact (the statement of g2_nearest_coop<true>: everything
// within the nearest spatial distance + mu -- without a partner: within the spatial radius + mu -- has
// been looked at, so the runner-up is at least that far). dist2 is formed exactly as the cooperative
// search forms it: ((c0 + dx^2) + dy^2) + dz^2 with c0 = (frame difference)^2, the same for every pair
// of the direction. One lane per query, LDS only.
struct TileHit {
  int idx;    // original index of the point, -1: none within the radius
  int j;      // where it is staged
  float bud;  // how far things may move before the answer can change
};
// seed_d2: the 4-D squared distance of a point KNOWN to be among the candidates and within the radius (the
// partner of the last iteration), or +inf: the search then starts from the ball through that point.
__device__ __forceinline__ TileHit tile_nearest(const TileGeom& t, float px, float py, float pz, float c0, float seed_d2) {
  const float fx = (px - t.lo0) * t.inv_h, fy = (py - t.lo1) * t.inv_h, fz = (pz - t.lo2) * t.inv_h;
  const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
  const int cx = (int)flx, cy = (int)fly, cz = (int)flz;
  // squared distance, in cells, from the query to the neighbouring cell at offset -1 / 0 / +1 per axis
  const float ux = fx - flx, uy = fy - fly, uz = fz - flz;
  const float gx0 = ux * ux, gx2 = (1.f - ux) * (1.f - ux);
  const float gy0 = uy * uy, gy2 = (1.f - uy) * (1.f - uy);
  const float gz0 = uz * uz, gz2 = (1.f - uz) * (1.f - uz);
  const float rs = sqrtf(fmaxf(t.radius2 - c0, 0.f));
  const float rinf2 = (rs + t.mu) * (rs + t.mu);
  // the slack (0.2 %) covers the rounding of the cell coordinates (a point may sit a few 1e-5 cells outside
  // its nominal cell)
  const float h2s = t.h * t.h * 0.998f;
  float bd = 0.f, cd2 = __builtin_inff();
  int bi = -1, bj = -1;
  float bound = rinf2 + t.margin;  // admissible |dxyz|^2 of a point that matters
  if (seed_d2 <= t.radius2) {
    const float r = sqrtf(fmaxf(seed_d2 - c0, 0.f)) + t.mu;
    bound = r * r + t.margin;
  }
  auto scan_bucket = [&](unsigned b) {
    const unsigned v = t_head[b];
    const int s = (int)(v >> 16), e = s + (int)(v & 0xFFFFu);
    bool better = false;
    for (int j = s; j < e; ++j) {
      const float4 c = t_pts[j];
      const float d1 = c.y - px, d2 = c.z - py, d3 = c.w - pz;
      float dist2 = c0;
      dist2 = dist2 + d1 * d1;
      dist2 = dist2 + d2 * d2;
      dist2 = dist2 + d3 * d3;
      const int oi = __float_as_int(c.x);
      if (oi == bi) continue;  // (cells that share a bucket: the best is met again)
      if (dist2 <= t.radius2 && (bi < 0 || dist2 < bd || (dist2 == bd && oi < bi))) {
        if (bi >= 0) cd2 = fminf(cd2, bd);
        bd = dist2;
        bi = oi;
        bj = j;
        better = true;
      } else {
        cd2 = fminf(cd2, dist2);
      }
    }
    if (better) {
      const float r = sqrtf(fmaxf(bd - c0, 0.f)) + t.mu;
      bound = fminf(bound, r * r + t.margin);
    }
  };
  // ---- the query's own cell, then the 26 around it that the bound still reaches
  scan_bucket(tile_bucket(cx, cy, cz));
#pragma unroll 1
  for (int dz = -1; dz <= 1; ++dz) {
    const float gz = dz < 0 ? gz0 : (dz > 0 ? gz2 : 0.f);
#pragma unroll 1
    for (int dy = -1; dy <= 1; ++dy) {
      const float gyz = gz + (dy < 0 ? gy0 : (dy > 0 ? gy2 : 0.f));
      if (gyz * h2s > bound) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        if ((dx | dy | dz) == 0) continue;
        const float g = gyz + (dx < 0 ? gx0 : (dx > 0 ? gx2 : 0.f));
        if (g * h2s > bound) continue;
        scan_bucket(tile_bucket(cx + dx, cy + dy, cz + dz));
      }
    }
  }
  // ---- further out: everything at Chebyshev distance >= 2 cells is at least h away. A query that has
  //      found nothing yet looks at EVERY occupied cell of the 27 blocks around it (whole = true): if it
  //      stays without partner it then knows the distance to the nearest point of the whole region, and that
  //      everything else lies beyond the region -- a budget that usually lasts for the rest of the
  //      registration, where the ball of radius + mu alone says "nothing within mu" and is searched again
  //      as soon as something has moved that far.
  float lb_none2 = fminf(cd2 - c0, rinf2);  // without partner: squared lower bound of the nearest point's distance
  if (!(h2s > bound)) {
    const bool whole = bi < 0 && !(seed_d2 <= t.radius2);  // (with a seed there IS a point within the bound)
    const int bx = cx >> 2, by = cy >> 2, bz = cz >> 2;
#pragma unroll 1
    for (int q = 0; q < 27; ++q) {
      const int X = bx + q % 3 - 1, Y = by + (q / 3) % 3 - 1, Z = bz + q / 9 - 1;
      unsigned long long m =
          t_bmask[tile_mix((unsigned)X * kTP2 ^ (unsigned)Y * kTP3 ^ (unsigned)Z * kTP1) & (kTileBlocks - 1)];
      while (m) {
        const int bit = __ffsll((long long)m) - 1;
        m &= m - 1ULL;
        const int x = 4 * X + (bit & 3), y = 4 * Y + ((bit >> 2) & 3), z = 4 * Z + (bit >> 4);
        const int ax = abs(x - cx), ay = abs(y - cy), az = abs(z - cz);
        const float ex = (float)max(ax - 1, 0), ey = (float)max(ay - 1, 0), ez = (float)max(az - 1, 0);
        const bool reach = !(((ex * ex + ey * ey) + ez * ez) * h2s > bound);
        if (max(max(ax, ay), az) <= 1) {
          // a near cell: looked at above unless the bound pruned it -- which only the whole-region scan undoes
          if (!(whole && bi < 0)) continue;
          const float g = (x < cx ? gx0 : (x > cx ? gx2 : 0.f)) + (y < cy ? gy0 : (y > cy ? gy2 : 0.f)) +
                          (z < cz ? gz0 : (z > cz ? gz2 : 0.f));
          if (!(g * h2s > bound)) continue;  // (it was within the bound, hence looked at)
        } else if (!reach && !(whole && bi < 0)) {
          continue;
        }
        scan_bucket(tile_bucket(x, y, z));
      }
    }
    if (whole && bi < 0) {
      // nothing within the radius; every point of the 3 x 3 x 3 blocks has been looked at (cd2), the rest
      // lies beyond their boundary
      const float eX = fminf(fx - (float)(4 * (bx - 1)), (float)(4 * (bx + 2)) - fx);
      const float eY = fminf(fy - (float)(4 * (by - 1)), (float)(4 * (by + 2)) - fy);
      const float eZ = fminf(fz - (float)(4 * (bz - 1)), (float)(4 * (bz + 2)) - fz);
      const float e = fmaxf(fminf(fminf(eX, eY), eZ), 0.f) * t.h * 0.999f;
      lb_none2 = fminf(cd2 - c0, e * e);
    } else if (bi < 0) {
      lb_none2 = fminf(cd2 - c0, rinf2);
    }
  }
  TileHit hit;
  hit.idx = bi;
  hit.j = bj;
  hit.bud = bi >= 0 ? icp_cache_budget(bd - c0, fminf(cd2 - c0, bound - t.margin), t.radius2)
                    : icp_cache_budget_none(lb_none2, t.radius2 - c0, t.radius2);
  return hit;
}

// The same search by a WHOLE WAVE for one query (uniform arguments): for the queries that have no partner to
// start from -- their ball is the full radius + mu, hundreds of cells, and without a partner every occupied
// cell of the 27 blocks (see above). One lane took ~100 k instructions for such a query, and a wave waits for
// its slowest lane. Here lanes 0..26 fetch the block masks, keep the cells the bound reaches, and the set bits
// of all 27 masks are dealt to the 64 lanes (rank r -> lane r % 64), so the walk takes ceil(cells / 64) rounds.
__device__ __forceinline__ TileHit tile_nearest_wave(const TileGeom& t, float px, float py, float pz, float c0,
                                                     float seed_d2) {
  const int lane = threadIdx.x & 63;
  const float fx = (px - t.lo0) * t.inv_h, fy = (py - t.lo1) * t.inv_h, fz = (pz - t.lo2) * t.inv_h;
  const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
  const int bx = cx >> 2, by = cy >> 2, bz = cz >> 2;
  const float rs = sqrtf(fmaxf(t.radius2 - c0, 0.f));
  const float rinf2 = (rs + t.mu) * (rs + t.mu);
  const float h2s = t.h * t.h * 0.998f;
  float bd = 0.f, cd2 = __builtin_inff();
  int bi = -1, bj = -1;
  float bound = rinf2 + t.margin;
  const bool seeded = seed_d2 <= t.radius2;  // a point known to be among the candidates, inside the radius
  if (seeded) {
    const float r = sqrtf(fmaxf(seed_d2 - c0, 0.f)) + t.mu;
    bound = r * r + t.margin;
  }
  auto scan_bucket = [&](unsigned b) {
    const unsigned v = t_head[b];
    const int s = (int)(v >> 16), e = s + (int)(v & 0xFFFFu);
    for (int j = s; j < e; ++j) {
      const float4 c = t_pts[j];
      const float d1 = c.y - px, d2 = c.z - py, d3 = c.w - pz;
      float dist2 = c0;
      dist2 = dist2 + d1 * d1;
      dist2 = dist2 + d2 * d2;
      dist2 = dist2 + d3 * d3;
      const int oi = __float_as_int(c.x);
      if (oi == bi) continue;
      if (dist2 <= t.radius2 && (bi < 0 || dist2 < bd || (dist2 == bd && oi < bi))) {
        if (bi >= 0) cd2 = fminf(cd2, bd);
        bd = dist2;
        bi = oi;
        bj = j;
      } else {
        cd2 = fminf(cd2, dist2);
      }
    }
  };
  const int X = bx + lane % 3 - 1, Y = by + (lane / 3) % 3 - 1, Z = bz + (lane / 9) % 3 - 1;
  unsigned long long mymask = 0ULL;
  if (lane < 27)
    mymask = t_bmask[tile_mix((unsigned)X * kTP2 ^ (unsigned)Y * kTP3 ^ (unsigned)Z * kTP1) & (kTileBlocks - 1)];
  // the cells of this lane's block within the bound: the squared bound separates into per-axis terms (small
  // integers), so the 64 tests are 16 thresholds on 4 values
  unsigned long long ball = 0ULL;
  {
    float ax2[4], ay2[4], az2[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float tx = (float)max(abs(4 * X + a - cx) - 1, 0), ty = (float)max(abs(4 * Y + a - cy) - 1, 0),
                  tz = (float)max(abs(4 * Z + a - cz) - 1, 0);
      ax2[a] = tx * tx; ay2[a] = ty * ty; az2[a] = tz * tz;
    }
    const float lim = (bound / h2s) * 1.00001f + 1e-5f;
#pragma unroll
    for (int zz = 0; zz < 4; ++zz) {
#pragma unroll
      for (int yy = 0; yy < 4; ++yy) {
        const float tt = lim - az2[zz] - ay2[yy];
        const unsigned xb = (unsigned)(ax2[0] <= tt) | ((unsigned)(ax2[1] <= tt) << 1) |
                            ((unsigned)(ax2[2] <= tt) << 2) | ((unsigned)(ax2[3] <= tt) << 3);
        ball |= (unsigned long long)xb << (4 * yy + 16 * zz);
      }
    }
  }
  // deal the cells `cells` (bits of lanes 0..26) to the 64 lanes and scan them
  auto walk = [&](unsigned long long cells) {
    const int n_mine = __popcll(cells);
    int incl = n_mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    const int total = __shfl(incl, 63, 64);
    const unsigned c_lo = (unsigned)(cells & 0xFFFFFFFFULL), c_hi = (unsigned)(cells >> 32);
    for (int base = 0; base < total; base += 64) {
      const int r = base + lane;
      const bool act = r < total;
      int lo = 0, hi = 26;  // owner: the first lane whose inclusive count exceeds r
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        const int mid = (lo + hi) >> 1;
        const int v = __shfl(incl, mid, 64);
        if (v <= r) lo = mid + 1; else hi = mid;
      }
      const int q = act ? min(lo, 26) : 0;
      const int q_incl = __shfl(incl, q, 64);
      const unsigned m_lo = __shfl(c_lo, q, 64), m_hi = __shfl(c_hi, q, 64);
      if (act) {
        int kk = r - (q_incl - (__popc(m_lo) + __popc(m_hi)));  // rank within the owner's mask
        unsigned w = m_lo;
        int pos = 0, c = __popc(m_lo);
        if (kk >= c) { kk -= c; w = m_hi; pos = 32; }
        c = __popc(w & 0xFFFFu);
        if (kk >= c) { kk -= c; w >>= 16; pos += 16; }
        w &= 0xFFFFu;
        c = __popc(w & 0xFFu);
        if (kk >= c) { kk -= c; w >>= 8; pos += 8; }
        w &= 0xFFu;
        c = __popc(w & 0xFu);
        if (kk >= c) { kk -= c; w >>= 4; pos += 4; }
        w &= 0xFu;
        c = __popc(w & 0x3u);
        if (kk >= c) { kk -= c; w >>= 2; pos += 2; }
        w &= 0x3u;
        if (kk >= (int)(w & 1u)) pos += 1;
        const int QX = bx + q % 3 - 1, QY = by + (q / 3) % 3 - 1, QZ = bz + q / 9 - 1;
        scan_bucket(tile_bucket(4 * QX + (pos & 3), 4 * QY + ((pos >> 2) & 3), 4 * QZ + (pos >> 4)));
      }
    }
  };
  // best of the wave; a lane whose own best lost keeps it as a non-best distance
  auto reduce = [&]() {
    const int mine = bi;
    const float mine_d = bd;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int oi = __shfl_xor(bi, o, 64);
      const float od = __shfl_xor(bd, o, 64);
      const int oj = __shfl_xor(bj, o, 64);
      const bool take = oi >= 0 && (bi < 0 || od < bd || (od == bd && oi < bi));
      if (take) { bi = oi; bd = od; bj = oj; }
    }
    if (mine >= 0 && mine != bi) cd2 = fminf(cd2, mine_d);
  };
  walk(mymask & ball);
  reduce();
  bool whole = false;
  if (bi < 0 && !seeded) {  // (uniform) nothing within the radius: the rest of the 27 blocks, for the budget
    walk(mymask & ~ball);
    reduce();
    whole = bi < 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cd2 = fminf(cd2, __shfl_xor(cd2, o, 64));
  TileHit hit;
  hit.idx = bi;
  hit.j = bj;
  if (bi >= 0) {
    const float r = sqrtf(fmaxf(bd - c0, 0.f)) + t.mu;
    hit.bud = icp_cache_budget(bd - c0, fminf(cd2 - c0, r * r), t.radius2);
  } else if (whole) {
    const float eX = fminf(fx - (float)(4 * (bx - 1)), (float)(4 * (bx + 2)) - fx);
    const float eY = fminf(fy - (float)(4 * (by - 1)), (float)(4 * (by + 2)) - fy);
    const float eZ = fminf(fz - (float)(4 * (bz - 1)), (float)(4 * (bz + 2)) - fz);
    const float e = fmaxf(fminf(fminf(eX, eY), eZ), 0.f) * t.h * 0.999f;
    hit.bud = icp_cache_budget_none(fminf(cd2 - c0, e * e), t.radius2 - c0, t.radius2);
  } else {
    hit.bud = icp_cache_budget_none(fminf(cd2 - c0, rinf2), t.radius2 - c0, t.radius2);
  }
  return hit;
}

// per direction: do the frames of the two clouds meet at all (the reference's frame-cell walk:
// round((f - lo0) / vs0) + 1 of the query, plus the frame offset, must be the candidates' frame cell) and
// the constant frame term of every distance
struct TileDir {
  bool ok;
  float c0;
};
__device__ __forceinline__ TileDir tile_dir(const IcpProblem& pr, const TileGeom& t, float f_query, float f_cand, int df) {
  const GridDesc g = pr.geom->g;
  TileDir d;
  const int fq = frame_cell(g, f_query) + df, fc = frame_cell(g, f_cand);
  const float dfl = fabsf(f_cand - f_query);
  d.c0 = dfl * dfl;
  const float rs = sqrtf(fmaxf(t.radius2 - d.c0, 0.f));
  const float budget = (rs + t.mu) * (rs + t.mu) + t.margin;
  d.ok = fq == fc && fc >= 0 && fc < kFrameCells && t.radius2 - d.c0 + t.margin >= 0.f;
  if (d.ok && (int)ceilf(sqrtf(budget) / t.h) > 4) {  // (cannot happen with the cell size icp_describe chooses)
    if (threadIdx.x == 0) atomicExch(&pr.st->err, 1);
    d.ok = false;
  }
  return d;
}

// The failed queries of one direction (pr.wl + list offset, n of them, unsorted) against the other cloud:
// FWD: moving points against the reference rows, else reference points against the moving rows. The list
// is first sorted by band (pr.wls), then every band with queries is staged and searched. Results: neighbour
// and budget of iteration k, the pair's moments.
template <bool FWD>
__device__ __forceinline__ void tile_search_pass(const IcpProblem& pr, int k, int Mn, int Rn, const TileGeom& t,
                                                 int n_list) {
  const float4* __restrict__ csrt = FWD ? pr.rsrt : pr.msrt;                // candidates, by rows
  const float4* __restrict__ qpts = FWD ? (const float4*)pr.mov : pr.ref;   // the queries' cloud, original order
  const int* crow = FWD ? t_rowr : t_rowm;
  const int qoff = FWD ? 0 : Mn, nqc = FWD ? Mn : Rn, ncc = FWD ? Rn : Mn;
  int32_t* __restrict__ nn_o = FWD ? pr.nn_f + (size_t)(k & 1) * pr.capM : pr.nn_b + (size_t)(k & 1) * pr.capR;
  float* __restrict__ bud_o = FWD ? pr.bud_f + (size_t)(k & 1) * pr.capM : pr.bud_b + (size_t)(k & 1) * pr.capR;
  const int32_t* __restrict__ list = pr.wl + qoff;
  int32_t* __restrict__ slist = pr.wls + qoff;
  const int32_t* __restrict__ seed = FWD ? pr.nn_f + (size_t)((k - 1) & 1) * pr.capM : pr.nn_b + (size_t)((k - 1) & 1) * pr.capR;
  const float4* __restrict__ cpts = FWD ? pr.ref : (const float4*)pr.mov;   // the candidates' cloud, original order
  if (n_list < 1) return;  // (uniform)
  const float f_ref = pr.bounds[0], f_mov = pr.bounds[8];  // (one frame value per cloud: checked before)
  const TileDir dir = FWD ? tile_dir(pr, t, f_mov, f_ref, pr.df) : tile_dir(pr, t, f_ref, f_mov, -pr.df);
  if (!dir.ok) {  // nothing can be in range: no neighbour, and nothing that could change that
    for (int w = threadIdx.x; w < n_list; w += kLoopBlock) {
      const int qi = (list[w] & ~kHeavyBit) - qoff;
      nn_o[qi] = -1;
      bud_o[qi] = 0.f;
      (FWD ? pr.aux_f : pr.aux_b)[qi] = NnAux{-1, -1, -1, -__builtin_inff()};
    }
    return;
  }
  const int nb = tile_plan_bands(crow, t.nrows);
  if (nb < 0) {  // three rows that do not fit (the moving cloud has contracted since the check)
    if (threadIdx.x == 0) atomicExch(&pr.st->err, 1);
    return;
  }
  __syncthreads();  // (the band table)
  // ---- the list by band: count, offsets, place
  auto band_of = [&](int qi) {
    int r = tile_row_of(t, qpts[qi]);
    if (r < 0) r = 0;  // (outside the rows: raised as an error when the cloud was sorted; any band will do)
    int lo = 0, hi = nb - 1;  // the last band that starts at or before row r
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((int)t_band_lo[mid] <= r) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  // (two classes per band, so that the waves are of a kind: queries that start from their old partner are
  //  done after a few cells; the others may walk every cell of their 27 blocks)
  for (int q = threadIdx.x; q < 2 * kTileBands; q += kLoopBlock) t_bcur[q] = 0;
  __syncthreads();
  for (int w = threadIdx.x; w < n_list; w += kLoopBlock) {
    const int q = list[w];
    atomicAdd(&t_bcur[2 * band_of((q & ~kHeavyBit) - qoff) + ((q & kHeavyBit) ? 1 : 0)], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = 0; b < 2 * nb; ++b) { const int c = t_bcur[b]; t_boff[b] = run; t_bcur[b] = run; run += c; }
    t_boff[2 * nb] = run;
  }
  __syncthreads();
  for (int w = threadIdx.x; w < n_list; w += kLoopBlock) {
    const int q = list[w];
    slist[atomicAdd(&t_bcur[2 * band_of((q & ~kHeavyBit) - qoff) + ((q & kHeavyBit) ? 1 : 0)], 1)] = q;
  }
  __syncthreads();
  // ---- band by band
  const MomTable mt = tile_mom_table(pr, k);
  const double lim2 = icp_limit_sq(Mn + Rn);
  for (int b = 0; b < nb; ++b) {
    const int w0 = t_boff[2 * b], w1 = t_boff[2 * b + 2];
    if (w1 <= w0) continue;  // (uniform)
    const int r0 = t_band_lo[b], r1 = t_band_lo[b + 1];
    const int s0 = crow[r0 > 0 ? r0 - 1 : 0], s1 = crow[r1 + 1 < t.nrows ? r1 + 1 : t.nrows];
    const int ns = s1 - s0;
    if (s0 < 0 || ns > kTileCap || s1 > ncc) {  // (cannot happen: the bands were planned on these rows)
      if (threadIdx.x == 0) atomicExch(&pr.st->err, 104);
      break;
    }
    tile_stage(csrt, s0, ns, t);
    // the queries that start from their old partner: one lane each
    const int wl1 = t_boff[2 * b + 1];
    for (int w = w0 + (int)threadIdx.x; w < wl1; w += kLoopBlock) {
      const int qi = slist[w] - qoff;
      if (qi < 0 || qi >= nqc) { atomicExch(&pr.st->err, 102); continue; }
      const float4 p = qpts[qi];
      float seed_d2 = __builtin_inff();
      const int sd = seed[qi];
      if (sd >= 0 && sd < ncc) {  // the old partner, where it is now (inside the radius: the budget test saw it)
        const float4 sp = cpts[sd];
        const float d1 = sp.y - p.y, d2 = sp.z - p.z, d3 = sp.w - p.w;
        float dist2 = dir.c0;
        dist2 = dist2 + d1 * d1;
        dist2 = dist2 + d2 * d2;
        dist2 = dist2 + d3 * d3;
        seed_d2 = dist2;
      }
      TileHit hit = tile_nearest(t, p.y, p.z, p.w, dir.c0, seed_d2);
      if (hit.idx < -1 || hit.idx >= ncc || (hit.idx >= 0 && (hit.j < 0 || hit.j >= ns))) {
        atomicExch(&pr.st->err, 101);  // (cannot happen: indices of staged points)
        hit.idx = -1;
        hit.bud = 0.f;
      }
      nn_o[qi] = hit.idx;
      bud_o[qi] = hit.bud;
      (FWD ? pr.aux_f : pr.aux_b)[qi] = NnAux{-1, -1, -1, -__builtin_inff()};  // (the tile searches keep no second level)
      if (hit.idx >= 0) {  // the pair as (moving point, reference point, the moving point's component)
        const float4 c = t_pts[hit.j];
        const int mc = pr.comp[FWD ? qi : hit.idx];
        if (mc >= 0 && mc < (int)pr.C) {
          if (FWD) mom_add_stream(pr, mt, p, c, mc, lim2);
          else mom_add_stream(pr, mt, c, p, mc, lim2);
        } else {
          atomicExch(&pr.st->err, 105);
        }
      }
    }
    // the others: one wave each
    for (int w = wl1 + (int)(threadIdx.x >> 6); w < w1; w += kLoopBlock / 64) {
      const int qi = (slist[w] & ~kHeavyBit) - qoff;  // (uniform in the wave)
      if (qi < 0 || qi >= nqc) { atomicExch(&pr.st->err, 102); continue; }
      const float4 p = qpts[qi];
      float seed_d2 = __builtin_inff();
      const int sd = k > 1 ? seed[qi] : -1;
      if (sd >= 0 && sd < ncc) {  // the old partner, where it is now
        const float4 sp = cpts[sd];
        const float d1 = sp.y - p.y, d2 = sp.z - p.z, d3 = sp.w - p.w;
        float dist2 = dir.c0;
        dist2 = dist2 + d1 * d1;
        dist2 = dist2 + d2 * d2;
        dist2 = dist2 + d3 * d3;
        seed_d2 = dist2;
      }
      TileHit hit = tile_nearest_wave(t, p.y, p.z, p.w, dir.c0, seed_d2);
      if (hit.idx < -1 || hit.idx >= ncc || (hit.idx >= 0 && (hit.j < 0 || hit.j >= ns))) {
        atomicExch(&pr.st->err, 101);
        hit.idx = -1;
        hit.bud = 0.f;
      }
      if ((threadIdx.x & 63) == 0) {
        nn_o[qi] = hit.idx;
        bud_o[qi] = hit.bud;
        (FWD ? pr.aux_f : pr.aux_b)[qi] = NnAux{-1, -1, -1, -__builtin_inff()};
        if (hit.idx >= 0) {
          const float4 c = t_pts[hit.j];
          const int mc = pr.comp[FWD ? qi : hit.idx];
          if (mc >= 0 && mc < (int)pr.C) {
            if (FWD) mom_add_stream(pr, mt, p, c, mc, lim2);
            else mom_add_stream(pr, mt, c, p, mc, lim2);
          } else {
            atomicExch(&pr.st->err, 105);
          }
        }
      }
    }
    __syncthreads();  // (the LDS is staged again)
  }
}

// ---- the phases of a tile iteration (separate functions: each gets the registers of a 1024-thread workgroup)
// first launch of a registration: the reference cloud by rows, the path's preconditions, the mode word
__device__ __noinline__ void tile_prepare_phase(const IcpProblem* __restrict__ prp, int Mn, int Rn) {
  const IcpProblem& pr = *prp;
  IcpState* st = pr.st;
  const TileGeom t = tile_geom(pr);
  bool ok = Mn >= 1 && Rn >= 1 && t.nrows <= kTileRows && pr.bounds[0] == pr.bounds[4] && pr.bounds[8] == pr.bounds[12];
  if (threadIdx.x == 0) { t_flag[0] = 0; t_flag[1] = 0; }
  __syncthreads();
  if (ok) {
    tile_bin(pr.ref, Rn, pr.rsrt, t_rowr, t, &t_flag[1]);
    const int mr = tile_max_triple(t_rowr, t.nrows);
    if ((threadIdx.x & 63) == 0) atomicMax(&t_flag[0], mr);
    tile_bin((const float4*)pr.mov, Mn, pr.msrt, t_rowm, t, &t_flag[1]);
    const int mm = tile_max_triple(t_rowm, t.nrows);
    if ((threadIdx.x & 63) == 0) atomicMax(&t_flag[0], mm + mm / 3);  // (the moving cloud may contract a little)
    __syncthreads();
    ok = t_flag[0] <= kTileCap && t_flag[1] == 0 && Rn <= kTileBands * (kTileCap / 2) && Mn <= kTileBands * (kTileCap / 2);
  }
  if (ok) {
    for (int r = threadIdx.x; r <= kTileRows; r += kLoopBlock) pr.trow[r] = t_rowr[r];
  }
  if (threadIdx.x == 0) {
    st->mode = ok ? 1 : 2;
    t_flag[2] = ok ? 1 : 2;  // (the workgroup reads the decision here, behind the barrier)
  }
  __syncthreads();
}
// start of an iteration: the row table of the reference cloud (a later launch of the registration), the
// moving cloud by rows, the step table of iteration k-1, an empty moment table, empty work lists
__device__ __noinline__ void tile_begin_phase(const IcpProblem* __restrict__ prp, int k, int Mn, int load_rows) {
  const IcpProblem& pr = *prp;
  const TileGeom t = tile_geom(pr);
  if (load_rows)
    for (int r = threadIdx.x; r <= kTileRows; r += kLoopBlock) t_rowr[r] = pr.trow[r];
  tile_bin((const float4*)pr.mov, Mn, pr.msrt, t_rowm, t, &pr.st->err);
  if (k > 1) {
    const unsigned int* __restrict__ sstep = pr.sstep + (size_t)((k - 1) & 1) * kStepH;
    for (int q = threadIdx.x; q < kStepH / 4; q += kLoopBlock) ((uint4*)t_step)[q] = ((const uint4*)sstep)[q];
  }
  mom_init(tile_mom_table(pr, k));
  if (threadIdx.x < 4) t_cnt[threadIdx.x] = 0;
  __syncthreads();
}
// the budget test of the cache path; the queries that fail it go to ONE list per direction (t_cnt[0], [1]).
// Iteration 1: every query.
__device__ __noinline__ void tile_budget_phase(const IcpProblem* __restrict__ prp, int k, int Mn, int Rn) {
  const IcpProblem& pr = *prp;
  if (k == 1) {
    for (int q = threadIdx.x; q < Mn + Rn; q += kLoopBlock) pr.wl[q] = q | kHeavyBit;
    if (threadIdx.x == 0) { t_cnt[0] = Mn; t_cnt[1] = Rn; }
    return;
  }
  const GridDesc g = pr.geom->g;
  icp_budget_phase<kLoopBlock, false>(pr, k, Mn, Rn, 0, 1, g, tile_mom_table(pr, k), t_step, t_cnt);
}
__device__ __noinline__ void tile_forward_phase(const IcpProblem* __restrict__ prp, int k, int Mn, int Rn) {
  const TileGeom t = tile_geom(*prp);
  tile_search_pass<true>(*prp, k, Mn, Rn, t, t_cnt[0]);
}
__device__ __noinline__ void tile_backward_phase(const IcpProblem* __restrict__ prp, int k, int Mn, int Rn) {
  const TileGeom t = tile_geom(*prp);
  tile_search_pass<false>(*prp, k, Mn, Rn, t, t_cnt[1]);
}
__device__ __noinline__ void tile_flush_phase(const IcpProblem* __restrict__ prp, int k) {
  mom_flush(tile_mom_table(*prp, k));
}
// every moving point moves by its component's step; the previous positions are kept (budget test, l1
// passes); the step table of iteration k takes every point's step under the place it moved to
__device__ __noinline__ void tile_apply_phase(const IcpProblem* __restrict__ prp, int k, int Mn) {
  const IcpProblem& pr = *prp;
  float4* __restrict__ mov = pr.mov;
  float4* __restrict__ prev = pr.prev;
  const int32_t* __restrict__ comp = pr.comp;
  const double* __restrict__ Rt = pr.Rt;
  const int64_t C = pr.C;
  const GridDesc d = pr.geom->g;
  if (threadIdx.x == 0) pr.st->iters = k;
  {  // the step table of iteration k+1 (nothing reads it during this phase)
    unsigned int* nxt = pr.sstep + (size_t)((k + 1) & 1) * kStepH;
    for (int q = threadIdx.x; q < kStepH; q += kLoopBlock) nxt[q] = 0u;
  }
  unsigned int* __restrict__ sstep = pr.sstep + (size_t)(k & 1) * kStepH;
  for (int i = threadIdx.x; i < Mn; i += kLoopBlock) {
    const int ci = comp[i];
    double R[9], t[3];
#pragma unroll
    for (int q = 0; q < 9; ++q) R[q] = Rt[rt_ix(q, ci, C)];
#pragma unroll
    for (int q = 0; q < 3; ++q) t[q] = Rt[rt_ix(9 + q, ci, C)];
    const float4 p = mov[i];
    prev[i] = p;
    const double x = p.y, y = p.z, z = p.w;
    const float nx = (float)(x * R[0] + y * R[1] + z * R[2] + t[0]);
    const float ny = (float)(x * R[3] + y * R[4] + z * R[5] + t[1]);
    const float nz = (float)(x * R[6] + y * R[7] + z * R[8] + t[2]);
    mov[i] = make_float4(p.x, nx, ny, nz);
    const float dx = nx - p.y, dy = ny - p.z, dz = nz - p.w;
    const float st1 = sqrtf((dx * dx + dy * dy) + dz * dz) * 1.000001f + 1e-7f;
    const int fx = (int)floorf((nx - d.lo[0]) / d.h), fy = (int)floorf((ny - d.lo[1]) / d.h),
              fz = (int)floorf((nz - d.lo[2]) / d.h);
    atomicMax(&sstep[icp_step_slot(fx >> 3, fy >> 3, fz >> 3)], __float_as_uint(st1));
  }
}

// One workgroup per batch entry, like icp_loop_kernel; phase_ticks: [0] rows of the moving cloud + budget
// test, [1] the two search passes + moment flush, [2] solve + stop rule, [3] apply.
__global__ void __launch_bounds__(kLoopBlock)
icp_tile_kernel(const IcpProblem* __restrict__ tab, Batch bt, int n_it, int wide_limit, unsigned int budget_ticks,
                IcpState* __restrict__ out) {
  const IcpProblem* prp = tab + bt.idx[blockIdx.x];
  IcpState* st = prp->st;
  __shared__ int s_go;
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
  if (st->mode == 2) return;  // (uniform) a registration of the cache path: icp_loop_kernel's
  int Mn, Rn, Ra;
  icp_sizes(*prp, Mn, Rn, Ra);
  int k = st->iters + 1;
  const int k_end = k + n_it;
  icp_decide(*prp, k - 1);
  int load_rows = 1;
  if (k <= wide_limit || k > prp->max_iter || k > st->stop_iter) return;  // (uniform) nothing for this loop to run now
  if (st->mode == 0) {
    tile_prepare_phase(prp, Mn, Rn);
    load_rows = 0;
    if (t_flag[2] == 2) return;  // (uniform: written before the barrier that ends the phase)
  }
  unsigned long long ticks[4] = {0ULL, 0ULL, 0ULL, 0ULL};
  int done = 0;
  for (; k < k_end; ++k) {
    __syncthreads();
    unsigned long long t0 = 0ULL;
    if (threadIdx.x == 0) {
      t0 = __builtin_amdgcn_s_memrealtime();
      bool go = !(k <= wide_limit || k > prp->max_iter || k > st->stop_iter);  // (k <= wide_limit: a wide launch's)
      if (go && budget_ticks && done > 0 && t0 - t_begin > (unsigned long long)budget_ticks) go = false;
      s_go = go ? 1 : 0;
    }
    __syncthreads();
    if (!s_go) break;
    tile_begin_phase(prp, k, Mn, load_rows);
    load_rows = 0;
    tile_budget_phase(prp, k, Mn, Rn);
    __syncthreads();
    unsigned long long t1 = 0ULL;
    if (threadIdx.x == 0) { t1 = __builtin_amdgcn_s_memrealtime(); ticks[0] += t1 - t0; }
    tile_forward_phase(prp, k, Mn, Rn);
    __syncthreads();
    tile_backward_phase(prp, k, Mn, Rn);
    __syncthreads();
    tile_flush_phase(prp, k);
    __syncthreads();
    unsigned long long t2 = 0ULL;
    if (threadIdx.x == 0) { t2 = __builtin_amdgcn_s_memrealtime(); ticks[1] += t2 - t1; }
    loop_solve_phase(prp, k);
    __syncthreads();
    icp_decide(*prp, k);
    unsigned long long t3 = 0ULL;
    if (threadIdx.x == 0) { t3 = __builtin_amdgcn_s_memrealtime(); ticks[2] += t3 - t2; }
    tile_apply_phase(prp, k, Mn);
    if (threadIdx.x == 0) ticks[3] += __builtin_amdgcn_s_memrealtime() - t3;
    ++done;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    st->loop_iters += done;
#pragma unroll
    for (int q = 0; q < 4; ++q) st->phase_ticks[q] += ticks[q];
    if (out) out[blockIdx.x] = *st;
  }
}

// coverage: moving points (final positions) with a reference point (stationary included) in range
__global__ void __launch_bounds__(kBlock)
icp_coverage_kernel(const IcpProblem* __restrict__ tab, Batch bt) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.y]];
  int Mn, Rn, Ra;
  icp_sizes(pr, Mn, Rn, Ra);
  if (Mn < 1 || Ra < 1 || !(pr.want & kIcpWantRatio)) return;
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
    double n = (double)acc[acc_ix(0, c, (int)C)];
    pr.l1_out[c] = (iters > 0 && n > 0.5 && (pr.want & kIcpWantL1)) ? ((double)acc[acc_ix(18, c, (int)C)] * kFxInv) / n : 0.0;
    // component_edge_count / (component_deg.float() + 1e-6)  (registration_utils.py:197-199)
    pr.ratio_out[c] = (pr.want & kIcpWantRatio) ? (float)pr.cov[c] / ((float)pr.comp_deg[c] + 1e-6f) : 0.f;
    for (int q = 0; q < 16; ++q) pr.T_out[c * 16 + q] = Tfin[tb_ix(q, c, C)];
  }
}

// ---- optional HIP-event timing of the loop kernel, for bench.py's roofline ----
struct KernelProfile {
  double ms = 0.0;
  double launches = 0.0;
  double bytes = 0.0;
  double unit_iters = 0.0;
  double phase_us[4] = {0.0, 0.0, 0.0, 0.0};  // budget test, searches, solve, apply: summed over the units (CU time)
  double bytes_8d = 0.0;       // SURVEY 8(d)'s figure: 112 (Mn + Rn) + 144 C_live per unit-iteration
  double many_iters = 0.0;     // unit-iterations of registrations with more than 5000 live components
  double many_bytes_8d = 0.0;  // their 8(d) bytes
};
static bool g_prof_on = false;
static KernelProfile g_prof;
static std::mutex g_prof_mu;
struct ProfEvents {
  hipEvent_t ev[2];
  bool have = false;
  bool used = false;  // the slot's last chunk carried the events
};
constexpr int kProfSlots = 8;  // chunks one thread may have in flight
static thread_local ProfEvents t_evs[kProfSlots];

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

// The schedule of the iterations (pcs_icp_set_schedule, include/pcs.h): how many iterations of a registration
// run as wide launches before a unit-resident loop takes over (INT_MAX: all of them, in chunks of 10 -- the
// default, measured fastest: DESIGN.md section 4), and which resident loop (tile path / cache path).
static std::atomic<int> g_wide_iters{INT_MAX};
static std::atomic<int> g_use_tiles{1};
static int icp_wide_limit() { return g_wide_iters.load(std::memory_order_relaxed); }
// iterations per chunk of wide launches (pcs_icp_set_chunk_iterations; the lock-step engines read their units' states
// once per chunk). 10: measured on one box, alternating, 3 rounds: 8 -> 964-979 ms per batch of 12, 10 -> 959-963, 12 ->
// 954-969 (a registration runs 24 iterations at most: 10 + 10 + 4)
static std::atomic<int> g_chunk_iters{10};
static bool icp_use_tiles() { return g_use_tiles.load(std::memory_order_relaxed) != 0; }

// end of a wide chunk: the stop rule of every entry's last iteration (the others were settled by the next
// budget-test launch), and the state mirrored where the host reads it (`out` may be pinned host memory: the
// host then needs no copy, only the stream synchronisation)
__global__ void __launch_bounds__(64)
icp_decide_kernel(const IcpProblem* __restrict__ tab, Batch bt, IcpState* __restrict__ out) {
  const IcpProblem& pr = tab[bt.idx[blockIdx.x]];
  icp_decide(pr, pr.st->iters);  // (iterations that did not run have no loss: iters is the last one that did)
  if (out && threadIdx.x == 0) out[blockIdx.x] = *pr.st;  // (one wave: thread 0 wrote the decision itself)
}

int icp_enqueue_chunk(const IcpProblem* tab, const Batch& bt, const BatchInts& kbase, int n_it,
                      unsigned budget_us, const IcpHint* hints, IcpState* out, hipStream_t s, int prof_slot) {
  if (bt.n < 1 || n_it < 1) return PCS_OK;
  if (prof_slot < 0 || prof_slot >= kProfSlots) prof_slot = 0;
  ProfEvents& t_ev = t_evs[prof_slot];
  const int wide_limit = icp_wide_limit();
  const bool prof = g_prof_on;
  if (prof && !t_ev.have) {
    for (int i = 0; i < 2; ++i) PCS_CHECK_HIP(hipEventCreate(&t_ev.ev[i]));
    t_ev.have = true;
  }
  t_ev.used = prof;
  // the events ride on the dispatches themselves (hipExtLaunchKernelGGL): start of the chunk's first iteration
  // launch to the stop of its last launch, the kernels' own begin-to-end time as rocprofv3 reports it
  hipEvent_t ev_start = prof ? t_ev.ev[0] : nullptr;
  hipEvent_t ev_stop = prof ? t_ev.ev[1] : nullptr;
  // ---- entries whose next iterations are wide: launches over those entries only
  static thread_local Batch bf;
  static thread_local BatchInts kf;
  static thread_local IcpHint hf[kMaxBatch];
  bf.n = 0;
  const int chunk_iters = g_chunk_iters.load(std::memory_order_relaxed);
  int n_wide = 0;  // iterations to launch: at most chunk_iters per chunk
  bool any_first = false, all_wide = true;
  for (int e = 0; e < bt.n; ++e) {
    if (kbase.v[e] < wide_limit) {
      hf[bf.n] = hints[e];
      kf.v[bf.n] = kbase.v[e];
      bf.idx[bf.n++] = bt.idx[e];
      const int left = wide_limit - kbase.v[e];
      if (left > n_wide) n_wide = left;
      any_first = any_first || kbase.v[e] == 0;
    }
    if ((int64_t)kbase.v[e] + chunk_iters > (int64_t)wide_limit) all_wide = false;  // (the resident loop may get a turn)
  }
  if (n_wide > chunk_iters) n_wide = chunk_iters;
  if (n_wide > n_it) n_wide = n_it;
  if (bf.n > 0 && n_wide > 0) {
    int64_t q = 1, m = 1, qsum = 0, cmax = 1;
    for (int e = 0; e < bf.n; ++e) {
      int64_t qe = (int64_t)hf[e].Mn + hf[e].Rn;
      if (qe > q) q = qe;
      qsum += qe;
      if (hf[e].Mn > m) m = hf[e].Mn;
      if (hf[e].C > cmax) cmax = hf[e].C;
    }
    const int nb_s = gx_for(cmax, kSolveBlock, 1024);  // solve: one thread per component
    // budget test: one lane per query. A workgroup resets and drains its moment table once whatever the
    // number of rounds it serves: once the launch is several generations of workgroups wide, each takes 2-4 rounds
    const int64_t wg_total = (qsum + kBlock - 1) / kBlock;
    const int rounds_per_wg = wg_total > 7168 ? 4 : (wg_total > 5120 ? 3 : (wg_total > 3072 ? 2 : 1));
    const int nb_q = gx_for((q + rounds_per_wg - 1) / rounds_per_wg, kBlock, 1024);
    // the searches are a queue over the work lists of all entries (items of 8 cooperative / 32 seeded
    // queries): iteration 1 searches for every query, later ones for ~6 % + ~12 % of them
    const int nb_c1 = (int)(qsum / 8 < 64 ? 64 : (qsum / 8 > 2048 ? 2048 : qsum / 8));
    const int nb_cs = (int)(qsum / 48 < 64 ? 64 : (qsum / 48 > 2048 ? 2048 : qsum / 48));
    const int nb_m = gx_for(m, kBlock, 4096);
    for (int j = 0; j < n_wide; ++j) {
      if (j == 0 && any_first) icp_first_kernel<<<dim3(gx_for(q, kBlock, 1024), bf.n), kBlock, 0, s>>>(tab, bf, kf);
      hipExtLaunchKernelGGL(icp_nn_kernel, dim3(nb_q, bf.n), dim3(kBlock), 0, s, j == 0 ? ev_start : nullptr, nullptr, 0, tab,
                            bf, kf, j, wide_limit);
      icp_nn_search_kernel<<<(j == 0 && any_first) ? nb_c1 : nb_cs, kBlock, 0, s>>>(tab, bf, kf, j, wide_limit);
      icp_solve_kernel<<<dim3(nb_s, bf.n), kSolveBlock, 0, s>>>(tab, bf, kf, j, wide_limit);
      hipExtLaunchKernelGGL(icp_apply_kernel, dim3(nb_m, bf.n), dim3(kBlock), 0, s, nullptr,
                            (all_wide && j == n_wide - 1) ? ev_stop : nullptr, 0, tab, bf, kf, j, wide_limit);
    }
    ev_start = nullptr;  // (taken)
  }
  if (all_wide) {
    icp_decide_kernel<<<bt.n, 64, 0, s>>>(tab, bt, out);
    PCS_LAUNCH_CHECK();
    return PCS_OK;
  }
  // ---- the unit-resident loops: one workgroup per entry (entries that began above go on from where the
  //      wide launches left them). The tile path first (it decides a new registration's path); the cache
  //      path's kernel then serves the registrations that did not qualify -- and settles every stop rule
  //      and mirrors every state.
  const unsigned long long bt_ticks = (unsigned long long)budget_us * 100ULL;
  const unsigned int budget_ticks = bt_ticks > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (unsigned int)bt_ticks;
  if (icp_use_tiles()) {
    hipExtLaunchKernelGGL(icp_tile_kernel, dim3(bt.n), dim3(kLoopBlock), 0, s, ev_start, nullptr, 0, tab, bt, n_it, wide_limit,
                          budget_ticks, out);
    ev_start = nullptr;
  }
  hipExtLaunchKernelGGL(icp_loop_kernel, dim3(bt.n), dim3(kLoopBlock), 0, s, ev_start, ev_stop, 0, tab, bt, n_it, wide_limit,
                        budget_ticks, !icp_use_tiles(), out);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// after the stream has been synchronised: add the chunk's loop launch to the profile
void icp_profile_account(double unit_iters, double bytes, int prof_slot, double bytes_8d, double many_iters,
                         double many_bytes_8d) {
  if (prof_slot < 0 || prof_slot >= kProfSlots) prof_slot = 0;
  ProfEvents& t_ev = t_evs[prof_slot];
  if (!g_prof_on || !t_ev.used) return;
  t_ev.used = false;
  if (!(unit_iters > 0.0)) return;  // (a launch whose entries had all stopped)
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, t_ev.ev[0], t_ev.ev[1]) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.ms += ms;
  g_prof.launches += 1.0;
  g_prof.bytes += bytes;
  g_prof.unit_iters += unit_iters;
  g_prof.bytes_8d += bytes_8d;
  g_prof.many_iters += many_iters;
  g_prof.many_bytes_8d += many_bytes_8d;
}

void icp_profile_phases(const unsigned long long* ticks4) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int q = 0; q < 4; ++q) g_prof.phase_us[q] += (double)ticks4[q] * 0.01;  // 100 MHz ticks
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

// Schedule of the ICP iterations (process-wide; takes effect for chunks enqueued afterwards; a registration
// may change schedule between chunks: every schedule computes the same iterations).
//   wide_iters < 0 or >= max_iter: every iteration as wide launches (default);
//   wide_iters >= 1: that many wide iterations, then the unit-resident loop: use_tiles != 0 the tile path
//   (LDS-staged searches; registrations that do not qualify take the cache path), else the cache path.
// diagnostics: the verified cache's counters since the last call (zero unless the library was built with
// -DPCS_ICP_STATS): host_out[16] = queries tested, budgets run out, of those kept by the second level, searched; then
// the same four counts of the backward queries (reference -> moving cloud) alone; [8], [9] = queries served by the
// cooperative / the eight-lane seeded search (every iteration, the first included). Synchronises the device.
int pcs_icp_cache_stats(unsigned long long* host_out) {
  if (!host_out) return PCS_ERR_ARG;
  unsigned long long v[16];
  PCS_CHECK_HIP(hipDeviceSynchronize());
  PCS_CHECK_HIP(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_icp_stats), sizeof(v)));
  for (int i = 0; i < 16; ++i) host_out[i] = v[i];
  unsigned long long z[16] = {0};
  PCS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_icp_stats), z, sizeof(z)));
  return PCS_OK;
}

// pcs_icp_set_cache_levels: 2 (default) = the verified cache with its second level (the cached points re-ranked when
// a budget has run out), 1 = budgets only: every query whose budget has run out is searched (the form of rounds 1-4;
// the cross-check of tests/test_gpu_graph_icp.py: same results bit for bit)
int pcs_icp_set_cache_levels(int levels) {
  if (levels != 1 && levels != 2) return PCS_ERR_ARG;
  g_cache_levels.store(levels, std::memory_order_relaxed);
  return PCS_OK;
}

// pcs_icp_set_chunk_iterations: iterations per chunk of the wide schedule (default 10; 1..64). Results do not depend on
// it: a chunk is a scheduling unit of the lock-step engines.
int pcs_icp_set_chunk_iterations(int n) {
  if (n < 1 || n > 64) return PCS_ERR_ARG;
  g_chunk_iters.store(n, std::memory_order_relaxed);
  return PCS_OK;
}

int pcs_icp_set_widening(double radius_divisor) {
  if (!(radius_divisor >= 1.0)) return PCS_ERR_ARG;
  g_mu_div.store(radius_divisor, std::memory_order_relaxed);
  return PCS_OK;
}

int pcs_icp_set_grid_mode(int rebuild_always) {
  g_rebuild_always.store(rebuild_always != 0 ? 1 : 0, std::memory_order_relaxed);
  return PCS_OK;
}

int pcs_icp_set_schedule(int wide_iters, int use_tiles) {
  g_wide_iters.store(wide_iters < 0 ? INT_MAX : wide_iters, std::memory_order_relaxed);
  g_use_tiles.store(use_tiles != 0 ? 1 : 0, std::memory_order_relaxed);
  return PCS_OK;
}

// bench.py support: HIP-event timing of the unit-resident loop kernel on its launch stream.
// enable = 1 resets and starts accumulating.
int pcs_profile_icp(int enable) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = enable != 0;
  g_prof = KernelProfile();
  return PCS_OK;
}

// out9 (host): [0..3] = {milliseconds, launches, algorithmic bytes, unit-iterations} of icp_loop_kernel
// accumulated since the last enable; [4..7] = microseconds its workgroups spent in the budget test, the
// searches, solve + stop rule, apply (summed over the units: CU time, not wall time; registrations that
// have ended); [8] = 0.
int pcs_profile_icp_read(double* out9);
// the same and, behind it: [9] the chunks' algorithmic bytes by SURVEY 8(d)'s formula (112 (Mn + Rn) + 144 C_live per
// unit and iteration; [2] is this implementation's own traffic model), [10] unit-iterations of registrations whose
// key numbers more than 5000 components (the 0.25 m key; few of them have a moving point in a given registration), [11] their 8(d) bytes. n = doubles the caller's array holds (>= 9).
int pcs_profile_icp_read_ex(double* out, int n) {
  if (!out || n < 9) return PCS_ERR_ARG;
  int rc = pcs_profile_icp_read(out);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (n > 9) out[9] = g_prof.bytes_8d;
  if (n > 10) out[10] = g_prof.many_iters;
  if (n > 11) out[11] = g_prof.many_bytes_8d;
  for (int q = 12; q < n; ++q) out[q] = 0.0;
  return PCS_OK;
}

int pcs_profile_icp_read(double* out9) {
  if (!out9) return PCS_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int q = 0; q < 9; ++q) out9[q] = 0.0;
  out9[0] = g_prof.ms;
  out9[1] = g_prof.launches;
  out9[2] = g_prof.bytes;
  out9[3] = g_prof.unit_iters;
  for (int q = 0; q < 4; ++q) out9[4 + q] = g_prof.phase_us[q];
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
// host every iteration, this reads one state record per `chunk` iterations; chunk <= 0: the whole loop
// in one launch).
//   mov_fxyz [Mn,4]  non-stationary moving points, updated in place
//   mov_comp [Mn]    int32 component of each moving point, in [0, C)
//   ref_fxyz [Rn,4]  non-stationary reference points; ref_all [Ra,4] all reference points
//   comp_deg [C]     int64 number of moving points (stationary included) per component
//   T [C,16] f64 out; l1_err [C] f64 out (last iteration); edge_ratio [C] f32 out
//   info[4] host out: iterations run, stop reason (1 = countdown, 0 = max_iter), iterations run by the
//                     unit-resident loop, 0
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
  if (chunk < 1 || chunk > max_iter) chunk = max_iter > 0 ? max_iter : 1;
  Batch bt;
  bt.n = 1;
  bt.idx[0] = 0;
  IcpHint hint;
  hint.Mn = (int)Mn; hint.Rn = (int)Rn; hint.Ra = (int)Ra; hint.C = C;
  hint.clear_words = host_pr.gr.slot_words + host_pr.ga.slot_words + 2 * host_pr.gm[0].slot_words;
  rc = icp_enqueue_begin(tab, bt, &hint, s);
  if (rc) return rc;
  IcpState host;
  memset(&host, 0, sizeof(host));
  host.stop_iter = INT_MAX;
  const double by_iter = 76.0 * (double)(Mn + Rn) + 272.0 * (double)C + 68.0 * (double)Mn + 416.0 * (double)C;
  while (host.iters < max_iter) {
    BatchInts kb;
    kb.v[0] = host.iters;
    const int iters_before = host.iters;
    rc = icp_enqueue_chunk(tab, bt, kb, chunk, 0u, &hint, pin, s, 0);
    if (rc) return rc;
    if (!pin) PCS_CHECK_HIP(hipMemcpyAsync(&host, host_pr.st, sizeof(IcpState), hipMemcpyDeviceToHost, s));
    PCS_CHECK_HIP(hipStreamSynchronize(s));
    if (pin) host = *pin;
    icp_profile_account((double)(host.iters - iters_before), by_iter * (double)(host.iters - iters_before), 0);
    if (host.iters == iters_before && host.stop_iter == INT_MAX && !host.err) return PCS_ERR_HIP;  // (no progress: cannot happen)
    if (host.err) return PCS_ERR_RANGE;
    if (host.stop_iter != INT_MAX) break;
  }
  icp_profile_phases(host.phase_ticks);
  info[0] = host.iters;
  info[1] = host.stop_iter != INT_MAX ? 1 : 0;
  info[2] = host.loop_iters;
  info[3] = 0;
  rc = icp_enqueue_finish(tab, bt, &hint, pin, s);
  if (rc) return rc;
  if (!pin) PCS_CHECK_HIP(hipMemcpyAsync(&host, host_pr.st, sizeof(IcpState), hipMemcpyDeviceToHost, s));
  PCS_CHECK_HIP(hipStreamSynchronize(s));
  if (pin) host = *pin;
  if (host.err) return PCS_ERR_RANGE;
  return PCS_OK;
}

}  // extern "C"
