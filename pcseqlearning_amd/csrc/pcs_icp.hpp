// Batched form of the fused per-component ICP ("TLS solve", registration_utils.py:83-206): shared
// between the stand-alone entry point (pcs_icp.hip: a batch of one) and the sequence tracker
// (pcs_tracker.hip: every tracking unit that is inside a registration joins the same launches).
//
// A registration call is described ONCE by an IcpProblem stored in a device table; every kernel of
// the loop takes (table, Batch) -- Batch is a by-value list of table slots, blockIdx.y selects the
// entry -- so one launch advances all problems that are iterating at that moment and one host thread
// drives them all. Sizes (Mn, Rn, Ra) may be given as pointers to device words (the tracker's
// sampling kernels write them; the host never reads them): the kernels are grid-stride over their
// work items, so the launch geometry is only an estimate and never changes a result.
#pragma once
#include "pcs_grid2.hpp"

#include <climits>

namespace pcs {

constexpr int kMaxBatch = 128;  // table slots one launch can address (the lists travel by value: 4 B per slot)

struct Batch {
  int n;
  int idx[kMaxBatch];
};
struct BatchInts {  // one by-value int per batch entry (iteration base of the chunk)
  int v[kMaxBatch];
};

struct IcpState {
  double last_error;
  double loss;
  int countdown;
  int stop_iter;  // iterations > stop_iter do not run (INT_MAX until the stop rule fires)
  int iters;      // iterations solved so far
  int err;
  int decided;    // last iteration whose stop rule has been evaluated
  // work lists of the current iteration, one per direction (0: forward queries, moving -> reference,
  // list entries [0, Mn); 1: backward, entries [Mn, Mn + Rn)): a search workgroup serves ONE direction
  int wl_n[2];    // queries waiting for the cooperative search
  int ws_n[2];    // queries waiting for the search around their old partner (directly behind wl_n: the two are addressed as int[4])
  unsigned int dmax[2];  // float bits: largest displacement of a moving point in iteration k (slot k & 1)
  // the moving cloud's grid (g2_move_point): gcur[k & 1] = which of gm[0..1] holds the moving points that the
  // searches of iteration k see; the solve phase of iteration k decides whether its apply phase moves the points
  // in place or rebuilds them into the other (empty) grid, and writes gcur[(k + 1) & 1] accordingly
  int gcur[2];
  int grebuild;          // the apply phase of the current iteration rebuilds (written by its solve phase)
  unsigned int gmoves;   // points that have changed their cell since the last rebuild
  int govf0;             // length of the overflow list right after the last rebuild
  int loop_iters;        // iterations the unit-resident loops have run (the others were wide launches)
  int mode;              // 0: not decided yet, 1: tile path (icp_tile_kernel), 2: cache path (icp_loop_kernel)
  int live_c;            // components that have a moving point (of the C ids the caller numbers them with)
  // measurement: 100 MHz ticks (s_memrealtime) the unit-resident loop spent in its four phases -- budget
  // test, searches, solve, apply -- summed over the iterations of this registration
  unsigned long long phase_ticks[4];
};

// per-component accumulators of one iteration: n, sum p (3), sum q (3), sum p(x)q (9), sum |p-q|^2,
// sum dist, sum clamped dist, pad. ALL are 64-bit integers: n a count, the rest fixed point with
// kFxBits fractional bits of coordinates taken relative to the component's origin (the position its
// first moving point had when the call began: components are metres across, so the products stay
// small and 2^-36 resolves them as finely as f64 resolves raw moments about a global origin).
// Integer adds commute, so the sums do not depend on the order in which lanes, waves or workgroups
// arrive: bit-identical run to run and schedule to schedule by construction. A coordinate too far
// from its origin for the 64-bit range (limit_sq below) raises the call's error word.
constexpr int kAcc = 20;
constexpr int kLossParts = 16;  // addresses the loss terms of one iteration are spread over
constexpr int kFxBits = 36;
constexpr double kFx = 68719476736.0;        // 2^36
constexpr double kFxInv = 1.0 / 68719476736.0;

// Steps of the moving points, by place: slot = hash of the 8 x 8 x 8-cell block the point moved INTO.
// Colliding blocks share a slot and hence the larger step: the table can only over-estimate.
constexpr int kStepH = 4096;
constexpr int kTileRows = 1024;  // block rows a registration of the tile path may span
__host__ __device__ inline int icp_step_slot(int sx, int sy, int sz) {
  unsigned h = (unsigned)sx * 73856093u ^ (unsigned)sy * 19349663u ^ (unsigned)sz * 83492791u;
  h ^= h >> 15;
  return (int)(h & (unsigned)(kStepH - 1));
}

struct IcpGeom {
  GridDesc g;
};
// largest admissible squared distance of a point from its component origin: ne products of that size
// stay below 2^62 in fixed point, and a single term (a product, or a squared distance of two such points)
// below 2^51, the range of the two-instruction rounding (to_fx) and of the carry-free wave sums
__host__ __device__ inline double icp_limit_sq(int ne) {
  const double by_sum = 4611686018427387904.0 / kFx / (double)(ne < 1 ? 1 : ne);
  return by_sum < 4096.0 ? by_sum : 4096.0;
}

struct IcpProblem {
  float4* mov;          // [Mn] moving points, updated in place
  float4* prev;         // [Mn] positions before the last applied update
  const float4* ref;    // [Rn] non-stationary reference points
  const float4* all;    // [Ra] every reference point (coverage ratio)
  const int32_t* comp;  // [Mn] component of each moving point
  const int64_t* comp_deg;  // [C] moving points per component, stationary included
  const int64_t* cnt3;  // device {Mn, Ra, Rn} or nullptr (then the immediates below hold)
  int32_t* first;       // [C] smallest moving-point index of the component (INT_MAX: none)
  int32_t *nn_f, *nn_b; // [2, capM] / [2, capR]: buffer k & 1 holds iteration k's neighbours
  float *bud_f, *bud_b; // same shape: how far things may still move before the neighbour can change
  NnAux *aux_f, *aux_b; // [capM] / [capR]: second level of the verified cache (the other cached points, c = L - budget)
  int32_t* wl;          // [capM + capR] queries handed to the cooperative search
  int32_t* wls;         // [capM + capR] queries handed to the search around their old partner
  unsigned int* sstep;  // [2, kStepH] float bits: largest step of a moving point per 8x8x8-cell block (lossy hash)
  int32_t* mloc;        // [capM] where moving point i lies in the moving grid (g2_insert_point / g2_move_point)
  float4* rsrt;         // [capR] tile path: the reference points by block row, (original index bits, x, y, z)
  float4* msrt;         // [capM] tile path: the moving points by block row, rebuilt every iteration
  int32_t* trow;        // [kTileRows + 1] tile path: first row of rsrt of every block row
  // the moments are one record per component (a flush adds 17 consecutive words: three lines of atomics); the
  // transforms and steps are COMPONENT-MINOR (word q of every component side by side): the solve phase's thread c
  // and the apply phase's points read them next to their neighbours' (acc_ix / tb_ix / rt_ix in pcs_icp.hip)
  long long* acc;       // [2, C, kAcc]
  double* Tb;           // [2, 16, C] accumulated transforms, buffer k & 1 after iteration k
  double* Rt;           // [12, C] rotation (9) and translation (3) of the current iteration's step
  long long* lossfx;    // [2][kLossParts] fixed-point partial sums of the components' loss terms (iteration parity), then [2][kLossParts] the same at 10 fractional bits (overflow check)
  float4* corg;         // [C] origin of the component's fixed-point moments
  int32_t* cov;         // [C] matched moving points in the final coverage query
  IcpState* st;
  IcpGeom* geom;
  float* bounds;        // [24] min/max of ref, mov, all
  Grid2 gr, ga, gm[2];  // reference (non-stationary / all), moving (one in use, one empty for the next rebuild: IcpState::gcur)
  double* T_out;        // [C,16]
  double* l1_out;       // [C]
  float* ratio_out;     // [C]
  int64_t C;
  double reg, delta;
  float rad, cell, pad;
  float mu;             // widening of every search (see g2_nearest_coop<true>)
  int Mn, Rn, Ra;       // immediates (cnt3 == nullptr) / capacities of nn_f, nn_b (cnt3 != nullptr)
  int capM, capR;
  int df, max_iter;
  int want;             // outputs the caller will read: bit 0 the l1 error, bit 1 the coverage ratio (the others are written as 0)
};
constexpr int kIcpWantL1 = 1, kIcpWantRatio = 2;
constexpr int kIcpNoSecondLevel = 8;  // (same word) budgets only: pcs_icp_set_cache_levels(1)
constexpr int kIcpRebuildAlways = 4;  // (same word) the moving grid is rebuilt at every iteration: pcs_icp_set_grid_mode

__device__ __forceinline__ void icp_sizes(const IcpProblem& pr, int& Mn, int& Rn, int& Ra) {
  if (pr.cnt3) {
    Mn = (int)pr.cnt3[0];
    Ra = (int)pr.cnt3[1];
    Rn = (int)pr.cnt3[2];
  } else {
    Mn = pr.Mn;
    Rn = pr.Rn;
    Ra = pr.Ra;
  }
}

// host-side layout of one call's workspace
struct IcpBufs {
  IcpGeom* geom;
  float* bounds;
  IcpState* state;
  Grid2 gr, ga, gm[2];
  int32_t *nn_f, *nn_b, *wl, *wls;
  unsigned int* sstep;
  int32_t* mloc;
  float4 *rsrt, *msrt;
  int32_t* trow;
  float *bud_f, *bud_b;
  NnAux *aux_f, *aux_b;
  float* prev;
  long long* acc;
  double* Tb;
  double* Rt;
  long long* lossfx;
  float4* corg;
  int32_t *first, *cov;
};
bool plan_icp(Arena& a, int64_t Mn, int64_t Rn, int64_t Ra, int64_t C, IcpBufs& b);

// what a caller fills in for one registration (device pointers; sizes are capacities when cnt3 is set)
struct IcpArgs {
  float* mov;
  const int32_t* comp;
  const float* ref;
  const float* all;
  const int64_t* comp_deg;
  const int64_t* cnt3;
  int64_t Mn, Rn, Ra, C;
  double radius;
  int frame_offset;
  double reg, delta;
  int max_iter;
  float cell_size;
  double *T, *l1;
  float* ratio;
  // outputs the caller will not read are not computed (their arrays are written as zeros): the l1 error costs two
  // passes over the matched pairs, the coverage ratio a grid over all reference points and a search per moving point
  bool want_l1 = true, want_ratio = true;
};
// descriptor of a call on workspace `ws` (host side only; the caller uploads it into its table slot)
int icp_describe(const IcpArgs& a, void* ws, size_t ws_bytes, IcpProblem* out);

// launch geometry hints of one batch entry: expected sizes (only the grid dimensions derive from them)
struct IcpHint {
  int Mn, Rn, Ra;
  int64_t C;
  size_t clear_words;
};
// optional HIP-event timing of the unit-resident loop kernel (bench.py's roofline): after the stream has
// been synchronised, account the calling thread's last chunk: unit_iters = iterations its entries ran in
// the loop kernel, bytes = their algorithmic bytes
void icp_profile_account(double unit_iters, double bytes, int prof_slot, double bytes_8d = 0.0, double many_iters = 0.0,
                         double many_bytes_8d = 0.0);
// a registration has ended: its IcpState::phase_ticks join the profile
void icp_profile_phases(const unsigned long long* ticks4);
// enqueue: reset + bounds + grid description + the three grids of every entry
int icp_enqueue_begin(const IcpProblem* tab, const Batch& bt, const IcpHint* hints, hipStream_t s);
// enqueue one chunk of iterations for every entry. By default (pcs_icp_set_schedule) every iteration is a set of
// wide launches serving all entries (blockIdx.y = entry), 8 iterations per chunk; with a resident schedule the
// first `wide_iters` iterations of a registration are wide (the full searches of a first iteration want the
// whole chip) and ONE launch per chunk then gives every entry a workgroup of its own that runs its iterations
// back to back -- budget test, searches, solve, apply, stop rule, no launch in between -- until the stop rule
// fires, max_iter, n_it iterations of this chunk, or budget_us microseconds (0: no limit) have passed.
// How the iterations are cut into chunks and which schedule runs them never changes a result. The states
// are mirrored to out[e] (may be pinned host memory, may be nullptr).
int icp_enqueue_chunk(const IcpProblem* tab, const Batch& bt, const BatchInts& kbase, int n_it,
                      unsigned budget_us, const IcpHint* hints, IcpState* out, hipStream_t s, int prof_slot);
// enqueue: l1 error of the last iteration, coverage ratio, outputs
int icp_enqueue_finish(const IcpProblem* tab, const Batch& bt, const IcpHint* hints, IcpState* out,
                       hipStream_t s);

}  // namespace pcs
