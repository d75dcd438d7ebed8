// Small-state optimisation loops of the tracking stage as single persistent kernels.
//
// smooth_velo, pcdet/models/registration/preprocessors/cluster_tracking.py:162-199: up to 300
// AdamW steps (lr 1e-2, MultiStepLR [100,200,300] x0.1, betas (0.9, 0.999), eps 1e-8, weight decay
// 1e-2 on the WHOLE [C,F,3] tensor) on
//   loss = mean((v - d)^2)[.., a:b+1, :2] + 10 * mean(|v_t - v_{t+1}|)[.., a:b, :2]
// with a 3-strike stop at loss improvement < 1e-3. The reference runs forward + backward + step +
// scheduler + loss.item() per iteration (about 40 launches and a host sync, 300 times, per tracked
// frame). Here: one block iterates the in-range elements with the gradients in closed form and the
// loss reduced through LDS; a second elementwise kernel applies the same sequence of decoupled
// weight-decay factors to the elements that never see a gradient.
#include "pcs_common.hpp"

namespace pcs {

struct SmoothState {
  int iters;
  float last_loss;
};

// ---- geometry of the multi-workgroup form (see sv2_body)
constexpr int SV2_BLOCK = 256;     // threads per workgroup
constexpr int SV2_ELEMS = 4;       // in-range elements per thread
constexpr int SV2_WINDOW = 64;     // AdamW steps per launch of the window pass
constexpr int SV2_MAX_BLOCKS = 4096;
constexpr int SV_MAX_ITR = 512;
constexpr int SV2_MAX_WINDOWS = SV_MAX_ITR / SV2_WINDOW;

// stop-rule state of one multi-workgroup problem, carried from window to window
struct SmoothWin {
  unsigned int arrive[SV2_MAX_WINDOWS];  // workgroups that have finished window w
  int decided;      // the stop rule has fired, or num_itr steps are done: iters is final
  int stop_window;  // the window in which that happened
  int countdown;
  float last_loss;
};

// one smooth_velo problem of a batched launch (the sequence tracker, pcs_tracker.hip)
struct SvJob {
  float* v;
  const float* d;
  SmoothState* st;
  SmoothWin* win;
  double* partial;  // [SV2_WINDOW, blocks] per-workgroup loss terms of the current window
  float* buf;       // kind 1: checkpoints [2][3][C * nf * 2]; kind 2: Adam moments m, sq [C * F * 2] each
  int C, F, a, b, cpb, blocks, kind;  // kind 0: one workgroup, LDS-resident, 1: multi-workgroup, 2: generic
};
constexpr int kSvMaxBatch = 48;  // jobs per launch (the list travels by value: 80 B each, < 4 KB of kernarg)
struct SvBatch {
  int n;
  SvJob j[kSvMaxBatch];
};
static_assert(sizeof(SvBatch) <= 3900, "smooth_velo job list must fit the kernel-argument segment");

// MultiStepLR([100, 200, 300], gamma 0.1); the optimiser forms its scalars in double (Python
// floats) and hands them to fp32 tensor ops
__device__ __forceinline__ double lr_at(int step) {
  double lr = 1e-2;
  if (step >= 100) lr *= 0.1;
  if (step >= 200) lr *= 0.1;
  if (step >= 300) lr *= 0.1;
  return lr;
}

constexpr int SV_BLOCK = 1024;

// v, d: [C, F, 3] f32. In-range elements: (c, t in [a, b], k in {0,1}). m, s: Adam moments for the
// in-range elements, laid out [C, nf, 2].
__device__ __forceinline__ void
smooth_velo_body(float* __restrict__ v, const float* __restrict__ d, int C, int F, int a, int b,
                 float weight0, float weight, int num_itr, float stopping,
                 float* __restrict__ m, float* __restrict__ s, SmoothState* __restrict__ st) {
  __shared__ double red[SV_BLOCK / 64];
  __shared__ int stop_flag;
  int nf = b - a + 1;
  int n_in = C * nf * 2;
  float inv_fit = 1.0f / (float)(C * nf * 2);
  float inv_sm = 1.0f / (float)(C * (nf - 1) * 2);
  float last_loss = 1e10f;
  int countdown = 3;
  int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int e = tid; e < n_in; e += SV_BLOCK) { m[e] = 0.f; s[e] = 0.f; }
  if (tid == 0) stop_flag = 0;
  __syncthreads();
  int it = 0;
  for (; it < num_itr; ++it) {
    double lr = lr_at(it);
    // ---- loss (f64 partial sums of fp32 terms) ----
    double part = 0.0;
    for (int e = tid; e < n_in; e += SV_BLOCK) {
      int k = e & 1;
      int t = (e >> 1) % nf;
      int c = (e >> 1) / nf;
      int64_t at = ((int64_t)c * F + (a + t)) * 3 + k;
      float x = v[at] - d[at];
      part += (double)(x * x) * inv_fit * weight0;
      if (t < nf - 1) {
        float y = v[at] - v[at + 3];
        part += (double)fabsf(y) * inv_sm * weight;
      }
    }
    part = wave_sum_f64(part);
    if (lane == 0) red[w] = part;
    __syncthreads();
    float loss = 0.f;
    if (tid == 0) {
      double tot = 0.0;
      for (int q = 0; q < SV_BLOCK / 64; ++q) tot += red[q];
      loss = (float)tot;
    }
    // ---- gradients at the current iterate + AdamW update (needs all reads before any write:
    //      each thread caches the neighbours it needs, then a barrier, then writes) ----
    double bc1 = 1.0 - pow(0.9, (double)(it + 1));
    double bc2 = 1.0 - pow(0.999, (double)(it + 1));
    float step_size = (float)(lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    float decay = (float)(1.0 - lr * 1e-2);
    // two-phase: compute new values into registers per strided element is not possible for
    // arbitrary n_in, so gradients are staged in s-adjacent scratch: reuse `red` barrier + global g
    for (int e = tid; e < n_in; e += SV_BLOCK) {
      int k = e & 1;
      int t = (e >> 1) % nf;
      int c = (e >> 1) / nf;
      int64_t at = ((int64_t)c * F + (a + t)) * 3 + k;
      float x = v[at];
      float g = (x - d[at]) * 2.f * (inv_fit * weight0);
      if (t < nf - 1) {
        float y = x - v[at + 3];
        g += (y > 0.f ? 1.f : (y < 0.f ? -1.f : 0.f)) * (inv_sm * weight);
      }
      if (t > 0) {
        float y = v[at - 3] - x;
        g -= (y > 0.f ? 1.f : (y < 0.f ? -1.f : 0.f)) * (inv_sm * weight);
      }
      // stash the gradient in the moment buffer's spare slot: m' = lerp(m, g, 0.1) is a pure
      // function of (m, g), so compute the new moments now and write them (they are private to e)
      float me = m[e] + 0.1f * (g - m[e]);
      float se = s[e] * 0.999f + 0.001f * g * g;
      m[e] = me;
      s[e] = se;
    }
    __syncthreads();  // every gradient has been formed from the OLD v before any v is written
    for (int e = tid; e < n_in; e += SV_BLOCK) {
      int k = e & 1;
      int t = (e >> 1) % nf;
      int c = (e >> 1) / nf;
      int64_t at = ((int64_t)c * F + (a + t)) * 3 + k;
      float x = v[at] * decay;
      float denom = sqrtf(s[e]) / bc2_sqrt + 1e-8f;
      v[at] = x - step_size * (m[e] / denom);
    }
    if (tid == 0) {
      if (last_loss - loss < stopping) countdown -= 1;
      else countdown = 3;
      if (countdown <= 0) stop_flag = 1;
      else last_loss = loss;
    }
    __syncthreads();
    if (stop_flag) { ++it; break; }
  }
  if (tid == 0) { st->iters = it; st->last_loss = last_loss; }
}

// The same loop with the state on chip: v in LDS, d and the Adam moments in registers (<= SV_ELEMS
// elements per thread), the per-step scalars tabulated once. Identical arithmetic and the same
// element -> thread map / reduction tree as smooth_velo_kernel; used when the in-range elements fit.
constexpr int SV_ELEMS = 8;

__device__ __forceinline__ void
smooth_velo_lds_body(float* __restrict__ sv /* LDS [n_in] current iterate */, float* __restrict__ v,
                     const float* __restrict__ d, int C, int F, int a, int b, float weight0, float weight,
                     int num_itr, float stopping, SmoothState* __restrict__ st) {
  __shared__ double red[SV_BLOCK / 64];
  __shared__ float t_step[SV_MAX_ITR], t_bc2[SV_MAX_ITR], t_decay[SV_MAX_ITR];
  __shared__ int stop_flag;
  const int nf = b - a + 1;
  const int n_in = C * nf * 2;
  const float inv_fit = 1.0f / (float)(C * nf * 2);
  const float inv_sm = 1.0f / (float)(C * (nf - 1) * 2);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float last_loss = 1e10f;
  int countdown = 3;
  float de[SV_ELEMS], me[SV_ELEMS], se[SV_ELEMS];
  int tpos[SV_ELEMS];  // 0: interior, bit0: first frame of the range, bit1: last frame
#pragma unroll
  for (int j = 0; j < SV_ELEMS; ++j) {
    int e = tid + j * SV_BLOCK;
    de[j] = 0.f; me[j] = 0.f; se[j] = 0.f; tpos[j] = 0;
    if (e < n_in) {
      int k = e & 1;
      int t = (e >> 1) % nf;
      int c = (e >> 1) / nf;
      int64_t at = ((int64_t)c * F + (a + t)) * 3 + k;
      sv[e] = v[at];
      de[j] = d[at];
      tpos[j] = (t == 0 ? 1 : 0) | (t == nf - 1 ? 2 : 0);
    }
  }
  for (int it = tid; it < num_itr; it += SV_BLOCK) {
    double lr = lr_at(it);
    double bc1 = 1.0 - pow(0.9, (double)(it + 1));
    double bc2 = 1.0 - pow(0.999, (double)(it + 1));
    t_step[it] = (float)(lr / bc1);
    t_bc2[it] = (float)sqrt(bc2);
    t_decay[it] = (float)(1.0 - lr * 1e-2);
  }
  if (tid == 0) stop_flag = 0;
  __syncthreads();
  int it = 0;
  for (; it < num_itr; ++it) {
    double part = 0.0;
    float g[SV_ELEMS];
#pragma unroll
    for (int j = 0; j < SV_ELEMS; ++j) {
      int e = tid + j * SV_BLOCK;
      g[j] = 0.f;
      if (e < n_in) {
        float x = sv[e];
        float r = x - de[j];
        part += (double)(r * r) * inv_fit * weight0;
        float gj = r * 2.f * (inv_fit * weight0);
        if (!(tpos[j] & 2)) {
          float y = x - sv[e + 2];
          part += (double)fabsf(y) * inv_sm * weight;
          gj += (y > 0.f ? 1.f : (y < 0.f ? -1.f : 0.f)) * (inv_sm * weight);
        }
        if (!(tpos[j] & 1)) {
          float y = sv[e - 2] - x;
          gj -= (y > 0.f ? 1.f : (y < 0.f ? -1.f : 0.f)) * (inv_sm * weight);
        }
        g[j] = gj;
      }
    }
    part = wave_sum_f64(part);
    if (lane == 0) red[w] = part;
    __syncthreads();  // partial sums visible; every read of the old iterate is done
    const float step_size = t_step[it], bc2_sqrt = t_bc2[it], decay = t_decay[it];
#pragma unroll
    for (int j = 0; j < SV_ELEMS; ++j) {
      int e = tid + j * SV_BLOCK;
      if (e < n_in) {
        me[j] = me[j] + 0.1f * (g[j] - me[j]);
        se[j] = se[j] * 0.999f + 0.001f * g[j] * g[j];
        float x = sv[e] * decay;
        float denom = sqrtf(se[j]) / bc2_sqrt + 1e-8f;
        sv[e] = x - step_size * (me[j] / denom);
      }
    }
    if (tid == 0) {
      double tot = 0.0;
      for (int q = 0; q < SV_BLOCK / 64; ++q) tot += red[q];
      float loss = (float)tot;
      if (last_loss - loss < stopping) countdown -= 1;
      else countdown = 3;
      if (countdown <= 0) stop_flag = 1;
      else last_loss = loss;
    }
    __syncthreads();
    if (stop_flag) { ++it; break; }
  }
#pragma unroll
  for (int j = 0; j < SV_ELEMS; ++j) {
    int e = tid + j * SV_BLOCK;
    if (e < n_in) {
      int k = e & 1;
      int t = (e >> 1) % nf;
      int c = (e >> 1) / nf;
      v[((int64_t)c * F + (a + t)) * 3 + k] = sv[e];
    }
  }
  if (tid == 0) { st->iters = it; st->last_loss = last_loss; }
}

// Many components (thousands of small clusters): the same loop spread over many 256-thread
// workgroups, each owning a contiguous range of components (so the t +- 1 neighbours stay local) with
// its slice of the iterate in LDS. The ONLY coupling between workgroups is the scalar loss of the
// 3-strike stop rule, and the trajectory of the iterate does not depend on when the rule fires. So
// no workgroup ever waits for another one:
//   window pass (sv2_body<false>, one launch per window of SV2_WINDOW steps): every workgroup runs the
//     window's steps on its slice, stores its loss term of each step (f64) in partial[step][workgroup]
//     and leaves its state (iterate + Adam moments) in a checkpoint, double-buffered by window parity.
//     The workgroup that arrives LAST (arrival counter; nobody polls it) sums the terms of every step
//     in workgroup order and walks the stop rule through the window; when it fires (or num_itr is
//     reached) it stores the final step count. The launches of later windows find `decided` and return.
//   replay (sv2_body<true>, one launch): every workgroup restarts from the checkpoint BEFORE the window
//     in which the rule fired, repeats exactly the steps up to the stop, and writes its slice back.
// Same arithmetic per element as the one-workgroup forms; the loss of a step is the sum of the
// workgroups' terms in workgroup order whichever workgroup happens to add them up.
template <bool REPLAY>
__device__ __forceinline__ void
sv2_body(float* __restrict__ sv /* LDS [SV2_BLOCK * SV2_ELEMS] */, const SvJob& j, int bx, int w, float weight0,
         float weight, int num_itr, float stopping) {
  __shared__ double red[SV2_BLOCK / 64];
  __shared__ double losses[SV2_WINDOW];
  __shared__ float t_step[SV2_WINDOW], t_bc2[SV2_WINDOW], t_decay[SV2_WINDOW];
  __shared__ int is_last;
  SmoothWin* __restrict__ win = j.win;
  int it0, it1;
  if (REPLAY) {
    w = win->stop_window;
    it0 = w * SV2_WINDOW;
    it1 = j.st->iters;
  } else {
    if (win->decided) return;  // (written by an earlier launch: the same for every workgroup)
    it0 = w * SV2_WINDOW;
    it1 = min(num_itr, it0 + SV2_WINDOW);
  }
  const int B = j.blocks;
  const int nf = j.b - j.a + 1;
  const int c0 = bx * j.cpb;
  const int c1 = min(j.C, c0 + j.cpb);
  const int n_loc = (c1 - c0) * nf * 2;
  const int64_t n_tot = (int64_t)j.C * nf * 2;
  const int64_t g0 = (int64_t)c0 * nf * 2;
  const float inv_fit = 1.0f / (float)(j.C * nf * 2);
  const float inv_sm = 1.0f / (float)(j.C * (nf - 1) * 2);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* __restrict__ ck_in = j.buf + (size_t)((w + 1) & 1) * 3 * n_tot;  // what window w - 1 left
  float* __restrict__ ck_out = j.buf + (size_t)(w & 1) * 3 * n_tot;
  float de[SV2_ELEMS], me[SV2_ELEMS], se[SV2_ELEMS];
  int tpos[SV2_ELEMS];  // 0: interior, bit0: first frame of the range, bit1: last frame
#pragma unroll
  for (int q = 0; q < SV2_ELEMS; ++q) {
    int e = tid + q * SV2_BLOCK;
    de[q] = 0.f; me[q] = 0.f; se[q] = 0.f; tpos[q] = 0;
    if (e < n_loc) {
      int k = e & 1;
      int t = (e >> 1) % nf;
      int c = c0 + (e >> 1) / nf;
      int64_t at = ((int64_t)c * j.F + (j.a + t)) * 3 + k;
      de[q] = j.d[at];
      tpos[q] = (t == 0 ? 1 : 0) | (t == nf - 1 ? 2 : 0);
      if (w == 0) {
        sv[e] = j.v[at];
      } else {
        sv[e] = ck_in[g0 + e];
        me[q] = ck_in[n_tot + g0 + e];
        se[q] = ck_in[2 * n_tot + g0 + e];
      }
    }
  }
  for (int q = tid; q < it1 - it0; q += SV2_BLOCK) {
    const int it = it0 + q;
    double lr = lr_at(it);
    double bc1 = 1.0 - pow(0.9, (double)(it + 1));
    double bc2 = 1.0 - pow(0.999, (double)(it + 1));
    t_step[q] = (float)(lr / bc1);
    t_bc2[q] = (float)sqrt(bc2);
    t_decay[q] = (float)(1.0 - lr * 1e-2);
  }
  __syncthreads();
  for (int it = it0; it < it1; ++it) {
    double part = 0.0;
    float gr[SV2_ELEMS];
#pragma unroll
    for (int q = 0; q < SV2_ELEMS; ++q) {
      int e = tid + q * SV2_BLOCK;
      gr[q] = 0.f;
      if (e < n_loc) {
        float x = sv[e];
        float r = x - de[q];
        if (!REPLAY) part += (double)(r * r) * inv_fit * weight0;
        float gj = r * 2.f * (inv_fit * weight0);
        if (!(tpos[q] & 2)) {
          float y = x - sv[e + 2];
          if (!REPLAY) part += (double)fabsf(y) * inv_sm * weight;
          gj += (y > 0.f ? 1.f : (y < 0.f ? -1.f : 0.f)) * (inv_sm * weight);
        }
        if (!(tpos[q] & 1)) {
          float y = sv[e - 2] - x;
          gj -= (y > 0.f ? 1.f : (y < 0.f ? -1.f : 0.f)) * (inv_sm * weight);
        }
        gr[q] = gj;
      }
    }
    if (!REPLAY) {
      part = wave_sum_f64(part);
      if (lane == 0) red[wv] = part;
    }
    __syncthreads();  // partial sums visible; every read of the old iterate is done
    if (!REPLAY && tid == 0) {
      double tot = 0.0;
      for (int q = 0; q < SV2_BLOCK / 64; ++q) tot += red[q];
      __hip_atomic_store(&j.partial[(size_t)(it - it0) * B + bx], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float step_size = t_step[it - it0], bc2_sqrt = t_bc2[it - it0], decay = t_decay[it - it0];
#pragma unroll
    for (int q = 0; q < SV2_ELEMS; ++q) {
      int e = tid + q * SV2_BLOCK;
      if (e < n_loc) {
        me[q] = me[q] + 0.1f * (gr[q] - me[q]);
        se[q] = se[q] * 0.999f + 0.001f * gr[q] * gr[q];
        float x = sv[e] * decay;
        float denom = sqrtf(se[q]) / bc2_sqrt + 1e-8f;
        sv[e] = x - step_size * (me[q] / denom);
      }
    }
    __syncthreads();
  }
  if (REPLAY) {
#pragma unroll
    for (int q = 0; q < SV2_ELEMS; ++q) {
      int e = tid + q * SV2_BLOCK;
      if (e < n_loc) {
        int k = e & 1;
        int t = (e >> 1) % nf;
        int c = c0 + (e >> 1) / nf;
        j.v[((int64_t)c * j.F + (j.a + t)) * 3 + k] = sv[e];
      }
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < SV2_ELEMS; ++q) {
    int e = tid + q * SV2_BLOCK;
    if (e < n_loc) {
      ck_out[g0 + e] = sv[e];
      ck_out[n_tot + g0 + e] = me[q];
      ck_out[2 * n_tot + g0 + e] = se[q];
    }
  }
  // the workgroup that arrives last walks the stop rule through the window (thread 0 wrote this
  // workgroup's terms itself, so its release covers them; nobody waits for anybody)
  if (tid == 0) {
    unsigned old = __hip_atomic_fetch_add(&win->arrive[w], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (old == (unsigned)(B - 1));
  }
  __syncthreads();
  if (!is_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const int n = it1 - it0;
  for (int q = tid; q < n; q += SV2_BLOCK) {
    double tot = 0.0;
    for (int p = 0; p < B; ++p)
      tot += __hip_atomic_load(&j.partial[(size_t)q * B + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    losses[q] = tot;
  }
  __syncthreads();
  if (tid == 0) {
    float last_loss = w == 0 ? 1e10f : win->last_loss;
    int countdown = w == 0 ? 3 : win->countdown;
    int stop_at = -1;
    for (int q = 0; q < n; ++q) {
      float loss = (float)losses[q];
      if (last_loss - loss < stopping) countdown -= 1;
      else countdown = 3;
      if (countdown <= 0) { stop_at = it0 + q + 1; break; }
      last_loss = loss;
    }
    if (stop_at < 0 && it1 == num_itr) stop_at = num_itr;
    if (stop_at >= 0) {
      j.st->iters = stop_at;
      j.st->last_loss = last_loss;
      win->stop_window = w;
      win->decided = 1;
    } else {
      win->last_loss = last_loss;
      win->countdown = countdown;
    }
  }
}

// elements outside (c, [a,b], {0,1}) only see the decoupled weight decay of every executed step
__device__ __forceinline__ void
smooth_velo_decay_body(float* __restrict__ v, int64_t total, int F, int a, int b,
                       const SmoothState* __restrict__ st) {
  int n = st->iters;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int k = (int)(i % 3);
    int t = (int)((i / 3) % F);
    if (k < 2 && t >= a && t <= b) continue;
    float x = v[i];
    for (int it = 0; it < n; ++it) x = x * (float)(1.0 - lr_at(it) * 1e-2);
    v[i] = x;
  }
}

// ---- batched launches (blockIdx.y = problem; pcs_smooth_velo is a batch of one)
__global__ void __launch_bounds__(SV_BLOCK)
smooth_velo_batch_kernel(SvBatch sb, int kind, float weight0, float weight, int num_itr, float stopping) {
  extern __shared__ float sv_dyn[];
  const SvJob& j = sb.j[blockIdx.y];
  if (j.kind != kind) return;
  if (kind == 0)
    smooth_velo_lds_body(sv_dyn, j.v, j.d, j.C, j.F, j.a, j.b, weight0, weight, num_itr, stopping, j.st);
  else
    smooth_velo_body(j.v, j.d, j.C, j.F, j.a, j.b, weight0, weight, num_itr, stopping, j.buf,
                     j.buf + (size_t)j.C * j.F * 2, j.st);
}
__global__ void __launch_bounds__(SV2_BLOCK)
smooth_velo_window_kernel(SvBatch sb, int w, float weight0, float weight, int num_itr, float stopping) {
  __shared__ float sv[SV2_BLOCK * SV2_ELEMS];
  const SvJob& j = sb.j[blockIdx.y];
  if (j.kind != 1 || (int)blockIdx.x >= j.blocks) return;
  sv2_body<false>(sv, j, blockIdx.x, w, weight0, weight, num_itr, stopping);
}
__global__ void __launch_bounds__(SV2_BLOCK)
smooth_velo_replay_kernel(SvBatch sb, float weight0, float weight, int num_itr, float stopping) {
  __shared__ float sv[SV2_BLOCK * SV2_ELEMS];
  const SvJob& j = sb.j[blockIdx.y];
  if (j.kind != 1 || (int)blockIdx.x >= j.blocks) return;
  sv2_body<true>(sv, j, blockIdx.x, 0, weight0, weight, num_itr, stopping);
}
__global__ void __launch_bounds__(64)
smooth_velo_batch_reset_kernel(SvBatch sb) {
  const SvJob& j = sb.j[blockIdx.x];
  if (threadIdx.x < SV2_MAX_WINDOWS) j.win->arrive[threadIdx.x] = 0u;
  if (threadIdx.x == 0) {
    j.win->decided = 0;
    j.win->stop_window = 0;
    j.st->iters = 0;
  }
}
__global__ void __launch_bounds__(kBlock)
smooth_velo_batch_decay_kernel(SvBatch sb) {
  const SvJob& j = sb.j[blockIdx.y];
  smooth_velo_decay_body(j.v, (int64_t)j.C * j.F * 3, j.F, j.a, j.b, j.st);
}


// l1_minimization, pcdet/models/registration/preprocessors/preprocessor_utils.py:313-350: ground
// height per pillar h[X,Y] minimising
//   mean|(h - min_z) w| + rigid * (left + up + t1 + t2),  each a mean of |second difference * (w+1e-2)|
// along x, y and the two diagonals, by AdamW (lr, MultiStepLR(decay_steps) x0.1, wd 1e-2), at most
// max_iters steps, stop when 3 consecutive improvements are < 1e-4. The reference does forward +
// backward + step + loss.item() per iteration (~60 launches and a sync, up to 10 000 times per
// sequence); here one block keeps h in LDS and iterates with gradients in closed form.
constexpr int L1_BLOCK = 1024;
constexpr int L1_MAX_CELLS = 12288;  // 48 KiB of LDS for h

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// Round 5: the Adam moments of a thread's cells live in registers (a thread owns cells tid, tid + 1024, ...: the same
// ones at every step), the stencil weights weight + 1e-2 and min_z in LDS beside h (144 of the CU's 160 KB) -- a
// step reads one cached global word per cell (it was two loads and two stores of m / v and a dozen weight loads). The same operations in the same order as before: the same bits.
constexpr int L1_PER_THREAD = L1_MAX_CELLS / L1_BLOCK;
__global__ void __launch_bounds__(L1_BLOCK)
l1_min_kernel(const float* __restrict__ min_z, const float* __restrict__ weight, int X, int Y,
              double lr0, const int* __restrict__ decay_steps, int n_decay, float rigid,
              int max_iters, float* __restrict__ h_out, float* __restrict__ m_unused,
              float* __restrict__ v_unused, SmoothState* __restrict__ st) {
  __shared__ float h[L1_MAX_CELLS];
  __shared__ float sw[L1_MAX_CELLS];  // weight + 1e-2: the stencils' factor
  __shared__ float smz[L1_MAX_CELLS];
  __shared__ double red[L1_BLOCK / 64];
  __shared__ int stop_flag;
  (void)m_unused;
  (void)v_unused;
  int n = X * Y;
  int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float mr[L1_PER_THREAD], vr[L1_PER_THREAD];
#pragma unroll
  for (int k = 0; k < L1_PER_THREAD; ++k) {
    const int e = tid + k * L1_BLOCK;
    mr[k] = 0.f;
    vr[k] = 0.f;
    if (e < n) { h[e] = 0.f; sw[e] = weight[e] + 1e-2f; smz[e] = min_z[e]; }
  }
  if (tid == 0) stop_flag = 0;
  __syncthreads();
  float i1 = 1.f / (float)(X * Y), iL = 1.f / (float)((X - 2) * Y), iU = 1.f / (float)(X * (Y - 2)),
        iT = 1.f / (float)((X - 2) * (Y - 2));
  float last_loss = 1e10f;
  int countdown = 3, it = 0;
  // beta^(it + 1) as running products: only IEEE multiplications, so the CPU restatement
  // (oracle/pcs_oracle_ground.cpp) forms the same bits (pow() is not the same function everywhere)
  double b1p = 1.0, b2p = 1.0;
  auto H = [&](int i, int j) { return h[i * Y + j]; };
  auto W = [&](int i, int j) { return sw[i * Y + j]; };
  for (; it < max_iters; ++it) {
    double lr = lr0;
    for (int q = 0; q < n_decay; ++q) if (it >= decay_steps[q]) lr *= 0.1;
    double part = 0.0;
    // gradient of every cell from the OLD h (h is written only after the barrier)
#pragma unroll
    for (int k = 0; k < L1_PER_THREAD; ++k) {
      const int e = tid + k * L1_BLOCK;
      if (e < n) {
        int i = e / Y, j = e - i * Y;
        float g = 0.f;
        const float wte = weight[e];  // (cached: the one global read of a cell's step)
        float a = (H(i, j) - smz[e]) * wte;
        part += (double)(fabsf(a) * i1);
        g += sgn(a) * wte * i1;
        // second differences along x: centres ci = i-1, i, i+1
#pragma unroll 1
        for (int dc = -1; dc <= 1; ++dc) {
          int ci = i + dc;
          if (ci >= 1 && ci <= X - 2) {
            float t = (H(ci - 1, j) - 2.f * H(ci, j) + H(ci + 1, j)) * W(ci, j);
            float coef = dc == 0 ? -2.f : 1.f;
            g += rigid * sgn(t) * W(ci, j) * coef * iL;
            if (dc == 0) part += (double)(rigid * fabsf(t) * iL);
          }
          int cj = j + dc;
          if (cj >= 1 && cj <= Y - 2) {
            float t = (H(i, cj - 1) - 2.f * H(i, cj) + H(i, cj + 1)) * W(i, cj);
            float coef = dc == 0 ? -2.f : 1.f;
            g += rigid * sgn(t) * W(i, cj) * coef * iU;
            if (dc == 0) part += (double)(rigid * fabsf(t) * iU);
          }
          // diagonal (i-1,j-1),(i,j),(i+1,j+1): this cell is the centre shifted by dc along it
          int di = i + dc, dj = j + dc;
          if (di >= 1 && di <= X - 2 && dj >= 1 && dj <= Y - 2) {
            float t = (H(di - 1, dj - 1) - 2.f * H(di, dj) + H(di + 1, dj + 1)) * W(di, dj);
            float coef = dc == 0 ? -2.f : 1.f;
            g += rigid * sgn(t) * W(di, dj) * coef * iT;
            if (dc == 0) part += (double)(rigid * fabsf(t) * iT);
          }
          // anti-diagonal (i+1,j-1),(i,j),(i-1,j+1)
          int ai = i - dc, aj = j + dc;
          if (ai >= 1 && ai <= X - 2 && aj >= 1 && aj <= Y - 2) {
            float t = (H(ai + 1, aj - 1) - 2.f * H(ai, aj) + H(ai - 1, aj + 1)) * W(ai, aj);
            float coef = dc == 0 ? -2.f : 1.f;
            g += rigid * sgn(t) * W(ai, aj) * coef * iT;
            if (dc == 0) part += (double)(rigid * fabsf(t) * iT);
          }
        }
        mr[k] = mr[k] + 0.1f * (g - mr[k]);
        vr[k] = vr[k] * 0.999f + 0.001f * g * g;
      }
    }
    part = wave_sum_f64(part);
    if (lane == 0) red[w] = part;
    __syncthreads();
    b1p *= 0.9;
    b2p *= 0.999;
    double bc1 = 1.0 - b1p;
    double bc2 = 1.0 - b2p;
    float step_size = (float)(lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    float decay = (float)(1.0 - lr * 1e-2);
#pragma unroll
    for (int k = 0; k < L1_PER_THREAD; ++k) {
      const int e = tid + k * L1_BLOCK;
      if (e < n) {
        float x = h[e] * decay;
        float denom = sqrtf(vr[k]) / bc2_sqrt + 1e-8f;
        h[e] = x - step_size * (mr[k] / denom);
      }
    }
    if (tid == 0) {
      double tot = 0.0;
      for (int q = 0; q < L1_BLOCK / 64; ++q) tot += red[q];
      float loss = (float)tot;
      if (last_loss - loss < 1e-4f) countdown -= 1;
      else countdown = 3;
      if (countdown == 0) stop_flag = 1;
      else last_loss = loss;
    }
    __syncthreads();
    if (stop_flag) { ++it; break; }
  }
  for (int e = tid; e < n; e += L1_BLOCK) h_out[e] = h[e];
  if (tid == 0 && st) { st->iters = it; st->last_loss = last_loss; }
}


// iterative_reweighted_ransac, pcdet/models/registration/preprocessors/preprocessor_utils.py:32-80:
// per pillar, <= 50 rounds of {weighted centre, weighted covariance / count, normal = eigenvector of
// the smallest eigenvalue, Cauchy weights on the point-plane error times a 0.5 m distance falloff},
// stopped when no weight moves by more than `delta`. The reference issues ~25 torch kernels, a
// batched eigh and a host sync per round (x 50 rounds x 30 height ratios per sequence).
// Here a round is two kernels: `irls_accumulate` (one pass over the voxels: new weight from the
// previous plane, raw weighted moments about the pillar's reference point via wave-aggregated f64
// atomics, max weight change) and `irls_solve` (one thread per pillar: centre, covariance, 3x3
// symmetric Jacobi eigen-solve). The convergence test runs on the device (rounds after it are
// skipped), so the host never synchronises inside the loop.
struct IrlsState {
  int stop_round;   // rounds > stop_round do not run
  int rounds;
  unsigned max_dw_bits;  // max |new_w - w| of the current round (float bits, >= 0)
  int pad;
};

constexpr int kIrlsAcc = 12;
constexpr int kIrlsSlots = 16;  // sum w, sum w x (3), sum w x x^T (6: xx xy xz yy yz zz), count, pad

__global__ void irls_init_kernel(IrlsState* st, double* acc, int64_t P) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i == 0) { st->stop_round = 0x7fffffff; st->rounds = 0; st->max_dw_bits = 0; }
  for (int64_t q = i; q < P * kIrlsAcc; q += (int64_t)gridDim.x * blockDim.x) acc[q] = 0.0;
}

// round r (1-based). r == 1 uses the given weights w0; later rounds derive the weight from the
// plane of round r-1 (centre / normal in `plane`), exactly the reference's update order.
__global__ void __launch_bounds__(kBlock)
irls_accumulate_kernel(int r, const float* __restrict__ xyz, const int64_t* __restrict__ pidx,
                       int64_t n, const float* __restrict__ origin, float sigma2, float delta,
                       float* __restrict__ w, const float* __restrict__ plane,
                       float* __restrict__ perr, double* __restrict__ acc,
                       IrlsState* __restrict__ st) {
  if (r > st->stop_round) return;
  __shared__ int skey[kIrlsSlots];
  __shared__ double sacc[kIrlsSlots][11];
  for (int q = threadIdx.x; q < kIrlsSlots; q += kBlock) skey[q] = -1;
  for (int q = threadIdx.x; q < kIrlsSlots * 11; q += kBlock) (&sacc[0][0])[q] = 0.0;
  __syncthreads();
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool active = i < n;
  int c = 0;
  double v[11];
#pragma unroll
  for (int q = 0; q < 11; ++q) v[q] = 0.0;
  float dw = 0.f;
  if (active) {
    c = (int)pidx[i];
    float x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    float wi = w[i];
    if (r > 1) {
      const float* pl = plane + (int64_t)c * 6;  // centre (3), normal (3) of the previous round
      float dx = x - pl[0], dy = y - pl[1], dz = z - pl[2];
      float e = fabsf(dx * pl[3] + dy * pl[4] + dz * pl[5]);
      perr[i] = e;
      float nw = sigma2 / (e * e + sigma2);
      float dist_w = 0.25f / (dx * dx + dy * dy + dz * dz + 0.25f);
      nw = nw * dist_w;
      dw = fabsf(nw - wi);
      // the reference tests max|new_w - w| < delta BEFORE adopting new_w: if the test passes the
      // loop ends with the old weights, so the accumulation below must then be discarded. The
      // solve kernel handles that (it ignores this round's sums when the test passed).
      wi = nw;
      w[i] = nw;
    }
    float ox = x - origin[c * 3], oy = y - origin[c * 3 + 1], oz = z - origin[c * 3 + 2];
    double W = wi;
    v[0] = W; v[1] = W * ox; v[2] = W * oy; v[3] = W * oz;
    v[4] = W * ox * ox; v[5] = W * ox * oy; v[6] = W * ox * oz;
    v[7] = W * oy * oy; v[8] = W * oy * oz; v[9] = W * oz * oz;
    v[10] = 1.0;
  }
  float mx = wave_max(dw);
  if ((threadIdx.x & 63) == 0 && r > 1) atomicMax(&st->max_dw_bits, __float_as_uint(mx));
  // per-workgroup LDS table keyed by pillar (voxels are sorted by pillar: a workgroup sees one or
  // two pillars), flushed with one global f64 atomic per (pillar, accumulator)
  int found = -1;
  if (active) {
    int slot = (int)(((unsigned)c * 2654435761u) >> 28) & (kIrlsSlots - 1);
    for (int t = 0; t < kIrlsSlots; ++t) {
      int old = atomicCAS(&skey[slot], -1, c);
      if (old == -1 || old == c) { found = slot; break; }
      slot = (slot + 1) & (kIrlsSlots - 1);
    }
    if (found >= 0) {
#pragma unroll
      for (int q = 0; q < 11; ++q) atomicAdd(&sacc[found][q], v[q]);
    } else {
#pragma unroll
      for (int q = 0; q < 11; ++q) atomicAdd(&acc[(int64_t)c * kIrlsAcc + q], v[q]);
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < kIrlsSlots * 11; q += kBlock) {
    int sl = q / 11, a = q - sl * 11;
    int key = skey[sl];
    if (key >= 0) atomicAdd(&acc[(int64_t)key * kIrlsAcc + a], sacc[sl][a]);
  }
}

// smallest-eigenvalue eigenvector of a symmetric 3x3 (cyclic Jacobi, f32 like the reference's eigh)
__device__ __forceinline__ void eig3_smallest(const float A_in[6], float nrm[3]) {
  // A = [[a0 a1 a2],[a1 a3 a4],[a2 a4 a5]]
  double A[3][3] = {{A_in[0], A_in[1], A_in[2]}, {A_in[1], A_in[3], A_in[4]}, {A_in[2], A_in[4], A_in[5]}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    double dia = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
    if (off <= 1e-14 * dia || off == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  // (selects, not A[m][m] / V[k][m] with a computed m: indexed, the matrices went through scratch memory)
  int m = 0;
  double dm = A[0][0];
  if (A[1][1] < dm) { m = 1; dm = A[1][1]; }
  if (A[2][2] < dm) m = 2;
  const double v0 = m == 0 ? V[0][0] : (m == 1 ? V[0][1] : V[0][2]);
  const double v1 = m == 0 ? V[1][0] : (m == 1 ? V[1][1] : V[1][2]);
  const double v2 = m == 0 ? V[2][0] : (m == 1 ? V[2][1] : V[2][2]);
  double sgn = v2 < 0 ? -1.0 : 1.0;  // canonical sign: normal points up
  nrm[0] = (float)(sgn * v0);
  nrm[1] = (float)(sgn * v1);
  nrm[2] = (float)(sgn * v2);
}

__global__ void __launch_bounds__(kBlock)
irls_solve_kernel(int r, int64_t P, const float* __restrict__ origin, float delta,
                  double* __restrict__ acc, float* __restrict__ plane, IrlsState* __restrict__ st) {
  if (r > st->stop_round) return;
  // convergence of the reference: round r computed new weights from plane r-1; if none moved by
  // delta or more the loop ends and plane r-1 is the result
  bool converged = r > 1 && __uint_as_float(st->max_dw_bits) < delta;
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c < P) {
    double* a = acc + c * kIrlsAcc;
    if (!converged) {
      double ws = a[0] + 1e-6;
      double cnt = a[10] > 0.5 ? a[10] : 1.0;
      // centre = sum(w x) / (sum w + 1e-6) with x = o + origin; m = centre - origin
      double m[3];
      for (int q = 0; q < 3; ++q) m[q] = (a[1 + q] + (double)origin[c * 3 + q] * a[0]) / ws - (double)origin[c * 3 + q];
      // sum w d d^T with d = x - centre = o - m:  S_oo - m S_o^T - S_o m^T + W m m^T
      double So[3] = {a[1], a[2], a[3]};
      double Soo[6] = {a[4], a[5], a[6], a[7], a[8], a[9]};
      int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
      float cov[6];
      for (int q = 0; q < 6; ++q) {
        int i = ia[q], j = ib[q];
        double sdd = Soo[q] - m[i] * So[j] - So[i] * m[j] + a[0] * m[i] * m[j];
        cov[q] = (float)(sdd / cnt);
      }
      float nrm[3];
      eig3_smallest(cov, nrm);
      float* pl = plane + c * 6;
      pl[0] = (float)(m[0] + origin[c * 3]);
      pl[1] = (float)(m[1] + origin[c * 3 + 1]);
      pl[2] = (float)(m[2] + origin[c * 3 + 2]);
      pl[3] = nrm[0]; pl[4] = nrm[1]; pl[5] = nrm[2];
    }
    for (int q = 0; q < kIrlsAcc; ++q) a[q] = 0.0;  // ready for the next round
  }
  __threadfence();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (converged) st->stop_round = r;
    else st->rounds = r;
  }
}

__global__ void irls_reset_dw_kernel(int r, IrlsState* st) {
  if (r > st->stop_round) return;
  if (threadIdx.x == 0) st->max_dw_bits = 0;
}

// final point-plane error against the last plane (the reference's `plane_fitting_error`)
__global__ void __launch_bounds__(kBlock)
irls_error_kernel(const float* __restrict__ xyz, const int64_t* __restrict__ pidx, int64_t n,
                  const float* __restrict__ plane, float* __restrict__ perr) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float* pl = plane + pidx[i] * 6;
    float dx = xyz[i * 3] - pl[0], dy = xyz[i * 3 + 1] - pl[1], dz = xyz[i * 3 + 2] - pl[2];
    perr[i] = fabsf(dx * pl[3] + dy * pl[4] + dz * pl[5]);
  }
}

// ---- segmented form: the reference sorts the voxels by pillar before the fits
// (preprocessor_utils.py:112-118), so a pillar is one contiguous run of rows. One workgroup per
// pillar does a whole round (new weights from the previous plane, weighted raw moments in a fixed
// order -- no atomics, bit-reproducible --, 3x3 eigen-solve) in ONE launch. The global
// convergence test of round r (max weight change over ALL rows < delta) is evaluated by every
// workgroup of launch r+1 from one word per round, so nothing synchronises inside a launch:
// planes are double-buffered and the plane of round r-1 survives until the test of round r is known.
constexpr int kIrlsMaxRounds = 62;
struct IrlsSegState {
  int stop_round;   // launches with r > stop_round return immediately
  int rounds;       // rounds whose plane was adopted
  int final_buf;    // plane buffer holding the result
  int pad;
  unsigned dw[kIrlsMaxRounds + 2];  // max |new_w - w| per round (float bits)
};

__global__ void irls_seg_init_kernel(IrlsSegState* st) {
  st += blockIdx.x;
  int i = threadIdx.x;
  if (i == 0) { st->stop_round = 0x7fffffff; st->rounds = 0; st->final_buf = 0; st->pad = 0; }
  if (i < kIrlsMaxRounds + 2) st->dw[i] = 0;
}

// round r (1-based); r == max_rounds + 1 with check_only: only the convergence test of the last round
// gridDim.y independent fits share xyz / seg_start / origin (the 30 height ratios of
// compute_min_height_from_ransac, preprocessor_utils.py:120-135): fit y has its own weights
// w[y*n..], planes, state; with `ratios` != nullptr its initial weights are formed here from
// cur_z = min_z * ratio + max_z * (1 - ratio) instead of being read from w.
// TWO WAVES per (pillar, fit): a 2 m pillar holds ~200 rows of the 10 cm voxels, and the round ends with a
// serial 3 x 3 eigen-solve in thread 0 -- a 256-thread workgroup kept four waves resident through it (5000
// pillars x 30 ratios = 600 k waves per round, 0.28-0.40 ms per round, 50 rounds per sequence). Batches of 12:
// 256 threads 1494-1510 ms, 64 threads 1451-1473 (but the launch ends with its largest pillar on one wave:
// the stage alone 20.8 -> 27.8 ms), 128 threads 1445-1448 ms (the stage alone 22 ms).
constexpr int kIrlsWg = 128;
__global__ void __launch_bounds__(kIrlsWg)
irls_seg_round_kernel(int r, int check_only, const float* __restrict__ xyz,
                      const int32_t* __restrict__ seg_start, const float* __restrict__ origin,
                      float sigma2, float delta, const float* __restrict__ w0, float* __restrict__ planes,
                      int64_t P, IrlsSegState* __restrict__ st, int64_t n,
                      const float* __restrict__ ratios, const float* __restrict__ min_z,
                      const float* __restrict__ max_z) {
  st += blockIdx.y;
  planes += (size_t)blockIdx.y * P * 12;
  if (r > st->stop_round) return;
  float* plane_prev = planes + (size_t)((r - 1) & 1) * P * 6;  // plane of round r-1
  float* plane_cur = planes + (size_t)(r & 1) * P * 6;         // plane of round r (and of r-2)
  if (r > 2 && __uint_as_float(st->dw[r - 1]) < delta) {
    // round r-1's new weights moved by less than delta: the loop ended with the plane of round r-2
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->stop_round = r - 1; st->rounds = r - 2; st->final_buf = r & 1; }
    return;
  }
  if (check_only) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->rounds = r - 1; st->final_buf = (r - 1) & 1; }
    return;
  }
  const int c = blockIdx.x;
  const int b = seg_start[c], e = seg_start[c + 1];
  const float o0 = origin[c * 3], o1 = origin[c * 3 + 1], o2 = origin[c * 3 + 2];
  // The weights are never stored: the weight of round r is a function of the plane of round r-1 (or of
  // the initial rule), and the previous weight -- needed only for the convergence test -- of the
  // plane of round r-2, which is still in the other plane buffer. (Stored, they were R x n floats
  // read and written every round: 600 MB per round for the 30-ratio batch, the launch's whole time.)
  float pl[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, pl2[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (r > 1) {
#pragma unroll
    for (int q = 0; q < 6; ++q) pl[q] = plane_prev[(int64_t)c * 6 + q];
  }
  if (r > 2) {
#pragma unroll
    for (int q = 0; q < 6; ++q) pl2[q] = plane_cur[(int64_t)c * 6 + q];  // round r-2 (overwritten at the end)
  }
  double v[11];
#pragma unroll
  for (int q = 0; q < 11; ++q) v[q] = 0.0;
  float dw = 0.f;
  float cur_z = 0.f;
  const bool from_ratio = ratios != nullptr;
  if (from_ratio && r <= 2) {
    float ratio = ratios[blockIdx.y];
    float omr = (float)(1.0 - (double)ratio);  // Python forms 1 - ratio in double
    cur_z = min_z[c] * ratio + max_z[c] * omr;
  }
  auto plane_weight = [&](const float* q, float x, float y, float z) {
    float dx = x - q[0], dy = y - q[1], dz = z - q[2];
    float er = fabsf(dx * q[3] + dy * q[4] + dz * q[5]);
    float nw = sigma2 / (er * er + sigma2);
    float dist_w = 0.25f / (dx * dx + dy * dy + dz * dz + 0.25f);
    return nw * dist_w;
  };
  for (int i = b + (int)threadIdx.x; i < e; i += kIrlsWg) {
    float x = xyz[(int64_t)i * 3], y = xyz[(int64_t)i * 3 + 1], z = xyz[(int64_t)i * 3 + 2];
    float wi;
    if (r <= 2) {  // the initial weight (round 1) / the previous weight of round 2
      if (from_ratio) {
        float zd = cur_z - z;
        wi = (1.f / (zd * zd + sigma2)) * sigma2;  // scalar / tensor is reciprocal() * scalar in torch
      } else {
        wi = w0[i];
      }
    } else {
      wi = plane_weight(pl2, x, y, z);
    }
    if (r > 1) {
      float nw = plane_weight(pl, x, y, z);
      dw = fmaxf(dw, fabsf(nw - wi));
      wi = nw;
    }
    float ox = x - o0, oy = y - o1, oz = z - o2;
    double W = wi;
    v[0] += W; v[1] += W * ox; v[2] += W * oy; v[3] += W * oz;
    v[4] += W * ox * ox; v[5] += W * ox * oy; v[6] += W * ox * oz;
    v[7] += W * oy * oy; v[8] += W * oy * oz; v[9] += W * oz * oz;
    v[10] += 1.0;
  }
  double a[11];
  float m4;
  if constexpr (kIrlsWg == 64) {
#pragma unroll
    for (int q = 0; q < 11; ++q) a[q] = wave_sum_f64(v[q]);  // (fixed tree: lane l held rows l, l + 64, ...)
    m4 = wave_max(dw);
  } else {  // several waves per fit: their sums are added in wave order by thread 0
    __shared__ double part[kIrlsWg / 64][11];
    __shared__ float sdw[kIrlsWg / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      double x = wave_sum_f64(v[q]);
      if (lane == 0) part[wv][q] = x;
    }
    float mx = wave_max(dw);
    if (lane == 0) sdw[wv] = mx;
    __syncthreads();
    m4 = sdw[0];
    for (int k = 1; k < kIrlsWg / 64; ++k) m4 = fmaxf(m4, sdw[k]);
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      double x = part[0][q];
      for (int k = 1; k < kIrlsWg / 64; ++k) x += part[k][q];
      a[q] = x;
    }
  }
  if (threadIdx.x == 0) {
    if (r > 1) atomicMax(&st->dw[r], __float_as_uint(m4));
    double ws = a[0] + 1e-6;
    double cnt = a[10] > 0.5 ? a[10] : 1.0;
    double og[3] = {(double)o0, (double)o1, (double)o2};
    double m[3];
    for (int q = 0; q < 3; ++q) m[q] = (a[1 + q] + og[q] * a[0]) / ws - og[q];
    double So[3] = {a[1], a[2], a[3]};
    double Soo[6] = {a[4], a[5], a[6], a[7], a[8], a[9]};
    const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
    float cov[6];
    for (int q = 0; q < 6; ++q) {
      int i = ia[q], j = ib[q];
      double sdd = Soo[q] - m[i] * So[j] - So[i] * m[j] + a[0] * m[i] * m[j];
      cov[q] = (float)(sdd / cnt);
    }
    float nrm[3];
    eig3_smallest(cov, nrm);
    float* out = plane_cur + (int64_t)c * 6;
    out[0] = (float)(m[0] + og[0]);
    out[1] = (float)(m[1] + og[1]);
    out[2] = (float)(m[2] + og[2]);
    out[3] = nrm[0]; out[4] = nrm[1]; out[5] = nrm[2];
  }
}

// ---- the 30 height-ratio fits of compute_min_height_from_ransac, CHUNKED (round 5). The form above gives every
// (pillar, fit) one workgroup per round: an 8 m super-pillar next to the sensor holds 24 000 of the sequence's
// 650 000 rows (342 super-pillars, 97 of them 94 % of the rows), so a round's launch ended with a few two-wave
// workgroups walking 190 rows per thread, each followed by a serial 3 x 3 eigen-solve in one lane -- 190 us per
// round, 51 launches. Here a workgroup owns a CHUNK of kIrlsChunk rows of one pillar for ALL fits:
//   * prologue: lane f < R adds the chunks' partial moments of round r-1 in chunk order and solves fit f's plane
//     -- the 30 eigen-solves of a pillar side by side in 30 lanes (every chunk of the pillar repeats them: a few
//     microseconds, no second launch, no grid barrier); chunk 0 stores the plane;
//   * the chunk's rows are staged in LDS once and serve all fits: thread (g, f) adds rows g, g + 8, ... of fit f
//     (32 lanes read one row: a broadcast), the two row slices of a wave meet by one shuffle, the four waves in
//     LDS in wave order, and the chunk's 11 moments of every fit go to the partial buffer of round r.
// Sums have a fixed tree (oracle/pcs_oracle_ground.cpp restates it), the convergence word of a round is an
// order-free maximum, so results do not depend on what runs where. R <= 32 fits.
static int g_irls_per_fit = 0;  // pcs_irls_per_fit_rounds: the form above for the ratio fits too (cross-check)
constexpr int kIrlsChunk = 1024;
constexpr int kIrlsChunkWg = 256;
constexpr int kIrlsFits = 32;
__global__ void __launch_bounds__(kBlock)
irls_chunk_count_kernel(const int32_t* __restrict__ seg_start, int64_t P, int32_t* __restrict__ nchunk) {
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < P; c += (int64_t)gridDim.x * blockDim.x) {
    const int rows = seg_start[c + 1] - seg_start[c];
    nchunk[c] = rows > 0 ? (rows + kIrlsChunk - 1) / kIrlsChunk : 1;  // (an empty pillar still has a plane)
  }
}
// moments of one (pillar, fit) -> plane (centre, normal): the tail of irls_seg_round_kernel
__device__ __forceinline__ void irls_plane_from_moments(const double a[11], float o0, float o1, float o2, float out[6]) {
  double ws = a[0] + 1e-6;
  double cnt = a[10] > 0.5 ? a[10] : 1.0;
  double og[3] = {(double)o0, (double)o1, (double)o2};
  double m[3];
  for (int q = 0; q < 3; ++q) m[q] = (a[1 + q] + og[q] * a[0]) / ws - og[q];
  double So[3] = {a[1], a[2], a[3]};
  double Soo[6] = {a[4], a[5], a[6], a[7], a[8], a[9]};
  float cov[6];
  // (written out: with the index pairs in arrays the operands went through scratch memory)
#define PCS_IRLS_COV(q, i, j) cov[q] = (float)((Soo[q] - m[i] * So[j] - So[i] * m[j] + a[0] * m[i] * m[j]) / cnt)
  PCS_IRLS_COV(0, 0, 0); PCS_IRLS_COV(1, 0, 1); PCS_IRLS_COV(2, 0, 2);
  PCS_IRLS_COV(3, 1, 1); PCS_IRLS_COV(4, 1, 2); PCS_IRLS_COV(5, 2, 2);
#undef PCS_IRLS_COV
  float nrm[3];
  eig3_smallest(cov, nrm);
  out[0] = (float)(m[0] + og[0]);
  out[1] = (float)(m[1] + og[1]);
  out[2] = (float)(m[2] + og[2]);
  out[3] = nrm[0]; out[4] = nrm[1]; out[5] = nrm[2];
}
// round r (1-based) of all R fits; r == max_rounds + 1: the plane of the last round and its convergence test only
// (grid: the chunks; `last` = that final launch). planes: [R][2][P][6] as above; part: [2][chunks][kIrlsFits][12].
__global__ void __launch_bounds__(kIrlsChunkWg)
irls_chunk_round_kernel(int r, int last, const float* __restrict__ xyz, const int32_t* __restrict__ seg_start,
                        const int32_t* __restrict__ chunk_off, const float* __restrict__ origin, float sigma2,
                        float delta, float* __restrict__ planes, int64_t P, int R, IrlsSegState* __restrict__ st,
                        double* __restrict__ part, int64_t part_stride, const float* __restrict__ ratios,
                        const float* __restrict__ min_z, const float* __restrict__ max_z) {
  __shared__ float s_rows[kIrlsChunk * 3];
  __shared__ double s_part[kIrlsChunkWg / 64][kIrlsFits][11];
  __shared__ float s_plane[2][kIrlsFits][6];  // [0] round r-1, [1] round r-2
  __shared__ int s_active[kIrlsFits];
  __shared__ int s_c, s_k;
  const int t = threadIdx.x;
  const int j = blockIdx.x;
  if (j >= chunk_off[P]) return;
  if (t == 0) {
    int lo = 0, hi = (int)P;  // the pillar of chunk j: chunk_off[lo] <= j < chunk_off[lo + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (chunk_off[mid] <= j) lo = mid; else hi = mid;
    }
    s_c = lo;
    s_k = j - chunk_off[lo];
  }
  __syncthreads();
  const int c = s_c, k = s_k;
  const int nk = chunk_off[c + 1] - chunk_off[c];
  const float o0 = origin[c * 3], o1 = origin[c * 3 + 1], o2 = origin[c * 3 + 2];
  // ---- prologue: which fits still run, and their planes of the two rounds before
  if (t < kIrlsFits) {
    const int f = t;
    int active = 0;
    if (f < R) {
      IrlsSegState* sf = st + f;
      active = r <= sf->stop_round;
      if (active && r > 2 && __uint_as_float(sf->dw[r - 1]) < delta) {
        // round r-1's new weights moved by less than delta: the loop ended with the plane of round r-2
        active = 0;
        if (j == 0) { sf->stop_round = r - 1; sf->rounds = r - 2; sf->final_buf = r & 1; }
      }
      if (active) {
        float* pf = planes + (size_t)f * P * 12;
        if (r > 1) {
          // the plane of round r-1 from its chunks' moments, in chunk order
          const double* pp = part + (size_t)((r - 1) & 1) * part_stride + ((size_t)chunk_off[c] * kIrlsFits + f) * 12;
          double a[11];
#pragma unroll
          for (int q = 0; q < 11; ++q) a[q] = pp[q];
          for (int kk = 1; kk < nk; ++kk) {
            const double* pq = pp + (size_t)kk * kIrlsFits * 12;
#pragma unroll
            for (int q = 0; q < 11; ++q) a[q] += pq[q];
          }
          float pl[6];
          irls_plane_from_moments(a, o0, o1, o2, pl);
#pragma unroll
          for (int q = 0; q < 6; ++q) s_plane[0][f][q] = pl[q];
          if (k == 0) {
            float* out = pf + (size_t)((r - 1) & 1) * P * 6 + (int64_t)c * 6;
#pragma unroll
            for (int q = 0; q < 6; ++q) out[q] = pl[q];
          }
        }
        if (r > 2) {
          const float* p2 = pf + (size_t)(r & 1) * P * 6 + (int64_t)c * 6;  // round r-2 (stored by launch r-1)
#pragma unroll
          for (int q = 0; q < 6; ++q) s_plane[1][f][q] = p2[q];
        }
        if (last && j == 0) { sf->rounds = r - 1; sf->final_buf = (r - 1) & 1; }
      }
    }
    s_active[f] = active;
  }
  if (last) return;
  // ---- the chunk's rows, staged once for all fits
  const int b = seg_start[c] + k * kIrlsChunk;
  const int e = min(seg_start[c + 1], b + kIrlsChunk);
  const int nrow = max(e - b, 0);
  for (int i = t; i < nrow * 3; i += kIrlsChunkWg) s_rows[i] = xyz[(int64_t)b * 3 + i];
  __syncthreads();
  const int f = t & 31, g = t >> 5;
  const bool on = s_active[f] != 0;
  float pl[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, pl2[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (on && r > 1) {
#pragma unroll
    for (int q = 0; q < 6; ++q) pl[q] = s_plane[0][f][q];
  }
  if (on && r > 2) {
#pragma unroll
    for (int q = 0; q < 6; ++q) pl2[q] = s_plane[1][f][q];
  }
  float cur_z = 0.f;
  if (on && r <= 2) {
    const float ratio = ratios[f];
    const float omr = (float)(1.0 - (double)ratio);  // Python forms 1 - ratio in double
    cur_z = min_z[c] * ratio + max_z[c] * omr;
  }
  auto plane_weight = [&](const float* q, float x, float y, float z) {
    float dx = x - q[0], dy = y - q[1], dz = z - q[2];
    float er = fabsf(dx * q[3] + dy * q[4] + dz * q[5]);
    float nw = sigma2 / (er * er + sigma2);
    float dist_w = 0.25f / (dx * dx + dy * dy + dz * dz + 0.25f);
    return nw * dist_w;
  };
  double v[11];
#pragma unroll
  for (int q = 0; q < 11; ++q) v[q] = 0.0;
  float dw = 0.f;
  if (on) {
    for (int i = g; i < nrow; i += kIrlsChunkWg / 32) {
      const float x = s_rows[i * 3], y = s_rows[i * 3 + 1], z = s_rows[i * 3 + 2];
      float wi;
      if (r <= 2) {  // the initial weight (round 1) / the previous weight of round 2
        float zd = cur_z - z;
        wi = (1.f / (zd * zd + sigma2)) * sigma2;  // scalar / tensor is reciprocal() * scalar in torch
      } else {
        wi = plane_weight(pl2, x, y, z);
      }
      if (r > 1) {
        float nw = plane_weight(pl, x, y, z);
        dw = fmaxf(dw, fabsf(nw - wi));
        wi = nw;
      }
      float ox = x - o0, oy = y - o1, oz = z - o2;
      double W = wi;
      v[0] += W; v[1] += W * ox; v[2] += W * oy; v[3] += W * oz;
      v[4] += W * ox * ox; v[5] += W * ox * oy; v[6] += W * ox * oz;
      v[7] += W * oy * oy; v[8] += W * oy * oz; v[9] += W * oz * oz;
      v[10] += 1.0;
    }
  }
  // the two row slices of a wave (lanes f and f + 32), then the waves in order
  const int wv = t >> 6;
#pragma unroll
  for (int q = 0; q < 11; ++q) {
    const double o = __shfl_xor(v[q], 32, 64);
    const double lo_v = (t & 32) ? o : v[q], hi_v = (t & 32) ? v[q] : o;
    v[q] = lo_v + hi_v;
  }
  dw = fmaxf(dw, __shfl_xor(dw, 32, 64));
  if ((t & 63) < 32) {
#pragma unroll
    for (int q = 0; q < 11; ++q) s_part[wv][f][q] = v[q];
  }
  // (the maximum over the chunk's rows of fit f: the four waves' lanes f)
  __shared__ float s_dw[kIrlsChunkWg / 64][kIrlsFits];
  if ((t & 63) < 32) s_dw[wv][f] = dw;
  __syncthreads();
  if (t < kIrlsFits && t < R && s_active[t]) {
    double* out = part + (size_t)(r & 1) * part_stride + ((size_t)j * kIrlsFits + t) * 12;
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      double x = s_part[0][t][q];
      for (int w = 1; w < kIrlsChunkWg / 64; ++w) x += s_part[w][t][q];
      out[q] = x;
    }
    float m4 = s_dw[0][t];
    for (int w = 1; w < kIrlsChunkWg / 64; ++w) m4 = fmaxf(m4, s_dw[w][t]);
    if (r > 1) atomicMax(&st[t].dw[r], __float_as_uint(m4));
  }
}

// result plane -> caller's buffer, point-plane error of every row against it; with `num_hit`
// also the rows of the pillar whose error is below `hit_below` (the hypothesis' confidence)
__global__ void __launch_bounds__(kBlock)
irls_seg_finish_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ seg_start, int64_t P,
                       const float* __restrict__ planes, const IrlsSegState* __restrict__ st,
                       float* __restrict__ plane_out, float* __restrict__ perr, float hit_below,
                       float* __restrict__ num_hit) {
  st += blockIdx.y;
  planes += (size_t)blockIdx.y * P * 12;
  plane_out += (size_t)blockIdx.y * P * 6;
  const int c = blockIdx.x;
  const float* pl = planes + (size_t)st->final_buf * P * 6 + (int64_t)c * 6;
  float q[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) q[k] = pl[k];
  if (threadIdx.x < 6) plane_out[(int64_t)c * 6 + threadIdx.x] = q[threadIdx.x];
  const int b = seg_start[c], e = seg_start[c + 1];
  int hits = 0;
  for (int i = b + (int)threadIdx.x; i < e; i += kBlock) {
    float dx = xyz[(int64_t)i * 3] - q[0], dy = xyz[(int64_t)i * 3 + 1] - q[1], dz = xyz[(int64_t)i * 3 + 2] - q[2];
    float er = fabsf(dx * q[3] + dy * q[4] + dz * q[5]);
    if (perr) perr[i] = er;
    hits += er < hit_below;
  }
  if (num_hit) {
    __shared__ int sh[kBlock / 64];
    hits = wave_sum_i32(hits);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = hits;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int k = 0; k < kBlock / 64; ++k) t += sh[k];
      num_hit[(size_t)blockIdx.y * P + c] = (float)t;
    }
  }
}

// sequential choice over the fits in order: a later fit replaces the best only with strictly
// more hits (preprocessor_utils.py:131-135)
__global__ void __launch_bounds__(kBlock)
irls_select_kernel(int R, int64_t P, const float* __restrict__ planes_out,
                   const float* __restrict__ num_hit, float* __restrict__ best_center,
                   float* __restrict__ best_normal, float* __restrict__ best_conf) {
  int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= P) return;
  float conf = 0.f;
  float ctr[3] = {0.f, 0.f, 0.f}, nrm[3] = {0.f, 0.f, 1.f};
  for (int y = 0; y < R; ++y) {
    float h = num_hit[(size_t)y * P + c];
    if (conf < h) {
      conf = h;
      const float* pl = planes_out + ((size_t)y * P + c) * 6;
      ctr[0] = pl[0]; ctr[1] = pl[1]; ctr[2] = pl[2];
      nrm[0] = pl[3]; nrm[1] = pl[4]; nrm[2] = pl[5];
    }
  }
  best_conf[c] = conf;
  for (int k = 0; k < 3; ++k) { best_center[c * 3 + k] = ctr[k]; best_normal[c * 3 + k] = nrm[k]; }
}

// points_in_boxes_cpu, pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:121-168: serial
// boxes x points double loop on the host in the reference; one thread per (box, point) here.
// Same arithmetic: |z - cz| <= dz/2 in double, rotation by cos/sin of the double angle rounded to
// fp32, |local| < d/2 + 1e-2 compared in double.
__global__ void __launch_bounds__(kBlock)
points_in_boxes_kernel(const float* __restrict__ pts, int64_t N, const float* __restrict__ boxes,
                       int B, int32_t* __restrict__ out) {
  int64_t total = (int64_t)B * N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int b = (int)(i / N);
    int64_t j = i - (int64_t)b * N;
    const float* bx = boxes + b * 7;
    float x = pts[j * 3], y = pts[j * 3 + 1], z = pts[j * 3 + 2];
    float cx = bx[0], cy = bx[1], cz = bx[2], dx = bx[3], dy = bx[4], dz = bx[5], rz = bx[6];
    int flag = 0;
    if (!((double)fabsf(z - cz) > (double)dz / 2.0)) {
      const float MARGIN = 1e-2;
      float cosa = (float)cos((double)(-rz)), sina = (float)sin((double)(-rz));
      float sx = x - cx, sy = y - cy;
      float lx = sx * cosa + sy * (-sina);
      float ly = sx * sina + sy * cosa;
      flag = ((double)fabsf(lx) < (double)dx / 2.0 + (double)MARGIN) &
             ((double)fabsf(ly) < (double)dy / 2.0 + (double)MARGIN);
    }
    out[i] = flag;
  }
}

__device__ __forceinline__ int point_in_box(float x, float y, float z, const float* __restrict__ bx) {
  float cx = bx[0], cy = bx[1], cz = bx[2], dx = bx[3], dy = bx[4], dz = bx[5], rz = bx[6];
  if ((double)fabsf(z - cz) > (double)dz / 2.0) return 0;
  const float MARGIN = 1e-2;
  float cosa = (float)cos((double)(-rz)), sina = (float)sin((double)(-rz));
  float sx = x - cx, sy = y - cy;
  float lx = sx * cosa + sy * (-sina);
  float ly = sx * sina + sy * cosa;
  return ((double)fabsf(lx) < (double)dx / 2.0 + (double)MARGIN) &
         ((double)fabsf(ly) < (double)dy / 2.0 + (double)MARGIN);
}

// The per-frame points_in_boxes calls of the evaluation loops (cluster_proposal.py:200-215,
// cluster_tracking.py:388-400) for all frames at once: boxes are grouped by frame
// (box_offsets [F+1]); a point is tested against the boxes of its own frame only. member [N,Bmax]
// (column = box index within the frame, 0 beyond the frame's box count) and / or first [N]
// (lowest box index within the frame that contains the point, -1 = none).
__global__ void __launch_bounds__(kBlock)
points_in_frame_boxes_kernel(const float* __restrict__ pts, const int64_t* __restrict__ frame,
                             int64_t N, const float* __restrict__ boxes,
                             const int32_t* __restrict__ box_offsets, int F, int Bmax,
                             int32_t* __restrict__ member, int32_t* __restrict__ first) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < N;
       j += (int64_t)gridDim.x * blockDim.x) {
    int64_t f = frame[j];
    int b0 = 0, nb = 0;
    if (f >= 0 && f < F) { b0 = box_offsets[f]; nb = box_offsets[f + 1] - b0; }
    float x = pts[j * 3], y = pts[j * 3 + 1], z = pts[j * 3 + 2];
    int fst = -1;
    for (int bl = 0; bl < Bmax; ++bl) {
      int in = 0;
      if (bl < nb) in = point_in_box(x, y, z, boxes + (int64_t)(b0 + bl) * 7);
      if (member) member[j * Bmax + bl] = in;
      if (in && fst < 0) fst = bl;
    }
    if (first) first[j] = fst;
  }
}

// The same test, but instead of the [N, Bmax] membership matrix (the largest array of the extraction: 1.5 M points x
// 30 boxes x 4 bytes written, read again by an index_add) the launch adds every hit straight into counts[group[j],
// box] -- the [G, Bmax] table component_box_iou_frames wants (integer adds: order-free). The boxes' cos / sin are
// formed once per workgroup (the same double-precision expressions, rounded to fp32 as point_in_box rounds them)
// instead of once per (point, box).
constexpr int kBoxTrig = 2048;
__global__ void __launch_bounds__(kBlock)
frame_box_counts_kernel(const float* __restrict__ pts, const int64_t* __restrict__ frame, int64_t N,
                        const float* __restrict__ boxes, const int32_t* __restrict__ box_offsets, int F, int Bmax,
                        const int64_t* __restrict__ group, int32_t* __restrict__ counts) {
  __shared__ float s_cos[kBoxTrig], s_sin[kBoxTrig];
  const int nbox = box_offsets[F];
  const bool table = nbox <= kBoxTrig;
  if (table)
    for (int b = threadIdx.x; b < nbox; b += kBlock) {
      const float rz = boxes[(int64_t)b * 7 + 6];
      s_cos[b] = (float)cos((double)(-rz));
      s_sin[b] = (float)sin((double)(-rz));
    }
  __syncthreads();
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < N;
       j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = frame[j];
    int b0 = 0, nb = 0;
    if (f >= 0 && f < F) { b0 = box_offsets[f]; nb = box_offsets[f + 1] - b0; }
    const float x = pts[j * 3], y = pts[j * 3 + 1], z = pts[j * 3 + 2];
    const int64_t g = group[j];
    for (int bl = 0; bl < nb && bl < Bmax; ++bl) {
      const float* __restrict__ bx = boxes + (int64_t)(b0 + bl) * 7;
      int in;
      if (table) {
        const float cx = bx[0], cy = bx[1], cz = bx[2], dx = bx[3], dy = bx[4], dz = bx[5];
        if ((double)fabsf(z - cz) > (double)dz / 2.0) continue;
        const float MARGIN = 1e-2;
        const float cosa = s_cos[b0 + bl], sina = s_sin[b0 + bl];
        const float sx = x - cx, sy = y - cy;
        const float lx = sx * cosa + sy * (-sina);
        const float ly = sx * sina + sy * cosa;
        in = ((double)fabsf(lx) < (double)dx / 2.0 + (double)MARGIN) & ((double)fabsf(ly) < (double)dy / 2.0 + (double)MARGIN);
      } else {
        in = point_in_box(x, y, z, bx);
      }
      if (in) atomicAdd(&counts[g * Bmax + bl], 1);
    }
  }
}

// workspace of one problem
struct SvWs {
  SmoothState* st;
  SmoothWin* win;
  double* partial;
  float* buf;
};
static inline int64_t sv2_max_blocks(int64_t C, int F) {
  const int per = SV2_BLOCK * SV2_ELEMS / (2 * (F < 2 ? 2 : F));  // components per workgroup at the longest range
  if (per < 1) return 1;
  int64_t b = (C + per - 1) / per;
  return b > SV2_MAX_BLOCKS ? 1 : b;
}
static bool sv_plan(Arena& a, int64_t C, int F, SvWs& w) {
  w.st = a.take<SmoothState>(1);
  w.win = a.take<SmoothWin>(1);
  w.partial = a.take<double>((size_t)SV2_WINDOW * sv2_max_blocks(C, F));
  w.buf = a.take<float>((size_t)C * F * 2 * 6);  // two checkpoints of (iterate, m, sq), or the generic form's moments
  return a.ok();
}

// The same for several problems at once (one per tracking unit; every problem takes the kernel form
// and geometry pcs_smooth_velo would give it, so the results are those of separate calls).
// comp_velos[e] [C,F,3] f32 updated in place; comp_center_diffs[e] [C,F,3]. frame_lo <= frame_hi are the
// ordered ends of the optimised range (the reference swaps them itself when tracking backwards).
// ws[e] / ws_bytes[e]: each problem's own workspace (pcs_smooth_velo_workspace_bytes).
// Nothing here waits for the device or for another host thread.
int smooth_velo_run_batch(int n, float* const* comp_velos, const float* const* comp_center_diffs,
                          const int64_t* C, const int* F, const int* frame_lo, const int* frame_hi,
                          float weight0, float weight, int num_itr, float stopping, void* const* ws,
                          const size_t* ws_bytes, hipStream_t s) {
  if (n < 1) return PCS_OK;
  if (num_itr < 0) return PCS_ERR_ARG;
  for (int e0 = 0; e0 < n; e0 += kSvMaxBatch) {
    const int e1 = e0 + kSvMaxBatch < n ? e0 + kSvMaxBatch : n;
    SvBatch sb;
    sb.n = 0;
    size_t lds0 = 0;
    int have[3] = {0, 0, 0}, blocks1 = 1;
    int64_t tot_max = 1;
    for (int e = e0; e < e1; ++e) {
      if (!comp_velos[e] || !comp_center_diffs[e] || !ws[e] || C[e] < 1 || F[e] < 1 || frame_lo[e] < 0 ||
          frame_hi[e] >= F[e] || frame_lo[e] > frame_hi[e])
        return PCS_ERR_ARG;
      if (frame_lo[e] == frame_hi[e] || num_itr == 0) continue;  // cluster_tracking.py:166-167
      if (C[e] * (int64_t)F[e] * 3 >= (1LL << 31)) return PCS_ERR_RANGE;
      Arena a(ws[e], ws_bytes[e]);
      SvWs w;
      if (!sv_plan(a, C[e], F[e], w)) return PCS_ERR_WORKSPACE;
      SvJob& j = sb.j[sb.n];
      j.st = w.st;
      j.win = w.win;
      j.partial = w.partial;
      j.buf = w.buf;
      j.v = comp_velos[e];
      j.d = comp_center_diffs[e];
      j.C = (int)C[e];
      j.F = F[e];
      j.a = frame_lo[e];
      j.b = frame_hi[e];
      const int nf = frame_hi[e] - frame_lo[e] + 1;
      const int64_t n_in = C[e] * (int64_t)nf * 2;
      j.cpb = (SV2_BLOCK * SV2_ELEMS) / (nf * 2);
      const int64_t blocks = j.cpb > 0 ? (C[e] + j.cpb - 1) / j.cpb : (int64_t)SV2_MAX_BLOCKS + 1;
      j.blocks = 1;
      if (n_in <= (int64_t)SV_BLOCK * SV_ELEMS && num_itr <= SV_MAX_ITR) {
        j.kind = 0;
        if ((size_t)n_in * sizeof(float) > lds0) lds0 = (size_t)n_in * sizeof(float);
      } else if (blocks <= SV2_MAX_BLOCKS && blocks <= sv2_max_blocks(C[e], F[e]) && num_itr <= SV_MAX_ITR) {
        j.kind = 1;
        j.blocks = (int)blocks;
        if (j.blocks > blocks1) blocks1 = j.blocks;
      } else {
        j.kind = 2;
      }
      have[j.kind] = 1;
      const int64_t tot = C[e] * (int64_t)F[e] * 3;
      if (tot > tot_max) tot_max = tot;
      ++sb.n;
    }
    if (sb.n == 0) continue;
    smooth_velo_batch_reset_kernel<<<sb.n, 64, 0, s>>>(sb);
    if (have[0])
      smooth_velo_batch_kernel<<<dim3(1, sb.n), SV_BLOCK, lds0, s>>>(sb, 0, weight0, weight, num_itr, stopping);
    if (have[1]) {
      const int windows = (num_itr + SV2_WINDOW - 1) / SV2_WINDOW;
      for (int w = 0; w < windows; ++w)
        smooth_velo_window_kernel<<<dim3(blocks1, sb.n), SV2_BLOCK, 0, s>>>(sb, w, weight0, weight, num_itr,
                                                                           stopping);
      smooth_velo_replay_kernel<<<dim3(blocks1, sb.n), SV2_BLOCK, 0, s>>>(sb, weight0, weight, num_itr, stopping);
    }
    if (have[2])
      smooth_velo_batch_kernel<<<dim3(1, sb.n), SV_BLOCK, 0, s>>>(sb, 2, weight0, weight, num_itr, stopping);
    smooth_velo_batch_decay_kernel<<<dim3(grid_for(tot_max), sb.n), kBlock, 0, s>>>(sb);
  }
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}


// ---- hypothesis pruning of the ground stage ("truncated least squares", preprocessor_utils.py:137-160
// of the reference: for each of 100 shrinking thresholds, drop the plane hypotheses whose mean 8-NN
// curvature is at or above it and recompute). The whole loop in ONE workgroup: the hypotheses (a
// few hundred to a couple of thousand super-pillar planes) live in LDS, every thread owns up to two
// of them, the k nearest are kept in registers by a fixed insertion network (ties by index), the
// threshold loop and its "skip while the threshold is above the largest curvature" test run on the
// device. The reference (and the host form this replaces) launches a k-NN graph and reads a maximum
// back to the host for each of the ~20-30 pruning events.
constexpr int kPruneMax = 2048;
constexpr int kPruneWg = 1024;
constexpr int kPruneK = 8;
struct PruneThresholds {
  float v[128];
};

// Round 5: the k-NN set of a hypothesis changes only when one of its members is dropped (dropping others leaves its
// k nearest what they were), so after a pruning event only the DIRTY hypotheses -- a handful -- are evaluated again,
// and an evaluation is a WAVE's job, not a thread's: every lane scans its share of the hypotheses with the same
// insertion network (f64 squared distances, ties by index), the 64 sorted lists are merged by k rounds of a wave
// arg-min, and the curvature is added up in that ascending order. The same neighbours and the same sums as the
// one-thread form (and as oracle/pcs_oracle_ground.cpp): 4.9 -> 0.3 ms per sequence.
__global__ void __launch_bounds__(kPruneWg)
prune_hypotheses_kernel(const float* __restrict__ xyz, const float* __restrict__ normal, int H, int K,
                        PruneThresholds thr, int T, uint8_t* __restrict__ keep) {
  __shared__ float sx[kPruneMax], sy[kPruneMax], sz[kPruneMax];
  __shared__ float snx[kPruneMax], sny[kPruneMax], snz[kPruneMax];
  __shared__ float curv[kPruneMax];
  __shared__ short nn[kPruneMax * kPruneK];  // the k nearest alive hypotheses of every hypothesis, ascending
  __shared__ short list[kPruneMax];          // hypotheses to evaluate
  __shared__ uint8_t alive[kPruneMax];
  __shared__ float red[kPruneWg / 64];
  __shared__ int redn[kPruneWg / 64];
  __shared__ float s_max;
  __shared__ int s_alive, s_nlist;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < H; i += kPruneWg) {
    sx[i] = xyz[3 * i]; sy[i] = xyz[3 * i + 1]; sz[i] = xyz[3 * i + 2];
    snx[i] = normal[3 * i]; sny[i] = normal[3 * i + 1]; snz[i] = normal[3 * i + 2];
    alive[i] = 1;
    curv[i] = 0.f;
    list[i] = (short)i;
  }
  if (tid == 0) { s_nlist = H; s_alive = H; }
  __syncthreads();
  // mean curvature of the listed hypotheses over their k nearest alive hypotheses (itself included, like
  // torch_cluster.knn(x, x, k)); then the maximum over the alive ones and their count (s_max / s_alive)
  auto evaluate = [&]() {
    const int nl = s_nlist, n_alive = s_alive;
    const int k = n_alive < K ? n_alive : K;
    for (int li = w; li < nl; li += kPruneWg / 64) {
      const int i = list[li];
      double bd[kPruneK];
      int bi[kPruneK];
#pragma unroll
      for (int q = 0; q < kPruneK; ++q) { bd[q] = 1e300; bi[q] = 0x7fffffff; }
      const double xi = sx[i], yi = sy[i], zi = sz[i];
      for (int j = lane; j < H; j += 64) {
        if (!alive[j]) continue;
        const double dx = (double)sx[j] - xi, dy = (double)sy[j] - yi, dz = (double)sz[j] - zi;
        double d = dx * dx + dy * dy + dz * dz;
        int id = j;
        if (d > bd[kPruneK - 1]) continue;
#pragma unroll
        for (int q = 0; q < kPruneK; ++q) {  // insert (d, id) into the lane's ascending list: fixed network
          const bool lt = d < bd[q] || (d == bd[q] && id < bi[q]);
          const double td = lt ? bd[q] : d;
          const int ti = lt ? bi[q] : id;
          bd[q] = lt ? d : bd[q];
          bi[q] = lt ? id : bi[q];
          d = td;
          id = ti;
        }
      }
      // merge: k rounds of (smallest head over the lanes, by (d, id)); the winning lane drops its head
      float sum = 0.f;
      for (int q = 0; q < k; ++q) {
        double md = bd[0];
        int mi = bi[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const double od = __shfl_xor(md, o, 64);
          const int oi = __shfl_xor(mi, o, 64);
          const bool take = od < md || (od == md && oi < mi);
          md = take ? od : md;
          mi = take ? oi : mi;
        }
        if (bi[0] == mi) {
#pragma unroll
          for (int u = 0; u + 1 < kPruneK; ++u) { bd[u] = bd[u + 1]; bi[u] = bi[u + 1]; }
          bd[kPruneK - 1] = 1e300;
          bi[kPruneK - 1] = 0x7fffffff;
        }
        const int j = mi;  // (< H: at least k hypotheses are alive)
        if (lane == 0) nn[i * kPruneK + q] = (short)j;
        const float dx = sx[j] - sx[i], dy = sy[j] - sy[i], dz = sz[j] - sz[i];
        const float p2p = fabsf((dx * snx[i] + dy * sny[i]) + dz * snz[i]);
        sum += p2p / (sqrtf((dx * dx + dy * dy) + dz * dz) + 1e-4f);
      }
      if (lane == 0) {
        curv[i] = sum / (float)k;
        for (int q = k; q < kPruneK; ++q) nn[i * kPruneK + q] = -1;
      }
    }
    __syncthreads();
    float mx = -__builtin_inff();
    int cnt = 0;
    for (int i = tid; i < H; i += kPruneWg)
      if (alive[i]) { mx = fmaxf(mx, curv[i]); ++cnt; }
    mx = wave_max(mx);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) { red[w] = mx; redn[w] = cnt; }
    __syncthreads();
    if (tid == 0) {
      float m = red[0];
      int c = redn[0];
      for (int q = 1; q < kPruneWg / 64; ++q) { m = fmaxf(m, red[q]); c += redn[q]; }
      s_max = m;
      s_alive = c;
      s_nlist = 0;
    }
    __syncthreads();
  };
  evaluate();
  for (int t = 0; t < T; ++t) {
    const float th = thr.v[t];
    if (th > s_max) continue;  // (uniform: s_max is only written inside evaluate(), between barriers)
    const int alive_before = s_alive;
    int dropped = 0;
    for (int i = tid; i < H; i += kPruneWg)
      if (alive[i] && !(curv[i] < th)) { alive[i] = 0; ++dropped; }
    for (int off = 32; off > 0; off >>= 1) dropped += __shfl_xor(dropped, off, 64);
    if (lane == 0) redn[w] = dropped;
    __syncthreads();
    if (tid == 0) {
      int c = 0;
      for (int q = 0; q < kPruneWg / 64; ++q) c += redn[q];
      s_alive = alive_before - c;  // (evaluate() divides by min(alive, K))
    }
    __syncthreads();
    // dirty: a member of the k nearest was dropped -- or fewer than K are left (the mean's divisor changes)
    const bool all = s_alive < K;
    for (int i = tid; i < H; i += kPruneWg) {
      if (!alive[i]) continue;
      bool dirty = all;
#pragma unroll
      for (int q = 0; q < kPruneK; ++q) {
        const int j = nn[i * kPruneK + q];
        dirty = dirty || (j >= 0 && !alive[j]);
      }
      if (dirty) list[atomicAdd(&s_nlist, 1)] = (short)i;
    }
    __syncthreads();
    evaluate();
  }
  for (int i = tid; i < H; i += kPruneWg) keep[i] = alive[i];
}

}  // namespace pcs

using namespace pcs;

extern "C" {

size_t pcs_smooth_velo_workspace_bytes(int64_t C, int F) {
  if (C < 1 || F < 1) return 0;
  Arena a(nullptr, 0);
  SvWs w;
  sv_plan(a, C, F, w);
  return a.off;
}

// comp_velos [C,F,3] f32 updated in place; comp_center_diffs [C,F,3]. frame_lo <= frame_hi are the
// ordered ends of the optimised range (the reference swaps them itself when tracking backwards).
int pcs_smooth_velo(float* comp_velos, const float* comp_center_diffs, int64_t C, int F,
                    int frame_lo, int frame_hi, float weight0, float weight, int num_itr,
                    float stopping, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!comp_velos || !comp_center_diffs || !ws) return PCS_ERR_ARG;
  return pcs::smooth_velo_run_batch(1, &comp_velos, &comp_center_diffs, &C, &F, &frame_lo, &frame_hi, weight0,
                                    weight, num_itr, stopping, &ws, &ws_bytes, s);
}

// several problems in one call (what the sequence tracker does at every step, one problem per unit)
int pcs_smooth_velo_batch(int n, float* const* comp_velos, const float* const* comp_center_diffs,
                          const int64_t* C, const int* F, const int* frame_lo, const int* frame_hi,
                          float weight0, float weight, int num_itr, float stopping, void* const* ws,
                          const size_t* ws_bytes, hipStream_t s) {
  if (n < 0 || (n > 0 && (!comp_velos || !comp_center_diffs || !C || !F || !frame_lo || !frame_hi || !ws ||
                          !ws_bytes)))
    return PCS_ERR_ARG;
  int rc = pcs::smooth_velo_run_batch(n, comp_velos, comp_center_diffs, C, F, frame_lo, frame_hi, weight0,
                                      weight, num_itr, stopping, ws, ws_bytes, s);
  if (rc != PCS_OK) return rc;
  PCS_CHECK_HIP(hipStreamSynchronize(s));
  return PCS_OK;
}

// out [B, N] int32 (1 = point inside box)
int pcs_points_in_boxes(const float* points, int64_t N, const float* boxes, int B, int32_t* out,
                        hipStream_t s) {
  if (N < 0 || B < 0 || ((N > 0 && B > 0) && (!points || !boxes || !out))) return PCS_ERR_ARG;
  if (N == 0 || B == 0) return PCS_OK;
  points_in_boxes_kernel<<<grid_for((int64_t)B * N), kBlock, 0, s>>>(points, N, boxes, B, out);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

int pcs_points_in_frame_boxes(const float* points, const int64_t* point_frame, int64_t N,
                              const float* boxes, const int32_t* box_offsets, int F, int Bmax,
                              int32_t* member, int32_t* first, hipStream_t s) {
  if (N < 0 || F < 0 || Bmax < 0 || (N > 0 && (!points || !point_frame || !box_offsets)) ||
      (!member && !first))
    return PCS_ERR_ARG;
  if (N == 0) return PCS_OK;
  if (Bmax > 0 && !boxes) return PCS_ERR_ARG;
  points_in_frame_boxes_kernel<<<grid_for(N), kBlock, 0, s>>>(points, point_frame, N, boxes, box_offsets,
                                                              F, Bmax, member, first);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// counts [G, Bmax] int32 += the hits of pcs_points_in_frame_boxes' test, point j under row group[j] (group [N] int64,
// values in [0, G)); the caller zeroes counts
int pcs_frame_box_counts(const float* points, const int64_t* point_frame, int64_t N, const float* boxes,
                         const int32_t* box_offsets, int F, int Bmax, const int64_t* group, int64_t G,
                         int32_t* counts, hipStream_t s) {
  if (N < 0 || F < 0 || Bmax < 0 || G < 0 || (N > 0 && (!points || !point_frame || !box_offsets || !group || !counts)))
    return PCS_ERR_ARG;
  if (N == 0 || Bmax == 0) return PCS_OK;
  if (!boxes) return PCS_ERR_ARG;
  frame_box_counts_kernel<<<grid_for(N), kBlock, 0, s>>>(points, point_frame, N, boxes, box_offsets, F, Bmax, group,
                                                         counts);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

size_t pcs_l1_minimization_workspace_bytes(int X, int Y) {
  Arena a(nullptr, 0);
  a.take<SmoothState>(1);
  a.take<float>((size_t)X * Y);
  a.take<float>((size_t)X * Y);
  a.take<int>(16);
  return a.off;
}

// pillar_min_z, weight, height_out: [X,Y] f32. decay_steps: HOST int array (MultiStepLR milestones).
int pcs_l1_minimization(const float* pillar_min_z, const float* weight, int X, int Y, double lr,
                        const int* decay_steps, int n_decay, float rigid_weight, int max_iters,
                        float* height_out, int32_t* d_iters, void* ws, size_t ws_bytes,
                        hipStream_t s) {
  if (!pillar_min_z || !weight || !height_out || !ws || X < 3 || Y < 3 || n_decay < 0 ||
      n_decay > 16 || (n_decay > 0 && !decay_steps) || max_iters < 0)
    return PCS_ERR_ARG;
  if ((int64_t)X * Y > L1_MAX_CELLS) return PCS_ERR_RANGE;
  Arena a(ws, ws_bytes);
  SmoothState* st = a.take<SmoothState>(1);
  float* m = a.take<float>((size_t)X * Y);
  float* v = a.take<float>((size_t)X * Y);
  int* dsteps = a.take<int>(16);
  if (!a.ok()) return PCS_ERR_WORKSPACE;
  if (n_decay > 0)
    PCS_CHECK_HIP(hipMemcpyAsync(dsteps, decay_steps, n_decay * sizeof(int), hipMemcpyHostToDevice, s));
  l1_min_kernel<<<1, L1_BLOCK, 0, s>>>(pillar_min_z, weight, X, Y, lr, dsteps, n_decay,
                                       rigid_weight, max_iters, height_out, m, v, st);
  if (d_iters)
    PCS_CHECK_HIP(hipMemcpyAsync(d_iters, &st->iters, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

size_t pcs_irls_plane_fit_sorted_workspace_bytes(int64_t n, int64_t P) {
  Arena a(nullptr, 0);
  a.take<IrlsSegState>(1);
  a.take<float>((size_t)(P > 0 ? P : 1) * 12);
  (void)n;
  return a.off;
}

// Rows grouped by pillar: seg_start [P+1] int32 row ranges (empty pillars allowed). Same outputs as
// pcs_irls_plane_fit. One launch per round, bit-reproducible (fixed-order sums). Never synchronises.
int pcs_irls_plane_fit_sorted(const float* xyz, const int32_t* seg_start, int64_t n, const float* w0,
                              const float* origin, int64_t P, float sigma2, float stopping_delta,
                              int max_rounds, float* plane, float* plane_err, int32_t* d_rounds,
                              void* ws, size_t ws_bytes, hipStream_t s) {
  if (!xyz || !seg_start || !w0 || !origin || !plane || !plane_err || !ws || n < 1 || P < 1 ||
      max_rounds < 1 || max_rounds > kIrlsMaxRounds)
    return PCS_ERR_ARG;
  if (P >= (1LL << 24) || n >= (1LL << 31)) return PCS_ERR_RANGE;
  Arena a(ws, ws_bytes);
  IrlsSegState* st = a.take<IrlsSegState>(1);
  float* planes = a.take<float>((size_t)P * 12);
  if (!a.ok()) return PCS_ERR_WORKSPACE;
  irls_seg_init_kernel<<<1, 64, 0, s>>>(st);
  for (int r = 1; r <= max_rounds; ++r)
    irls_seg_round_kernel<<<(int)P, kIrlsWg, 0, s>>>(r, 0, xyz, seg_start, origin, sigma2, stopping_delta,
                                                    w0, planes, P, st, n, nullptr, nullptr, nullptr);
  irls_seg_round_kernel<<<1, 64, 0, s>>>(max_rounds + 1, 1, xyz, seg_start, origin, sigma2,
                                         stopping_delta, w0, planes, P, st, n, nullptr, nullptr, nullptr);
  irls_seg_finish_kernel<<<(int)P, kBlock, 0, s>>>(xyz, seg_start, P, planes, st, plane, plane_err, 0.f,
                                                   nullptr);
  if (d_rounds)
    PCS_CHECK_HIP(hipMemcpyAsync(d_rounds, &st->rounds, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// most chunks a call can have: every pillar one, and one more per kIrlsChunk rows
static inline int64_t irls_max_chunks(int64_t n, int64_t P) { return (n > 0 ? n : 0) / kIrlsChunk + (P > 0 ? P : 1); }
size_t pcs_ransac_plane_hypotheses_workspace_bytes(int64_t n, int64_t P, int R) {
  Arena a(nullptr, 0);
  if (R < 1) R = 1;
  if (P < 1) P = 1;
  a.take<IrlsSegState>(R);
  a.take<float>((size_t)R * P * 12);
  a.take<float>((size_t)R * P * 6);
  a.take<float>((size_t)R * P);
  a.take<float>(R);
  a.take<int32_t>(P + 1);
  a.take<int32_t>(P + 1);
  a.take<char>(scan_i32_temp_bytes(P + 1));
  a.take<double>((size_t)2 * irls_max_chunks(n, P) * kIrlsFits * 12);
  return a.off;
}

// The hypothesis loop of compute_min_height_from_ransac (preprocessor_utils.py:120-135): for every
// height ratio (HOST array, R <= 64) the IRLS plane fit of every pillar from the initial weights
// sigma2 / ((min_z*ratio + max_z*(1-ratio) - z)^2 + sigma2), its hit count (rows with point-plane
// error < hit_below), and per pillar the first fit with the most hits. The R fits are independent
// and run side by side: one launch per round for all of them (grid = pillars x ratios).
int pcs_ransac_plane_hypotheses(const float* xyz, const int32_t* seg_start, int64_t n,
                                const float* min_z, const float* max_z, const float* ratios, int R,
                                const float* origin, int64_t P, float sigma2, float stopping_delta,
                                int max_rounds, float hit_below, float* best_center,
                                float* best_normal, float* best_conf, void* ws, size_t ws_bytes,
                                hipStream_t s) {
  if (!xyz || !seg_start || !min_z || !max_z || !ratios || !origin || !best_center || !best_normal ||
      !best_conf || !ws || n < 1 || P < 1 || R < 1 || R > 64 || max_rounds < 1 ||
      max_rounds > kIrlsMaxRounds)
    return PCS_ERR_ARG;
  if (P >= 65536 * 16 || n >= (1LL << 31)) return PCS_ERR_RANGE;
  Arena a(ws, ws_bytes);
  IrlsSegState* st = a.take<IrlsSegState>(R);
  float* planes = a.take<float>((size_t)R * P * 12);
  const float* w = nullptr;  // the weights of the ratio fits are never stored
  float* planes_out = a.take<float>((size_t)R * P * 6);
  float* num_hit = a.take<float>((size_t)R * P);
  float* d_ratios = a.take<float>(R);
  if (!a.ok()) return PCS_ERR_WORKSPACE;
  PCS_CHECK_HIP(hipMemcpyAsync(d_ratios, ratios, (size_t)R * sizeof(float), hipMemcpyHostToDevice, s));
  int32_t* nchunk = a.take<int32_t>(P + 1);
  int32_t* chunk_off = a.take<int32_t>(P + 1);
  const size_t scan_bytes = scan_i32_temp_bytes(P + 1);
  void* scan_tmp = a.take<char>(scan_bytes);
  const int64_t max_chunks = irls_max_chunks(n, P);
  double* part = a.take<double>((size_t)2 * max_chunks * kIrlsFits * 12);
  if (!a.ok()) return PCS_ERR_WORKSPACE;
  irls_seg_init_kernel<<<R, 64, 0, s>>>(st);
  dim3 grid((unsigned)P, (unsigned)R);
  if (R <= kIrlsFits && !g_irls_per_fit) {
    // a workgroup per chunk of rows, all fits (irls_chunk_round_kernel)
    PCS_CHECK_HIP(hipMemsetAsync(nchunk + P, 0, sizeof(int32_t), s));
    irls_chunk_count_kernel<<<grid_for(P), kBlock, 0, s>>>(seg_start, P, nchunk);
    int rc = exclusive_scan_i32(scan_tmp, scan_bytes, nchunk, chunk_off, (size_t)P + 1, s);
    if (rc) return rc;
    const size_t part_stride = (size_t)max_chunks * kIrlsFits * 12;
    for (int r = 1; r <= max_rounds + 1; ++r)
      irls_chunk_round_kernel<<<(int)max_chunks, kIrlsChunkWg, 0, s>>>(
          r, r == max_rounds + 1, xyz, seg_start, chunk_off, origin, sigma2, stopping_delta, planes, P, R, st,
          part, (int64_t)part_stride, d_ratios, min_z, max_z);
  } else {
    for (int r = 1; r <= max_rounds; ++r)
      irls_seg_round_kernel<<<grid, kIrlsWg, 0, s>>>(r, 0, xyz, seg_start, origin, sigma2, stopping_delta, w,
                                                    planes, P, st, n, d_ratios, min_z, max_z);
    irls_seg_round_kernel<<<dim3(1, (unsigned)R), 64, 0, s>>>(max_rounds + 1, 1, xyz, seg_start, origin,
                                                              sigma2, stopping_delta, w, planes, P, st, n,
                                                              d_ratios, min_z, max_z);
  }
  irls_seg_finish_kernel<<<grid, kBlock, 0, s>>>(xyz, seg_start, P, planes, st, planes_out, nullptr,
                                                 hit_below, num_hit);
  irls_select_kernel<<<grid_for(P), kBlock, 0, s>>>(R, P, planes_out, num_hit, best_center, best_normal,
                                                    best_conf);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// test hook: 1 = the ratio fits on one workgroup per (pillar, fit) and round (irls_seg_round_kernel), the form before
// the chunked rounds: another summation tree, results equal to a few last bits
int pcs_irls_per_fit_rounds(int on) {
  g_irls_per_fit = on ? 1 : 0;
  return PCS_OK;
}

size_t pcs_irls_plane_fit_workspace_bytes(int64_t n, int64_t P) {
  Arena a(nullptr, 0);
  a.take<IrlsState>(1);
  a.take<double>((size_t)(P > 0 ? P : 1) * kIrlsAcc);
  a.take<float>((size_t)(n > 0 ? n : 1));
  return a.off;
}

// xyz [n,3] f32, pillar_idx [n] int64 in [0,P), w0 [n] f32 initial weights, origin [P,3] f32 (any
// point near each pillar; only conditions the raw moments). Outputs: plane [P,6] (centre, normal),
// plane_err [n] f32 (error of every point against the returned plane), *d_rounds (optional device
// int32) = rounds whose plane was adopted. Never synchronises.
int pcs_irls_plane_fit(const float* xyz, const int64_t* pillar_idx, int64_t n, const float* w0,
                       const float* origin, int64_t P, float sigma2, float stopping_delta,
                       int max_rounds, float* plane, float* plane_err, int32_t* d_rounds, void* ws,
                       size_t ws_bytes, hipStream_t s) {
  if (!xyz || !pillar_idx || !w0 || !origin || !plane || !plane_err || !ws || n < 1 || P < 1 ||
      max_rounds < 1)
    return PCS_ERR_ARG;
  Arena a(ws, ws_bytes);
  IrlsState* st = a.take<IrlsState>(1);
  double* acc = a.take<double>((size_t)P * kIrlsAcc);
  float* w = a.take<float>((size_t)n);
  if (!a.ok()) return PCS_ERR_WORKSPACE;
  PCS_CHECK_HIP(hipMemcpyAsync(w, w0, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
  irls_init_kernel<<<grid_for(P * kIrlsAcc), kBlock, 0, s>>>(st, acc, P);
  int nb = (int)((n + kBlock - 1) / kBlock);
  int nbp = (int)((P + kBlock - 1) / kBlock);
  // the reference runs `for itr in range(50)`: round r fits plane r from weights r; the weights of
  // round r+1 come from plane r; it breaks (keeping plane r) when they moved by less than delta
  for (int r = 1; r <= max_rounds; ++r) {
    irls_reset_dw_kernel<<<1, 64, 0, s>>>(r, st);
    irls_accumulate_kernel<<<nb, kBlock, 0, s>>>(r, xyz, pillar_idx, n, origin, sigma2,
                                                 stopping_delta, w, plane, plane_err, acc, st);
    irls_solve_kernel<<<nbp, kBlock, 0, s>>>(r, P, origin, stopping_delta, acc, plane, st);
  }
  irls_error_kernel<<<grid_for(n), kBlock, 0, s>>>(xyz, pillar_idx, n, plane, plane_err);
  if (d_rounds)
    PCS_CHECK_HIP(hipMemcpyAsync(d_rounds, &st->rounds, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}


/* prune_plane_hypotheses: see include/pcs.h */
int pcs_prune_plane_hypotheses(const float* xyz, const float* normal, int64_t H, int K,
                               const float* thresholds, int T, uint8_t* keep, hipStream_t s) {
  if (!xyz || !normal || !thresholds || !keep || H < 1 || K < 1 || K > kPruneK || T < 0 || T > 128)
    return PCS_ERR_ARG;
  if (H > kPruneMax) return PCS_ERR_RANGE;  // the caller keeps its host-driven form for that
  PruneThresholds th;
  for (int t = 0; t < T; ++t) th.v[t] = thresholds[t];
  prune_hypotheses_kernel<<<1, kPruneWg, 0, s>>>(xyz, normal, (int)H, K, th, T, keep);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

}  // extern "C"
