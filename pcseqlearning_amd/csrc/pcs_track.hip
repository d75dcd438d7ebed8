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
__global__ void __launch_bounds__(SV_BLOCK)
smooth_velo_kernel(float* __restrict__ v, const float* __restrict__ d, int C, int F, int a, int b,
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

// elements outside (c, [a,b], {0,1}) only see the decoupled weight decay of every executed step
__global__ void __launch_bounds__(kBlock)
smooth_velo_decay_kernel(float* __restrict__ v, int64_t total, int F, int a, int b,
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

__global__ void __launch_bounds__(L1_BLOCK)
l1_min_kernel(const float* __restrict__ min_z, const float* __restrict__ weight, int X, int Y,
              double lr0, const int* __restrict__ decay_steps, int n_decay, float rigid,
              int max_iters, float* __restrict__ h_out, float* __restrict__ m,
              float* __restrict__ v, SmoothState* __restrict__ st) {
  __shared__ float h[L1_MAX_CELLS];
  __shared__ double red[L1_BLOCK / 64];
  __shared__ int stop_flag;
  int n = X * Y;
  int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int e = tid; e < n; e += L1_BLOCK) { h[e] = 0.f; m[e] = 0.f; v[e] = 0.f; }
  if (tid == 0) stop_flag = 0;
  __syncthreads();
  float i1 = 1.f / (float)(X * Y), iL = 1.f / (float)((X - 2) * Y), iU = 1.f / (float)(X * (Y - 2)),
        iT = 1.f / (float)((X - 2) * (Y - 2));
  float last_loss = 1e10f;
  int countdown = 3, it = 0;
  auto H = [&](int i, int j) { return h[i * Y + j]; };
  auto W = [&](int i, int j) { return weight[i * Y + j] + 1e-2f; };
  for (; it < max_iters; ++it) {
    double lr = lr0;
    for (int q = 0; q < n_decay; ++q) if (it >= decay_steps[q]) lr *= 0.1;
    double part = 0.0;
    // gradient of every cell from the OLD h (kept in registers-free form: written to v-adjacent
    // moments only after the barrier)
    for (int e = tid; e < n; e += L1_BLOCK) {
      int i = e / Y, j = e - i * Y;
      float g = 0.f;
      float a = (H(i, j) - min_z[e]) * weight[e];
      part += (double)(fabsf(a) * i1);
      g += sgn(a) * weight[e] * i1;
      // second differences along x: centres ci = i-1, i, i+1
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
      m[e] = m[e] + 0.1f * (g - m[e]);
      v[e] = v[e] * 0.999f + 0.001f * g * g;
    }
    part = wave_sum_f64(part);
    if (lane == 0) red[w] = part;
    __syncthreads();
    double bc1 = 1.0 - pow(0.9, (double)(it + 1));
    double bc2 = 1.0 - pow(0.999, (double)(it + 1));
    float step_size = (float)(lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    float decay = (float)(1.0 - lr * 1e-2);
    for (int e = tid; e < n; e += L1_BLOCK) {
      float x = h[e] * decay;
      float denom = sqrtf(v[e]) / bc2_sqrt + 1e-8f;
      h[e] = x - step_size * (m[e] / denom);
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
  if (tid == 0) { st->iters = it; st->last_loss = last_loss; }
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
__device__ inline void eig3_smallest(const float A_in[6], float nrm[3]) {
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
  int m = 0;
  if (A[1][1] < A[m][m]) m = 1;
  if (A[2][2] < A[m][m]) m = 2;
  double sgn = V[2][m] < 0 ? -1.0 : 1.0;  // canonical sign: normal points up
  for (int k = 0; k < 3; ++k) nrm[k] = (float)(sgn * V[k][m]);
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

}  // namespace pcs

using namespace pcs;

extern "C" {

size_t pcs_smooth_velo_workspace_bytes(int64_t C, int F) {
  Arena a(nullptr, 0);
  a.take<SmoothState>(1);
  a.take<float>((size_t)C * F * 2);
  a.take<float>((size_t)C * F * 2);
  return a.off;
}

// comp_velos [C,F,3] f32 updated in place; comp_center_diffs [C,F,3]. frame_lo <= frame_hi are the
// ordered ends of the optimised range (the reference swaps them itself when tracking backwards).
int pcs_smooth_velo(float* comp_velos, const float* comp_center_diffs, int64_t C, int F,
                    int frame_lo, int frame_hi, float weight0, float weight, int num_itr,
                    float stopping, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!comp_velos || !comp_center_diffs || !ws || C < 1 || F < 1 || frame_lo < 0 ||
      frame_hi >= F || frame_lo > frame_hi || num_itr < 0)
    return PCS_ERR_ARG;
  if (frame_lo == frame_hi) return PCS_OK;  // cluster_tracking.py:166-167
  if (C * (int64_t)F * 3 >= (1LL << 31)) return PCS_ERR_RANGE;
  Arena a(ws, ws_bytes);
  SmoothState* st = a.take<SmoothState>(1);
  float* m = a.take<float>((size_t)C * F * 2);
  float* sq = a.take<float>((size_t)C * F * 2);
  if (!a.ok()) return PCS_ERR_WORKSPACE;
  smooth_velo_kernel<<<1, SV_BLOCK, 0, s>>>(comp_velos, comp_center_diffs, (int)C, F, frame_lo,
                                            frame_hi, weight0, weight, num_itr, stopping, m, sq, st);
  smooth_velo_decay_kernel<<<grid_for(C * (int64_t)F * 3), kBlock, 0, s>>>(
      comp_velos, C * (int64_t)F * 3, F, frame_lo, frame_hi, st);
  PCS_LAUNCH_CHECK();
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

}  // extern "C"
