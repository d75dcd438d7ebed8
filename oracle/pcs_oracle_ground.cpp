// CPU restatement of the ground stage's HIP kernels, arithmetic for arithmetic.
//
// TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).
//
// The ground-height estimation of pcdet/models/registration/preprocessors/preprocessor_utils.py:21-419
// is iterative floating-point work with discrete decisions on top (IRLS convergence rounds, hypothesis
// choice by hit count, pruning events, the 3-strike stop of 10 000 AdamW steps), so "the same points
// are kept" needs "the same bits" all the way. oracle/ground_oracle.py (numpy, pinned to the
// reference's own outputs with a tolerance) fixes WHAT is computed; this file fixes the ORDER of every
// floating-point operation to the one the HIP kernels use (pcs_track.hip: irls_seg_round_kernel,
// irls_seg_finish_kernel, irls_select_kernel, prune_hypotheses_kernel, l1_min_kernel; pcs_voxel.hip:
// segment_reduce_{wave,block}_kernel), serially: a "thread" is a loop index, a 64-lane butterfly is
// the same tree of additions, so tests can demand identical kept-point masks (differ == 0).
// Only + - * / sqrt on IEEE types are used on both sides, both are built with -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

constexpr int kBlock = 256;
constexpr int kIrlsWg = 128;  // irls_seg_round_kernel: one wave per (pillar, fit)

// all lanes of a 64-lane xor butterfly end with the same tree; lane 0's value
template <typename T>
T butterfly64(T* x) {
  for (int o = 32; o > 0; o >>= 1) {
    T y[64];
    for (int l = 0; l < 64; ++l) y[l] = x[l] + x[l ^ o];
    for (int l = 0; l < 64; ++l) x[l] = y[l];
  }
  return x[0];
}

// smallest-eigenvalue eigenvector of a symmetric 3x3 (cyclic Jacobi in f64; pcs_track.hip: eig3_smallest)
void eig3_smallest(const float A_in[6], float nrm[3]) {
  double A[3][3] = {{A_in[0], A_in[1], A_in[2]}, {A_in[1], A_in[3], A_in[4]}, {A_in[2], A_in[4], A_in[5]}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
    double dia = std::fabs(A[0][0]) + std::fabs(A[1][1]) + std::fabs(A[2][2]);
    if (off <= 1e-14 * dia || off == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
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
  double sgn = V[2][m] < 0 ? -1.0 : 1.0;
  for (int k = 0; k < 3; ++k) nrm[k] = (float)(sgn * V[k][m]);
}

inline float plane_weight(const float* q, float x, float y, float z, float sigma2) {
  float dx = x - q[0], dy = y - q[1], dz = z - q[2];
  float er = std::fabs(dx * q[3] + dy * q[4] + dz * q[5]);
  float nw = sigma2 / (er * er + sigma2);
  float dist_w = 0.25f / (dx * dx + dy * dy + dz * dz + 0.25f);
  return nw * dist_w;
}

inline float sgnf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

}  // namespace

extern "C" {

// pcs_segment_reduce_f32 in its three forms. mode 0: serial per segment (segment_mean_kernel);
// 1: one wave per segment (lane l adds rows l, l + 64, ... then the butterfly); 2: one 256-thread
// workgroup per segment (thread t adds rows t, t + 256, ..., butterfly per wave, waves in order).
void orc_segment_reduce_f32_mode(const float* data, int d, const int32_t* order, const int32_t* seg_start,
                                 int64_t v, int mean, int mode, float* out) {
  for (int64_t s = 0; s < v; ++s) {
    const int b = seg_start[s], e = seg_start[s + 1];
    for (int c = 0; c < d; ++c) {
      float x;
      if (mode == 0) {
        x = 0.f;
        for (int p = b; p < e; ++p) x = x + data[(int64_t)order[p] * d + c];
      } else {
        const int T = mode == 1 ? 64 : kBlock;
        float acc[kBlock];
        for (int t = 0; t < T; ++t) {
          float a = 0.f;
          for (int p = b + t; p < e; p += T) a = a + data[(int64_t)order[p] * d + c];
          acc[t] = a;
        }
        x = butterfly64(acc);
        for (int w = 1; w < T / 64; ++w) x = x + butterfly64(acc + 64 * w);
      }
      out[s * d + c] = mean ? x / (float)(e - b) : x;
    }
  }
}

// pcs_ransac_plane_hypotheses (irls_seg_round_kernel x rounds, irls_seg_finish_kernel, irls_select_kernel)
void orc_ransac_plane_hypotheses(const float* xyz, const int32_t* seg_start, int64_t n, const float* min_z,
                                 const float* max_z, const float* ratios, int R, const float* origin, int64_t P,
                                 float sigma2, float delta, int max_rounds, float hit_below, float* best_center,
                                 float* best_normal, float* best_conf) {
  (void)n;
  std::vector<float> planes_out((size_t)R * P * 6), num_hit((size_t)R * P);
  std::vector<float> planes((size_t)P * 12);
  for (int y = 0; y < R; ++y) {
    std::fill(planes.begin(), planes.end(), 0.f);
    int stop_round = 0x7fffffff, final_buf = 0;
    std::vector<float> dw_round(max_rounds + 3, 0.f);
    const float ratio = ratios[y];
    const float omr = (float)(1.0 - (double)ratio);
    for (int r = 1; r <= max_rounds + 1; ++r) {
      const bool check_only = r == max_rounds + 1;
      if (r > stop_round) continue;
      float* plane_prev = planes.data() + (size_t)((r - 1) & 1) * P * 6;
      float* plane_cur = planes.data() + (size_t)(r & 1) * P * 6;
      if (r > 2 && dw_round[r - 1] < delta) {
        stop_round = r - 1;
        final_buf = r & 1;
        continue;
      }
      if (check_only) {
        final_buf = (r - 1) & 1;
        continue;
      }
      float dw_all = 0.f;
      std::vector<float> new_cur((size_t)P * 6);
      for (int64_t c = 0; c < P; ++c) {
        const int b = seg_start[c], e = seg_start[c + 1];
        const float o0 = origin[c * 3], o1 = origin[c * 3 + 1], o2 = origin[c * 3 + 2];
        float pl[6] = {0, 0, 0, 0, 0, 0}, pl2[6] = {0, 0, 0, 0, 0, 0};
        if (r > 1) for (int q = 0; q < 6; ++q) pl[q] = plane_prev[c * 6 + q];
        if (r > 2) for (int q = 0; q < 6; ++q) pl2[q] = plane_cur[c * 6 + q];
        float cur_z = 0.f;
        if (r <= 2) cur_z = min_z[c] * ratio + max_z[c] * omr;
        // irls_chunk_round_kernel: chunks of kIrlsChunk rows; in a chunk eight row slices (slice g adds rows g, g + 8,
        // ...), the slices of a wave pairwise, the four waves in order; the chunks of a pillar in order
        constexpr int kIrlsChunk = 1024, kSlices = 8;
        double a[11];
        for (int q = 0; q < 11; ++q) a[q] = 0.0;
        float dws_all = 0.f;
        const int nk = e > b ? (e - b + kIrlsChunk - 1) / kIrlsChunk : 1;
        for (int k = 0; k < nk; ++k) {
          const int cb = b + k * kIrlsChunk, ce = std::min(e, cb + kIrlsChunk);
          double sl[kSlices][11];
          for (int g = 0; g < kSlices; ++g) {
            double v[11];
            for (int q = 0; q < 11; ++q) v[q] = 0.0;
            for (int i = cb + g; i < ce; i += kSlices) {
              float x = xyz[(int64_t)i * 3], yy = xyz[(int64_t)i * 3 + 1], z = xyz[(int64_t)i * 3 + 2];
              float wi;
              if (r <= 2) {
                float zd = cur_z - z;
                wi = (1.f / (zd * zd + sigma2)) * sigma2;
              } else {
                wi = plane_weight(pl2, x, yy, z, sigma2);
              }
              if (r > 1) {
                float nw = plane_weight(pl, x, yy, z, sigma2);
                dws_all = std::fmax(dws_all, std::fabs(nw - wi));
                wi = nw;
              }
              float ox = x - o0, oy = yy - o1, oz = z - o2;
              double W = wi;
              v[0] += W; v[1] += W * ox; v[2] += W * oy; v[3] += W * oz;
              v[4] += W * ox * ox; v[5] += W * ox * oy; v[6] += W * ox * oz;
              v[7] += W * oy * oy; v[8] += W * oy * oz; v[9] += W * oz * oz;
              v[10] += 1.0;
            }
            for (int q = 0; q < 11; ++q) sl[g][q] = v[q];
          }
          for (int q = 0; q < 11; ++q) {
            double x = sl[0][q] + sl[1][q];
            for (int w = 1; w < kSlices / 2; ++w) x += sl[2 * w][q] + sl[2 * w + 1][q];
            a[q] = k == 0 ? x : a[q] + x;
          }
        }
        const float m4 = dws_all;
        if (r > 1) dw_all = std::fmax(dw_all, m4);
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
        float* out = new_cur.data() + c * 6;
        out[0] = (float)(m[0] + og[0]);
        out[1] = (float)(m[1] + og[1]);
        out[2] = (float)(m[2] + og[2]);
        out[3] = nrm[0]; out[4] = nrm[1]; out[5] = nrm[2];
      }
      std::memcpy(plane_cur, new_cur.data(), sizeof(float) * P * 6);
      dw_round[r] = dw_all;
    }
    // finish: the result plane, hits
    for (int64_t c = 0; c < P; ++c) {
      const float* q = planes.data() + (size_t)final_buf * P * 6 + c * 6;
      for (int k = 0; k < 6; ++k) planes_out[((size_t)y * P + c) * 6 + k] = q[k];
      int hits = 0;
      for (int i = seg_start[c]; i < seg_start[c + 1]; ++i) {
        float dx = xyz[(int64_t)i * 3] - q[0], dy = xyz[(int64_t)i * 3 + 1] - q[1], dz = xyz[(int64_t)i * 3 + 2] - q[2];
        float er = std::fabs(dx * q[3] + dy * q[4] + dz * q[5]);
        hits += er < hit_below;
      }
      num_hit[(size_t)y * P + c] = (float)hits;
    }
  }
  for (int64_t c = 0; c < P; ++c) {
    float conf = 0.f;
    float ctr[3] = {0.f, 0.f, 0.f}, nrm[3] = {0.f, 0.f, 1.f};
    for (int y = 0; y < R; ++y) {
      float h = num_hit[(size_t)y * P + c];
      if (conf < h) {
        conf = h;
        const float* pl = planes_out.data() + ((size_t)y * P + c) * 6;
        ctr[0] = pl[0]; ctr[1] = pl[1]; ctr[2] = pl[2];
        nrm[0] = pl[3]; nrm[1] = pl[4]; nrm[2] = pl[5];
      }
    }
    for (int k = 0; k < 3; ++k) { best_center[c * 3 + k] = ctr[k]; best_normal[c * 3 + k] = nrm[k]; }
    best_conf[c] = conf;
  }
}

// pcs_prune_plane_hypotheses (prune_hypotheses_kernel): keep [H] uint8
void orc_prune_plane_hypotheses(const float* xyz, const float* normal, int H, int K, const float* thr, int T,
                                uint8_t* keep) {
  std::vector<uint8_t> alive(H, 1);
  std::vector<float> curv(H, 0.f);
  float s_max = 0.f;
  auto evaluate = [&]() {
    float mx = -INFINITY;
    for (int i = 0; i < H; ++i) {
      curv[i] = 0.f;
      if (!alive[i]) continue;
      double bd[8];
      int bi[8];
      for (int q = 0; q < 8; ++q) { bd[q] = 1e300; bi[q] = 0x7fffffff; }
      const double xi = xyz[3 * i], yi = xyz[3 * i + 1], zi = xyz[3 * i + 2];
      int n_alive = 0;
      for (int j = 0; j < H; ++j) {
        if (!alive[j]) continue;
        ++n_alive;
        const double dx = (double)xyz[3 * j] - xi, dy = (double)xyz[3 * j + 1] - yi, dz = (double)xyz[3 * j + 2] - zi;
        double d = dx * dx + dy * dy + dz * dz;
        int id = j;
        if (d > bd[7]) continue;
        for (int q = 0; q < 8; ++q) {
          const bool lt = d < bd[q] || (d == bd[q] && id < bi[q]);
          const double td = lt ? bd[q] : d;
          const int ti = lt ? bi[q] : id;
          bd[q] = lt ? d : bd[q];
          bi[q] = lt ? id : bi[q];
          d = td;
          id = ti;
        }
      }
      const int k = n_alive < K ? n_alive : K;
      float sum = 0.f;
      for (int q = 0; q < 8; ++q) {
        if (q < k) {
          const int j = bi[q];
          const float dx = xyz[3 * j] - xyz[3 * i], dy = xyz[3 * j + 1] - xyz[3 * i + 1], dz = xyz[3 * j + 2] - xyz[3 * i + 2];
          const float p2p = std::fabs((dx * normal[3 * i] + dy * normal[3 * i + 1]) + dz * normal[3 * i + 2]);
          sum += p2p / (std::sqrt((dx * dx + dy * dy) + dz * dz) + 1e-4f);
        }
      }
      curv[i] = sum / (float)k;
      mx = std::fmax(mx, curv[i]);
    }
    s_max = mx;
  };
  evaluate();
  for (int t = 0; t < T; ++t) {
    const float th = thr[t];
    if (th > s_max) continue;
    for (int i = 0; i < H; ++i)
      if (alive[i] && !(curv[i] < th)) alive[i] = 0;
    evaluate();
  }
  for (int i = 0; i < H; ++i) keep[i] = alive[i];
}

// pcs_l1_minimization (l1_min_kernel): h_out [X*Y]; returns the number of AdamW steps
int orc_l1_minimization(const float* min_z, const float* weight, int X, int Y, double lr0, const int* decay_steps,
                        int n_decay, float rigid, int max_iters, float* h_out) {
  constexpr int L1_BLOCK = 1024;
  const int n = X * Y;
  std::vector<float> h(n, 0.f), m(n, 0.f), v(n, 0.f), hn(n);
  const float i1 = 1.f / (float)(X * Y), iL = 1.f / (float)((X - 2) * Y), iU = 1.f / (float)(X * (Y - 2)),
              iT = 1.f / (float)((X - 2) * (Y - 2));
  float last_loss = 1e10f;
  int countdown = 3, it = 0;
  double b1p = 1.0, b2p = 1.0;  // 0.9^(it+1), 0.999^(it+1) as running products (like the kernel)
  auto H = [&](int i, int j) { return h[i * Y + j]; };
  auto W = [&](int i, int j) { return weight[i * Y + j] + 1e-2f; };
  for (; it < max_iters; ++it) {
    double lr = lr0;
    for (int q = 0; q < n_decay; ++q) if (it >= decay_steps[q]) lr *= 0.1;
    double parts[L1_BLOCK];
    for (int t = 0; t < L1_BLOCK; ++t) parts[t] = 0.0;
    for (int e = 0; e < n; ++e) {
      double& part = parts[e % L1_BLOCK];
      int i = e / Y, j = e - i * Y;
      float g = 0.f;
      float a = (H(i, j) - min_z[e]) * weight[e];
      part += (double)(std::fabs(a) * i1);
      g += sgnf(a) * weight[e] * i1;
      for (int dc = -1; dc <= 1; ++dc) {
        int ci = i + dc;
        if (ci >= 1 && ci <= X - 2) {
          float t = (H(ci - 1, j) - 2.f * H(ci, j) + H(ci + 1, j)) * W(ci, j);
          float coef = dc == 0 ? -2.f : 1.f;
          g += rigid * sgnf(t) * W(ci, j) * coef * iL;
          if (dc == 0) part += (double)(rigid * std::fabs(t) * iL);
        }
        int cj = j + dc;
        if (cj >= 1 && cj <= Y - 2) {
          float t = (H(i, cj - 1) - 2.f * H(i, cj) + H(i, cj + 1)) * W(i, cj);
          float coef = dc == 0 ? -2.f : 1.f;
          g += rigid * sgnf(t) * W(i, cj) * coef * iU;
          if (dc == 0) part += (double)(rigid * std::fabs(t) * iU);
        }
        int di = i + dc, dj = j + dc;
        if (di >= 1 && di <= X - 2 && dj >= 1 && dj <= Y - 2) {
          float t = (H(di - 1, dj - 1) - 2.f * H(di, dj) + H(di + 1, dj + 1)) * W(di, dj);
          float coef = dc == 0 ? -2.f : 1.f;
          g += rigid * sgnf(t) * W(di, dj) * coef * iT;
          if (dc == 0) part += (double)(rigid * std::fabs(t) * iT);
        }
        int ai = i - dc, aj = j + dc;
        if (ai >= 1 && ai <= X - 2 && aj >= 1 && aj <= Y - 2) {
          float t = (H(ai + 1, aj - 1) - 2.f * H(ai, aj) + H(ai - 1, aj + 1)) * W(ai, aj);
          float coef = dc == 0 ? -2.f : 1.f;
          g += rigid * sgnf(t) * W(ai, aj) * coef * iT;
          if (dc == 0) part += (double)(rigid * std::fabs(t) * iT);
        }
      }
      m[e] = m[e] + 0.1f * (g - m[e]);
      v[e] = v[e] * 0.999f + 0.001f * g * g;
    }
    double tot = 0.0;
    for (int w = 0; w < L1_BLOCK / 64; ++w) tot += butterfly64(parts + 64 * w);
    b1p *= 0.9;
    b2p *= 0.999;
    double bc1 = 1.0 - b1p, bc2 = 1.0 - b2p;
    float step_size = (float)(lr / bc1);
    float bc2_sqrt = (float)std::sqrt(bc2);
    float decay = (float)(1.0 - lr * 1e-2);
    for (int e = 0; e < n; ++e) {
      float x = h[e] * decay;
      float denom = std::sqrt(v[e]) / bc2_sqrt + 1e-8f;
      hn[e] = x - step_size * (m[e] / denom);
    }
    h.swap(hn);
    float loss = (float)tot;
    bool stop = false;
    if (last_loss - loss < 1e-4f) countdown -= 1;
    else countdown = 3;
    if (countdown == 0) stop = true;
    else last_loss = loss;
    if (stop) { ++it; break; }
  }
  for (int e = 0; e < n; ++e) h_out[e] = h[e];
  return it;
}

}  // extern "C"
