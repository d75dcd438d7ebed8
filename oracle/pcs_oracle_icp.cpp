// pcs_oracle_icp.cpp -- C++ restatement of register_to_next_frame for the CPU baseline.
//
// TEST INFRASTRUCTURE ONLY (see pcs_oracle.cpp): used by bench.py's cpu_baseline leg and pinned by
// tests/test_oracle_registration.py against oracle/reg_oracle.py (the numpy restatement, itself
// pinned to the reference's own Python by tests/golden/reg_*.npz).
//
// Follows pcdet/models/registration/preprocessors/registration_utils.py:83-206 the way the
// reference executes it: per iteration two 1-NN radius graphs on the literal torch_hash table
// (pcdet/ops/torch_hash/src/torch_hash_kernel.cu:31-91,224-409: cells of one radius, 27-cell walk,
// linear probing), sorted-segment means, per-component covariance, 3x3 SVD, R = V diag(1,1,det) U^T.
// `threads` = 1 is the serial baseline; > 1 spreads the query walks and the per-edge terms over
// OpenMP threads (the sums stay in a fixed order: per-thread partials are combined serially).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include <omp.h>

namespace {

struct Table {
  int64_t H = 0;
  std::vector<int64_t> keys, rev;
  std::vector<float> vals;
  float lo[4], vs[4];
  int64_t dims[4];
};

inline int64_t map2key(const int64_t* c, const int64_t* dims) {
  int64_t ans = 0;
  for (int i = 0; i < 4; ++i) {
    int64_t k = c[i];
    if (k >= dims[i]) k = dims[i];
    if (k < 0) k = 0;
    ans = ans * dims[i] + k;
  }
  return ans;
}
inline int64_t hashkey(int64_t key, int64_t H) { return ((key % H) * 999269 + 999437) % H; }

// graph_utils.py:169-183 + hash_insert_gpu
void build_table(const float* ref, int64_t N, const float* qry, int64_t M, float radius, Table& t) {
  t.vs[0] = (float)(1 - 1e-3);
  t.vs[1] = t.vs[2] = t.vs[3] = radius;
  float hi[4];
  for (int d = 0; d < 4; ++d) { t.lo[d] = INFINITY; hi[d] = -INFINITY; }
  for (int64_t i = 0; i < N; ++i)
    for (int d = 0; d < 4; ++d) { t.lo[d] = std::fmin(t.lo[d], ref[i * 4 + d]); hi[d] = std::fmax(hi[d], ref[i * 4 + d]); }
  for (int64_t i = 0; i < M; ++i)
    for (int d = 0; d < 4; ++d) { t.lo[d] = std::fmin(t.lo[d], qry[i * 4 + d]); hi[d] = std::fmax(hi[d], qry[i * 4 + d]); }
  for (int d = 0; d < 4; ++d) {
    t.lo[d] = t.lo[d] - t.vs[d] * 2.f;
    hi[d] = hi[d] + t.vs[d] * 2.f;
    t.dims[d] = (int64_t)std::nearbyint((hi[d] - t.lo[d]) / t.vs[d]) + 3;
  }
  t.H = (int64_t)((double)N / 0.5);
  if (t.H < 1) t.H = 1;
  t.keys.assign(t.H, -1);
  t.rev.assign(t.H, 0);
  t.vals.assign((size_t)t.H * 4, 0.f);
  for (int64_t i = 0; i < N; ++i) {
    int64_t c[4];
    for (int d = 0; d < 4; ++d) c[d] = (int64_t)std::nearbyint((ref[i * 4 + d] - t.lo[d]) / t.vs[d]) + 1;
    int64_t key = map2key(c, t.dims);
    int64_t slot = hashkey(key, t.H);
    while (t.keys[slot] != -1) slot = (slot + 1) % t.H;
    t.keys[slot] = key;
    for (int d = 0; d < 4; ++d) t.vals[slot * 4 + d] = ref[i * 4 + d];
    t.rev[slot] = i;
  }
}

// sorted 1-NN of every query within the radius (radius_graph_gpu with max_nbr 1, SORT_BY_DIST,
// qmin/qmax = [fo,-1,-1,-1]/[fo,1,1,1]); canonical tie order (dist2, ref index). nn[q] = ref index or -1.
void nn1(const Table& t, const float* qry, int64_t M, float radius, int fo, int64_t* nn, int threads) {
  const float r2 = radius * radius;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t q = 0; q < M; ++q) {
    int64_t qc[4];
    for (int d = 0; d < 4; ++d) qc[d] = (int64_t)std::nearbyint((qry[q * 4 + d] - t.lo[d]) / t.vs[d]) + 1;
    int64_t best = -1;
    float bd = 0.f;
    for (int c = 0; c < 27; ++c) {
      int64_t cc[4] = {qc[0] + fo, qc[1] + (c % 3 - 1), qc[2] + ((c / 3) % 3 - 1), qc[3] + (c / 9 - 1)};
      int64_t key = map2key(cc, t.dims);
      int64_t slot = hashkey(key, t.H);
      while (t.keys[slot] != -1) {
        if (t.keys[slot] == key) {
          float dist2 = 0.f;
          for (int d = 0; d < 4; ++d) {
            float di = t.vals[slot * 4 + d] - qry[q * 4 + d];
            dist2 = dist2 + di * di;
          }
          if (dist2 <= r2) {
            int64_t idx = t.rev[slot];
            if (best < 0 || dist2 < bd || (dist2 == bd && idx < best)) { best = idx; bd = dist2; }
          }
        }
        slot = (slot + 1) % t.H;
      }
    }
    nn[q] = best;
  }
}

// one-sided Jacobi SVD of a 3x3 (f64): M = U diag(S) V^T
void svd3(const double M[9], double U[9], double S[3], double V[9]) {
  double A[9];
  for (int i = 0; i < 9; ++i) { A[i] = M[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < 3; ++i) {
          al += A[i * 3 + p] * A[i * 3 + p];
          be += A[i * 3 + q] * A[i * 3 + q];
          ga += A[i * 3 + p] * A[i * 3 + q];
        }
        if (ga == 0.0 || std::fabs(ga) <= 1e-17 * std::sqrt(al * be)) continue;
        off = std::fmax(off, std::fabs(ga) / (std::sqrt(al * be) + 1e-300));
        double zeta = (be - al) / (2.0 * ga);
        double tt = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + tt * tt), s = c * tt;
        for (int i = 0; i < 3; ++i) {
          double ap = A[i * 3 + p], aq = A[i * 3 + q];
          A[i * 3 + p] = c * ap - s * aq;
          A[i * 3 + q] = s * ap + c * aq;
          double vp = V[i * 3 + p], vq = V[i * 3 + q];
          V[i * 3 + p] = c * vp - s * vq;
          V[i * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-16) break;
  }
  for (int j = 0; j < 3; ++j) {
    S[j] = std::sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    for (int i = 0; i < 3; ++i) U[i * 3 + j] = S[j] > 1e-300 ? A[i * 3 + j] / S[j] : (i == j ? 1.0 : 0.0);
  }
}

inline double det3(const double A[9]) {
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

}  // namespace

extern "C" {

// mov [Mn,4] f32 (non-stationary moving points, updated in place), comp [Mn] i64, ref [Rn,4] f32
// (non-stationary reference), ref_all [Ra,4], comp_deg [C] i64. T [C,16] f64, l1 [C] f64, ratio [C] f32.
// Returns the number of iterations.
int orc_register_to_next_frame(float* mov, const int64_t* comp, int64_t Mn, const float* ref, int64_t Rn,
                               const float* ref_all, int64_t Ra, const int64_t* comp_deg, int64_t C,
                               double radius, int fo, double angle_regularizer, int max_iter,
                               double stopping_delta, double* T, double* l1, float* ratio, int threads) {
  if (threads < 1) threads = 1;
  const float rad = (float)std::sqrt(radius * radius + (double)fo * (double)fo);
  for (int64_t c = 0; c < C; ++c)
    for (int k = 0; k < 16; ++k) T[c * 16 + k] = (k % 5 == 0) ? 1.0 : 0.0;
  std::fill(l1, l1 + C, 0.0);
  std::vector<int64_t> nn_f(Mn), nn_b(Rn);
  std::vector<int64_t> e_mov, e_ref;
  std::vector<double> mc(C * 3), rc(C * 3), cov(C * 9), cnt(C);
  std::vector<double> dist;
  double last_error = 1e10;
  int countdown = 3, iters = 0;
  Table tref;  // the reference cloud does not move; its cell origin depends on both clouds, though
  for (int itr = 0; itr < max_iter; ++itr) {
    ++iters;
    build_table(ref, Rn, mov, Mn, rad, tref);
    nn1(tref, mov, Mn, rad, fo, nn_f.data(), threads);
    Table tmov;
    build_table(mov, Mn, ref, Rn, rad, tmov);
    nn1(tmov, ref, Rn, rad, -fo, nn_b.data(), threads);
    e_mov.clear();
    e_ref.clear();
    for (int64_t m = 0; m < Mn; ++m)
      if (nn_f[m] >= 0) { e_mov.push_back(m); e_ref.push_back(nn_f[m]); }
    for (int64_t r = 0; r < Rn; ++r)
      if (nn_b[r] >= 0) { e_mov.push_back(nn_b[r]); e_ref.push_back(r); }
    const int64_t E = (int64_t)e_mov.size();
    // centres: f64 sums, mean, rounded once to fp32 (efficient_robust_mean on fp32 input), then double
    std::fill(mc.begin(), mc.end(), 0.0);
    std::fill(rc.begin(), rc.end(), 0.0);
    std::fill(cnt.begin(), cnt.end(), 0.0);
    for (int64_t e = 0; e < E; ++e) {
      int64_t c = comp[e_mov[e]];
      cnt[c] += 1.0;
      for (int d = 0; d < 3; ++d) {
        mc[c * 3 + d] += (double)mov[e_mov[e] * 4 + 1 + d];
        rc[c * 3 + d] += (double)ref[e_ref[e] * 4 + 1 + d];
      }
    }
    for (int64_t c = 0; c < C; ++c)
      for (int d = 0; d < 3; ++d) {
        mc[c * 3 + d] = cnt[c] > 0 ? (double)(float)(mc[c * 3 + d] / cnt[c]) : 0.0;
        rc[c * 3 + d] = cnt[c] > 0 ? (double)(float)(rc[c * 3 + d] / cnt[c]) : 0.0;
      }
    std::fill(cov.begin(), cov.end(), 0.0);
    dist.assign(E, 0.0);
    double loss = 0.0;
    for (int64_t e = 0; e < E; ++e) {
      int64_t c = comp[e_mov[e]];
      double P[3], Q[3], d2 = 0.0;
      for (int d = 0; d < 3; ++d) {
        P[d] = (double)mov[e_mov[e] * 4 + 1 + d] - mc[c * 3 + d];
        Q[d] = (double)ref[e_ref[e] * 4 + 1 + d] - rc[c * 3 + d];
        d2 += (P[d] - Q[d]) * (P[d] - Q[d]);
      }
      dist[e] = std::sqrt(d2);
      loss += dist[e] * dist[e];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) cov[c * 9 + a * 3 + b] += P[a] * Q[b];
    }
    // truncated_robust_mean of dist (registration_utils.py:44-58)
    {
      std::vector<double> mean(C, 0.0), mean2(C, 0.0);
      for (int64_t e = 0; e < E; ++e) mean[comp[e_mov[e]]] += dist[e];
      for (int64_t c = 0; c < C; ++c) mean[c] = cnt[c] > 0 ? mean[c] / cnt[c] : 0.0;
      for (int64_t e = 0; e < E; ++e) {
        int64_t c = comp[e_mov[e]];
        mean2[c] += std::fmin(std::fmax(dist[e], mean[c] - 0.3), mean[c] + 0.3);
      }
      for (int64_t c = 0; c < C; ++c) l1[c] = cnt[c] > 0 ? mean2[c] / cnt[c] : 0.0;
    }
    std::vector<double> Rall(C * 9), tall(C * 3);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t c = 0; c < C; ++c) {
      double M[9], U[9], S[3], V[9];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
          double cv = cnt[c] > 0.5 ? cov[c * 9 + a * 3 + b] / cnt[c] : cov[c * 9 + a * 3 + b];
          M[a * 3 + b] = cv + T[c * 16 + a * 4 + b] * angle_regularizer;
        }
      svd3(M, U, S, V);
      double VUt[9];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
          VUt[a * 3 + b] = V[a * 3 + 0] * U[b * 3 + 0] + V[a * 3 + 1] * U[b * 3 + 1] + V[a * 3 + 2] * U[b * 3 + 2];
      // sign of the column belonging to the SMALLEST singular value (LAPACK orders descending)
      int jmin = 0;
      for (int j = 1; j < 3; ++j)
        if (S[j] < S[jmin]) jmin = j;
      double sg = det3(VUt);
      double* R = &Rall[c * 9];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
          double acc = 0.0;
          for (int j = 0; j < 3; ++j) acc += (j == jmin ? sg : 1.0) * V[a * 3 + j] * U[b * 3 + j];
          R[a * 3 + b] = acc;
        }
      for (int a = 0; a < 3; ++a)
        tall[c * 3 + a] = rc[c * 3 + a] - (mc[c * 3 + 0] * R[a * 3 + 0] + mc[c * 3 + 1] * R[a * 3 + 1] +
                                           mc[c * 3 + 2] * R[a * 3 + 2]);
      double Tn[16], *Tc = &T[c * 16];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 4; ++b)
          Tn[a * 4 + b] = R[a * 3 + 0] * Tc[0 * 4 + b] + R[a * 3 + 1] * Tc[1 * 4 + b] + R[a * 3 + 2] * Tc[2 * 4 + b] +
                          tall[c * 3 + a] * Tc[3 * 4 + b];
      for (int b = 0; b < 4; ++b) Tn[12 + b] = Tc[12 + b];
      for (int k = 0; k < 16; ++k) Tc[k] = Tn[k];
    }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t m = 0; m < Mn; ++m) {
      const double* R = &Rall[comp[m] * 9];
      const double* t = &tall[comp[m] * 3];
      double x = mov[m * 4 + 1], y = mov[m * 4 + 2], z = mov[m * 4 + 3];
      mov[m * 4 + 1] = (float)(x * R[0] + y * R[1] + z * R[2] + t[0]);
      mov[m * 4 + 2] = (float)(x * R[3] + y * R[4] + z * R[5] + t[1]);
      mov[m * 4 + 3] = (float)(x * R[6] + y * R[7] + z * R[8] + t[2]);
    }
    if (last_error - loss < stopping_delta) countdown -= 1;
    else countdown = 3;
    if (countdown <= 0) break;
    last_error = loss;
  }
  // edge coverage against ALL reference points (registration_utils.py:191-199)
  {
    Table tall;
    build_table(ref_all, Ra, mov, Mn, rad, tall);
    std::vector<int64_t> nn(Mn);
    nn1(tall, mov, Mn, rad, fo, nn.data(), threads);
    std::vector<int64_t> c2(C, 0);
    for (int64_t m = 0; m < Mn; ++m)
      if (nn[m] >= 0) c2[comp[m]] += 1;
    for (int64_t c = 0; c < C; ++c) ratio[c] = (float)c2[c] / ((float)comp_deg[c] + 1e-6f);
  }
  return iters;
}

int orc_max_threads() { return omp_get_max_threads(); }

}  // extern "C"

// ---- CPU twin of pcs_register_to_next_frame (include/pcs_cpu.h): the same argument list, host pointers
extern "C" int pcs_cpu_register_to_next_frame(float* mov_fxyz, const int32_t* mov_comp, int64_t Mn,
                                              const float* ref_fxyz, int64_t Rn, const float* ref_all, int64_t Ra,
                                              const int64_t* comp_deg, int64_t C, double radius, int frame_offset,
                                              double angle_regularizer, int max_iter, double stopping_delta,
                                              float cell_size, int chunk, double* T, double* l1_err,
                                              float* edge_ratio, int32_t* info, void* ws, size_t ws_bytes,
                                              void* stream) {
  (void)cell_size; (void)chunk; (void)ws; (void)ws_bytes; (void)stream;
  if (!mov_fxyz || !mov_comp || !comp_deg || !T || !l1_err || !edge_ratio || !info || C < 1) return -1;
  std::vector<int64_t> comp64((size_t)Mn);
  for (int64_t i = 0; i < Mn; ++i) comp64[i] = mov_comp[i];
  int iters = orc_register_to_next_frame(mov_fxyz, comp64.data(), Mn, ref_fxyz, Rn, ref_all, Ra, comp_deg, C, radius,
                                         frame_offset, angle_regularizer, max_iter, stopping_delta, T, l1_err,
                                         edge_ratio, 1);
  info[0] = iters;
  info[1] = iters < max_iter ? 1 : 0;
  info[2] = info[3] = 0;
  return 0;
}
