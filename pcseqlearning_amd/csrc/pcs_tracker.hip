// Native host runtime + kernels of ClusterTracking.track_frame for gfx950.
//
// Reference: pcdet/models/registration/preprocessors/cluster_tracking.py:430-787 (track_frame;
// SURVEY.md A.6) with its helpers sample_frame (:39-51), component_center / component_diameter /
// filter_components (:89-160), dist_compensate (:80-87) and smooth_velo (:162-199).
//
// In the reference every step of the tracking loop is ~450 small framework ops issued from the
// interpreter. A tracking unit (anchor frame x component key) is a chain of small, latency-bound
// launches, and 9 independent units exist per sequence; issued from interpreter threads they
// serialise on the interpreter lock (tools/launch_scaling.py: 115 k launches/s from 9 threads),
// issued from native threads they scale (1.7 M launches/s). So the whole loop of one unit lives
// here: one call, one stream, all state in the caller's workspace, a handful of fused kernels per
// step and three small host reads per step (loop condition, sample counts, ICP stop word).
//
// Arithmetic follows the host-side restatement (pcdet/.../cluster_tracking.py of this package) op
// for op: fp32 where torch computes in fp32, f64 for the transforms, the same fixed-order segment
// reductions (pcs_segment_reduce_f32 serial / wave modes), so both produce identical ids.
#include "pcs_common.hpp"

#include "pcs.h"

namespace pcs {

constexpr int kMaxLevels = 4;

struct TrkState {
  int32_t active;      // any component still tracked (loop condition of the next step)
  int32_t err;
  int64_t ex_count;    // extracted points so far
  int64_t counts[4];   // per level: voxels(anchor), non-stationary(anchor), voxels(next), non-stationary(next)
};

// ------------------------------------------------------------------ anchor set-up
__global__ void __launch_bounds__(kBlock)
trk_load_anchor_kernel(const float4* __restrict__ src, const int64_t* __restrict__ comp_in,
                       const uint8_t* __restrict__ stat_in, int64_t base, int n,
                       float4* __restrict__ xyzf, int64_t* __restrict__ comp64,
                       int32_t* __restrict__ comp32, float* __restrict__ stat_f) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  xyzf[i] = src[i];
  int64_t c = comp_in[i] - base;
  comp64[i] = c;
  comp32[i] = (int32_t)c;
  stat_f[i] = stat_in[i] ? 1.f : 0.f;
}

__global__ void __launch_bounds__(kBlock)
trk_stat_to_float_kernel(const uint8_t* __restrict__ stat_in, int n, float* __restrict__ stat_f) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stat_f[i] = stat_in[i] ? 1.f : 0.f;
}

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
  if (deg_f) { deg_f[c] = nf; deg_i[c] = n; }
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

// per component: masks, identity transforms, output defaults, centre of the anchor frame
__global__ void __launch_bounds__(kBlock)
trk_comp_init_kernel(int C, int F, int W, int frame_id, float max_diam, const float* __restrict__ deg_f,
                     const float* __restrict__ diam, const float* __restrict__ ctr,
                     uint8_t* __restrict__ vcm, double* __restrict__ transforms,
                     int64_t* __restrict__ cmin, int64_t* __restrict__ cmax,
                     float* __restrict__ centers) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bool valid = deg_f[c] > 0.5f;
  if (max_diam > 0.f) valid = valid && (diam[c] < max_diam);
  vcm[c] = valid;
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

// ------------------------------------------------------------------ per direction
__global__ void __launch_bounds__(kBlock)
trk_dir_init_kernel(int C, int n, const uint8_t* __restrict__ vcm, uint8_t* __restrict__ stopped,
                    uint8_t* __restrict__ moving, const float4* __restrict__ anchor,
                    float4* __restrict__ xyzf, float4* __restrict__ last, TrkState* st) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) {
    bool v = vcm[i];
    stopped[i] = !v;
    moving[i] = v;
    if (v) st->active = 1;
  }
  if (i < n) {
    float4 p = anchor[i];
    xyzf[i] = p;
    last[i] = p;
  }
}

// ------------------------------------------------------------------ per step
// transforms[:, col] = transforms[:, col - dir]; constant-velocity prediction for the components
// still tracked (cluster_tracking.py:520-534)
__global__ void __launch_bounds__(kBlock)
trk_step_begin_kernel(int C, int n, int F, int W, int col, int dir, int last_col,
                      const uint8_t* __restrict__ stopped, const float* __restrict__ velos,
                      const int32_t* __restrict__ comp, float4* __restrict__ xyzf,
                      double* __restrict__ transforms) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) {
    double* dst = transforms + ((int64_t)i * W + col) * 16;
    const double* src = transforms + ((int64_t)i * W + (col - dir)) * 16;
    for (int k = 0; k < 16; ++k) dst[k] = src[k];
    if (last_col >= 0 && !stopped[i]) {
      const float* v = velos + ((int64_t)i * F + last_col) * 3;
      float fd = (float)dir;
      dst[3] += (double)(v[0] * fd);
      dst[7] += (double)(v[1] * fd);
      dst[11] += (double)(v[2] * fd);
    }
  }
  if (i < n && last_col >= 0) {
    int c = comp[i];
    if (!stopped[c]) {
      const float* v = velos + ((int64_t)c * F + last_col) * 3;
      float fd = (float)dir;
      float4 p = xyzf[i];
      p.y = p.y + v[0] * fd;
      p.z = p.z + v[1] * fd;
      p.w = p.w + v[2] * fd;
      xyzf[i] = p;
    }
  }
}

// ---- anchor + target frame sampled by ONE key / sort / segment pipeline -------------------------
// Both clouds get their grid_cluster ids (same arithmetic as grid_ids_kernel, each with its own
// bounds) under a set bit on top of the key, so one sort yields the anchor's voxels first, then the
// target's, every voxel's rows in ascending original index: exactly the two separate pipelines,
// half the launches.
__global__ void __launch_bounds__(kBlock)
trk_bounds2_kernel(const float4* __restrict__ a, int na, const float4* __restrict__ b, int nb,
                   float* __restrict__ out16) {
  const float4* pts = blockIdx.y == 0 ? a : b;
  int n = blockIdx.y == 0 ? na : nb;
  float* out8 = out16 + 8 * blockIdx.y;
  float mn[4] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
  float mx[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 p = pts[i];
    mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
    mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
    mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
    mn[3] = fminf(mn[3], p.w); mx[3] = fmaxf(mx[3], p.w);
  }
  __shared__ float smn[kBlock / 64][4], smx[kBlock / 64][4];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    float x = wave_min(mn[d]), y = wave_max(mx[d]);
    if (lane == 0) { smn[w][d] = x; smx[w][d] = y; }
  }
  __syncthreads();
  if (threadIdx.x < 4 && n > 0) {
    int d = threadIdx.x;
    float x = smn[0][d], y = smx[0][d];
    for (int k = 1; k < kBlock / 64; ++k) { x = fminf(x, smn[k][d]); y = fmaxf(y, smx[k][d]); }
    atomic_min_f32(&out8[d], x);
    atomic_max_f32(&out8[4 + d], y);
  }
}

__global__ void trk_bounds_reset_kernel(float* out16) {
  int t = threadIdx.x;
  if (t < 16) out16[t] = ((t & 7) < 4) ? __builtin_inff() : -__builtin_inff();
}

__global__ void __launch_bounds__(kBlock)
trk_keys2_kernel(const float4* __restrict__ a, int na, const float4* __restrict__ b, int nb,
                 const float* __restrict__ bounds16, float4 size, float margin0,
                 uint64_t* __restrict__ keys, int32_t* __restrict__ vals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= na + nb) return;
  int set = i >= na;
  const float* b8 = bounds16 + 8 * set;
  float st[4], en[4], sz[4] = {size.x, size.y, size.z, size.w};
#pragma unroll
  for (int d = 0; d < 4; ++d) { st[d] = b8[d]; en[d] = b8[4 + d]; }
  st[0] = st[0] - margin0;
  en[0] = en[0] + margin0;
  int64_t stride[4];
  stride[0] = 1;
#pragma unroll
  for (int d = 0; d < 3; ++d) stride[d + 1] = stride[d] * ((int64_t)((en[d] - st[d]) / sz[d]) + 1);
  float4 p = set ? b[i - na] : a[i];
  int64_t id = (int64_t)((p.x - st[0]) / sz[0]) * stride[0] + (int64_t)((p.y - st[1]) / sz[1]) * stride[1] +
               (int64_t)((p.z - st[2]) / sz[2]) * stride[2] + (int64_t)((p.w - st[3]) / sz[3]) * stride[3];
  keys[i] = ((uint64_t)set << 62) | (uint64_t)id;
  vals[i] = i;
}

// per voxel of the combined structure: serial mean of fxyz, stationary vote, (anchor only) median
// component and the per-component voxel count; also finds where the target's voxels begin
__global__ void __launch_bounds__(kBlock)
trk_sample2_emit_kernel(const float4* __restrict__ a, const float* __restrict__ stat_a,
                        const int64_t* __restrict__ comp64, int na, const float4* __restrict__ b,
                        const float* __restrict__ stat_b, const int32_t* __restrict__ order,
                        const int32_t* __restrict__ seg, const int64_t* __restrict__ d_num, int n,
                        float4* __restrict__ sub, int32_t* __restrict__ sub_comp,
                        int32_t* __restrict__ flag_ns, unsigned long long* __restrict__ sub_deg,
                        int64_t* __restrict__ d_va) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  int64_t v = *d_num;
  if (s >= v) { flag_ns[s] = 0; return; }
  int bb = seg[s], e = seg[s + 1];
  bool tgt = order[bb] >= na;
  // first target voxel (or v if there is none)
  if (tgt && (s == 0 || order[seg[s - 1]] < na)) *d_va = s;
  if (!tgt && s == v - 1) *d_va = v;
  // a coarse voxel holds dozens of rows and every row is two dependent loads (order -> point): eight
  // rows are requested together and then added in their original order (same sums, an eighth of the
  // round trips); the anchor's component ids ride along
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, st = 0.f;
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
        if (tgt) {
          q[j] = b[o[j] - na];
          sv[j] = stat_b[o[j] - na];
        } else {
          q[j] = a[o[j]];
          sv[j] = stat_a[o[j]];
          cc[j] = comp64[o[j]];
        }
      }
    }
    if (p0 == bb) first = cc[0];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (o[j] >= 0) {
        a0 = a0 + q[j].x; a1 = a1 + q[j].y; a2 = a2 + q[j].z; a3 = a3 + q[j].w;
        st = st + sv[j];
        same = same && (cc[j] == first);
      }
    }
  }
  float cnt = (float)(e - bb);
  sub[s] = make_float4(a0 / cnt, a1 / cnt, a2 / cnt, a3 / cnt);
  flag_ns[s] = !((st / cnt) > 0.5f);
  int64_t comp_of = -1;  // anchor voxels: the voxel's (median) component
  if (!tgt) {
    int k = (e - bb) / 2;
    int64_t res = first;
    if (!same) {
      for (int p = bb; p < e; ++p) {
        int64_t x = comp64[order[p]];
        int lt = 0, eq = 0;
        for (int q = bb; q < e; ++q) {
          int64_t y = comp64[order[q]];
          lt += (y < x);
          eq += (y == x);
        }
        if (lt <= k && k < lt + eq) { res = x; break; }
      }
    }
    sub_comp[s] = (int32_t)res;
    comp_of = res;
  }
  // per-component voxel counts: neighbouring voxels mostly share a component and a large component
  // owns thousands of voxels, so one atomic per voxel queues up on a few addresses (~12 ns each);
  // the lanes of a wave that agree on the component send one add between them
  {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(comp_of >= 0);
    while (todo) {
      int leader = __ffsll((long long)todo) - 1;
      long long lc = __shfl((long long)comp_of, leader, 64);
      unsigned long long grp = __ballot(comp_of == lc);
      if (lane == leader) atomicAdd(&sub_deg[lc], (unsigned long long)__popcll(grp));
      todo &= ~grp;
    }
  }
}

// split + compact: anchor voxels -> mov (the non-stationary ones), target -> sub_n (all) / ref_ns;
// counts = {anchor voxels, anchor non-stationary, target voxels, target non-stationary}
__global__ void __launch_bounds__(kBlock)
trk_compact2_kernel(const float4* __restrict__ sub, const int32_t* __restrict__ sub_comp,
                    const int32_t* __restrict__ flag, const int32_t* __restrict__ pos, int n,
                    const int64_t* __restrict__ d_num, const int64_t* __restrict__ d_va,
                    float4* __restrict__ mov, int32_t* __restrict__ mov_comp,
                    float4* __restrict__ sub_n, float4* __restrict__ ref_ns,
                    int64_t* __restrict__ counts) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  int64_t v = *d_num, va = *d_va;
  int total_ns = pos[n - 1] + flag[n - 1];
  int ns_a = va < n ? pos[va] : total_ns;  // non-stationary anchor voxels = rank of the first target voxel
  if (s < v) {
    float4 q = sub[s];
    if (s < va) {
      if (flag[s]) { mov[pos[s]] = q; mov_comp[pos[s]] = sub_comp[s]; }
    } else {
      sub_n[s - va] = q;
      if (flag[s]) ref_ns[pos[s] - ns_a] = q;
    }
  }
  if (s == 0) {
    counts[0] = va;
    counts[1] = ns_a;
    counts[2] = v - va;
    counts[3] = total_ns - ns_a;
  }
}

// frame.fxyz[:, 1:] = (R @ xyz.double()).float() + t.float(); transforms[:, col] = T @ transforms[:, col]
__global__ void __launch_bounds__(kBlock)
trk_apply_T_kernel(int C, int n, int W, int col, const double* __restrict__ T,
                   const int32_t* __restrict__ comp, float4* __restrict__ xyzf,
                   double* __restrict__ transforms) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double* M = T + (int64_t)comp[i] * 16;
    float4 p = xyzf[i];
    double x = (double)p.y, y = (double)p.z, z = (double)p.w;
    float rx = (float)((M[0] * x + M[1] * y) + M[2] * z);
    float ry = (float)((M[4] * x + M[5] * y) + M[6] * z);
    float rz = (float)((M[8] * x + M[9] * y) + M[10] * z);
    p.y = rx + (float)M[3];
    p.z = ry + (float)M[7];
    p.w = rz + (float)M[11];
    xyzf[i] = p;
  }
  if (i < C) {
    const double* A = T + (int64_t)i * 16;
    double* B = transforms + ((int64_t)i * W + col) * 16;
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

// point_velo = (xyz - last_xyz) * dir, laid out [n,4] (4th column 0) for the d = 4 reductions
__global__ void __launch_bounds__(kBlock)
trk_point_velo_kernel(const float4* __restrict__ xyzf, const float4* __restrict__ last, int n, int dir,
                      float4* __restrict__ pv) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = xyzf[i], q = last[i];
  float fd = (float)dir;
  pv[i] = make_float4((p.y - q.y) * fd, (p.z - q.z) * fd, (p.w - q.w) * fd, 0.f);
}

// per present component: centre of the moved points, mean point velocity (z zeroed), centre
// difference (cluster_tracking.py:631-648)
__global__ void __launch_bounds__(kBlock)
trk_comp_motion_kernel(const int64_t* __restrict__ uniq, const int32_t* __restrict__ seg, int Cu,
                       int F, int next, int dir, const float* __restrict__ sum4,
                       const float* __restrict__ mean4, float* __restrict__ centers,
                       float* __restrict__ velos, float* __restrict__ diffs,
                       float* __restrict__ velo_pre) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= Cu) return;
  int64_t c = uniq[u];
  float nf = (float)(seg[u + 1] - seg[u]);
  float* cn = centers + ((int64_t)c * F + next) * 3;
  const float* cp = centers + ((int64_t)c * F + (next - dir)) * 3;
  float* vn = velos + ((int64_t)c * F + next) * 3;
  float* dn = diffs + ((int64_t)c * F + next) * 3;
  float fd = (float)dir;
  for (int k = 0; k < 3; ++k) {
    float ctr = sum4[u * 4 + k] / nf;
    cn[k] = ctr;
    dn[k] = (ctr - cp[k]) * fd;
    float v = k < 2 ? mean4[u * 4 + k] : 0.f;
    vn[k] = v;
    velo_pre[c * 3 + k] = v;
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
// (cluster_tracking.py:650-700) and the per-step records
__global__ void __launch_bounds__(kBlock)
trk_after_smooth_kernel(int C, int n, int F, int W, int col, int frame_id, int next, int dir,
                        int last_col, int min_move_frame, float reg_coeff, float angle_thr,
                        const int32_t* __restrict__ comp, float4* __restrict__ xyzf,
                        float4* __restrict__ last, float* __restrict__ velos,
                        const float* __restrict__ velo_pre, const float* __restrict__ centers,
                        const float* __restrict__ diam, const int64_t* __restrict__ deg_i,
                        const double* __restrict__ l1, const float* __restrict__ ratio,
                        uint8_t* __restrict__ stopped, uint8_t* __restrict__ moving,
                        double* __restrict__ transforms, int64_t* __restrict__ cmin,
                        int64_t* __restrict__ cmax, float* __restrict__ reg_errors,
                        float* __restrict__ edge_ratios, TrkState* st) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float fd = (float)dir;
  if (i < n) {
    int c = comp[i];
    const float* vn = velos + ((int64_t)c * F + next) * 3;
    const float* vp = velo_pre + (int64_t)c * 3;
    float4 p = xyzf[i];
    p.y = p.y + (vn[0] - vp[0]) * fd;
    p.z = p.z + (vn[1] - vp[1]) * fd;
    p.w = p.w + (vn[2] - vp[2]) * fd;
    xyzf[i] = p;
    last[i] = p;
  }
  if (i < C) {
    int c = i;
    float* vn = velos + ((int64_t)c * F + next) * 3;
    const float* vp = velo_pre + (int64_t)c * 3;
    double* T = transforms + ((int64_t)c * W + col) * 16;
    T[3] += (double)((vn[0] - vp[0]) * fd);
    T[7] += (double)((vn[1] - vp[1]) * fd);
    T[11] += (double)((vn[2] - vp[2]) * fd);
    bool stop = stopped[c];
    float dm = diam[c];
    float thr = (reg_coeff * dm) * (1.f + dist_compensate(deg_i[c]));
    stop = stop || (l1[c] > (double)thr);
    stop = stop || (ratio[c] < 0.5f);
    if ((next - frame_id) * dir == min_move_frame) {
      const float* cn = centers + ((int64_t)c * F + next) * 3;
      const float* c0 = centers + ((int64_t)c * F + frame_id) * 3;
      float moved = norm3(cn[0] - c0[0], cn[1] - c0[1], cn[2] - c0[2]);
      moving[c] = moving[c] && (moved > 0.08f * dm);
    }
    float v0 = vn[0], v1 = vn[1], v2 = vn[2];
    if (last_col >= 0) {
      const float* lv = velos + ((int64_t)c * F + last_col) * 3;
      float dev = norm3(v0 - lv[0], v1 - lv[1], v2 - lv[2]);
      stop = stop || (dev > 0.24f * dm);
      const float* pv = velos + ((int64_t)c * F + (next - dir)) * 3;
      float nrm = norm3(v0, v1, v2) * norm3(pv[0], pv[1], pv[2]);
      nrm = fmaxf(nrm, 1e-6f);
      float cosv = ((v0 * pv[0] + v1 * pv[1]) + v2 * pv[2]) / nrm;
      cosv = fminf(fmaxf(cosv, -1.f), 1.f);
      float angle = acosf(cosv) / 3.14159265358979323846f * 180.0f;
      float nxy = sqrtf(v0 * v0 + v1 * v1);
      stop = stop || ((angle > angle_thr) && (nxy > 0.01f));
    }
    if (next == frame_id - 1) {
      float* vf = velos + ((int64_t)c * F + frame_id) * 3;
      vf[0] = v0; vf[1] = v1; vf[2] = v2;
    }
    stopped[c] = stop;
    if (!stop) {
      if (dir == -1) cmin[c] = next; else cmax[c] = next;
      st->active = 1;
    }
    reg_errors[(int64_t)c * F + next] = (float)l1[c];
    edge_ratios[(int64_t)c * F + next] = ratio[c];
  }
}

__global__ void __launch_bounds__(kBlock)
trk_set_frame_kernel(float4* __restrict__ xyzf, int n, float f) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) xyzf[i].x = f;
}

// harvest flags: target-frame point q is kept when its nearest moved anchor point belongs to a
// component that is still tracked (cluster_tracking.py:702-741)
__global__ void __launch_bounds__(kBlock)
trk_harvest_flag_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ deg, int m,
                        const int32_t* __restrict__ comp, const uint8_t* __restrict__ stopped,
                        int32_t* __restrict__ flag) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) return;
  int f = 0;
  if (deg[q] > 0) f = !stopped[comp[nbr[q]]];
  flag[q] = f;
}

// anchor's own points: kept when their component passed filter_components
__global__ void __launch_bounds__(kBlock)
trk_anchor_flag_kernel(const int32_t* __restrict__ comp, const uint8_t* __restrict__ vcm, int n,
                       int32_t* __restrict__ flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = vcm[comp[i]] ? 1 : 0;
}

// append the kept rows: (index into the sequence, index within the frame, component)
__global__ void __launch_bounds__(kBlock)
trk_harvest_emit_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ pos, int m,
                        const int32_t* __restrict__ nbr, const int32_t* __restrict__ comp,
                        int64_t frame_off, int64_t cap, int64_t* __restrict__ ex_index,
                        int64_t* __restrict__ ex_local, int64_t* __restrict__ ex_comp,
                        const int64_t* __restrict__ d_base, int64_t* __restrict__ d_next,
                        TrkState* st) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) return;
  int64_t base = *d_base;
  if (flag[q]) {
    int64_t o = base + pos[q];
    if (o < cap) {
      ex_index[o] = frame_off + q;
      ex_local[o] = q;
      ex_comp[o] = comp[nbr ? nbr[q] : q];
    } else {
      st->err = 2;
    }
  }
  if (q == m - 1) *d_next = base + pos[q] + flag[q];
}

__global__ void trk_copy_count_kernel(const int64_t* src, int64_t* dst) { *dst = *src; }

// the state words the host is about to read, mirrored into pinned host memory (the host then needs
// the stream synchronisation only)
__global__ void trk_publish_kernel(const TrkState* __restrict__ st, TrkState* __restrict__ out) { *out = *st; }

int radius_nn1_view(int64_t N, int64_t M, void* ws, size_t ws_bytes, const int32_t** nbr_idx,
                    const int32_t** deg, const int32_t** err);
int smooth_velo_run(float* comp_velos, const float* comp_center_diffs, int64_t C, int F, int frame_lo,
                    int frame_hi, float weight0, float weight, int num_itr, float stopping, void* ws,
                    size_t ws_bytes, int* err, hipStream_t s);

struct TrkPlan {
  TrkState* st;
  int64_t* exc;  // [2] ping-pong extraction counters
  // anchor
  float4 *xyzf, *last, *pv;
  int64_t* comp64;
  int32_t* comp32;
  float *stat_a, *dist;
  int32_t *order_c, *seg_c;
  int64_t *uniq_c, *d_numc;
  // per component
  float *deg_f, *diam, *ctr, *centers, *diffs, *velo_pre, *ratio, *ratio_tmp, *red4a, *red4b, *red1;
  int64_t* deg_i;
  unsigned long long* sub_deg;
  uint8_t* stopped;
  double *Tl, *l1, *l1_tmp;
  // sampling (a = anchor, n = target frame)
  float *b8;
  int32_t* seg_s;
  int64_t* d_nums;
  float4 *mov, *sub_n, *ref_ns, *sub2;
  int32_t *sub_comp, *mov_comp, *flag, *pos;
  float* stat_n;
  void *ws_unique, *ws_icp, *ws_rg, *ws_sv, *tmp_scan;
  size_t unique_bytes, icp_bytes, rg_bytes, sv_bytes, scan_bytes;
  size_t comp_zero_begin, comp_zero_end;
};

static bool plan_track(Arena& a, int64_t na, int64_t nm, int64_t C, int F, TrkPlan& p) {
  if (na < 1) na = 1;
  if (nm < 1) nm = 1;
  p.st = a.take<TrkState>(1);
  p.exc = a.take<int64_t>(2);
  p.xyzf = a.take<float4>(na + 1);  // + 1: the d = 4 reductions read one float past the last row
  p.last = a.take<float4>(na);
  p.pv = a.take<float4>(na);
  p.comp64 = a.take<int64_t>(na);
  p.comp32 = a.take<int32_t>(na);
  p.stat_a = a.take<float>(na);
  p.dist = a.take<float>(na);
  p.order_c = a.take<int32_t>(na);
  p.seg_c = a.take<int32_t>(na + 1);
  p.uniq_c = a.take<int64_t>(na);
  p.d_numc = a.take<int64_t>(1);
  p.comp_zero_begin = a.off;
  p.deg_f = a.take<float>(C);
  p.diam = a.take<float>(C);
  p.ctr = a.take<float>(C * 4);
  p.centers = a.take<float>(C * (int64_t)F * 3);
  p.diffs = a.take<float>(C * (int64_t)F * 3);
  p.velo_pre = a.take<float>(C * 3);
  p.ratio = a.take<float>(C);
  p.ratio_tmp = a.take<float>(C);
  p.deg_i = a.take<int64_t>(C);
  p.sub_deg = a.take<unsigned long long>(C);
  p.stopped = a.take<uint8_t>(C);
  p.l1 = a.take<double>(C);
  p.l1_tmp = a.take<double>(C);
  p.comp_zero_end = a.off;
  p.Tl = a.take<double>(C * 16);
  p.red4a = a.take<float>(na * 4);
  p.red4b = a.take<float>(na * 4);
  p.red1 = a.take<float>(na);
  p.b8 = a.take<float>(16);
  p.seg_s = a.take<int32_t>(na + nm + 1);
  p.d_nums = a.take<int64_t>(2);
  p.sub2 = a.take<float4>(na + nm);
  p.mov = a.take<float4>(na);
  p.sub_n = a.take<float4>(nm);
  p.ref_ns = a.take<float4>(nm);
  p.sub_comp = a.take<int32_t>(na + nm);
  p.mov_comp = a.take<int32_t>(na);
  p.flag = a.take<int32_t>(na + nm);
  p.pos = a.take<int32_t>(na + nm);
  p.stat_n = a.take<float>(nm);
  p.scan_bytes = scan_i32_temp_bytes(na + nm);
  p.tmp_scan = a.take<char>(p.scan_bytes);
  p.unique_bytes = pcs_unique_i64_workspace_bytes(na + nm);
  p.ws_unique = a.take<char>(p.unique_bytes);
  p.icp_bytes = pcs_register_to_next_frame_workspace_bytes(na, nm, nm, C);
  p.ws_icp = a.take<char>(p.icp_bytes);
  p.rg_bytes = pcs_radius_graph_workspace_bytes(na, nm, 1);
  p.ws_rg = a.take<char>(p.rg_bytes);
  p.sv_bytes = pcs_smooth_velo_workspace_bytes(C, F);
  p.ws_sv = a.take<char>(p.sv_bytes);
  return a.ok();
}

}  // namespace pcs

using namespace pcs;

#define TRK_RC(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc) return _rc;        \
  } while (0)

static inline int nblk(int64_t n) { return (int)((n < 1 ? 1 : n) + kBlock - 1) / kBlock; }

extern "C" {

size_t pcs_track_frame_workspace_bytes(int64_t n_anchor, int64_t n_frame_max, int64_t C,
                                       int num_frames) {
  Arena a(nullptr, 0);
  TrkPlan p;
  plan_track(a, n_anchor, n_frame_max, C < 1 ? 1 : C, num_frames < 1 ? 1 : num_frames, p);
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
  if (!prm || !seq_fxyz || !seq_component || !seq_stationary || !frame_offsets || !ex_index ||
      !ex_local || !ex_component || !transforms || !reg_errors || !comp_edge_ratios || !comp_velos ||
      !comp_min_frame_id || !comp_max_frame_id || !moving || !valid_comp || !info || !ws)
    return PCS_ERR_ARG;
  if (C < 1 || num_frames < 1 || anchor_frame < 0 || anchor_frame >= num_frames ||
      min_frame_id < 0 || max_frame_id >= num_frames || min_frame_id > anchor_frame ||
      max_frame_id < anchor_frame || prm->num_levels < 1 || prm->num_levels > kMaxLevels)
    return PCS_ERR_ARG;
  const int F = max_frame_id + 1;
  const int W = max_frame_id - min_frame_id + 1;
  const int L = prm->num_levels;
  const int64_t off_a = frame_offsets[anchor_frame];
  const int64_t na = frame_offsets[anchor_frame + 1] - off_a;
  if (na < 1) return PCS_ERR_ARG;
  int64_t nm = 1;
  for (int f = min_frame_id; f <= max_frame_id; ++f) {
    int64_t c = frame_offsets[f + 1] - frame_offsets[f];
    if (c < 0) return PCS_ERR_ARG;
    if (f != anchor_frame && c > nm) nm = c;
  }
  if (na >= (1LL << 30) || nm >= (1LL << 30) || C >= (1LL << 30)) return PCS_ERR_RANGE;
  Arena a(ws, ws_bytes);
  TrkPlan p;
  if (!plan_track(a, na, nm, C, F, p)) return PCS_ERR_WORKSPACE;
  // optional: the ICP iterations of the concurrent tracking units share launches for the duration
  // of the call (prm->share_launches; see IcpBatcher in pcs_icp.hip for when that pays)
  struct ShareGuard {
    bool on;
    explicit ShareGuard(int group) : on(group != 0) { if (on) pcs_icp_share_launches(group); }
    ~ShareGuard() { if (on) pcs_icp_share_launches(0); }
  } share_guard(prm->share_launches);

  const float4* anchor = (const float4*)seq_fxyz + off_a;
  const int gA = nblk(na), gC = nblk(C);
  const int gAC = nblk(na > C ? na : C);
  info[0] = info[1] = info[2] = info[3] = 0;

  // ---- set-up: components of the anchor frame, their size, centre, diameter and validity
  PCS_CHECK_HIP(hipMemsetAsync(p.st, 0, sizeof(TrkState), s));
  PCS_CHECK_HIP(hipMemsetAsync(p.exc, 0, 2 * sizeof(int64_t), s));
  PCS_CHECK_HIP(hipMemsetAsync((char*)ws + p.comp_zero_begin, 0, p.comp_zero_end - p.comp_zero_begin, s));
  PCS_CHECK_HIP(hipMemsetAsync(p.xyzf + na, 0, sizeof(float4), s));
  PCS_CHECK_HIP(hipMemsetAsync(comp_velos, 0, (size_t)C * F * 3 * sizeof(float), s));
  PCS_CHECK_HIP(hipMemsetAsync(reg_errors, 0, (size_t)C * F * sizeof(float), s));
  PCS_CHECK_HIP(hipMemsetAsync(comp_edge_ratios, 0, (size_t)C * F * sizeof(float), s));
  trk_load_anchor_kernel<<<gA, kBlock, 0, s>>>(anchor, seq_component + off_a, seq_stationary + off_a,
                                               comp_base, (int)na, p.xyzf, p.comp64, p.comp32, p.stat_a);
  TRK_RC(pcs_unique_i64(p.comp64, na, 64, p.uniq_c, nullptr, p.order_c, p.seg_c, p.d_numc,
                        p.ws_unique, p.unique_bytes, s));
  int64_t Cu64 = 0;
  PCS_CHECK_HIP(hipMemcpyAsync(&Cu64, p.d_numc, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  PCS_CHECK_HIP(hipStreamSynchronize(s));
  const int Cu = (int)Cu64;
  if (Cu < 1 || Cu > C) return PCS_ERR_ARG;
  const int gU = nblk(Cu);
  const float* xyz1 = (const float*)p.xyzf + 1;  // rows (x, y, z, next row's frame) with stride 4
  TRK_RC(pcs_segment_reduce_f32(xyz1, 4, p.order_c, p.seg_c, Cu, 2, p.red4a, s));
  trk_comp_center_kernel<<<gU, kBlock, 0, s>>>(p.uniq_c, p.seg_c, Cu, p.red4a, p.deg_f, p.deg_i, p.ctr);
  trk_point_dist_kernel<<<gA, kBlock, 0, s>>>(p.xyzf, p.comp32, (int)na, p.ctr, p.dist);
  TRK_RC(pcs_segment_minmax_f32(p.dist, 1, p.order_c, p.seg_c, Cu, 1, p.red1, s));
  trk_comp_diam_kernel<<<gU, kBlock, 0, s>>>(p.uniq_c, Cu, p.red1, p.diam);
  trk_comp_init_kernel<<<gC, kBlock, 0, s>>>((int)C, F, W, anchor_frame, prm->max_diameter, p.deg_f,
                                             p.diam, p.ctr, valid_comp, transforms,
                                             comp_min_frame_id, comp_max_frame_id, p.centers);
  // the anchor's own points
  int ex_par = 0;
  trk_anchor_flag_kernel<<<gA, kBlock, 0, s>>>(p.comp32, valid_comp, (int)na, p.flag);
  TRK_RC(exclusive_scan_i32(p.tmp_scan, p.scan_bytes, p.flag, p.pos, na, s));
  trk_harvest_emit_kernel<<<gA, kBlock, 0, s>>>(p.flag, p.pos, (int)na, nullptr, p.comp32, off_a,
                                                ex_cap, ex_index, ex_local, ex_component,
                                                p.exc + ex_par, p.exc + (ex_par ^ 1), p.st);
  ex_par ^= 1;

  const float nn_radius = prm->nn_radius;
  const double nn_r = (double)nn_radius;
  const float nn_cell = (float)(sqrt(nn_r * nn_r + 3e-5 * nn_r * nn_r + 3e-6) / 4 * 1.0005);
  int64_t steps = 0, reg_calls = 0;
  TrkState host;
  // pinned page: bytes [0, 256) are the ICP's, the tracker's mirror sits behind them
  char* page = (char*)thread_pinned_page();
  TrkState* pin = page ? (TrkState*)(page + 1024) : nullptr;
  auto read_state = [&]() -> int {
    if (pin) {
      trk_publish_kernel<<<1, 1, 0, s>>>(p.st, pin);
    } else if (hipMemcpyAsync(&host, p.st, sizeof(TrkState), hipMemcpyDeviceToHost, s) != hipSuccess) {
      return PCS_ERR_HIP;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return PCS_ERR_HIP;
    if (pin) host = *pin;
    return PCS_OK;
  };

  for (int dir = -1; dir <= 1; dir += 2) {
    int next = anchor_frame + dir;
    int last_col = (dir == 1 && anchor_frame > 0) ? anchor_frame : -1;
    PCS_CHECK_HIP(hipMemsetAsync(&p.st->active, 0, sizeof(int32_t), s));
    trk_dir_init_kernel<<<gAC, kBlock, 0, s>>>((int)C, (int)na, valid_comp, p.stopped, moving, anchor,
                                               p.xyzf, p.last, p.st);
    while (next >= min_frame_id && next <= max_frame_id) {
      TRK_RC(read_state());
      if (host.err || host.counts[0] < 0) return PCS_ERR_RANGE;
      if (!host.active) break;
      PCS_CHECK_HIP(hipMemsetAsync(&p.st->active, 0, sizeof(int32_t), s));
      ++steps;
      const int64_t off_n = frame_offsets[next];
      const int64_t nn = frame_offsets[next + 1] - off_n;
      const float4* nxt = (const float4*)seq_fxyz + off_n;
      const int col = next - min_frame_id;
      const int gN = nblk(nn);
      trk_step_begin_kernel<<<gAC, kBlock, 0, s>>>((int)C, (int)na, F, W, col, dir, last_col, p.stopped,
                                                   comp_velos, p.comp32, p.xyzf, transforms);
      if (nn > 0)
        trk_stat_to_float_kernel<<<gN, kBlock, 0, s>>>(seq_stationary + off_n, (int)nn, p.stat_n);
      for (int lv = 0; lv < L; ++lv) {
        float size4[4] = {1.f, prm->voxel_size[lv][0], prm->voxel_size[lv][1], prm->voxel_size[lv][2]};
        // ---- sample_frame(anchor) and sample_frame(target) in one pipeline: combined keys, one
        //      sort, one segment pass; then the non-stationary subsets and the per-component counts
        {
          const int64_t n2 = na + nn;
          const int g2 = nblk(n2);
          uint64_t* keys = nullptr;
          int32_t* vals = nullptr;
          TRK_RC(unique_prepared_ptrs(p.ws_unique, p.unique_bytes, n2, &keys, &vals));
          trk_bounds_reset_kernel<<<1, 64, 0, s>>>(p.b8);
          {
            int64_t nmax2 = na > nn ? na : nn;
            int gb = nblk(nmax2);
            if (gb > 128) gb = 128;
            trk_bounds2_kernel<<<dim3(gb, 2), kBlock, 0, s>>>(p.xyzf, (int)na, nxt, (int)nn, p.b8);
          }
          trk_keys2_kernel<<<g2, kBlock, 0, s>>>(p.xyzf, (int)na, nxt, (int)nn, p.b8,
                                                 make_float4(size4[0], size4[1], size4[2], size4[3]), 0.5f,
                                                 keys, vals);
          const int32_t* order2 = nullptr;
          TRK_RC(unique_prepared_run(p.ws_unique, p.unique_bytes, n2, 64, p.seg_s, p.d_nums, &order2, s));
          PCS_CHECK_HIP(hipMemsetAsync(p.sub_deg, 0, (size_t)C * sizeof(unsigned long long), s));
          trk_sample2_emit_kernel<<<g2, kBlock, 0, s>>>(p.xyzf, p.stat_a, p.comp64, (int)na, nxt, p.stat_n,
                                                        order2, p.seg_s, p.d_nums, (int)n2, p.sub2,
                                                        p.sub_comp, p.flag, p.sub_deg, p.d_nums + 1);
          TRK_RC(exclusive_scan_i32(p.tmp_scan, p.scan_bytes, p.flag, p.pos, n2, s));
          trk_compact2_kernel<<<g2, kBlock, 0, s>>>(p.sub2, p.sub_comp, p.flag, p.pos, (int)n2, p.d_nums,
                                                    p.d_nums + 1, p.mov, p.mov_comp, p.sub_n, p.ref_ns,
                                                    &p.st->counts[0]);
        }
        TRK_RC(read_state());
        const int64_t Mn = host.counts[1], Ra = host.counts[2], Rn = host.counts[3];
        // ---- register_to_next_frame
        int32_t icp_info[4];
        double* l1_out = (lv == L - 1) ? p.l1 : p.l1_tmp;
        float* ratio_out = (lv == 0) ? p.ratio : p.ratio_tmp;
        TRK_RC(pcs_register_to_next_frame((float*)p.mov, p.mov_comp, Mn, (const float*)p.ref_ns, Rn,
                                          (const float*)p.sub_n, Ra, (const int64_t*)p.sub_deg, C,
                                          prm->radius[lv], next - anchor_frame, prm->angle_regularizer,
                                          prm->max_iter, prm->stopping_delta[lv], 0.f, 8, p.Tl, l1_out,
                                          ratio_out, icp_info, p.ws_icp, p.icp_bytes, s));
        ++reg_calls;
        trk_apply_T_kernel<<<gAC, kBlock, 0, s>>>((int)C, (int)na, W, col, p.Tl, p.comp32, p.xyzf,
                                                  transforms);
      }
      // ---- component motion, velocity smoothing, stop rules
      TRK_RC(pcs_segment_reduce_f32(xyz1, 4, p.order_c, p.seg_c, Cu, 2, p.red4a, s));
      trk_point_velo_kernel<<<gA, kBlock, 0, s>>>(p.xyzf, p.last, (int)na, dir, p.pv);
      TRK_RC(pcs_segment_reduce_f32((const float*)p.pv, 4, p.order_c, p.seg_c, Cu, 3, p.red4b, s));
      trk_comp_motion_kernel<<<gU, kBlock, 0, s>>>(p.uniq_c, p.seg_c, Cu, F, next, dir, p.red4a,
                                                   p.red4b, p.centers, comp_velos, p.diffs, p.velo_pre);
      {
        int a0 = anchor_frame + dir, b0 = next;
        if (a0 != b0) {
          int lo = a0 < b0 ? a0 : b0, hi = a0 < b0 ? b0 : a0;
          TRK_RC(smooth_velo_run(comp_velos, p.diffs, C, F, lo, hi, prm->smooth_weight0,
                                 prm->smooth_weight, prm->smooth_num_itr, prm->smooth_stopping,
                                 p.ws_sv, p.sv_bytes, &p.st->err, s));
        }
      }
      trk_after_smooth_kernel<<<gAC, kBlock, 0, s>>>(
          (int)C, (int)na, F, W, col, anchor_frame, next, dir, last_col, prm->min_move_frame,
          prm->reg_error_coeff, prm->angle_threshold, p.comp32, p.xyzf, p.last, comp_velos,
          p.velo_pre, p.centers, p.diam, p.deg_i, p.l1, p.ratio, p.stopped, moving, transforms,
          comp_min_frame_id, comp_max_frame_id, reg_errors, comp_edge_ratios, p.st);
      last_col = next;
      // ---- harvest the target frame's points next to the moved anchor points
      if (nn > 0) {
        trk_set_frame_kernel<<<gA, kBlock, 0, s>>>(p.xyzf, (int)na, (float)next);
        int64_t* d_edges = &p.st->counts[0];  // scratch word, rewritten by the next level-0 sample
        TRK_RC(pcs_radius_graph_search((const float*)p.xyzf, na, (const float*)nxt, nn, nn_radius,
                                       nn_cell, 0, 0, 1, 1, d_edges, p.ws_rg, p.rg_bytes, s));
        const int32_t *nbr = nullptr, *deg = nullptr, *rerr = nullptr;
        TRK_RC(radius_nn1_view(na, nn, p.ws_rg, p.rg_bytes, &nbr, &deg, &rerr));
        trk_harvest_flag_kernel<<<gN, kBlock, 0, s>>>(nbr, deg, (int)nn, p.comp32, p.stopped, p.flag);
        TRK_RC(exclusive_scan_i32(p.tmp_scan, p.scan_bytes, p.flag, p.pos, nn, s));
        trk_harvest_emit_kernel<<<gN, kBlock, 0, s>>>(p.flag, p.pos, (int)nn, nbr, p.comp32, off_n,
                                                      ex_cap, ex_index, ex_local, ex_component,
                                                      p.exc + ex_par, p.exc + (ex_par ^ 1), p.st);
        ex_par ^= 1;
        trk_set_frame_kernel<<<gA, kBlock, 0, s>>>(p.xyzf, (int)na, (float)anchor_frame);
        // a fine-cell overflow of the 1-NN grid leaves a negative edge count in counts[0]: seen by
        // the next read of the state (top of the loop / end of the call)
      }
      next += dir;
    }
  }
  trk_copy_count_kernel<<<1, 1, 0, s>>>(p.exc + ex_par, &p.st->ex_count);
  PCS_LAUNCH_CHECK();
  TRK_RC(read_state());
  if (host.err || host.counts[0] < 0) return PCS_ERR_RANGE;
  info[0] = host.ex_count;
  info[1] = reg_calls;
  info[2] = steps;
  {
    double t2[2] = {0, 0};
    pcs_profile_thread_icp_read(t2);
    info[3] = (int64_t)(t2[0] * 1e6) * 1000000LL + (int64_t)(t2[1] * 1e6);  // us in launcher | us in ICP
  }
  return PCS_OK;
}

}  // extern "C"
