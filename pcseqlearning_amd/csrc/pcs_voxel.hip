// Voxel keys, sorted-unique with inverse, ordered segment means, dynamic and hard voxelizers.
//
// Reference behaviour restated (not copied) from:
//   pcdet/models/model_utils/grid_sampling.py:22-46      (GridSampling3D: grid_cluster -> unique -> mean)
//   pcdet/models/registration/simple_reg.py:119-130      (subsample: max original index per cell)
//   pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:51-76 (dynamic voxelization)
//   pcdet/ops/spconv/include/spconv/point2voxel.h:33-92  (hard voxelizer, first-come voxel ids)
//
// Layout: points are row-major [n, 4] f32 (frame|batch, x, y, z) so one lane reads one 16-B row
// (64 lanes = 1 KiB per wave-instruction, fully coalesced). All per-voxel float sums run in
// ascending original-point order (stable key sort), so results are run-to-run deterministic and
// bit-identical to the CPU restatement in oracle/.
#include "pcs_common.hpp"

namespace pcs {

// ---------------------------------------------------------------- min / max of a [n,4] array
__global__ void minmax4_init_kernel(float* out8) {
  int t = threadIdx.x;
  if (t < 4) out8[t] = __builtin_inff();
  else if (t < 8) out8[t] = -__builtin_inff();
}

__global__ void __launch_bounds__(kBlock) minmax4_kernel(const float4* __restrict__ pts, int64_t n,
                                                         float* out8) {
  float mn[4] = {__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()};
  float mx[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
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

// ---------------------------------------------------------------- grid_cluster ids
// id = sum_d trunc((p_d - start_d) / size_d) * stride_d, stride_0 = 1,
// stride_{d+1} = stride_d * (trunc((end_d - start_d) / size_d) + 1); fp32 subtract + IEEE divide.
__global__ void __launch_bounds__(kBlock)
grid_ids_kernel(const float4* __restrict__ pts, int64_t n, const float* __restrict__ bounds8,
                float4 size, float margin0, int64_t* __restrict__ ids) {
  float st[4], en[4], sz[4] = {size.x, size.y, size.z, size.w};
#pragma unroll
  for (int d = 0; d < 4; ++d) { st[d] = bounds8[d]; en[d] = bounds8[4 + d]; }
  st[0] = st[0] - margin0;
  en[0] = en[0] + margin0;
  int64_t stride[4];
  stride[0] = 1;
#pragma unroll
  for (int d = 0; d < 3; ++d) stride[d + 1] = stride[d] * ((int64_t)((en[d] - st[d]) / sz[d]) + 1);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 p = pts[i];
    int64_t id = (int64_t)((p.x - st[0]) / sz[0]) * stride[0] +
                 (int64_t)((p.y - st[1]) / sz[1]) * stride[1] +
                 (int64_t)((p.z - st[2]) / sz[2]) * stride[2] +
                 (int64_t)((p.w - st[3]) / sz[3]) * stride[3];
    ids[i] = id;
  }
}

// ---------------------------------------------------------------- sorted unique with inverse
__global__ void __launch_bounds__(kBlock)
iota_keys_kernel(const int64_t* __restrict__ ids, int64_t n, uint64_t* __restrict__ keys,
                 int32_t* __restrict__ vals) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    keys[i] = (uint64_t)ids[i];
    vals[i] = (int32_t)i;
  }
}

// head[i] = 1 where a new key run starts (keys equal to `sentinel` are excluded: head 0).
__global__ void __launch_bounds__(kBlock)
head_flags_kernel(const uint64_t* __restrict__ skeys, int64_t n, uint64_t sentinel,
                  int32_t* __restrict__ head) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t k = skeys[i];
    int h = (k != sentinel) && (i == 0 || skeys[i - 1] != k);
    head[i] = h;
  }
}

// rank_incl = inclusive scan of head. Emits uniq, inv, seg_start and the voxel count.
__global__ void __launch_bounds__(kBlock)
unique_emit_kernel(const uint64_t* __restrict__ skeys, const int32_t* __restrict__ sidx,
                   const int32_t* __restrict__ head, const int32_t* __restrict__ rank_incl,
                   int64_t n, uint64_t sentinel, int64_t* __restrict__ uniq,
                   int64_t* __restrict__ inv, int32_t* __restrict__ seg_start,
                   int64_t* __restrict__ d_num) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t k = skeys[i];
    int r = rank_incl[i] - 1;
    if (k != sentinel) {
      if (inv) inv[sidx[i]] = r;
      if (head[i]) {
        if (uniq) uniq[r] = (int64_t)k;
        seg_start[r] = (int32_t)i;
      }
    } else if (inv) {
      inv[sidx[i]] = -1;
    }
    if (i == n - 1) {
      int v = rank_incl[i];
      // number of valid (non-sentinel) rows = first sentinel position
      *d_num = v;
    }
    // close the last segment: position of the first sentinel (or n)
    if (k != sentinel && (i == n - 1 || skeys[i + 1] == sentinel)) seg_start[rank_incl[i]] = (int32_t)(i + 1);
  }
}

// ---------------------------------------------------------------- ordered segment mean / sum
// One thread per (segment, column): sequential fp32 adds in ascending sorted position.
__global__ void __launch_bounds__(kBlock)
segment_mean_kernel(const float* __restrict__ data, int d, const int32_t* __restrict__ order,
                    const int32_t* __restrict__ seg_start, int64_t v, float* __restrict__ out,
                    int mean) {
  int64_t total = v * d;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t seg = t / d;
    int c = (int)(t - seg * d);
    int b = seg_start[seg], e = seg_start[seg + 1];
    float acc = 0.f;
    for (int p = b; p < e; ++p) acc = acc + data[(int64_t)order[p] * d + c];
    if (mean) acc = acc / (float)(e - b);
    out[t] = acc;
  }
}

// Large segments (pillars, components): one wave per segment, lane l sums elements l, l+64, ...
// in order, then a 64-lane butterfly. Deterministic (fixed order), d <= 16 columns.
__global__ void __launch_bounds__(kBlock)
segment_reduce_wave_kernel(const float* __restrict__ data, int d, const int32_t* __restrict__ order,
                           const int32_t* __restrict__ seg_start, int64_t v,
                           float* __restrict__ out, int mean) {
  int lane = threadIdx.x & 63;
  int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t seg = wave; seg < v; seg += nwaves) {
    int b = seg_start[seg], e = seg_start[seg + 1];
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    for (int p = b + lane; p < e; p += 64) {
      const float* row = data + (int64_t)order[p] * d;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < d) acc[c] = acc[c] + row[c];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c < d) {
        float x = acc[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x = x + __shfl_xor(x, o, 64);
        if (lane == 0) out[seg * d + c] = mean ? x / (float)(e - b) : x;
      }
    }
  }
}

// Very large segments (8 m super-pillars of the ground fit: thousands of rows each, few hundred
// segments): one 256-thread block per segment, thread-strided partial sums, wave butterflies, then
// the 4 wave results added in wave order. Fixed order -> reproducible. d <= 16.
__global__ void __launch_bounds__(kBlock)
segment_reduce_block_kernel(const float* __restrict__ data, int d, const int32_t* __restrict__ order,
                            const int32_t* __restrict__ seg_start, int64_t v,
                            float* __restrict__ out, int mean) {
  __shared__ float part[kBlock / 64][16];
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t seg = blockIdx.x; seg < v; seg += gridDim.x) {
    int b = seg_start[seg], e = seg_start[seg + 1];
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    for (int p = b + (int)threadIdx.x; p < e; p += kBlock) {
      const float* row = data + (int64_t)order[p] * d;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < d) acc[c] = acc[c] + row[c];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c < d) {
        float x = acc[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x = x + __shfl_xor(x, o, 64);
        if (lane == 0) part[w][c] = x;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < d) {
      float x = part[0][threadIdx.x];
      for (int q = 1; q < kBlock / 64; ++q) x = x + part[q][threadIdx.x];
      out[seg * d + threadIdx.x] = mean ? x / (float)(e - b) : x;
    }
    __syncthreads();
  }
}

// last (= max original index) / first (= min original index) row of every segment
__global__ void __launch_bounds__(kBlock)
segment_pick_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ seg_start,
                    int64_t v, int pick_last, int64_t* __restrict__ out) {
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < v;
       s += (int64_t)gridDim.x * blockDim.x) {
    out[s] = pick_last ? order[seg_start[s + 1] - 1] : order[seg_start[s]];
  }
}

// ---------------------------------------------------------------- dynamic voxelization keys
// c = floor((p - lo) / vs) per axis; in-range mask; key = b*XYZ + cx*YZ + cy*Z + cz
// (x-major, as dynamic_mean_vfe.py:57-60). Out-of-range points get the sentinel key.
__global__ void __launch_bounds__(kBlock)
dyn_voxel_keys_kernel(const float* __restrict__ pts, int64_t n, int ncol, float lox, float loy,
                      float loz, float vx, float vy, float vz, int gx, int gy, int gz,
                      uint64_t* __restrict__ keys, int32_t* __restrict__ vals) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = pts + i * ncol;
    int b = (int)p[0];
    int cx = (int)floorf((p[1] - lox) / vx);
    int cy = (int)floorf((p[2] - loy) / vy);
    int cz = (int)floorf((p[3] - loz) / vz);
    bool ok = cx >= 0 && cx < gx && cy >= 0 && cy < gy && cz >= 0 && cz < gz;
    uint64_t key = ~0ULL;
    if (ok) key = (uint64_t)b * ((uint64_t)gx * gy * gz) + (uint64_t)cx * ((uint64_t)gy * gz) +
                  (uint64_t)cy * (uint64_t)gz + (uint64_t)cz;
    keys[i] = key;
    vals[i] = (int32_t)i;
  }
}

// decode sorted-unique keys into (b, z, y, x) int32 rows; the voxel count stays on the device
__global__ void __launch_bounds__(kBlock)
dyn_voxel_coords_kernel(const uint64_t* __restrict__ skeys, const int32_t* __restrict__ seg_start,
                        const int64_t* __restrict__ d_num, int gx, int gy, int gz,
                        int32_t* __restrict__ coords) {
  int64_t v = *d_num;
  uint64_t xyz = (uint64_t)gx * gy * gz, yz = (uint64_t)gy * gz;
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < v;
       s += (int64_t)gridDim.x * blockDim.x) {
    uint64_t k = skeys[seg_start[s]];
    int b = (int)(k / xyz);
    uint64_t r = k % xyz;
    int cx = (int)(r / yz);
    r = r % yz;
    int cy = (int)(r / gz);
    int cz = (int)(r % gz);
    ((int4*)coords)[s] = make_int4(b, cz, cy, cx);
  }
}

// mean of columns 1..ncol-1 per voxel, ascending original index order
__global__ void __launch_bounds__(kBlock)
dyn_voxel_mean_kernel(const float* __restrict__ pts, int ncol, const int32_t* __restrict__ order,
                      const int32_t* __restrict__ seg_start, const int64_t* __restrict__ d_num,
                      float* __restrict__ feat) {
  int d = ncol - 1;
  int64_t total = (*d_num) * d;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t seg = t / d;
    int c = (int)(t - seg * d);
    int b = seg_start[seg], e = seg_start[seg + 1];
    float acc = 0.f;
    for (int p = b; p < e; ++p) acc = acc + pts[(int64_t)order[p] * ncol + 1 + c];
    feat[t] = acc / (float)(e - b);
  }
}

// ---------------------------------------------------------------- hard voxelizer
// key = flat (z, y, x) cell of floor((p - lo)/vs); rejected points get the sentinel.
__global__ void __launch_bounds__(kBlock)
hard_voxel_keys_kernel(const float* __restrict__ pts, int64_t n, int ncol, float lox, float loy,
                       float loz, float vx, float vy, float vz, int gx, int gy, int gz,
                       uint64_t* __restrict__ keys, int32_t* __restrict__ vals) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = pts + i * ncol;
    int cx = (int)floorf((p[0] - lox) / vx);
    int cy = (int)floorf((p[1] - loy) / vy);
    int cz = (int)floorf((p[2] - loz) / vz);
    bool ok = cx >= 0 && cx < gx && cy >= 0 && cy < gy && cz >= 0 && cz < gz;
    uint64_t key = ~0ULL;
    if (ok) key = ((uint64_t)cz * gy + cy) * gx + cx;
    keys[i] = key;
    vals[i] = (int32_t)i;
  }
}

// per sorted-unique cell: key for the first-come ordering = smallest original index in the cell
__global__ void __launch_bounds__(kBlock)
hard_voxel_first_kernel(const int32_t* __restrict__ sidx, const int32_t* __restrict__ seg_start,
                        const int64_t* __restrict__ d_cells, int64_t n,
                        uint32_t* __restrict__ first, int32_t* __restrict__ cell) {
  int64_t v = *d_cells;
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n;
       s += (int64_t)gridDim.x * blockDim.x) {
    first[s] = s < v ? (uint32_t)sidx[seg_start[s]] : 0xFFFFFFFFu;  // padding sorts last
    cell[s] = (int32_t)s;
  }
}

// voxel r (first-come rank) = cell scell[r]; copy up to max_points rows, coords (z,y,x), counts
__global__ void __launch_bounds__(kBlock)
hard_voxel_emit_kernel(const float* __restrict__ pts, int ncol, const uint64_t* __restrict__ skeys,
                       const int32_t* __restrict__ sidx, const int32_t* __restrict__ seg_start,
                       const int32_t* __restrict__ scell, const int64_t* __restrict__ d_nvox,
                       int max_points, int gx, int gy, float* __restrict__ voxels,
                       int32_t* __restrict__ coords, int32_t* __restrict__ num) {
  int64_t nvox = *d_nvox;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nvox;
       r += (int64_t)gridDim.x * blockDim.x) {
    int c = scell[r];
    int b = seg_start[c], e = seg_start[c + 1];
    uint64_t k = skeys[b];
    int cx = (int)(k % gx);
    int cy = (int)((k / gx) % gy);
    int cz = (int)(k / ((uint64_t)gx * gy));
    coords[r * 3 + 0] = cz;
    coords[r * 3 + 1] = cy;
    coords[r * 3 + 2] = cx;
    int cnt = e - b;
    if (cnt > max_points) cnt = max_points;
    num[r] = cnt;
    for (int j = 0; j < cnt; ++j) {
      const float* src = pts + (int64_t)sidx[b + j] * ncol;
      float* dst = voxels + ((int64_t)r * max_points + j) * ncol;
      for (int q = 0; q < ncol; ++q) dst[q] = src[q];
    }
  }
}

__global__ void clamp_count_kernel(const int64_t* d_in, int64_t cap, int64_t* d_out) {
  int64_t v = *d_in;
  *d_out = v < cap ? v : cap;
}

// shared sort->heads->scan->emit pipeline over (keys, vals) already in workspace
struct UniquePlan {
  uint64_t *k0, *k1;
  int32_t *v0, *v1, *head, *rank;
  void* tmp;
  size_t tmp_bytes;
};

static bool plan_unique(Arena& a, int64_t n, UniquePlan& p) {
  p.k0 = a.take<uint64_t>(n);
  p.k1 = a.take<uint64_t>(n);
  p.v0 = a.take<int32_t>(n);
  p.v1 = a.take<int32_t>(n);
  p.head = a.take<int32_t>(n);
  p.rank = a.take<int32_t>(n);
  size_t t1 = sort_pairs_u64_temp_bytes(n), t2 = scan_i32_temp_bytes(n);
  p.tmp_bytes = t1 > t2 ? t1 : t2;
  p.tmp = a.take<char>(p.tmp_bytes);
  return a.ok();
}

static int run_unique(UniquePlan& p, int64_t n, int key_bits, uint64_t sentinel, int64_t* uniq,
                      int64_t* inv, int32_t* order, int32_t* seg_start, int64_t* d_num,
                      hipStream_t s) {
  int rc = sort_pairs_u64(p.tmp, p.tmp_bytes, p.k0, p.k1, p.v0, p.v1, n, 0, key_bits, s);
  if (rc) return rc;
  int g = grid_for(n);
  head_flags_kernel<<<g, kBlock, 0, s>>>(p.k1, n, sentinel, p.head);
  rc = inclusive_scan_i32(p.tmp, p.tmp_bytes, p.head, p.rank, n, s);
  if (rc) return rc;
  unique_emit_kernel<<<g, kBlock, 0, s>>>(p.k1, p.v1, p.head, p.rank, n, sentinel, uniq, inv,
                                          seg_start, d_num);
  if (order) PCS_CHECK_HIP(hipMemcpyAsync(order, p.v1, n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// ---- sorted unique on keys the caller writes straight into the workspace (saves the key copy and
// the copy of the sorted order): used by the native tracker's combined anchor + target sampling
int unique_prepared_ptrs(void* ws, size_t ws_bytes, int64_t n, uint64_t** keys, int32_t** vals) {
  Arena a(ws, ws_bytes);
  UniquePlan p;
  if (!plan_unique(a, n > 0 ? n : 1, p)) return PCS_ERR_WORKSPACE;
  *keys = p.k0;
  *vals = p.v0;
  return PCS_OK;
}

// seg_start [v+1] boundaries into *sorted_vals (original indices ordered by (key, index)), *d_num = v
int unique_prepared_run(void* ws, size_t ws_bytes, int64_t n, int key_bits, int32_t* seg_start,
                        int64_t* d_num, const int32_t** sorted_vals, hipStream_t s) {
  if (n < 1) return PCS_ERR_ARG;
  Arena a(ws, ws_bytes);
  UniquePlan p;
  if (!plan_unique(a, n, p)) return PCS_ERR_WORKSPACE;
  *sorted_vals = p.v1;
  return run_unique(p, n, key_bits, ~0ULL, nullptr, nullptr, nullptr, seg_start, d_num, s);
}

}  // namespace pcs

using namespace pcs;

extern "C" {

int pcs_minmax4_f32(const float* pts, int64_t n, float* out8, hipStream_t s) {
  if (!pts || !out8 || n < 0) return PCS_ERR_ARG;
  minmax4_init_kernel<<<1, 64, 0, s>>>(out8);
  if (n > 0) minmax4_kernel<<<grid_for(n), kBlock, 0, s>>>((const float4*)pts, n, out8);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

int pcs_grid_cluster_ids(const float* pts, int64_t n, const float* bounds8, const float* size4,
                         float margin0, int64_t* ids, hipStream_t s) {
  if (!pts || !bounds8 || !size4 || !ids || n < 0) return PCS_ERR_ARG;
  if (n == 0) return PCS_OK;
  float4 sz = make_float4(size4[0], size4[1], size4[2], size4[3]);
  grid_ids_kernel<<<grid_for(n), kBlock, 0, s>>>((const float4*)pts, n, bounds8, sz, margin0, ids);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

size_t pcs_unique_i64_workspace_bytes(int64_t n) {
  Arena a(nullptr, 0);
  UniquePlan p;
  plan_unique(a, n > 0 ? n : 1, p);
  return a.off;
}

int pcs_unique_i64(const int64_t* ids, int64_t n, int key_bits, int64_t* uniq, int64_t* inv,
                   int32_t* order, int32_t* seg_start, int64_t* d_num, void* ws, size_t ws_bytes,
                   hipStream_t s) {
  if (!ids || !seg_start || !d_num || n < 0 || !ws) return PCS_ERR_ARG;
  if (n >= (1LL << 31)) return PCS_ERR_RANGE;
  if (n == 0) { PCS_CHECK_HIP(hipMemsetAsync(d_num, 0, sizeof(int64_t), s)); return PCS_OK; }
  if (key_bits <= 0 || key_bits > 64) key_bits = 64;
  Arena a(ws, ws_bytes);
  UniquePlan p;
  if (!plan_unique(a, n, p)) return PCS_ERR_WORKSPACE;
  iota_keys_kernel<<<grid_for(n), kBlock, 0, s>>>(ids, n, p.k0, p.v0);
  // ids are non-negative; the sentinel (all ones) can only collide with id == -1, never produced
  return run_unique(p, n, key_bits, ~0ULL, uniq, inv, order, seg_start, d_num, s);
}

int pcs_segment_reduce_f32(const float* data, int d, const int32_t* order, const int32_t* seg_start,
                           int64_t v, int mean, float* out, hipStream_t s) {
  if (!data || !order || !seg_start || !out || d <= 0 || v < 0) return PCS_ERR_ARG;
  if (v == 0) return PCS_OK;
  if ((mean & 4) && d <= 16) {
    segment_reduce_block_kernel<<<(int)(v > 4096 ? 4096 : v), kBlock, 0, s>>>(data, d, order,
                                                                             seg_start, v, out, mean & 1);
    PCS_LAUNCH_CHECK();
    return PCS_OK;
  }
  if ((mean & 2) && d <= 16) {
    int64_t blocks = (v * 64 + kBlock - 1) / kBlock;
    segment_reduce_wave_kernel<<<(int)(blocks > 2048 ? 2048 : blocks), kBlock, 0, s>>>(
        data, d, order, seg_start, v, out, mean & 1);
    PCS_LAUNCH_CHECK();
    return PCS_OK;
  }
  mean &= 1;
  segment_mean_kernel<<<grid_for(v * d), kBlock, 0, s>>>(data, d, order, seg_start, v, out, mean);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

int pcs_segment_pick(const int32_t* order, const int32_t* seg_start, int64_t v, int pick_last,
                     int64_t* out, hipStream_t s) {
  if (!order || !seg_start || !out || v < 0) return PCS_ERR_ARG;
  if (v == 0) return PCS_OK;
  segment_pick_kernel<<<grid_for(v), kBlock, 0, s>>>(order, seg_start, v, pick_last, out);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

size_t pcs_dynamic_voxelize_workspace_bytes(int64_t n) {
  Arena a(nullptr, 0);
  UniquePlan p;
  plan_unique(a, n > 0 ? n : 1, p);
  a.take<int32_t>((n > 0 ? n : 1) + 1);
  return a.off;
}

// points [n, ncol] = (batch, x, y, z, feat...). Outputs sized for the worst case (n voxels):
// voxel_feat [n, ncol-1], voxel_coords [n, 4] (b, z, y, x), inv [n] (-1 for dropped points).
int pcs_dynamic_voxelize(const float* pts, int64_t n, int ncol, const float* range6,
                         const float* voxel3, const int32_t* grid3, float* voxel_feat,
                         int32_t* voxel_coords, int64_t* inv, int64_t* d_num, void* ws,
                         size_t ws_bytes, hipStream_t s) {
  if (!pts || !range6 || !voxel3 || !grid3 || !voxel_feat || !voxel_coords || !d_num || !ws ||
      ncol < 4 || n < 0)
    return PCS_ERR_ARG;
  if (n >= (1LL << 31)) return PCS_ERR_RANGE;
  if (n == 0) { PCS_CHECK_HIP(hipMemsetAsync(d_num, 0, sizeof(int64_t), s)); return PCS_OK; }
  Arena a(ws, ws_bytes);
  UniquePlan p;
  if (!plan_unique(a, n, p)) return PCS_ERR_WORKSPACE;
  int32_t* seg = a.take<int32_t>(n + 1);
  if (!a.ok()) return PCS_ERR_WORKSPACE;
  int g = grid_for(n);
  dyn_voxel_keys_kernel<<<g, kBlock, 0, s>>>(pts, n, ncol, range6[0], range6[1], range6[2],
                                             voxel3[0], voxel3[1], voxel3[2], grid3[0], grid3[1],
                                             grid3[2], p.k0, p.v0);
  int rc = run_unique(p, n, 64, ~0ULL, nullptr, inv, nullptr, seg, d_num, s);
  if (rc) return rc;
  dyn_voxel_coords_kernel<<<g, kBlock, 0, s>>>(p.k1, seg, d_num, grid3[0], grid3[1], grid3[2],
                                               voxel_coords);
  dyn_voxel_mean_kernel<<<grid_for(n * (ncol - 1)), kBlock, 0, s>>>(pts, ncol, p.v1, seg, d_num,
                                                                      voxel_feat);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}


size_t pcs_hard_voxelize_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  Arena a(nullptr, 0);
  UniquePlan p;
  plan_unique(a, n, p);
  a.take<int32_t>(n + 1);   // seg_start
  a.take<int64_t>(1);       // cell count
  a.take<uint32_t>(n);      // first
  a.take<uint32_t>(n);      // first sorted
  a.take<int32_t>(n);       // cell
  a.take<int32_t>(n);       // cell sorted
  a.take<char>(sort_pairs_u32_temp_bytes(n));
  return a.off;
}

// points [n, ncol] = (x, y, z, feat...). voxels [max_voxels, max_points, ncol], coords
// [max_voxels, 3] (z, y, x), num [max_voxels]; d_num = number of voxels produced.
int pcs_hard_voxelize(const float* pts, int64_t n, int ncol, const float* range6,
                      const float* voxel3, int max_points, int max_voxels, float* voxels,
                      int32_t* coords, int32_t* num, int64_t* d_num, void* ws, size_t ws_bytes,
                      hipStream_t s) {
  if (!pts || !range6 || !voxel3 || !voxels || !coords || !num || !d_num || !ws || ncol < 3 ||
      n < 0 || max_points < 1 || max_voxels < 1)
    return PCS_ERR_ARG;
  if (n >= (1LL << 31)) return PCS_ERR_RANGE;
  PCS_CHECK_HIP(hipMemsetAsync(voxels, 0, (size_t)max_voxels * max_points * ncol * sizeof(float), s));
  PCS_CHECK_HIP(hipMemsetAsync(coords, 0, (size_t)max_voxels * 3 * sizeof(int32_t), s));
  PCS_CHECK_HIP(hipMemsetAsync(num, 0, (size_t)max_voxels * sizeof(int32_t), s));
  PCS_CHECK_HIP(hipMemsetAsync(d_num, 0, sizeof(int64_t), s));
  if (n == 0) return PCS_OK;
  // grid = round((hi - lo) / vs): computed in fp32 like the reference's DType=float instantiation
  int g3[3];
  for (int d = 0; d < 3; ++d) g3[d] = (int)roundf((range6[3 + d] - range6[d]) / voxel3[d]);
  Arena a(ws, ws_bytes);
  UniquePlan p;
  if (!plan_unique(a, n, p)) return PCS_ERR_WORKSPACE;
  int32_t* seg = a.take<int32_t>(n + 1);
  int64_t* d_cells = a.take<int64_t>(1);
  uint32_t* first = a.take<uint32_t>(n);
  uint32_t* first_s = a.take<uint32_t>(n);
  int32_t* cell = a.take<int32_t>(n);
  int32_t* cell_s = a.take<int32_t>(n);
  size_t tb = sort_pairs_u32_temp_bytes(n);
  void* tmp = a.take<char>(tb);
  if (!a.ok()) return PCS_ERR_WORKSPACE;
  int g = grid_for(n);
  hard_voxel_keys_kernel<<<g, kBlock, 0, s>>>(pts, n, ncol, range6[0], range6[1], range6[2],
                                              voxel3[0], voxel3[1], voxel3[2], g3[0], g3[1], g3[2],
                                              p.k0, p.v0);
  int rc = run_unique(p, n, 64, ~0ULL, nullptr, nullptr, nullptr, seg, d_cells, s);
  if (rc) return rc;
  hard_voxel_first_kernel<<<g, kBlock, 0, s>>>(p.v1, seg, d_cells, n, first, cell);
  rc = sort_pairs_u32(tmp, tb, first, first_s, cell, cell_s, n, 0, 32, s);
  if (rc) return rc;
  clamp_count_kernel<<<1, 1, 0, s>>>(d_cells, (int64_t)max_voxels, d_num);
  hard_voxel_emit_kernel<<<grid_for(max_voxels), kBlock, 0, s>>>(
      pts, ncol, p.k1, p.v1, seg, cell_s, d_num, max_points, g3[0], g3[1], voxels, coords, num);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

}  // extern "C"
