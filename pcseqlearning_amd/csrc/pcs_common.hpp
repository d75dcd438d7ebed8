// Shared device/host helpers for the gfx950 kernels behind the C-ABI in include/pcs.h.
// Everything here is written for wave64 / CDNA4 only.
#pragma once
#include <cstring>
#include <cstdint>
#include <cstddef>
#include <hip/hip_runtime.h>

#define PCS_OK 0
#define PCS_ERR_ARG (-1)       // bad argument (null pointer, negative size, unsupported shape)
#define PCS_ERR_WORKSPACE (-2) // caller-provided workspace too small
#define PCS_ERR_HIP (-3)       // a HIP runtime call failed (launch error, rocPRIM error)
#define PCS_ERR_RANGE (-4)     // a size exceeds what the kernels index with (e.g. E >= 2^31)

#define PCS_CHECK_HIP(expr)                      \
  do {                                           \
    hipError_t _e = (expr);                      \
    if (_e != hipSuccess) return PCS_ERR_HIP;    \
  } while (0)

#define PCS_LAUNCH_CHECK()                                  \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return PCS_ERR_HIP; \
  } while (0)

namespace pcs {

constexpr int kBlock = 256;

// Memory-bound grid sizing: cap at 256 CUs x 8 blocks and grid-stride the rest.
static inline int grid_for(int64_t n, int block = kBlock) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace; never allocates device memory itself.
struct Arena {
  char* base;
  size_t cap;
  size_t off;
  bool dry;  // size-query mode: only counts
  Arena(void* ws, size_t bytes) : base((char*)ws), cap(bytes), off(0), dry(ws == nullptr) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = align_up(count * sizeof(T));
    size_t at = off;
    off += bytes;
    if (dry) return nullptr;
    if (off > cap) return nullptr;
    return (T*)(base + at);
  }
  bool ok() const { return dry || off <= cap; }
};

// Per-thread page of pinned, device-visible host memory for the few status words a blocking entry
// point reads back (implemented in pcs_prims.hip): the last kernel of a phase writes them there, so
// the host needs only the stream synchronisation, not a device-to-host copy. nullptr if pinning fails.
// Layout: [0, 4 KiB) ICP states, [4 KiB, 8 KiB) tracker states, [8 KiB, 64 KiB) staging of
// descriptors on their way to the device (hipMemcpyAsync from pinned memory does not block).
constexpr size_t kPinnedPageBytes = 65536;
constexpr size_t kPinnedTrackerOff = 4096;
constexpr size_t kPinnedStageOff = 8192;
void* thread_pinned_page();

// sorted unique on keys written by the caller into the workspace (pcs_voxel.hip)
int unique_prepared_ptrs(void* ws, size_t ws_bytes, int64_t n, uint64_t** keys, int32_t** vals);
int unique_prepared_run(void* ws, size_t ws_bytes, int64_t n, int key_bits, int32_t* seg_start,
                        int64_t* d_num, const int32_t** sorted_vals, hipStream_t s);

// ---- prims implemented in pcs_prims.hip (the only TU that includes rocPRIM) ----
size_t sort_pairs_u64_temp_bytes(size_t n);
int sort_pairs_u64(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout,
                   const int32_t* vin, int32_t* vout, size_t n, int begin_bit, int end_bit,
                   hipStream_t s);
size_t sort_pairs_u32_temp_bytes(size_t n);
int sort_pairs_u32(void* tmp, size_t tmp_bytes, const uint32_t* kin, uint32_t* kout,
                   const int32_t* vin, int32_t* vout, size_t n, int begin_bit, int end_bit,
                   hipStream_t s);
size_t scan_i32_temp_bytes(size_t n);
int exclusive_scan_i32(void* tmp, size_t tmp_bytes, const int32_t* in, int32_t* out, size_t n,
                       hipStream_t s);
int inclusive_scan_i32(void* tmp, size_t tmp_bytes, const int32_t* in, int32_t* out, size_t n,
                       hipStream_t s);
size_t scan_i64_temp_bytes(size_t n);
int exclusive_scan_i64(void* tmp, size_t tmp_bytes, const int64_t* in, int64_t* out, size_t n,
                       hipStream_t s);

// connected components over a neighbour table (pcs_segment.hip): node q -- nbr_idx[q, e] for e < deg[q] with
// nbr_d[q, e] <= r2cut[i], for n_cut cuts at once; component [n_cut, n], d_num [n_cut]
size_t cc_table_workspace_bytes(int64_t n, int n_cut);
int cc_table_components(const int32_t* nbr_idx, const float* nbr_d, const int32_t* deg, int64_t n, int K,
                        const float* r2cut, int n_cut, int64_t* component, int64_t* d_num, void* ws,
                        size_t ws_bytes, hipStream_t s);

// 64-lane butterfly reductions (all lanes end with the result).
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// float atomics on ordered-int views: valid for non-NaN input.
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
  if (v >= 0.f) atomicMin((int*)addr, __float_as_int(v));
  else atomicMax((unsigned int*)addr, __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned int*)addr, __float_as_uint(v));
}

// 64-bit mix used by the open-addressing coordinate tables (rulebook, cell lookup).
__device__ __forceinline__ uint32_t mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return (uint32_t)k;
}

}  // namespace pcs
