// One-workgroup (1024 threads) building blocks shared by the tracker's sampling kernels (pcs_tracker.hip) and the
// conv kernels' row ordering (pcs_spconv.hip): an exclusive scan and a stable LSD radix sort of packed 64-bit words.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcs {

constexpr int kWg = 1024;  // one-workgroup kernels (sort, compaction)

// exclusive scan of one int per thread; *total receives the sum. `sh` is LDS scratch [kWg / 64 + 1].
__device__ __forceinline__ int wg_excl_scan(int v, int* total, int* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  __syncthreads();  // sh may still be read from a previous call
  if (lane == 63) sh[w] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int q = 0; q < kWg / 64; ++q) { int t = sh[q]; sh[q] = run; run += t; }
    sh[kWg / 64] = run;
  }
  __syncthreads();
  *total = sh[kWg / 64];
  return sh[w] + incl - v;
}


// A stable LSD radix sort of n packed words by one workgroup, 8 bits per pass, the items cut into 16 contiguous
// wave slices (rank inside a wave by ballots: no atomics on the data path).
// The same sort on PACKED words (key << ib | original index, ib = bits of n - 1): one 8-byte word per
// item instead of a key and a value (a third less traffic through the one CU that runs the sort), and
// the histogram of the NEXT digit is counted while a pass scatters (the destination says which wave
// slice the item will be in), so a pass reads the data once instead of twice. On entry hist_a holds
// the per-slice histogram of the first digit (the caller counts it while it writes the words) and the
// workgroup has passed a barrier. Sorts by bits [ib, ib + nbits); returns 0 / 1 like wg_radix_sort.
__device__ __forceinline__ int wg_radix_sort_packed(uint64_t* w0, uint64_t* w1, int n, int ib, int nbits,
                                                    int* hist_a, int* hist_b, int* sh) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int per_wave = ((n + kWg - 1) / kWg) * 64;
  const int begin = w * per_wave;
  const int end = begin + per_wave < n ? begin + per_wave : n;
  uint64_t *win = w0, *wout = w1;
  int *cur = hist_a, *nxt = hist_b;
  int which = 0;
  constexpr int U = 8;  // rounds whose loads are issued together
  for (int b = 0; b < nbits; b += 8) {
    const bool more = b + 8 < nbits;
    const int sh0 = ib + b, sh1 = ib + b + 8;
    // exclusive scan in (digit, wave) order: thread t owns entries 4t .. 4t+3 of that order
    int c[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int e = 4 * tid + q;
      c[q] = cur[(e & 15) * 256 + (e >> 4)];
      sum += c[q];
    }
    int tot;
    int run = wg_excl_scan(sum, &tot, sh);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int e = 4 * tid + q;
      cur[(e & 15) * 256 + (e >> 4)] = run;
      run += c[q];
    }
    for (int q = tid; q < 16 * 256; q += kWg) nxt[q] = 0;
    __syncthreads();
    for (int i0 = begin; i0 < end; i0 += 64 * U) {
      uint64_t kk[U];
#pragma unroll
      for (int r = 0; r < U; ++r) {
        int i = i0 + r * 64 + lane;
        kk[r] = i < end ? win[i] : ~0ULL;
      }
#pragma unroll
      for (int r = 0; r < U; ++r) {
        int i = i0 + r * 64 + lane;
        bool in = i < end;
        int d = (int)((kk[r] >> sh0) & 255);
        unsigned long long same = __ballot(in);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
          unsigned long long bm = __ballot((d >> bit) & 1);
          same &= ((d >> bit) & 1) ? bm : ~bm;
        }
        int rank = __popcll(same & ((1ULL << lane) - 1ULL));
        int cnt = __popcll(same);
        int base = 0;
        if (in) base = cur[w * 256 + d];
        if (in && rank == 0) cur[w * 256 + d] = base + cnt;  // after every lane of the group has read
        if (in) {
          const int dest = base + rank;
          wout[dest] = kk[r];
          if (more) atomicAdd(&nxt[(dest / per_wave) * 256 + (int)((kk[r] >> sh1) & 255)], 1);
        }
      }
    }
    __syncthreads();
    uint64_t* tk = win; win = wout; wout = tk;
    int* th = cur; cur = nxt; nxt = th;
    which ^= 1;
  }
  return which;
}

}  // namespace pcs
