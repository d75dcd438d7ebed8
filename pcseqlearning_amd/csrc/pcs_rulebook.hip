// Sparse-conv rulebook (spconv "indice pairs") build for gfx950.
//
// Restates the semantics of the reference CPU builders
//   pcdet/ops/spconv/include/spconv/geometry.h:25-85   (getValidOutPos: kernel-offset numbering)
//   pcdet/ops/spconv/include/spconv/geometry.h:145-194 (getIndicePairsConv: first-encounter out ids)
//   pcdet/ops/spconv/include/spconv/geometry.h:87-142, 197-245 (getValidOutPosTranspose, getIndicePairsDeConv)
//   pcdet/ops/spconv/include/spconv/geometry.h:248-297 (getIndicePairsSubM)
//   pcdet/ops/spconv/include/spconv/spconv_ops.h:27-141 (getIndicePair: shapes, subM forces s=1,p=k/2)
// with a different data structure: instead of the dense `gridOut[batch*volume]` (371 MB memset at
// the Waymo grid) an open-addressing coordinate table of 2x the active count lives in the caller's
// workspace, and pairs are emitted by count -> scan -> fill so that, per kernel offset, pairs are in
// ascending input index -- the reference CPU functor order -- by construction (no atomics on the
// pair slots, so the rulebook is run-to-run deterministic, unlike indice.cu.h's atomicAdd slots).
//
// Pair (j, k): out position o_d = (in_d + pad_d - k_d*dil_d) / stride_d when divisible and in
// range; kernel offset id k = (kz*KY + ky)*KX + kx. For a fixed input the reference enumerates
// valid outputs in ascending k, so the first-encounter order of output voxels is the order of
// min(j*KV + k) over the emissions that hit them. Transposed conv: o_d = in_d*stride_d - pad_d + k_d*dil_d
// when in range; the reference walks a fixed input's footprint from `uppers` downward, i.e. in
// DESCENDING k, so there the first-encounter order is the order of min(j*KV + (KV - 1 - k)).
#include "pcs_common.hpp"

namespace pcs {

constexpr uint64_t kEmptyKey = ~0ULL;

struct ConvGeom {
  int ks[3], st[3], pd[3], dl[3];
  int in_shape[3], out_shape[3];
  int kv;
  int transpose;
};

// Probe sequence of a key. First slot: the four cells of an aligned run along x (the fastest axis of the
// key) share a 32-byte run of slots, so the three x-neighbours a 3 x 3 x 3 footprint probes per (z, y) mostly
// come with one sector instead of three (the probes are what these kernels wait for). Further slots: a
// key-dependent stride of 4 * odd (double hashing; every slot of the key's residue class is reached) --
// occupied runs of x leave runs of full slots, and a +1 walk through them made the LONGEST walk of a wave
// (what a launch at one frame per batch waits for) 10-20 slots. Tables have >= 64 slots.
struct Probe { uint32_t slot, step; };
__device__ __forceinline__ Probe probe_begin(uint64_t key, uint32_t hmask) {
  const uint32_t m = mix64(key >> 2);
  Probe p;
  p.slot = ((m << 2) | ((uint32_t)key & 3u)) & hmask;
  p.step = ((m >> 13) | 1u) << 2;
  return p;
}

__device__ __forceinline__ uint32_t table_find(const uint64_t* __restrict__ keys, uint32_t hmask,
                                               uint64_t key) {
  Probe p = probe_begin(key, hmask);
  while (true) {
    uint64_t k = keys[p.slot];
    if (k == key) return p.slot;
    if (k == kEmptyKey) return 0xFFFFFFFFu;
    p.slot = (p.slot + p.step) & hmask;
  }
}

__device__ __forceinline__ uint32_t table_insert(uint64_t* __restrict__ keys, uint32_t hmask,
                                                 uint64_t key, bool* claimed = nullptr) {
  Probe p = probe_begin(key, hmask);
  while (true) {
    uint64_t prev = atomicCAS((unsigned long long*)&keys[p.slot], (unsigned long long)kEmptyKey,
                              (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) {
      if (claimed) *claimed = prev == kEmptyKey;
      return p.slot;
    }
    p.slot = (p.slot + p.step) & hmask;
  }
}

// out position key of input voxel `c` (b,z,y,x) under kernel offset k, or kEmptyKey if invalid
__device__ __forceinline__ uint64_t out_key(const ConvGeom& g, int4 c, int k) {
  int kx = k % g.ks[2];
  int t = k / g.ks[2];
  int ky = t % g.ks[1];
  int kz = t / g.ks[1];
  int kk[3] = {kz, ky, kx};
  int in[3] = {c.y, c.z, c.w};
  int o[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    int q;
    if (g.transpose) {
      q = in[d] * g.st[d] - g.pd[d] + kk[d] * g.dl[d];
      if (q < 0) return kEmptyKey;
    } else {
      int num = in[d] + g.pd[d] - kk[d] * g.dl[d];
      if (num < 0) return kEmptyKey;
      q = num / g.st[d];
      if (q * g.st[d] != num) return kEmptyKey;
    }
    if (q >= g.out_shape[d]) return kEmptyKey;
    o[d] = q;
  }
  uint64_t vol = (uint64_t)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
  return (uint64_t)c.x * vol + ((uint64_t)o[0] * g.out_shape[1] + o[1]) * g.out_shape[2] + o[2];
}

// output coordinate along axis d of an input coordinate under kernel offset kd, or -1
__device__ __forceinline__ int out_pos(const ConvGeom& g, int d, int in, int kd) {
  int q;
  if (g.transpose) {
    q = in * g.st[d] - g.pd[d] + kd * g.dl[d];
    if (q < 0) return -1;
  } else {
    const int num = in + g.pd[d] - kd * g.dl[d];
    if (num < 0) return -1;
    q = num / g.st[d];
    if (q * g.st[d] != num) return -1;
  }
  return q < g.out_shape[d] ? q : -1;
}

// ---- subM: table of the input voxels themselves, value = input index (last writer = max j)
__global__ void __launch_bounds__(kBlock)
insert_inputs_kernel(const int4* __restrict__ idx, int V, ConvGeom g, uint64_t* __restrict__ keys,
                     int32_t* __restrict__ vals, uint32_t hmask) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V) return;
  int4 c = idx[j];
  uint64_t vol = (uint64_t)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
  uint64_t key = (uint64_t)c.x * vol + ((uint64_t)c.y * g.out_shape[1] + c.z) * g.out_shape[2] + c.w;
  uint32_t slot = table_insert(keys, hmask, key);
  atomicMax(&vals[slot], j);
}

// ---- strided conv P1: every emission claims its output cell with min(j*KV + k)
__global__ void __launch_bounds__(kBlock)
insert_outputs_kernel(const int4* __restrict__ idx, int V, ConvGeom g, uint64_t* __restrict__ keys,
                      uint32_t* __restrict__ minseq, uint32_t hmask, int32_t* __restrict__ lslot,
                      int32_t* __restrict__ counter) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V) return;
  int4 c = idx[j];
  for (int k = 0; k < g.kv; ++k) {
    uint64_t key = out_key(g, c, k);
    if (key == kEmptyKey) continue;
    bool claimed = false;
    uint32_t slot = table_insert(keys, hmask, key, &claimed);
    atomicMin(&minseq[slot], (uint32_t)j * (uint32_t)g.kv + (uint32_t)(g.transpose ? g.kv - 1 - k : k));
    // the claimer lists the slot (order irrelevant: sorted by minseq afterwards), so the 16 V-slot
    // table never has to be scanned for its occupied entries
    if (claimed) lslot[atomicAdd(counter, 1)] = (int32_t)slot;
  }
}

// ---- strided conv P2a: final first-encounter key of every listed output slot
__global__ void __launch_bounds__(kBlock)
list_outputs_kernel(const uint32_t* __restrict__ minseq, const int32_t* __restrict__ lslot,
                    const int32_t* __restrict__ counter, uint32_t* __restrict__ lseq) {
  int U = *counter;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < U; p += gridDim.x * blockDim.x)
    lseq[p] = minseq[lslot[p]];
}

// ---- strided conv P2b: rank r (first-encounter order) -> out coords; slot value <- rank
__global__ void __launch_bounds__(kBlock)
assign_outputs_kernel(const int32_t* __restrict__ sslot, const int32_t* __restrict__ counter,
                      const uint64_t* __restrict__ keys, ConvGeom g, int32_t* __restrict__ vals,
                      int32_t* __restrict__ out_ids, int64_t* __restrict__ d_num_out) {
  int U = *counter;
  if (blockIdx.x == 0 && threadIdx.x == 0) *d_num_out = U;
  uint64_t vol = (uint64_t)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < U; r += gridDim.x * blockDim.x) {
    int slot = sslot[r];
    uint64_t key = keys[slot];
    vals[slot] = r;
    int b = (int)(key / vol);
    uint64_t rem = key % vol;
    int x = (int)(rem % g.out_shape[2]);
    rem /= g.out_shape[2];
    int y = (int)(rem % g.out_shape[1]);
    int z = (int)(rem / g.out_shape[1]);
    ((int4*)out_ids)[r] = make_int4(b, z, y, x);
  }
}

// ---- strided / transposed conv, packed form (batch * output volume < 2^32: every grid on the path)
// A table word is  key << 32 | first-encounter sequence  (empty: all ones). One emission = a plain load
// and, only where it can change the word, ONE atomic: the CAS of the claimer, or a 64-bit atomicMin with
// the same key in the high half (the smaller sequence wins). Inputs are served roughly in ascending j, so
// the claimer usually holds the smallest sequence already and the other emissions into the cell (up to
// eight for k = 3, s = 2) read it and leave. (The two-array form below costs a CAS and an atomicMin per
// emission on a table sized for 8 outputs per input: 1.7 ms per launch at 1.3 M inputs.)
// The launch also leaves what pair_count_kernel would compute: for a strided conv every valid emission
// hits an existing output, so the hit mask is the validity mask.
// A stale plain load is harmless: a slot's key never changes once claimed, so a stale word is either
// empty (the CAS then returns the truth) or carries a sequence that is too large (a needless atomicMin).
constexpr int kMaxProbe = 256;  // longer runs mean a table that is (nearly) full: flagged, the host retries larger

__device__ __forceinline__ uint32_t packed_find(const uint64_t* __restrict__ tab, uint32_t hmask, uint32_t key) {
  Probe p = probe_begin(key, hmask);
  while (true) {  // (only called for keys that were inserted)
    if ((uint32_t)(tab[p.slot] >> 32) == key) return p.slot;
    p.slot = (p.slot + p.step) & hmask;
  }
}

// The claimed slots are listed per WORKGROUP (LDS) and handed to the global list with one add per workgroup:
// one returning atomic per claim on a single counter word was the launch (the word takes ~90 of them per
// microsecond from the whole chip: 1.3 M claims = the 2-4 ms the launch took, whatever the table cost).
constexpr int kClaimList = 4096;  // per workgroup; 256 inputs x 8 emissions fit, a transposed 3 x 3 x 3 may pass it
__global__ void __launch_bounds__(kBlock)
insert_outputs_packed_kernel(const int4* __restrict__ idx, int V, ConvGeom g, uint64_t* __restrict__ tab,
                             uint32_t hmask, int32_t* __restrict__ lslot, int32_t* __restrict__ counter,
                             uint32_t* __restrict__ masks, int32_t* __restrict__ blockcnt, int nblocks) {
  __shared__ int wcnt[kBlock / 64][32];
  __shared__ int s_list[kClaimList];
  __shared__ int s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t mask = 0;
  if (j < V) {
    const int4 c = idx[j];
    const uint32_t vol = (uint32_t)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
    const uint32_t base = (uint32_t)c.x * vol;
    const int in[3] = {c.y, c.z, c.w};
    bool over = false;
    for (int kz = 0; kz < g.ks[0] && !over; ++kz) {
      const int oz = out_pos(g, 0, in[0], kz);
      if (oz < 0) continue;
      for (int ky = 0; ky < g.ks[1] && !over; ++ky) {
        const int oy = out_pos(g, 1, in[1], ky);
        if (oy < 0) continue;
        for (int kx = 0; kx < g.ks[2]; ++kx) {
          const int ox = out_pos(g, 2, in[2], kx);
          if (ox < 0) continue;
          const int k = (kz * g.ks[1] + ky) * g.ks[2] + kx;
          mask |= 1u << k;
          const uint32_t key = base + ((uint32_t)oz * g.out_shape[1] + oy) * g.out_shape[2] + ox;
          const uint32_t seq = (uint32_t)j * (uint32_t)g.kv + (uint32_t)(g.transpose ? g.kv - 1 - k : k);
          const uint64_t word = ((uint64_t)key << 32) | seq;
          const Probe pb = probe_begin(key, hmask);
          uint32_t slot = pb.slot;
          int probe = 0;
          for (; probe < kMaxProbe; ++probe) {
            uint64_t cur = tab[slot];
            if (cur == kEmptyKey) {
              cur = atomicCAS((unsigned long long*)&tab[slot], (unsigned long long)kEmptyKey, (unsigned long long)word);
              if (cur == kEmptyKey) {  // claimed: list the slot (order irrelevant, sorted by sequence afterwards)
                const int at = atomicAdd(&s_n, 1);
                if (at < kClaimList) s_list[at] = (int)slot;
                else lslot[atomicAdd(&counter[0], 1)] = (int32_t)slot;
                break;
              }
            }
            if ((uint32_t)(cur >> 32) == key) {
              if ((uint32_t)cur > seq) atomicMin((unsigned long long*)&tab[slot], (unsigned long long)word);
              break;
            }
            slot = (slot + pb.step) & hmask;
          }
          if (probe == kMaxProbe) { counter[1] = 1; over = true; break; }
        }
      }
    }
    masks[j] = mask;
  }
  for (int k = 0; k < g.kv; ++k) {
    unsigned long long b = __ballot((mask >> k) & 1u);
    if (lane == 0) wcnt[w][k] = __popcll(b);
  }
  __syncthreads();
  const int listed = s_n < kClaimList ? s_n : kClaimList;
  if (threadIdx.x == 0 && listed > 0) s_base = atomicAdd(&counter[0], listed);
  if (threadIdx.x < g.kv) {
    int k = threadIdx.x, sum = 0;
#pragma unroll
    for (int q = 0; q < kBlock / 64; ++q) sum += wcnt[q][k];
    blockcnt[(int64_t)k * nblocks + blockIdx.x] = sum;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < listed; q += kBlock) lslot[s_base + q] = s_list[q];
}

__global__ void __launch_bounds__(kBlock)
list_outputs_packed_kernel(const uint64_t* __restrict__ tab, const int32_t* __restrict__ lslot, int U,
                           uint32_t* __restrict__ lseq) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < U; p += gridDim.x * blockDim.x)
    lseq[p] = (uint32_t)tab[lslot[p]];
}

__global__ void __launch_bounds__(kBlock)
assign_outputs_packed_kernel(const int32_t* __restrict__ sslot, int U, const uint64_t* __restrict__ tab,
                             ConvGeom g, int32_t* __restrict__ vals, int32_t* __restrict__ out_ids,
                             int64_t* __restrict__ d_num_out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *d_num_out = U;
  const uint32_t vol = (uint32_t)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < U; r += gridDim.x * blockDim.x) {
    const int slot = sslot[r];
    const uint32_t key = (uint32_t)(tab[slot] >> 32);
    vals[slot] = r;
    const int b = (int)(key / vol);
    uint32_t rem = key % vol;
    const int x = (int)(rem % g.out_shape[2]);
    rem /= g.out_shape[2];
    const int y = (int)(rem % g.out_shape[1]);
    const int z = (int)(rem / g.out_shape[1]);
    ((int4*)out_ids)[r] = make_int4(b, z, y, x);
  }
}

// fill from the masks the insert left (no second validity pass): position = offset base of (k, block) +
// rank of this input among the block's hits; the output index is one table lookup per pair
__global__ void __launch_bounds__(kBlock)
pair_fill_packed_kernel(const int4* __restrict__ idx, int V, ConvGeom g, const uint64_t* __restrict__ tab,
                        const int32_t* __restrict__ vals, uint32_t hmask, const uint32_t* __restrict__ masks,
                        const int32_t* __restrict__ blockoff, int nblocks, int32_t* __restrict__ pairs,
                        int32_t* __restrict__ num) {
  __shared__ int wcnt[kBlock / 64][32];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t mask = (j < V) ? masks[j] : 0u;
  for (int k = 0; k < g.kv; ++k) {
    unsigned long long b = __ballot((mask >> k) & 1u);
    if (lane == 0) wcnt[w][k] = __popcll(b);
  }
  __syncthreads();
  const int4 c = (j < V) ? idx[j] : make_int4(0, 0, 0, 0);
  const uint32_t vol = (uint32_t)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
  for (int k = 0; k < g.kv; ++k) {
    unsigned long long b = __ballot((mask >> k) & 1u);
    if ((mask >> k) & 1u) {
      int pre = __popcll(b & ((1ULL << lane) - 1ULL));
      for (int q = 0; q < w; ++q) pre += wcnt[q][k];
      const int64_t kb = (int64_t)k * nblocks;
      const int pos = blockoff[kb + blockIdx.x] - blockoff[kb] + pre;
      const int kx = k % g.ks[2], t = k / g.ks[2], ky = t % g.ks[1], kz = t / g.ks[1];
      const uint32_t key = (uint32_t)c.x * vol +
                           ((uint32_t)out_pos(g, 0, c.y, kz) * g.out_shape[1] + out_pos(g, 1, c.z, ky)) * g.out_shape[2] +
                           out_pos(g, 2, c.w, kx);
      pairs[((int64_t)k * 2 + 0) * V + pos] = j;
      pairs[((int64_t)k * 2 + 1) * V + pos] = vals[packed_find(tab, hmask, key)];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < g.kv) {
    const int k = threadIdx.x;
    num[k] = blockoff[(int64_t)(k + 1) * nblocks] - blockoff[(int64_t)k * nblocks];
  }
}

// ---- count: one thread per input voxel, 27-bit hit mask, per-block per-offset counts
__global__ void __launch_bounds__(kBlock)
pair_count_kernel(const int4* __restrict__ idx, int V, ConvGeom g,
                  const uint64_t* __restrict__ keys, uint32_t hmask, uint32_t* __restrict__ masks,
                  int32_t* __restrict__ blockcnt, int nblocks) {
  __shared__ int wcnt[kBlock / 64][32];
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t mask = 0;
  if (j < V) {
    int4 c = idx[j];
    for (int k = 0; k < g.kv; ++k) {
      uint64_t key = out_key(g, c, k);
      if (key != kEmptyKey && table_find(keys, hmask, key) != 0xFFFFFFFFu) mask |= 1u << k;
    }
    masks[j] = mask;
  }
  for (int k = 0; k < g.kv; ++k) {
    unsigned long long b = __ballot((mask >> k) & 1u);
    if (lane == 0) wcnt[w][k] = __popcll(b);
  }
  __syncthreads();
  if (threadIdx.x < g.kv) {
    int k = threadIdx.x, sum = 0;
#pragma unroll
    for (int q = 0; q < kBlock / 64; ++q) sum += wcnt[q][k];
    blockcnt[(int64_t)k * nblocks + blockIdx.x] = sum;
  }
}

// ---- subM with the conv's neighbour table: ONE probing pass. The voxel n found at c_j + pad - k is the
// partner of pair (in j, out n, offset k) and, seen from output j, its input at the mirrored offset
// kv - 1 - k (odd kernel sizes, no dilation): the pass writes row j of the output-major table the conv
// kernels read (pcs_build_neighbor_table's layout; -1 where nothing is there) next to the hit mask, and the
// fill takes the partners back from that row instead of probing again. A voxel that lost its coordinate to
// a later duplicate (insert_inputs_kernel keeps the largest index) is never an output: its row stays -1.
__global__ void __launch_bounds__(kBlock)
pair_count_table_kernel(const int4* __restrict__ idx, int V, ConvGeom g, const uint64_t* __restrict__ keys,
                        const int32_t* __restrict__ vals, uint32_t hmask, uint32_t* __restrict__ masks,
                        int32_t* __restrict__ blockcnt, int nblocks, int32_t* __restrict__ table) {
  __shared__ int wcnt[kBlock / 64][32];
  // the workgroup's rows are gathered in LDS and leave as one contiguous run (a thread writing its own 27
  // ints at a 108-byte stride made every store a partial sector: 0.8 ms per launch at 1.3 M voxels)
  __shared__ int32_t s_rows[kBlock * 32];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t mask = 0;
  if (j < V) {
    const int4 c = idx[j];
    const uint64_t vol = (uint64_t)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
    const uint64_t base = (uint64_t)c.x * vol;
    const int pz = g.ks[0] / 2, py = g.ks[1] / 2, px = g.ks[2] / 2;
    int32_t* row = s_rows + threadIdx.x * g.kv;  // (kv is odd here: the rows do not collide on the banks)
    const bool is_output = vals[table_find(keys, hmask, base + ((uint64_t)c.y * g.out_shape[1] + c.z) * g.out_shape[2] + c.w)] == j;
    int k = 0;
    for (int kz = 0; kz < g.ks[0]; ++kz) {
      const int z = c.y + pz - kz;
      for (int ky = 0; ky < g.ks[1]; ++ky) {
        const int y = c.z + py - ky;
        for (int kx = 0; kx < g.ks[2]; ++kx, ++k) {
          const int x = c.w + px - kx;
          int n = -1;
          if (z >= 0 && z < g.out_shape[0] && y >= 0 && y < g.out_shape[1] && x >= 0 && x < g.out_shape[2]) {
            const uint32_t slot = table_find(keys, hmask, base + ((uint64_t)z * g.out_shape[1] + y) * g.out_shape[2] + x);
            if (slot != 0xFFFFFFFFu) n = vals[slot];
          }
          if (n >= 0) mask |= 1u << k;
          row[g.kv - 1 - k] = is_output ? n : -1;
        }
      }
    }
    masks[j] = mask;
  }
  for (int k = 0; k < g.kv; ++k) {
    unsigned long long b = __ballot((mask >> k) & 1u);
    if (lane == 0) wcnt[w][k] = __popcll(b);
  }
  __syncthreads();
  if (threadIdx.x < g.kv) {
    int k = threadIdx.x, sum = 0;
#pragma unroll
    for (int q = 0; q < kBlock / 64; ++q) sum += wcnt[q][k];
    blockcnt[(int64_t)k * nblocks + blockIdx.x] = sum;
  }
  const int64_t first = (int64_t)blockIdx.x * blockDim.x;
  const int rows = (int)((V - first) < (int64_t)blockDim.x ? (V - first) : (int64_t)blockDim.x);
  int32_t* __restrict__ dst = table + first * g.kv;
  for (int q = threadIdx.x; q < rows * g.kv; q += kBlock) dst[q] = s_rows[q];
}

// ---- submanifold 3 x 3 x 3 on a packed table (batch * volume < 2^32): a word is  key << 32 | voxel index
// (the largest index of a coordinate wins, as in insert_inputs_kernel), so a hit needs no second load, and
// the 27 first probes of a voxel are 27 INDEPENDENT loads issued together -- at one frame per batch the
// launch is a chain of memory latencies (54 dependent loads per voxel in the loop form: 150 us for 66 k
// voxels), at twenty frames it is bound by the sectors the probes touch. Only a first slot that holds
// another key (a collision) walks on serially.
__global__ void __launch_bounds__(kBlock)
insert_inputs_packed_kernel(const int4* __restrict__ idx, int V, ConvGeom g, uint64_t* __restrict__ tab, uint32_t hmask) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= V) return;
  const int4 c = idx[j];
  const uint32_t vol = (uint32_t)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
  const uint32_t key = (uint32_t)c.x * vol + ((uint32_t)c.y * g.out_shape[1] + c.z) * g.out_shape[2] + c.w;
  const uint64_t word = ((uint64_t)key << 32) | (uint32_t)j;
  Probe p = probe_begin(key, hmask);
  while (true) {
    uint64_t cur = tab[p.slot];
    if (cur == kEmptyKey) {
      cur = atomicCAS((unsigned long long*)&tab[p.slot], (unsigned long long)kEmptyKey, (unsigned long long)word);
      if (cur == kEmptyKey) return;
    }
    if ((uint32_t)(cur >> 32) == key) {
      if ((uint32_t)cur < (uint32_t)j) atomicMax((unsigned long long*)&tab[p.slot], (unsigned long long)word);
      return;
    }
    p.slot = (p.slot + p.step) & hmask;
  }
}

__global__ void __launch_bounds__(kBlock)
pair_count_table333_kernel(const int4* __restrict__ idx, int V, ConvGeom g, const uint64_t* __restrict__ tab,
                           uint32_t hmask, uint32_t* __restrict__ masks, int32_t* __restrict__ blockcnt, int nblocks,
                           int32_t* __restrict__ table) {
  constexpr int KV = 27;
  __shared__ int wcnt[kBlock / 64][32];
  __shared__ int32_t s_rows[kBlock * KV];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t mask = 0;
  if (j < V) {
    const int4 c = idx[j];
    const uint32_t vol = (uint32_t)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
    const uint32_t base = (uint32_t)c.x * vol;
    uint32_t key[KV], slot[KV];
    uint64_t wd[KV];
    uint32_t valid = 0;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      const int z = c.y + 1 - k / 9, y = c.z + 1 - (k / 3) % 3, x = c.w + 1 - k % 3;
      const bool ok = z >= 0 && z < g.out_shape[0] && y >= 0 && y < g.out_shape[1] && x >= 0 && x < g.out_shape[2];
      key[k] = base + ((uint32_t)z * g.out_shape[1] + y) * g.out_shape[2] + x;
      slot[k] = probe_begin(key[k], hmask).slot;
      if (ok) valid |= 1u << k;
    }
    // rounds: every unresolved probe advances one slot per round, the loads of a round issued together (a
    // probe that walks holds up only itself: the chain is the LONGEST walk of the 27, not their sum)
    int nb[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) nb[k] = -1;
    uint32_t pending = valid;
    while (pending) {
#pragma unroll
      for (int k = 0; k < KV; ++k)
        if ((pending >> k) & 1u) wd[k] = tab[slot[k]];
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        if ((pending >> k) & 1u) {
          const uint64_t cur = wd[k];
          if (cur == kEmptyKey) {
            pending &= ~(1u << k);
          } else if ((uint32_t)(cur >> 32) == key[k]) {
            nb[k] = (int)(uint32_t)cur;
            mask |= 1u << k;
            pending &= ~(1u << k);
          } else {
            slot[k] = (slot[k] + probe_begin(key[k], hmask).step) & hmask;
          }
        }
      }
    }
    const bool is_output = nb[13] == j;
    int32_t* row = s_rows + threadIdx.x * KV;
#pragma unroll
    for (int k = 0; k < KV; ++k) row[KV - 1 - k] = is_output ? nb[k] : -1;
    masks[j] = mask;
  }
  for (int k = 0; k < KV; ++k) {
    unsigned long long b = __ballot((mask >> k) & 1u);
    if (lane == 0) wcnt[w][k] = __popcll(b);
  }
  __syncthreads();
  if (threadIdx.x < KV) {
    int k = threadIdx.x, sum = 0;
#pragma unroll
    for (int q = 0; q < kBlock / 64; ++q) sum += wcnt[q][k];
    blockcnt[(int64_t)k * nblocks + blockIdx.x] = sum;
  }
  const int64_t first = (int64_t)blockIdx.x * blockDim.x;
  const int rows = (int)((V - first) < (int64_t)blockDim.x ? (V - first) : (int64_t)blockDim.x);
  int32_t* __restrict__ dst = table + first * KV;
  for (int q = threadIdx.x; q < rows * KV; q += kBlock) dst[q] = s_rows[q];
}

__global__ void __launch_bounds__(kBlock)
pair_fill_table_kernel(int V, ConvGeom g, const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                       uint32_t hmask, const int4* __restrict__ idx, const uint32_t* __restrict__ masks,
                       const int32_t* __restrict__ blockoff, int nblocks, const int32_t* __restrict__ table,
                       int32_t* __restrict__ pairs, int32_t* __restrict__ num, int packed) {
  __shared__ int wcnt[kBlock / 64][32];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t mask = (j < V) ? masks[j] : 0u;
  for (int k = 0; k < g.kv; ++k) {
    unsigned long long b = __ballot((mask >> k) & 1u);
    if (lane == 0) wcnt[w][k] = __popcll(b);
  }
  __syncthreads();
  const int32_t* __restrict__ row = table + (int64_t)(j < V ? j : 0) * g.kv;
  // (a voxel that is not an output has an empty row: its partners are probed for, as before)
  const bool row_ok = (mask == 0u) || row[g.kv / 2] == j;
  const int4 c = (j < V) ? idx[j] : make_int4(0, 0, 0, 0);
  for (int k = 0; k < g.kv; ++k) {
    unsigned long long b = __ballot((mask >> k) & 1u);
    if ((mask >> k) & 1u) {
      int pre = __popcll(b & ((1ULL << lane) - 1ULL));
      for (int q = 0; q < w; ++q) pre += wcnt[q][k];
      const int64_t kb = (int64_t)k * nblocks;
      const int pos = blockoff[kb + blockIdx.x] - blockoff[kb] + pre;
      int n;
      if (row_ok) n = row[g.kv - 1 - k];
      else if (packed) n = (int)(uint32_t)keys[packed_find(keys, hmask, (uint32_t)out_key(g, c, k))];
      else n = vals[table_find(keys, hmask, out_key(g, c, k))];
      pairs[((int64_t)k * 2 + 0) * V + pos] = j;
      pairs[((int64_t)k * 2 + 1) * V + pos] = n;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < g.kv) {
    const int k = threadIdx.x;
    num[k] = blockoff[(int64_t)(k + 1) * nblocks] - blockoff[(int64_t)k * nblocks];
  }
}

// ---- fill: position = offset base of (k, block) + rank of this input among the block's hits
__global__ void __launch_bounds__(kBlock)
pair_fill_kernel(const int4* __restrict__ idx, int V, ConvGeom g,
                 const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals,
                 uint32_t hmask, const uint32_t* __restrict__ masks,
                 const int32_t* __restrict__ blockoff, int nblocks, int32_t* __restrict__ pairs,
                 int32_t* __restrict__ num) {
  __shared__ int wcnt[kBlock / 64][32];
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t mask = (j < V) ? masks[j] : 0u;
  for (int k = 0; k < g.kv; ++k) {
    unsigned long long b = __ballot((mask >> k) & 1u);
    if (lane == 0) wcnt[w][k] = __popcll(b);
  }
  __syncthreads();
  int4 c = (j < V) ? idx[j] : make_int4(0, 0, 0, 0);
  for (int k = 0; k < g.kv; ++k) {
    unsigned long long b = __ballot((mask >> k) & 1u);
    if ((mask >> k) & 1u) {
      int pre = __popcll(b & ((1ULL << lane) - 1ULL));
      for (int q = 0; q < w; ++q) pre += wcnt[q][k];
      int64_t kb = (int64_t)k * nblocks;
      int pos = blockoff[kb + blockIdx.x] - blockoff[kb] + pre;
      uint32_t slot = table_find(keys, hmask, out_key(g, c, k));
      pairs[((int64_t)k * 2 + 0) * V + pos] = j;
      pairs[((int64_t)k * 2 + 1) * V + pos] = vals[slot];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < g.kv) {
    int k = threadIdx.x;
    num[k] = blockoff[(int64_t)(k + 1) * nblocks] - blockoff[(int64_t)k * nblocks];
  }
}

// Everything a rulebook build resets, in ONE launch (they were six to eight hipMemsetAsync / fill /
// copy calls per geometry: at one frame per batch the backbone's forward is launch-bound and a third
// of its launches were these).
struct RulebookInit {
  int32_t* pairs; int64_t n_pairs;     // <- -1
  int32_t* num; int kv;                // <- 0
  uint64_t* keys; int32_t* vals; uint32_t* minseq; int64_t H;  // <- all ones (minseq: strided only)
  uint32_t* lseq; int64_t cap;         // <- 0xFFFFFFFF (strided only)
  int32_t* counter;                    // <- 0 (strided only)
  int32_t* blockcnt_tail;              // <- 0
  int64_t* d_num_out; int64_t num_out; // <- num_out when >= 0 (subM: the outputs are the inputs)
};
__global__ void __launch_bounds__(kBlock) rulebook_init_kernel(RulebookInit a) {
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = tid; q < a.n_pairs; q += nt) a.pairs[q] = -1;
  for (int64_t q = tid; q < a.H; q += nt) {
    a.keys[q] = ~0ULL;
    if (a.vals) a.vals[q] = -1;
    if (a.minseq) a.minseq[q] = 0xFFFFFFFFu;
  }
  if (a.lseq)
    for (int64_t q = tid; q < a.cap; q += nt) a.lseq[q] = 0xFFFFFFFFu;
  if (tid < a.kv) a.num[tid] = 0;
  if (tid == 0) {
    if (a.counter) a.counter[0] = a.counter[1] = 0;
    *a.blockcnt_tail = 0;
    if (a.num_out >= 0) *a.d_num_out = a.num_out;
  }
}

// exclusive scan of n <= kScanOneWg ints by ONE workgroup (the per-(offset, block) pair counts: a few
// thousand entries; the library scan is three launches)
constexpr int kScanWg = 1024;
constexpr size_t kScanOneWg = 1u << 17;
__global__ void __launch_bounds__(kScanWg) scan_one_wg_kernel(const int32_t* __restrict__ in,
                                                              int32_t* __restrict__ out, int n) {
  __shared__ int wsum[kScanWg / 64];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  constexpr int PER = 4;  // consecutive entries per thread and round
  for (int base = 0; base < n; base += kScanWg * PER) {
    int v[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      int i = base + tid * PER + q;
      v[q] = i < n ? in[i] : 0;
      sum += v[q];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int carry = carry_s;
    int before = 0;
    for (int q = 0; q < w; ++q) before += wsum[q];
    int run = carry + before + incl - sum;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      int i = base + tid * PER + q;
      if (i < n) out[i] = run;
      run += v[q];
    }
    __syncthreads();
    if (tid == kScanWg - 1) carry_s = run;
    __syncthreads();
  }
}

static inline uint32_t pow2_at_least(uint64_t x) {
  uint32_t h = 64;
  while (h < x) h <<= 1;
  return h;
}

struct RulebookPlan {
  uint32_t H;
  int64_t cap;  // max distinct outputs (strided conv)
  uint64_t* keys;
  int32_t* vals;
  uint32_t *minseq, *lseq, *lseq_s, *masks;
  int32_t *lslot, *lslot_s, *counter, *blockcnt, *blockoff;
  void* tmp;
  size_t tmp_bytes;
  int nblocks;
};

static int64_t emissions_per_input(const int* ks, const int* st, int transpose = 0) {
  int64_t e = 1;
  for (int d = 0; d < 3; ++d) e *= transpose ? ks[d] : (ks[d] + st[d] - 1) / st[d];
  return e;
}

static bool plan_rulebook(Arena& a, int64_t V, int kv, int subm, int64_t cap, RulebookPlan& p) {
  if (V < 1) V = 1;
  p.cap = subm ? V : cap;
  // (submanifold: four slots per voxel -- most of the 27 probes of a voxel MISS, and a miss walks to the next
  // empty slot; strided: two per possible output, the retry size of the packed form)
  p.H = pow2_at_least((subm ? 4 : 2) * (uint64_t)p.cap);
  p.nblocks = (int)((V + kBlock - 1) / kBlock);
  p.keys = a.take<uint64_t>(p.H);
  p.vals = a.take<int32_t>(p.H);
  p.masks = a.take<uint32_t>(V);
  size_t nb = (size_t)kv * p.nblocks + 1;
  p.blockcnt = a.take<int32_t>(nb);
  p.blockoff = a.take<int32_t>(nb);
  size_t t = scan_i32_temp_bytes(nb);
  if (!subm) {
    p.minseq = a.take<uint32_t>(p.H);
    p.lseq = a.take<uint32_t>(p.cap);
    p.lseq_s = a.take<uint32_t>(p.cap);
    p.lslot = a.take<int32_t>(p.cap);
    p.lslot_s = a.take<int32_t>(p.cap);
    p.counter = a.take<int32_t>(2);  // claims; (packed form) the table-too-small flag
    size_t t2 = sort_pairs_u32_temp_bytes(p.cap);
    if (t2 > t) t = t2;
  }
  p.tmp_bytes = t;
  p.tmp = a.take<char>(t);
  return a.ok();
}

}  // namespace pcs

using namespace pcs;

extern "C" {

// upper bound on the number of output voxels of a strided conv over V inputs (rows to allocate
// for out_ids); for subM the outputs are the inputs. `subm`: 0 strided, 1 submanifold, 2 TRANSPOSED
// conv (every input emits its whole kernel footprint).
int64_t pcs_indice_pairs_out_capacity(int64_t V, const int* ksize3, const int* stride3, int subm) {
  if (subm == 1) return V;
  return V * emissions_per_input(ksize3, stride3, subm == 2);
}

size_t pcs_get_indice_pairs_workspace_bytes(int64_t V, const int* ksize3, const int* stride3,
                                            int subm) {
  Arena a(nullptr, 0);
  RulebookPlan p;
  int kv = ksize3[0] * ksize3[1] * ksize3[2];
  plan_rulebook(a, V, kv, subm == 1, pcs_indice_pairs_out_capacity(V > 0 ? V : 1, ksize3, stride3, subm), p);
  return a.off;
}

// 1 when pcs_get_indice_pairs_table fills `table` for this geometry (submanifold, odd kernel sizes, no
// dilation), 0 when the caller has to build it from the pairs (pcs_build_neighbor_table)
int pcs_indice_pairs_gives_table(const int* ksize3, const int* dil3, int subm) {
  if (!subm || !ksize3 || !dil3) return 0;
  for (int d = 0; d < 3; ++d)
    if ((ksize3[d] & 1) == 0 || dil3[d] != 1) return 0;
  return 1;
}

int pcs_get_indice_pairs_table(const int32_t* indices, int64_t V, int batch, const int* out_shape3,
                               const int* spatial_shape3, const int* ksize3, const int* stride3,
                               const int* pad3, const int* dil3, int subm, int transpose,
                               int32_t* out_ids, int64_t out_cap, int32_t* pairs, int32_t* num,
                               int64_t* d_num_out, int64_t* h_num_out, int32_t* table, void* ws, size_t ws_bytes,
                               hipStream_t s);

int pcs_get_indice_pairs(const int32_t* indices, int64_t V, int batch, const int* out_shape3,
                         const int* spatial_shape3, const int* ksize3, const int* stride3,
                         const int* pad3, const int* dil3, int subm, int transpose,
                         int32_t* out_ids, int64_t out_cap, int32_t* pairs, int32_t* num,
                         int64_t* d_num_out, void* ws, size_t ws_bytes, hipStream_t s) {
  return pcs_get_indice_pairs_table(indices, V, batch, out_shape3, spatial_shape3, ksize3, stride3, pad3, dil3, subm,
                                    transpose, out_ids, out_cap, pairs, num, d_num_out, nullptr, nullptr, ws, ws_bytes,
                                    s);
}

// The same, and (where pcs_indice_pairs_gives_table says so) row j of `table` [V, kv] = the inputs of
// output j by kernel offset, -1 where none: what pcs_build_neighbor_table would make of the pairs, from the
// rulebook's own probing pass (no second pass over the pairs, no reset of the table). `table` may be NULL.
// `h_num_out` (HOST pointer, may be NULL): the number of output voxels where the call knows it on return
// (submanifold: V; the packed strided form: it has synchronised the stream for it), else -1 = read *d_num_out.
int pcs_get_indice_pairs_table(const int32_t* indices, int64_t V, int batch, const int* out_shape3,
                               const int* spatial_shape3, const int* ksize3, const int* stride3,
                               const int* pad3, const int* dil3, int subm, int transpose,
                               int32_t* out_ids, int64_t out_cap, int32_t* pairs, int32_t* num,
                               int64_t* d_num_out, int64_t* h_num_out, int32_t* table, void* ws, size_t ws_bytes,
                               hipStream_t s) {
  if (h_num_out) *h_num_out = -1;
  if (!out_shape3 || !spatial_shape3 || !ksize3 || !stride3 || !pad3 || !dil3 || !num ||
      !d_num_out || !ws || V < 0 || batch < 1 || (V > 0 && (!indices || !pairs)))
    return PCS_ERR_ARG;
  if (transpose && subm) return PCS_ERR_ARG;
  ConvGeom g;
  g.kv = 1;
  g.transpose = transpose ? 1 : 0;
  for (int d = 0; d < 3; ++d) {
    g.ks[d] = ksize3[d];
    g.dl[d] = dil3[d];
    g.st[d] = subm ? 1 : stride3[d];
    g.pd[d] = subm ? ksize3[d] / 2 : pad3[d];
    g.in_shape[d] = spatial_shape3[d];
    g.out_shape[d] = out_shape3[d];
    if (g.ks[d] < 1 || g.st[d] < 1 || g.dl[d] < 1) return PCS_ERR_ARG;
    if (g.st[d] != 1 && g.dl[d] != 1) return PCS_ERR_ARG;  // as ops.py:70-71
    g.kv *= g.ks[d];
  }
  if (g.kv > 32) return PCS_ERR_ARG;  // hit masks are 32-bit; the path uses 3x3x3 and 3x1x1
  if (V >= (1LL << 31) / g.kv) return PCS_ERR_RANGE;
  uint64_t vol = (uint64_t)g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
  if (vol * (uint64_t)batch >= (1ULL << 62)) return PCS_ERR_RANGE;
  if (V == 0) {
    PCS_CHECK_HIP(hipMemsetAsync(num, 0, g.kv * sizeof(int32_t), s));
    PCS_CHECK_HIP(hipMemsetAsync(d_num_out, 0, sizeof(int64_t), s));
    if (h_num_out) *h_num_out = 0;
    return PCS_OK;
  }
  int64_t cap = pcs_indice_pairs_out_capacity(V, ksize3, stride3, transpose ? 2 : (subm ? 1 : 0));
  if (!subm && (!out_ids || out_cap < cap)) return PCS_ERR_ARG;
  Arena a(ws, ws_bytes);
  RulebookPlan p;
  if (!plan_rulebook(a, V, g.kv, subm, cap, p)) return PCS_ERR_WORKSPACE;
  uint32_t hmask = p.H - 1;
  const int4* idx = (const int4*)indices;
  int nb = p.nblocks;
  size_t nscan = (size_t)g.kv * nb + 1;
  // the packed strided form needs every output key below 2^32 - 1 (any grid of the path: 237 M cells at
  // twenty Waymo frames); larger grids take the two-array form
  const bool packed = !subm && vol * (uint64_t)batch < 0xFFFFFFFFULL;
  auto init = [&](uint32_t H, bool with_pairs) {
    RulebookInit in;
    in.pairs = pairs; in.n_pairs = with_pairs ? (int64_t)g.kv * 2 * V : 0;
    in.num = num; in.kv = g.kv;
    in.keys = p.keys; in.vals = packed ? nullptr : p.vals; in.minseq = (subm || packed) ? nullptr : p.minseq; in.H = H;
    in.lseq = (subm || packed) ? nullptr : p.lseq; in.cap = (subm || packed) ? 0 : p.cap;
    in.counter = subm ? nullptr : p.counter;
    in.blockcnt_tail = p.blockcnt + (nscan - 1);
    in.d_num_out = d_num_out; in.num_out = subm ? V : -1;
    int64_t longest = in.n_pairs > in.H ? in.n_pairs : in.H;
    if (in.cap > longest) longest = in.cap;
    int gi = grid_for(longest);
    if (gi > 4096) gi = 4096;
    rulebook_init_kernel<<<gi, kBlock, 0, s>>>(in);
  };
  if (packed) {
    // First try: a table of 4-8 slots per input (16-32 for the transposed form) instead of two per
    // POSSIBLE output (8 per input, 27 transposed): where a down-sampling conv has 0.2-1 outputs per input
    // the table is a few tens of MB (Infinity Cache) instead of half a GB. The output count decides the
    // rest of the work (and whether the table was large enough), so the stream is synchronised here once:
    // the caller needs that count on the host before it can use the outputs anyway.
    int32_t* h_cnt = (int32_t*)thread_pinned_page();
    int32_t h_pageable[2];
    if (!h_cnt) h_cnt = h_pageable;
    const uint32_t H_big = p.H;
    uint32_t H = pow2_at_least((uint64_t)V * (transpose ? 16 : 4));
    if (H > H_big) H = H_big;
    int U = 0;
    for (int attempt = 0;; ++attempt) {
      init(H, attempt == 0);
      insert_outputs_packed_kernel<<<nb, kBlock, 0, s>>>(idx, (int)V, g, p.keys, H - 1, p.lslot, p.counter, p.masks,
                                                         p.blockcnt, nb);
      PCS_CHECK_HIP(hipMemcpyAsync(h_cnt, p.counter, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      PCS_CHECK_HIP(hipStreamSynchronize(s));
      U = h_cnt[0];
      const bool crowded = h_cnt[1] != 0 || (uint64_t)U * 10 > (uint64_t)H * 7;
      if (!crowded || H == H_big) {
        if (h_cnt[1] != 0) return PCS_ERR_RANGE;  // (cannot happen: H_big holds two slots per possible output)
        break;
      }
      H = H_big;
    }
    const uint32_t hm = H - 1;
    if (U > 0) {
      list_outputs_packed_kernel<<<grid_for(U), kBlock, 0, s>>>(p.keys, p.lslot, U, p.lseq);
      int bits = 1;
      while (bits < 32 && ((uint64_t)V * g.kv) >> bits) ++bits;  // the sequences are below V * kv
      int rc = sort_pairs_u32(p.tmp, p.tmp_bytes, p.lseq, p.lseq_s, p.lslot, p.lslot_s, U, 0, bits, s);
      if (rc) return rc;
    }
    assign_outputs_packed_kernel<<<grid_for(U > 0 ? U : 1), kBlock, 0, s>>>(p.lslot_s, U, p.keys, g, p.vals, out_ids,
                                                                          d_num_out);
    if (nscan <= kScanOneWg) {
      scan_one_wg_kernel<<<1, kScanWg, 0, s>>>(p.blockcnt, p.blockoff, (int)nscan);
    } else {
      int rc = exclusive_scan_i32(p.tmp, p.tmp_bytes, p.blockcnt, p.blockoff, nscan, s);
      if (rc) return rc;
    }
    pair_fill_packed_kernel<<<nb, kBlock, 0, s>>>(idx, (int)V, g, p.keys, p.vals, hm, p.masks, p.blockoff, nb, pairs,
                                                  num);
    PCS_LAUNCH_CHECK();
    if (h_num_out) *h_num_out = U;
    return PCS_OK;
  }
  if (subm && h_num_out) *h_num_out = V;
  const bool with_table = table && pcs_indice_pairs_gives_table(ksize3, dil3, subm);
  const bool sub333 = with_table && g.kv == 27 && g.ks[0] == 3 && g.ks[1] == 3 && vol * (uint64_t)batch < 0xFFFFFFFFULL;
  init(p.H, true);
  if (sub333) {
    insert_inputs_packed_kernel<<<nb, kBlock, 0, s>>>(idx, (int)V, g, p.keys, hmask);
  } else if (subm) {
    insert_inputs_kernel<<<nb, kBlock, 0, s>>>(idx, (int)V, g, p.keys, p.vals, hmask);
  } else {
    insert_outputs_kernel<<<nb, kBlock, 0, s>>>(idx, (int)V, g, p.keys, p.minseq, hmask, p.lslot,
                                                p.counter);
    list_outputs_kernel<<<grid_for(p.cap), kBlock, 0, s>>>(p.minseq, p.lslot, p.counter, p.lseq);
    int rc = sort_pairs_u32(p.tmp, p.tmp_bytes, p.lseq, p.lseq_s, p.lslot, p.lslot_s, p.cap, 0, 32, s);
    if (rc) return rc;
    assign_outputs_kernel<<<grid_for(p.cap), kBlock, 0, s>>>(p.lslot_s, p.counter, p.keys, g,
                                                             p.vals, out_ids, d_num_out);
  }
  if (sub333)
    pair_count_table333_kernel<<<nb, kBlock, 0, s>>>(idx, (int)V, g, p.keys, hmask, p.masks, p.blockcnt, nb, table);
  else if (with_table)
    pair_count_table_kernel<<<nb, kBlock, 0, s>>>(idx, (int)V, g, p.keys, p.vals, hmask, p.masks, p.blockcnt, nb, table);
  else
    pair_count_kernel<<<nb, kBlock, 0, s>>>(idx, (int)V, g, p.keys, hmask, p.masks, p.blockcnt, nb);
  if (nscan <= kScanOneWg) {
    scan_one_wg_kernel<<<1, kScanWg, 0, s>>>(p.blockcnt, p.blockoff, (int)nscan);
  } else {
    int rc = exclusive_scan_i32(p.tmp, p.tmp_bytes, p.blockcnt, p.blockoff, nscan, s);
    if (rc) return rc;
  }
  if (with_table)
    pair_fill_table_kernel<<<nb, kBlock, 0, s>>>((int)V, g, p.keys, p.vals, hmask, idx, p.masks, p.blockoff, nb, table,
                                                 pairs, num, sub333 ? 1 : 0);
  else
    pair_fill_kernel<<<nb, kBlock, 0, s>>>(idx, (int)V, g, p.keys, p.vals, hmask, p.masks,
                                           p.blockoff, nb, pairs, num);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

}  // extern "C"
