// Rulebook-indexed sparse convolution (spconv "indice conv") for gfx950, fp32.
//
// Reference: pcdet/ops/spconv/include/spconv/spconv_ops.h:260-361 (indiceConv<T>): per kernel
// offset k, gather nHot input rows -> torch::mm with W[k] -> scatter-add into the output
// (26-27 gather/GEMM/scatter triples and one D->H sync per layer), and
// fused_spconv_ops.h:29-132 (same loop with the output pre-filled by bias).
//
// Here the rulebook pairs are first inverted into an output-major neighbour table
// T[out][k] = in (or -1): every output row appears at most once per offset (that is what makes
// the reference's non-atomic scatter-add legal, reordering.cu.h:147-156), so T is a plain
// scatter. The conv is then ONE output-stationary kernel: a 64-row output tile accumulates
// sum_k feat[T[o][k]] . W[k] with the gathered rows and W[k] staged in LDS; no atomics, no
// intermediate buffers, fixed accumulation order (subM: centre offset first, then k ascending;
// input channel ascending) -> bitwise reproducible. Bias / folded BN / ReLU run in the epilogue.
#include "pcs_common.hpp"
#include "pcs_wgsort.hpp"

namespace pcs {

__global__ void __launch_bounds__(kBlock)
build_table_kernel(const int32_t* __restrict__ pairs, int64_t pair_stride,
                   const int32_t* __restrict__ num, int kv, int inverse, int64_t num_out,
                   int32_t* __restrict__ table) {
  int k = blockIdx.y;
  int n = num[k];
  const int32_t* pin = pairs + ((int64_t)k * 2 + (inverse ? 1 : 0)) * pair_stride;
  const int32_t* pout = pairs + ((int64_t)k * 2 + (inverse ? 0 : 1)) * pair_stride;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    int o = pout[p];
    if (o >= 0 && o < num_out) table[(int64_t)o * kv + k] = pin[p];
  }
}

constexpr int TM = 64;       // output rows per workgroup
constexpr int CIN_MAX = 64;  // input channels staged per offset

// NC = Cout / 16 output columns per thread; thread (ty, tx) owns rows ty*4..ty*4+3 and
// columns tx + 16*n.
template <int NC>
__global__ void __launch_bounds__(kBlock)
indice_conv_tile_kernel(const float* __restrict__ feat, int cin, const float* __restrict__ W,
                        int kv, int centre, const int32_t* __restrict__ table, int64_t num_out,
                        const float* __restrict__ bias, const float* __restrict__ scale,
                        const float* __restrict__ shift, int relu, const float* __restrict__ residual,
                        float* __restrict__ out) {
  constexpr int COUT = NC * 16;
  __shared__ float As[TM][CIN_MAX + 1];
  __shared__ float Ws[CIN_MAX][COUT];
  __shared__ int Tk[TM];
  int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  int64_t row0 = (int64_t)blockIdx.x * TM;
  float acc[4][NC];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[i][n] = 0.f;

  for (int step = 0; step < kv; ++step) {
    // subM: centre offset first (spconv_ops.h:300-303), then the others ascending
    int k = step;
    if (centre >= 0) k = (step == 0) ? centre : (step <= centre ? step - 1 : step);
    int any = 0;
    if (tid < TM) {
      int64_t r = row0 + tid;
      int t = (r < num_out) ? table[r * kv + k] : -1;
      Tk[tid] = t;
      any = t >= 0;
    }
    any = __syncthreads_or(any);
    if (!any) continue;  // uniform: no row of this tile has a partner at offset k
    for (int e = tid; e < TM * cin; e += kBlock) {
      int r = e / cin, c = e - r * cin;
      int in = Tk[r];
      As[r][c] = in >= 0 ? feat[(int64_t)in * cin + c] : 0.f;
    }
    const float* Wk = W + (int64_t)k * cin * COUT;
    for (int e = tid; e < cin * COUT; e += kBlock) Ws[e / COUT][e % COUT] = Wk[e];
    __syncthreads();
    for (int c = 0; c < cin; ++c) {
      float a[4], b[NC];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[ty * 4 + i][c];
#pragma unroll
      for (int n = 0; n < NC; ++n) b[n] = Ws[c][tx + 16 * n];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < NC; ++n) acc[i][n] = fmaf(a[i], b[n], acc[i][n]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int64_t r = row0 + ty * 4 + i;
    if (r >= num_out) continue;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      int col = tx + 16 * n;
      float v = acc[i][n];
      if (bias) v = v + bias[col];
      if (scale) v = fmaf(v, scale[col], shift[col]);
      if (residual) v = v + residual[r * COUT + col];
      if (relu) v = fmaxf(v, 0.f);
      out[r * COUT + col] = v;
    }
  }
}


// MFMA variant (Cin % 4 == 0, Cout in {16,32,64,128}): same 64-row output tile and LDS staging,
// the inner product runs on the matrix cores with v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate:
// bit-for-bit a k-ordered fmaf chain, so results equal the FMA kernel's). Wave w owns rows
// 16w..16w+15 of the tile and all Cout columns (NC accumulators of 4 VGPRs).
// Fragment maps (cdna_hip_programming.md 3): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// D: col = l&15, row = (l>>4)*4 + reg.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int A_LD = CIN_MAX + 4;  // row stride 68: 4*row + k spreads the 16x4 fragment over banks

template <int NC, bool WIDE = false>
__global__ void __launch_bounds__(kBlock)
indice_conv_mfma_kernel(const float* __restrict__ feat, int cin, const float* __restrict__ W,
                        int kv, int centre, const int32_t* __restrict__ table, int64_t num_out,
                        const float* __restrict__ bias, const float* __restrict__ scale,
                        const float* __restrict__ shift, int relu, const float* __restrict__ residual,
                        float* __restrict__ out) {
  constexpr int COUT = NC * 16;
  constexpr int W_LD = COUT + 16;  // k and k+1 land 16 banks apart: conflict-free 32-lane groups
  constexpr int NA = TM * CIN_MAX / 4 / kBlock;     // float4 A loads per thread at Cin = 64
  constexpr int NW = CIN_MAX * COUT / 4 / kBlock;   // float4 W loads per thread at Cin = 64
  __shared__ __align__(16) float As[TM][A_LD];
  __shared__ __align__(16) float Ws[CIN_MAX][W_LD];
  __shared__ int Tt[TM][32];   // the tile's rows of the neighbour table (kv <= 32)
  __shared__ int steps[32];    // active offsets, accumulation order (centre first for subM)
  __shared__ int nsteps;
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int fi = lane & 15, fk = lane >> 4;
  int64_t row0 = (int64_t)blockIdx.x * TM;
  for (int e = tid; e < TM * kv; e += kBlock) {
    int r = e / kv, k = e - r * kv;
    int64_t gr = row0 + r;
    Tt[r][k] = gr < num_out ? table[gr * kv + k] : -1;
  }
  __syncthreads();
  if (tid < 64) {  // one wave: which offsets have a partner in this tile, in accumulation order
    int n = 0;
    for (int step = 0; step < kv; ++step) {
      int k = step;
      if (centre >= 0) k = (step == 0) ? centre : (step <= centre ? step - 1 : step);
      bool any = __ballot(Tt[tid][k] >= 0) != 0;
      if (any) { if (tid == 0) steps[n] = k; ++n; }
    }
    if (tid == 0) nsteps = n;
  }
  __syncthreads();
  // more than CIN_MAX input channels: an offset is staged in slices of CIN_MAX channels (a step is an
  // (offset, slice) pair; up to CIN_MAX channels the loop below is what it always was)
  const int slices = WIDE ? (cin + CIN_MAX - 1) / CIN_MAX : 1;
  int ns = nsteps * slices;
  int cs = 0, cin4 = 0, a_cnt = 0, w_cnt = 0;  // channels / float4 counts of the slice being staged
  float4 ra[NA], rw[NW];
  auto prefetch = [&](int step) {
    const int k = steps[step / slices], h = step - (step / slices) * slices;
    const int ch = min(CIN_MAX, cin - h * CIN_MAX), ch4 = ch >> 2;
    const int na = TM * ch4, nw = ch * COUT / 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int e = tid + i * kBlock;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < na) {
        int r = e / ch4, c4 = e - r * ch4;
        int in = Tt[r][k];
        if (in >= 0) v = *(const float4*)(feat + (int64_t)in * cin + h * CIN_MAX + c4 * 4);
      }
      ra[i] = v;
    }
    const float4* Wk = (const float4*)(W + ((int64_t)k * cin + h * CIN_MAX) * COUT);
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      int e = tid + i * kBlock;
      rw[i] = e < nw ? Wk[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  f32x4 acc[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (ns > 0) prefetch(0);
  for (int si = 0; si < ns; ++si) {
    // stage the prefetched tiles, then put the next step's loads in flight under the MFMAs
    {
      const int h = si - (si / slices) * slices;
      cs = min(CIN_MAX, cin - h * CIN_MAX);
      cin4 = cs >> 2;
      a_cnt = TM * cin4;
      w_cnt = cs * COUT / 4;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int e = tid + i * kBlock;
      if (e < a_cnt) { int r = e / cin4, c4 = e - r * cin4; *(float4*)&As[r][c4 * 4] = ra[i]; }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      int e = tid + i * kBlock;
      if (e < w_cnt) { int r = (e * 4) / COUT, c = (e * 4) - r * COUT; *(float4*)&Ws[r][c] = rw[i]; }
    }
    __syncthreads();
    if (si + 1 < ns) prefetch(si + 1);
    for (int c0 = 0; c0 < cs; c0 += 4) {
      float a = As[wv * 16 + fi][c0 + fk];
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        float b = Ws[c0 + fk][n * 16 + fi];
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    int col = n * 16 + fi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t r = row0 + wv * 16 + fk * 4 + j;
      if (r >= num_out) continue;
      float v = acc[n][j];
      if (bias) v = v + bias[col];
      if (scale) v = fmaf(v, scale[col], shift[col]);
      if (residual) v = v + residual[r * COUT + col];
      if (relu) v = fmaxf(v, 0.f);
      out[r * COUT + col] = v;
    }
  }
}

// ---- round 5: the gathered rows go straight into the MFMA operand registers (16, 32 or 64 input channels: every
// matrix-core layer of VoxelBackBone8x). v_mfma_f32_16x16x4_f32 takes A[i = l & 15][k = l >> 4]: four lanes share a
// row and each supplies ONE channel per instruction. Which channel is free as long as A and B agree: lane group g of
// a 16-channel chunk loads the row's float4 at channels 4 g .. 4 g + 3 (one 16-byte load) and the chunk's four MFMAs
// take its components one after the other -- instruction j multiplies channels {4 g + j}. The weights of KW offsets
// sit in LDS (36 KB; rows permuted so that the four lane groups read neighbouring rows, 8 banks apart), the four
// waves of a workgroup (128 output rows, 32 each) run through a round's offsets without meeting, two offsets'
// table entries and row pieces in flight at a time, and a wave skips an offset none of its rows has. Against the
// staged 64-row form above: no LDS traffic for the rows, a quarter of the weight bytes per output row, two
// barriers per round instead of two per offset. The products are the same, their order inside a 16-channel chunk is
// not: results agree with the staged form to fp32 rounding (both are pinned to the oracle by tolerance).
constexpr int kDirF32Block = 256;
constexpr int kDirF32Rows = 128;
constexpr int kDirF32W = 9216;  // floats of the weights' LDS buffer (36 KB)
template <int NC>
__global__ void __launch_bounds__(kDirF32Block)
indice_conv_f32_direct_kernel(const float* __restrict__ feat, int cin, const float* __restrict__ W, int kv,
                              int centre, const int32_t* __restrict__ table, int64_t num_out,
                              const float* __restrict__ bias, const float* __restrict__ scale,
                              const float* __restrict__ shift, int relu, const float* __restrict__ residual,
                        float* __restrict__ out,
                              const int32_t* __restrict__ order) {
  constexpr int COUT = NC * 16;
  constexpr int MT = 2;
  constexpr int WLD = COUT + 8;
  __shared__ __align__(16) float Wl[kDirF32W];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * kDirF32Rows + wv * 32;
  const int C16 = cin >> 4;  // 16-channel chunks (1, 2 or 4)
  int KW = min(kv, kDirF32W / (cin * WLD));
  if (KW > 8) KW = 8;
  if (KW > 1) KW &= ~1;
  const int rounds = (kv + KW - 1) / KW;
  auto kmap = [&](int step) {  // accumulation order: the centre first for subM
    return centre >= 0 ? (step == 0 ? centre : (step <= centre ? step - 1 : step)) : step;
  };
  // the fragment rows: positions row0 .. of `order` (rows sorted by their set of occupied offsets, see
  // pcs_neighbor_table_order: a fragment's sixteen rows then want the same offsets) or the rows themselves
  int64_t rm[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int64_t pos = row0 + m * 16 + fi;
    rm[m] = pos < num_out ? (order ? (int64_t)order[pos] : pos) : num_out;
    if ((uint64_t)rm[m] >= (uint64_t)num_out) rm[m] = num_out;  // (an entry that is no row: nothing read)
  }
  f32x4 acc[MT][NC];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int w4 = cin * (COUT / 4);  // float4 pieces of one offset's weights
  for (int rd = 0; rd < rounds; ++rd) {
    const int nk = min(KW, kv - rd * KW);
    // the round's weights: channel ch = 16 c + 4 g + j goes to LDS row 16 c + 4 j + g
    for (int e = tid; e < nk * w4; e += kDirF32Block) {
      const int kw = e / w4, r = e - kw * w4;
      const int ch = r / (COUT / 4), c4 = r - ch * (COUT / 4);
      const float4 v = *(const float4*)(W + ((int64_t)kmap(rd * KW + kw) * cin + ch) * COUT + c4 * 4);
      const int lr = (ch & ~15) | ((ch & 3) << 2) | ((ch >> 2) & 3);
      *(float4*)&Wl[(size_t)(kw * cin + lr) * WLD + c4 * 4] = v;
    }
    __syncthreads();
    for (int kb = 0; kb < nk; kb += 2) {
      int in[2][MT];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const bool on = kb + g < nk;
        const int k = kmap(on ? rd * KW + kb + g : 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) in[g][m] = (on && rm[m] < num_out) ? table[rm[m] * kv + k] : -1;
      }
      float4 a[2][MT][4];
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in[g][m] >= 0 && c < C16) v = *(const float4*)(feat + (int64_t)in[g][m] * cin + c * 16 + fg * 4);
            a[g][m][c] = v;
          }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        bool on_m[MT];  // a 16-row fragment none of whose rows has this offset adds nothing
        bool any = false;
#pragma unroll
        for (int m = 0; m < MT; ++m) { on_m[m] = __ballot(in[g][m] >= 0) != 0; any = any || on_m[m]; }
        if (!any) continue;
        const float* Wt = Wl + (size_t)(kb + g) * cin * WLD;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c < C16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int n = 0; n < NC; ++n) {
                const float b = Wt[(c * 16 + j * 4 + fg) * WLD + n * 16 + fi];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                  const float av = j == 0 ? a[g][m][c].x : (j == 1 ? a[g][m][c].y : (j == 2 ? a[g][m][c].z : a[g][m][c].w));
                  if (on_m[m]) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[m][n], 0, 0, 0);
                }
              }
            }
          }
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int col = n * 16 + fi;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t pos = row0 + m * 16 + fg * 4 + j;
        if (pos >= num_out) continue;
        const int64_t r = order ? (int64_t)order[pos] : pos;
        if ((uint64_t)r >= (uint64_t)num_out) continue;
        float v = acc[m][n][j];
        if (bias) v = v + bias[col];
        if (scale) v = fmaf(v, scale[col], shift[col]);
        if (residual) v = v + residual[r * COUT + col];
        if (relu) v = fmaxf(v, 0.f);
        out[r * COUT + col] = v;
      }
    }
  }
}

// a few input channels (the backbone's first conv: 5 -> 16): one THREAD per output row, its COUT accumulators in
// registers, the filters in LDS (every thread of a wave reads the same word: a broadcast), three offsets' table entries
// and partner channels requested together. The tile kernel stages 64 rows x 27 offsets x 5 channels through LDS with
// two barriers per offset for 80 multiply-adds per row and offset. Same order of operations per element (offsets
// centre-first, channels ascending, one fmaf each): the same bits as the tile / generic kernels.
template <int COUT>
__global__ void __launch_bounds__(kBlock)
indice_conv_f32_rows_kernel(const float* __restrict__ feat, int cin, const float* __restrict__ W, int kv,
                            int centre, const int32_t* __restrict__ table, int64_t num_out,
                            const float* __restrict__ bias, const float* __restrict__ scale,
                            const float* __restrict__ shift, int relu, const float* __restrict__ residual,
                        float* __restrict__ out) {
  constexpr int kMaxW = 32 * 8 * COUT;  // kv <= 32, cin <= 8
  __shared__ float Wf[kMaxW];
  for (int e = threadIdx.x; e < kv * cin * COUT; e += kBlock) Wf[e] = W[e];
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= num_out) return;
  float acc[COUT];
#pragma unroll
  for (int j = 0; j < COUT; ++j) acc[j] = 0.f;
  for (int s0 = 0; s0 < kv; s0 += 3) {
    int in[3], kk[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int step = s0 + q;
      int k = step;
      if (centre >= 0) k = (step == 0) ? centre : (step <= centre ? step - 1 : step);
      kk[q] = k;
      in[q] = step < kv ? table[r * kv + k] : -1;
    }
    float fv[3][8];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int c = 0; c < 8; ++c) fv[q][c] = (in[q] >= 0 && c < cin) ? feat[(int64_t)in[q] * cin + c] : 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (in[q] < 0) continue;
      const float* w = Wf + kk[q] * cin * COUT;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < cin) {
#pragma unroll
          for (int j = 0; j < COUT; ++j) acc[j] = fmaf(fv[q][c], w[c * COUT + j], acc[j]);
        }
      }
    }
  }
  float* o = out + r * COUT;
#pragma unroll
  for (int j4 = 0; j4 < COUT / 4; ++j4) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = j4 * 4 + j;
      float x = acc[col];
      if (bias) x = x + bias[col];
      if (scale) x = fmaf(x, scale[col], shift[col]);
      if (residual) x = x + residual[r * COUT + col];
      if (relu) x = fmaxf(x, 0.f);
      v[j] = x;
    }
    *(float4*)(o + j4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// any (Cin, Cout): one thread per output element, same accumulation order
__global__ void __launch_bounds__(kBlock)
indice_conv_generic_kernel(const float* __restrict__ feat, int cin, const float* __restrict__ W,
                           int kv, int cout, int centre, const int32_t* __restrict__ table,
                           int64_t num_out, const float* __restrict__ bias,
                           const float* __restrict__ scale, const float* __restrict__ shift,
                           int relu, const float* __restrict__ residual, float* __restrict__ out) {
  int64_t total = num_out * cout;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = t / cout;
    int col = (int)(t - r * cout);
    float acc = 0.f;
    for (int step = 0; step < kv; ++step) {
      int k = step;
      if (centre >= 0) k = (step == 0) ? centre : (step <= centre ? step - 1 : step);
      int in = table[r * kv + k];
      if (in < 0) continue;
      const float* f = feat + (int64_t)in * cin;
      const float* w = W + ((int64_t)k * cin) * cout + col;
      for (int c = 0; c < cin; ++c) acc = fmaf(f[c], w[(int64_t)c * cout], acc);
    }
    if (bias) acc = acc + bias[col];
    if (scale) acc = fmaf(acc, scale[col], shift[col]);
    if (residual) acc = acc + residual[t];
    if (relu) acc = fmaxf(acc, 0.f);
    out[t] = acc;
  }
}

// pcs_neighbor_table_order: ONE workgroup per block of kOrderBlock consecutive rows sorts the block's rows (stably)
// by the bit set of their occupied offsets -- word = set << 12 | row in block, wg_radix_sort_packed on the set's kv bits.
// Rows of a block are neighbours in space (the tables are in voxel order), so the gathers of a reordered tile still
// meet in the caches, which a sort over all rows gave away (measured: 13.1 -> 15.2 ms per forward).
constexpr int kOrderBlock = 4096;
__global__ void __launch_bounds__(kWg)
table_order_kernel(const int32_t* __restrict__ table, int64_t num_out, int kv, uint64_t* __restrict__ w0,
                   uint64_t* __restrict__ w1, int32_t* __restrict__ order) {
  __shared__ int hist[16 * 256], hist2[16 * 256];
  __shared__ int sh[kWg / 64 + 1];
  const int64_t base = (int64_t)blockIdx.x * kOrderBlock;
  const int n = (int)min((int64_t)kOrderBlock, num_out - base);
  uint64_t* k0 = w0 + base;
  uint64_t* k1 = w1 + base;
  const int tid = threadIdx.x;
  for (int q = tid; q < 16 * 256; q += kWg) hist[q] = 0;
  __syncthreads();
  const int per_wave = ((n + kWg - 1) / kWg) * 64;
  for (int i = tid; i < n; i += kWg) {
    uint32_t m = 0u;
    const int32_t* t = table + (base + i) * kv;
    for (int k = 0; k < kv; ++k) m |= (t[k] >= 0 ? 1u : 0u) << k;
    k0[i] = ((uint64_t)m << 12) | (uint64_t)i;
    atomicAdd(&hist[(i / per_wave) * 256 + (int)(m & 255u)], 1);
  }
  __syncthreads();
  const int which = wg_radix_sort_packed(k0, k1, n, 12, kv, hist, hist2, sh);
  const uint64_t* sw = which ? k1 : k0;
  for (int i = tid; i < n; i += kWg) order[base + i] = (int32_t)(base + (int64_t)(sw[i] & 4095ULL));
}

static int g_use_mfma = 1;

// eval-mode BatchNorm1d as y = x * scale + shift from inv_std = rsqrt(var + eps) (left to the tensor
// library: its rsqrt is not the device's rsqrtf bit for bit): the three to five elementwise tensor ops
// that follow, each rounded to fp32, nothing contracted
__global__ void bn_fold_kernel(const float* __restrict__ weight, const float* __restrict__ bias,
                               const float* __restrict__ mean, const float* __restrict__ inv_std,
                               int C, float* __restrict__ scale, float* __restrict__ shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float inv = inv_std[c];
  float sc = weight ? __fmul_rn(inv, weight[c]) : inv;
  float sh = __fmul_rn(-mean[c], sc);
  if (bias) sh = __fadd_rn(sh, bias[c]);
  scale[c] = sc;
  shift[c] = sh;
}

}  // namespace pcs

using namespace pcs;

extern "C" {

// 1 (default): layers with Cin >= 16, Cin % 4 == 0 run on v_mfma_f32_16x16x4_f32 (16 / 32 / 64 input channels: rows
// gathered straight into the operand registers, indice_conv_f32_direct_kernel); 2: the matrix cores on round 4's
// staged 64-row tiles everywhere; 3: the direct-gather form whatever the number of rows (1 keeps the staged tiles
// below 100 000 rows); 0: FMA tiles everywhere (for A/B measurements and cross-checks).
int pcs_conv_use_mfma(int enable) {
  g_use_mfma = (enable == 2 || enable == 3) ? enable : (enable != 0);
  return PCS_OK;
}

// scale / shift [C] of an eval-mode BatchNorm1d (weight / bias may be null) in one launch
int pcs_bn_fold_f32(const float* weight, const float* bias, const float* running_mean,
                    const float* inv_std, int C, float* scale, float* shift, hipStream_t s) {
  if (!running_mean || !inv_std || !scale || !shift || C < 0) return PCS_ERR_ARG;
  if (C == 0) return PCS_OK;
  bn_fold_kernel<<<grid_for(C), kBlock, 0, s>>>(weight, bias, running_mean, inv_std, C, scale, shift);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

size_t pcs_neighbor_table_bytes(int64_t num_out, int kv) {
  return align_up((size_t)(num_out > 0 ? num_out : 1) * kv * sizeof(int32_t));
}

// table [num_out, kv] int32, -1 where the output has no partner at that offset
int pcs_build_neighbor_table(const int32_t* pairs, int64_t pair_stride, const int32_t* num, int kv,
                             int inverse, int64_t num_out, int32_t* table, hipStream_t s) {
  if (!pairs || !num || !table || kv < 1 || num_out < 0 || pair_stride < 0) return PCS_ERR_ARG;
  if (num_out == 0) return PCS_OK;
  PCS_CHECK_HIP(hipMemsetAsync(table, 0xFF, (size_t)num_out * kv * sizeof(int32_t), s));
  if (pair_stride == 0) return PCS_OK;
  dim3 grid(grid_for(pair_stride), kv);
  build_table_kernel<<<grid, kBlock, 0, s>>>(pairs, pair_stride, num, kv, inverse, num_out, table);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// out[o,:] = epilogue( sum_k feat[table[o,k],:] . W[k] ), W = filters viewed [kv, cin, cout].
// centre >= 0 selects the offset accumulated first (subM); -1 = plain ascending k.
// epilogue: (+bias) then (*scale + shift) then relu, each optional (null / 0).
static int conv_table_f32(const float* features, int cin, const float* filters, int kv,
                          int cout, int centre, const int32_t* table, int64_t num_out,
                          const float* bias, const float* scale, const float* shift, int relu,
                          float* out, const int32_t* order, hipStream_t s, const float* residual = nullptr) {
  if (!features || !filters || !table || !out || cin < 1 || cout < 1 || kv < 1 || num_out < 0 ||
      centre >= kv || (scale && !shift))
    return PCS_ERR_ARG;
  if (num_out == 0) return PCS_OK;
  int64_t nb = (num_out + TM - 1) / TM;
  if (nb >= (1LL << 31)) return PCS_ERR_RANGE;
  bool cout_ok = cout == 16 || cout == 32 || cout == 64 || cout == 128;
  bool tiled = cin <= CIN_MAX && cout_ok;
  // matrix cores where the tile is dense enough to feed them: 4-aligned input channels (the
  // first layer, Cin = 3..5, stays on the FMA path: it is bandwidth-bound); wider
  // inputs (the 128-channel residual stage, the input gradient of 64 -> 128) in slices of CIN_MAX
  // (the matrix-core kernel stages a tile's rows of the neighbour table in LDS, 32 offsets wide:
  //  wider kernels -- 5x5x5, 3x3x5 ... from an externally built rulebook -- take the tile / generic
  //  kernels, which read the table from memory)
  // (one frame's layers -- tens of thousands of rows -- do not fill the chip with 128-row workgroups that each walk
  //  all 27 offsets: there the 64-row staged tiles stay ahead, 1.92 against 2.13 ms per forward at B = 1)
  if (cout_ok && (g_use_mfma == 3 || (g_use_mfma == 1 && num_out >= 100000)) && kv <= 32 &&
      (cin == 16 || cin == 32 || cin == 64)) {
    const int nbd = (int)((num_out + kDirF32Rows - 1) / kDirF32Rows);
#define PCS_F32_DIRECT(NCV)                                                                                       \
  indice_conv_f32_direct_kernel<NCV><<<nbd, kDirF32Block, 0, s>>>(features, cin, filters, kv, centre, table, num_out, \
                                                                 bias, scale, shift, relu, residual, out, order)
    switch (cout) {
      case 16: PCS_F32_DIRECT(1); break;
      case 32: PCS_F32_DIRECT(2); break;
      case 64: PCS_F32_DIRECT(4); break;
      default: PCS_F32_DIRECT(8); break;
    }
#undef PCS_F32_DIRECT
    PCS_LAUNCH_CHECK();
    return PCS_OK;
  }
  if (cout_ok && g_use_mfma && kv <= 32 && (cin % 4 == 0) && cin >= 16 && cin <= 8 * CIN_MAX) {
#define PCS_MFMA_LAUNCH(NCV, WIDEV)                                                                   \
  indice_conv_mfma_kernel<NCV, WIDEV><<<(int)nb, kBlock, 0, s>>>(features, cin, filters, kv, centre, \
                                                                 table, num_out, bias, scale, shift, relu, residual, out)
    const bool wide = cin > CIN_MAX;
    switch (cout) {
      case 16:
        if (wide) PCS_MFMA_LAUNCH(1, true); else PCS_MFMA_LAUNCH(1, false);
        break;
      case 32:
        if (wide) PCS_MFMA_LAUNCH(2, true); else PCS_MFMA_LAUNCH(2, false);
        break;
      case 64:
        if (wide) PCS_MFMA_LAUNCH(4, true); else PCS_MFMA_LAUNCH(4, false);
        break;
      default:
        if (wide) PCS_MFMA_LAUNCH(8, true); else PCS_MFMA_LAUNCH(8, false);
        break;
    }
#undef PCS_MFMA_LAUNCH
    PCS_LAUNCH_CHECK();
    return PCS_OK;
  }
  if (cin <= 8 && kv <= 32 && (cout == 16 || cout == 32) && g_use_mfma != 2) {
    const int nbr = (int)((num_out + kBlock - 1) / kBlock);
    if (cout == 16)
      indice_conv_f32_rows_kernel<16><<<nbr, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, scale, shift, relu, residual, out);
    else
      indice_conv_f32_rows_kernel<32><<<nbr, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, scale, shift, relu, residual, out);
    PCS_LAUNCH_CHECK();
    return PCS_OK;
  }
  if (tiled) {
    switch (cout) {
      case 16:
        indice_conv_tile_kernel<1><<<(int)nb, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, scale, shift, relu, residual, out);
        break;
      case 32:
        indice_conv_tile_kernel<2><<<(int)nb, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, scale, shift, relu, residual, out);
        break;
      case 64:
        indice_conv_tile_kernel<4><<<(int)nb, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, scale, shift, relu, residual, out);
        break;
      default:
        indice_conv_tile_kernel<8><<<(int)nb, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, scale, shift, relu, residual, out);
        break;
    }
  } else {
    indice_conv_generic_kernel<<<grid_for(num_out * cout), kBlock, 0, s>>>(
        features, cin, filters, kv, cout, centre, table, num_out, bias, scale, shift, relu, residual, out);
  }
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

int pcs_indice_conv_table_f32(const float* features, int cin, const float* filters, int kv,
                              int cout, int centre, const int32_t* table, int64_t num_out,
                              const float* bias, const float* scale, const float* shift, int relu,
                              float* out, hipStream_t s) {
  return conv_table_f32(features, cin, filters, kv, cout, centre, table, num_out, bias, scale, shift, relu, out,
                        nullptr, s);
}

// the same with a residual row added before the ReLU: relu((conv * scale + shift) + residual[o,:]) -- the tail of
// SparseBasicBlock.forward (pcdet/models/blocks/spconv_blocks.py:47-64 of the reference: bn2(conv2(.)) + identity,
// then relu) in the conv's epilogue instead of two more passes over the block's output. residual [num_out, cout],
// may alias nothing the call writes; NULL = pcs_indice_conv_table_f32.
int pcs_indice_conv_table_residual_f32(const float* features, int cin, const float* filters, int kv,
                                       int cout, int centre, const int32_t* table, int64_t num_out,
                                       const float* bias, const float* scale, const float* shift,
                                       const float* residual, int relu, float* out, hipStream_t s) {
  if (residual && residual == out) return PCS_ERR_ARG;
  return conv_table_f32(features, cin, filters, kv, cout, centre, table, num_out, bias, scale, shift, relu, out,
                        nullptr, s, residual);
}

// the same with the output rows taken in the order of pcs_neighbor_table_order (rows with the same set of occupied
// offsets side by side: the matrix-core kernels skip a 16-row fragment at an offset none of its rows has, and in
// table order only every second (fragment, offset) product is of a pair that exists). Every output row is the same
// sum in the same order: the same bits as without `order`. order == NULL: table order.
int pcs_indice_conv_table_ordered_f32(const float* features, int cin, const float* filters, int kv,
                                      int cout, int centre, const int32_t* table, int64_t num_out,
                                      const float* bias, const float* scale, const float* shift, int relu,
                                      float* out, const int32_t* order, hipStream_t s) {
  return conv_table_f32(features, cin, filters, kv, cout, centre, table, num_out, bias, scale, shift, relu, out,
                        order, s);
}

// order [num_out] int32: within every block of 4096 consecutive rows, the rows sorted (stably) by the bit set of their
// occupied offsets; kv <= 32
size_t pcs_neighbor_table_order_workspace_bytes(int64_t num_out) {
  const size_t n = (size_t)(num_out > 0 ? num_out : 1);
  return 2 * align_up(n * sizeof(uint64_t));
}
int pcs_neighbor_table_order(const int32_t* table, int64_t num_out, int kv, int32_t* order, void* ws,
                             size_t ws_bytes, hipStream_t s) {
  if (!table || !order || !ws || num_out < 0 || kv < 1 || kv > 32) return PCS_ERR_ARG;
  if (num_out >= (1LL << 31)) return PCS_ERR_RANGE;
  if (ws_bytes < pcs_neighbor_table_order_workspace_bytes(num_out)) return PCS_ERR_WORKSPACE;
  if (num_out == 0) return PCS_OK;
  Arena a(ws, ws_bytes);
  uint64_t* w0 = a.take<uint64_t>((size_t)num_out);
  uint64_t* w1 = a.take<uint64_t>((size_t)num_out);
  if (!w0 || !w1) return PCS_ERR_WORKSPACE;
  const int64_t nb = (num_out + kOrderBlock - 1) / kOrderBlock;
  table_order_kernel<<<(int)nb, kWg, 0, s>>>(table, num_out, kv, w0, w1, order);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// Reference-shaped entry (sparse_conv_ext.indice_conv_fp32, src/all.cc:30-33): builds the table
// in the workspace, then runs the conv. `subm` selects centre-first accumulation.
int pcs_indice_conv_f32(const float* features, int cin, const float* filters, int kv, int cout,
                        const int32_t* pairs, int64_t pair_stride, const int32_t* num,
                        int64_t num_out, int inverse, int subm, float* out, void* ws,
                        size_t ws_bytes, hipStream_t s) {
  if (!ws || ws_bytes < pcs_neighbor_table_bytes(num_out, kv)) return PCS_ERR_WORKSPACE;
  int rc = pcs_build_neighbor_table(pairs, pair_stride, num, kv, inverse, num_out, (int32_t*)ws, s);
  if (rc) return rc;
  int centre = (subm && (kv & 1)) ? kv / 2 : -1;
  return pcs_indice_conv_table_f32(features, cin, filters, kv, cout, centre, (const int32_t*)ws,
                                   num_out, nullptr, nullptr, nullptr, 0, out, s);
}

}  // extern "C"

// ---- indiceConvBackward (spconv_ops.h:363-449) --------------------------------
// input gradient: the same output-stationary conv, run over the TRANSPOSED rulebook (feature-row
// major table of out_grad rows) with W[k]^T -- no scatter-add, no atomics. filter gradient: per
// offset k, dW[k] = sum over pairs of in[i]^T (x) out_grad[o]; the pairs of an offset are cut into
// chunks of kGradChunk, one workgroup per (chunk, k, 32x32 tile) accumulates its outer products in
// registers and stores a partial; a second kernel adds the partials in ascending chunk order, so
// the result does not depend on scheduling (the reference: one cuBLAS GEMM per offset).
namespace pcs {
constexpr int kGradChunk = 1024;
constexpr int kGT = 32;  // tile edge (input channels x output channels)

__global__ void __launch_bounds__(kBlock)
transpose_filters_kernel(const float* __restrict__ W, int kv, int cin, int cout, float* __restrict__ Wt) {
  int64_t total = (int64_t)kv * cin * cout;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = t / ((int64_t)cin * cout);
    int r = (int)(t - k * cin * cout);
    int ci = r / cout, co = r - ci * cout;
    Wt[(k * cout + co) * cin + ci] = W[t];
  }
}

__global__ void __launch_bounds__(kBlock)
filter_grad_partial_kernel(const float* __restrict__ feat, int cin, const float* __restrict__ og, int cout,
                           const int32_t* __restrict__ pairs, int64_t pair_stride,
                           const int32_t* __restrict__ num, int side_in, int tiles_j,
                           float* __restrict__ partial, int kv) {
  const int chunk = blockIdx.x, k = blockIdx.y;
  const int ti0 = (blockIdx.z / tiles_j) * kGT, tj0 = (blockIdx.z % tiles_j) * kGT;
  const int n = num[k];
  const int p0 = chunk * kGradChunk;
  if (p0 >= n) return;
  const int p1 = min(n, p0 + kGradChunk);
  const int32_t* pin = pairs + ((int64_t)k * 2 + side_in) * pair_stride;
  const int32_t* pout = pairs + ((int64_t)k * 2 + (1 - side_in)) * pair_stride;
  __shared__ float s_in[kGT][kGT + 1];
  __shared__ float s_og[kGT][kGT + 1];
  // thread (ty, tx) owns dW[ti0 + 2 ty + {0,1}][tj0 + 2 tx + {0,1}]
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
  for (int pb = p0; pb < p1; pb += kGT) {
    // stage 32 pairs x 32 channels of both sides (zero beyond the ranges)
    for (int q = threadIdx.x; q < kGT * kGT; q += kBlock) {
      int pr = q >> 5, c = q & 31;
      int p = pb + pr;
      float vi = 0.f, vo = 0.f;
      if (p < p1) {
        int i = pin[p], o = pout[p];
        if (ti0 + c < cin) vi = feat[(int64_t)i * cin + ti0 + c];
        if (tj0 + c < cout) vo = og[(int64_t)o * cout + tj0 + c];
      }
      s_in[pr][c] = vi;
      s_og[pr][c] = vo;
    }
    __syncthreads();
#pragma unroll 8
    for (int pr = 0; pr < kGT; ++pr) {
      float i0 = s_in[pr][2 * ty], i1 = s_in[pr][2 * ty + 1];
      float o0 = s_og[pr][2 * tx], o1 = s_og[pr][2 * tx + 1];
      a00 = fmaf(i0, o0, a00); a01 = fmaf(i0, o1, a01);
      a10 = fmaf(i1, o0, a10); a11 = fmaf(i1, o1, a11);
    }
    __syncthreads();
  }
  float* dst = partial + ((int64_t)chunk * kv + k) * cin * cout;
  int ci = ti0 + 2 * ty, cj = tj0 + 2 * tx;
  if (ci < cin && cj < cout) dst[(int64_t)ci * cout + cj] = a00;
  if (ci < cin && cj + 1 < cout) dst[(int64_t)ci * cout + cj + 1] = a01;
  if (ci + 1 < cin && cj < cout) dst[(int64_t)(ci + 1) * cout + cj] = a10;
  if (ci + 1 < cin && cj + 1 < cout) dst[(int64_t)(ci + 1) * cout + cj + 1] = a11;
}

// The same partial sums on the matrix cores (v_mfma_f32_16x16x4_f32): dW[k] = in^T . out_grad is a
// GEMM with M = input channels, N = output channels, K = pairs. A workgroup owns a (RB*16) x (NC*16)
// tile; its four waves are RB row blocks x (4 / RB) pair slices, the slices summed in LDS in fixed
// order at the end. 32 pairs are staged per round, [pair][channel], row stride + 16 floats so that the
// two 16-lane halves of an LDS access land 16 banks apart.
template <int NC, int RB>
__global__ void __launch_bounds__(kBlock)
filter_grad_mfma_kernel(const float* __restrict__ feat, int cin, const float* __restrict__ og, int cout,
                        const int32_t* __restrict__ pairs, int64_t pair_stride,
                        const int32_t* __restrict__ num, int side_in, int tiles_j,
                        float* __restrict__ partial, int kv) {
  constexpr int PS = 4 / RB, TI = RB * 16, TJ = NC * 16, PB = 32;
  constexpr int LDI = TI + 16, LDJ = TJ + 16;
  const int chunk = blockIdx.x, k = blockIdx.y;
  const int ti0 = (blockIdx.z / tiles_j) * TI, tj0 = (blockIdx.z % tiles_j) * TJ;
  const int n = num[k];
  const int p0 = chunk * kGradChunk;
  if (p0 >= n) return;
  const int p1 = min(n, p0 + kGradChunk);
  const int32_t* pin = pairs + ((int64_t)k * 2 + side_in) * pair_stride;
  const int32_t* pout = pairs + ((int64_t)k * 2 + (1 - side_in)) * pair_stride;
  __shared__ __align__(16) float s_in[PB][LDI];
  __shared__ __align__(16) float s_og[PB][LDJ];
  __shared__ float s_red[PS > 1 ? 4 : 1][NC][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 15, fk = lane >> 4;
  const int rb = wv % RB, ps = wv / RB;
  f32x4 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int pb = p0; pb < p1; pb += PB) {
    for (int e = tid; e < PB * (TI / 4); e += kBlock) {
      int pr = e / (TI / 4), c4 = e - pr * (TI / 4);
      int p = pb + pr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < p1 && ti0 + c4 * 4 < cin) v = *(const float4*)(feat + (int64_t)pin[p] * cin + ti0 + c4 * 4);
      *(float4*)&s_in[pr][c4 * 4] = v;
    }
    for (int e = tid; e < PB * (TJ / 4); e += kBlock) {
      int pr = e / (TJ / 4), c4 = e - pr * (TJ / 4);
      int p = pb + pr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < p1 && tj0 + c4 * 4 < cout) v = *(const float4*)(og + (int64_t)pout[p] * cout + tj0 + c4 * 4);
      *(float4*)&s_og[pr][c4 * 4] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PB / PS; q += 4) {
      int pp = ps * (PB / PS) + q + fk;
      float a = s_in[pp][rb * 16 + fi];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float b = s_og[pp][c * 16 + fi];
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (PS > 1) {
    // pair slices of one row block: slice 0 adds the others in ascending order
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) s_red[wv][c][lane][j] = acc[c][j];
    __syncthreads();
    if (ps != 0) return;
    for (int o = 1; o < PS; ++o)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = acc[c][j] + s_red[o * RB + rb][c][lane][j];
  }
  float* dst = partial + ((int64_t)chunk * kv + k) * cin * cout;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int col = tj0 + c * 16 + fi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int row = ti0 + rb * 16 + fk * 4 + j;
      if (row < cin && col < cout) dst[(int64_t)row * cout + col] = acc[c][j];
    }
  }
}

__global__ void __launch_bounds__(kBlock)
filter_grad_reduce_kernel(const float* __restrict__ partial, const int32_t* __restrict__ num, int kv,
                          int cin, int cout, float* __restrict__ dW) {
  int64_t per = (int64_t)cin * cout;
  int64_t total = (int64_t)kv * per;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int k = (int)(t / per);
    int64_t r = t - (int64_t)k * per;
    int chunks = (num[k] + kGradChunk - 1) / kGradChunk;
    float acc = 0.f;
    for (int c = 0; c < chunks; ++c) acc = acc + partial[((int64_t)c * kv + k) * per + r];
    dW[t] = acc;
  }
}
}  // namespace pcs

extern "C" size_t pcs_indice_conv_backward_workspace_bytes(int64_t num_in, int64_t pair_stride, int kv,
                                                           int cin, int cout) {
  if (num_in < 1) num_in = 1;
  if (pair_stride < 1) pair_stride = 1;
  int64_t chunks = (pair_stride + pcs::kGradChunk - 1) / pcs::kGradChunk;
  size_t b = pcs::align_up((size_t)num_in * kv * sizeof(int32_t));
  b += pcs::align_up((size_t)kv * cin * cout * sizeof(float));
  b += pcs::align_up((size_t)chunks * kv * cin * cout * sizeof(float));
  return b;
}

extern "C" int pcs_indice_conv_backward_f32(const float* features, int cin, const float* filters, int kv,
                                            int cout, const float* out_grad, const int32_t* pairs,
                                            int64_t pair_stride, const int32_t* num, int64_t num_in,
                                            int64_t num_out, int inverse, int subm, float* in_grad,
                                            float* filters_grad, void* ws, size_t ws_bytes, hipStream_t s) {
  using namespace pcs;
  if (!features || !filters || !out_grad || !pairs || !num || !in_grad || !filters_grad || cin < 1 ||
      cout < 1 || kv < 1 || num_in < 0 || num_out < 0 || pair_stride < 0)
    return PCS_ERR_ARG;
  if (!ws || ws_bytes < pcs_indice_conv_backward_workspace_bytes(num_in, pair_stride, kv, cin, cout))
    return PCS_ERR_WORKSPACE;
  PCS_CHECK_HIP(hipMemsetAsync(filters_grad, 0, (size_t)kv * cin * cout * sizeof(float), s));
  if (num_in == 0) return PCS_OK;
  if (num_out == 0 || pair_stride == 0) {
    PCS_CHECK_HIP(hipMemsetAsync(in_grad, 0, (size_t)num_in * cin * sizeof(float), s));
    return PCS_OK;
  }
  char* w = (char*)ws;
  int32_t* table_t = (int32_t*)w;
  w += align_up((size_t)num_in * kv * sizeof(int32_t));
  float* Wt = (float*)w;
  w += align_up((size_t)kv * cin * cout * sizeof(float));
  float* partial = (float*)w;
  // ---- input gradient: in_grad[i] = sum_k out_grad[T'[i][k]] . W[k]^T
  int rc = pcs_build_neighbor_table(pairs, pair_stride, num, kv, inverse ? 0 : 1, num_in, table_t, s);
  if (rc) return rc;
  transpose_filters_kernel<<<grid_for((int64_t)kv * cin * cout), kBlock, 0, s>>>(filters, kv, cin, cout, Wt);
  int centre = (subm && (kv & 1)) ? kv / 2 : -1;
  rc = pcs_indice_conv_table_f32(out_grad, cout, Wt, kv, cin, centre, table_t, num_in, nullptr, nullptr,
                                 nullptr, 0, in_grad, s);
  if (rc) return rc;
  // ---- filter gradient
  int64_t chunks = (pair_stride + kGradChunk - 1) / kGradChunk;
  int tiles_i = (cin + kGT - 1) / kGT, tiles_j = (cout + kGT - 1) / kGT;
  if (chunks >= 65536 || kv >= 65536) return PCS_ERR_RANGE;
  const int rbv = cin >= 64 ? 4 : (cin >= 32 ? 2 : 1), ncv = cout >= 64 ? 4 : (cout >= 32 ? 2 : 1);
  if (g_use_mfma && cin >= 16 && cin % (rbv * 16) == 0 && cout >= 16 && cout % (ncv * 16) == 0) {
    const int mi = cin / (rbv * 16), mj = cout / (ncv * 16);
    dim3 grid((unsigned)chunks, kv, mi * mj);
#define PCS_FG_LAUNCH(NCV, RBV)                                                                       \
  filter_grad_mfma_kernel<NCV, RBV><<<grid, kBlock, 0, s>>>(features, cin, out_grad, cout, pairs,      \
                                                            pair_stride, num, inverse ? 1 : 0, mj, partial, kv)
    if (ncv == 4) { if (rbv == 4) PCS_FG_LAUNCH(4, 4); else if (rbv == 2) PCS_FG_LAUNCH(4, 2); else PCS_FG_LAUNCH(4, 1); }
    else if (ncv == 2) { if (rbv == 4) PCS_FG_LAUNCH(2, 4); else if (rbv == 2) PCS_FG_LAUNCH(2, 2); else PCS_FG_LAUNCH(2, 1); }
    else { if (rbv == 4) PCS_FG_LAUNCH(1, 4); else if (rbv == 2) PCS_FG_LAUNCH(1, 2); else PCS_FG_LAUNCH(1, 1); }
#undef PCS_FG_LAUNCH
  } else {
    filter_grad_partial_kernel<<<dim3((unsigned)chunks, kv, tiles_i * tiles_j), kBlock, 0, s>>>(
        features, cin, out_grad, cout, pairs, pair_stride, num, inverse ? 1 : 0, tiles_j, partial, kv);
  }
  filter_grad_reduce_kernel<<<grid_for((int64_t)kv * cin * cout), kBlock, 0, s>>>(partial, num, kv, cin,
                                                                                 cout, filters_grad);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// ---- indiceMaxPool / indiceMaxPoolBackward (pool_ops.h:26-98, src/maxpool.cc:22-64; row f4) ------
// The reference runs one scatter pass per kernel offset over an output that starts at zero. Over the
// output-major table both directions are gathers: out[o][c] = max(0, max_k in[T[o][k]][c]); the input
// gradient sums, in ascending offset (the reference's order, so the fp32 sum is the same), the
// gradients of the outputs whose pooled value equals the input -- no atomics.
namespace pcs {
__global__ void __launch_bounds__(kBlock)
maxpool_fwd_kernel(const float* __restrict__ feat, int C, const int32_t* __restrict__ table, int kv,
                   int64_t num_out, float* __restrict__ out) {
  int64_t total = num_out * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t o = t / C;
    int c = (int)(t - o * C);
    float acc = 0.f;
    for (int k = 0; k < kv; ++k) {
      int i = table[o * kv + k];
      if (i < 0) continue;
      float v = feat[(int64_t)i * C + c];
      if (acc < v) acc = v;
    }
    out[t] = acc;
  }
}

__global__ void __launch_bounds__(kBlock)
maxpool_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ outf,
                   const float* __restrict__ og, int C, const int32_t* __restrict__ table_t, int kv,
                   int64_t num_in, float* __restrict__ din) {
  int64_t total = num_in * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = t / C;
    int c = (int)(t - i * C);
    float v = feat[t], acc = 0.f;
    for (int k = 0; k < kv; ++k) {
      int o = table_t[i * kv + k];
      if (o < 0) continue;
      if (outf[(int64_t)o * C + c] == v) acc = acc + og[(int64_t)o * C + c];
    }
    din[t] = acc;
  }
}
}  // namespace pcs

extern "C" int pcs_indice_maxpool_f32(const float* features, int C, const int32_t* pairs, int64_t pair_stride,
                                      const int32_t* num, int kv, int64_t num_out, float* out, void* ws,
                                      size_t ws_bytes, hipStream_t s) {
  using namespace pcs;
  if (!features || !pairs || !num || !out || C < 1 || kv < 1 || num_out < 0) return PCS_ERR_ARG;
  if (num_out == 0) return PCS_OK;
  if (!ws || ws_bytes < pcs_neighbor_table_bytes(num_out, kv)) return PCS_ERR_WORKSPACE;
  int rc = pcs_build_neighbor_table(pairs, pair_stride, num, kv, 0, num_out, (int32_t*)ws, s);
  if (rc) return rc;
  maxpool_fwd_kernel<<<grid_for(num_out * C), kBlock, 0, s>>>(features, C, (const int32_t*)ws, kv, num_out, out);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

extern "C" int pcs_indice_maxpool_backward_f32(const float* features, const float* out_features,
                                               const float* out_grad, int C, const int32_t* pairs,
                                               int64_t pair_stride, const int32_t* num, int kv,
                                               int64_t num_in, float* in_grad, void* ws, size_t ws_bytes,
                                               hipStream_t s) {
  using namespace pcs;
  if (!features || !out_features || !out_grad || !pairs || !num || !in_grad || C < 1 || kv < 1 || num_in < 0)
    return PCS_ERR_ARG;
  if (num_in == 0) return PCS_OK;
  if (!ws || ws_bytes < pcs_neighbor_table_bytes(num_in, kv)) return PCS_ERR_WORKSPACE;
  int rc = pcs_build_neighbor_table(pairs, pair_stride, num, kv, 1, num_in, (int32_t*)ws, s);
  if (rc) return rc;
  maxpool_bwd_kernel<<<grid_for(num_in * C), kBlock, 0, s>>>(features, out_features, out_grad, C,
                                                            (const int32_t*)ws, kv, num_in, in_grad);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// ---- SparseConvTensor.dense() + HeightCompression (pcdet/ops/spconv/structure.py:5-18,55-64;
// pcdet/models/backbones_2d/map_to_bev/height_compression.py:10-27): features [V,C] at
// indices [V,4] (b,z,y,x) -> channels-first dense [B,C,D,H,W] (== [B, C*D, H, W] viewed), zeros
// elsewhere. The reference builds [B,D,H,W,C], permutes and copies; here one fill + one scatter.
namespace pcs {
__global__ void __launch_bounds__(kBlock)
sparse_to_dense_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx, int64_t V, int C,
                       int B, int D, int H, int W, float* __restrict__ out, int* __restrict__ status) {
  int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= V * C) return;
  int64_t v = t / C;
  int c = (int)(t - v * C);
  int4 q = ((const int4*)idx)[v];
  if (q.x < 0 || q.x >= B || q.y < 0 || q.y >= D || q.z < 0 || q.z >= H || q.w < 0 || q.w >= W) {
    if (status) atomicExch(status, 1);
    return;
  }
  out[((((int64_t)q.x * C + c) * D + q.y) * H + q.z) * W + q.w] = feat[t];
}
}  // namespace pcs

extern "C" int pcs_sparse_to_dense(const float* features, const int32_t* indices, int64_t V, int C, int B,
                                   int D, int H, int W, float* out, int* status, hipStream_t s) {
  if (V < 0 || C < 1 || B < 1 || D < 1 || H < 1 || W < 1 || !out) return PCS_ERR_ARG;
  if (V > 0 && (!features || !indices)) return PCS_ERR_ARG;
  size_t bytes = (size_t)B * C * D * H * W * sizeof(float);
  PCS_CHECK_HIP(hipMemsetAsync(out, 0, bytes, s));
  if (status) PCS_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int), s));
  if (V == 0) return PCS_OK;
  int64_t n = V * C;
  pcs::sparse_to_dense_kernel<<<(unsigned)((n + pcs::kBlock - 1) / pcs::kBlock), pcs::kBlock, 0, s>>>(
      features, indices, V, C, B, D, H, W, out, status);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// ---- half-precision entry points (sparse_conv_ext.indice_conv_half / fused_indice_conv_half /
// indice_conv_backward_half / indice_maxpool_half / indice_maxpool_backward_half, src/all.cc:35-50) ----
// The reference instantiates its gather - GEMM - scatter-add templates for at::Half: fp16 storage, each
// offset's GEMM result rounded to fp16 and added into the fp16 output. Here the fp16 rows are gathered
// as they are (half the bytes of the fp32 path -- the convs are gather-bound), the products run on the
// matrix cores' fp16 pipe (v_mfma_f32_16x16x32_f16: exact products, fp32 accumulation over ALL offsets and
// channels) and the result is rounded to fp16 ONCE: at least as accurate as the reference's per-offset
// roundings, the storage format of the caller unchanged, no conversion sweeps, no fp32 copies. The pooling
// is exact either way (a maximum of fp16 values is one of them).
#include <hip/hip_fp16.h>
namespace pcs {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int H_LD = CIN_MAX + 8;  // halves per LDS row: 144 B (a multiple of 16 B; rows start 4 banks apart)

// Cin % 16 == 0, Cout in {16, 32, 64, 128}: the f32 kernel's 64-row output tile and staging scheme. The gathered
// rows sit in LDS as [row][channel], W[k] TRANSPOSED as [column][channel], so that a lane's eight consecutive
// k of either operand are one 16-byte LDS read. Fragment maps (cdna_hip_programming.md 3): A[i = l & 15]
// [k = 8 (l >> 4) + j], B[k = 8 (l >> 4) + j][col = l & 15], D: col = l & 15, row = (l >> 4) * 4 + reg.
// Channels are staged in slices of at most CIN_MAX, padded with zeros to a multiple of 32 (the K of one MFMA).
template <int NC>
__global__ void __launch_bounds__(kBlock)
indice_conv_f16_mfma_kernel(const __half* __restrict__ feat, int cin, const __half* __restrict__ W, int kv,
                            int centre, const int32_t* __restrict__ table, int64_t num_out,
                            const __half* __restrict__ bias, int relu, const __half* __restrict__ residual,
                            __half* __restrict__ out) {
  constexpr int COUT = NC * 16;
  __shared__ __align__(16) _Float16 As[TM][H_LD];
  __shared__ __align__(16) _Float16 Wt[COUT][H_LD];
  __shared__ int Tt[TM][32];   // the tile's rows of the neighbour table (kv <= 32)
  __shared__ int steps[32];    // active offsets, accumulation order (centre first for subM)
  __shared__ int nsteps;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 15, fk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TM;
  for (int e = tid; e < TM * kv; e += kBlock) {
    int r = e / kv, k = e - r * kv;
    int64_t gr = row0 + r;
    Tt[r][k] = gr < num_out ? table[gr * kv + k] : -1;
  }
  __syncthreads();
  if (tid < 64) {  // one wave: which offsets have a partner in this tile, in accumulation order
    int n = 0;
    for (int step = 0; step < kv; ++step) {
      int k = step;
      if (centre >= 0) k = (step == 0) ? centre : (step <= centre ? step - 1 : step);
      bool any = __ballot(Tt[tid][k] >= 0) != 0;
      if (any) { if (tid == 0) steps[n] = k; ++n; }
    }
    if (tid == 0) nsteps = n;
  }
  __syncthreads();
  const int slices = (cin + CIN_MAX - 1) / CIN_MAX;
  const int ns = nsteps * slices;
  f32x4 acc[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint4 zero16 = make_uint4(0u, 0u, 0u, 0u);
  for (int si = 0; si < ns; ++si) {
    const int k = steps[si / slices], h = si - (si / slices) * slices;
    const int cs = min(CIN_MAX, cin - h * CIN_MAX);  // channels of this slice (a multiple of 16)
    const int cs32 = (cs + 31) & ~31;                // ... as the MFMAs see them
    const int ch8 = cs32 >> 3;
    // the gathered rows: 16 bytes (8 channels) per thread and step of the loop, zeros past the slice / the row
    for (int e = tid; e < TM * ch8; e += kBlock) {
      const int r = e / ch8, c8 = e - r * ch8;
      uint4 v = zero16;
      const int in = Tt[r][k];
      if (in >= 0 && c8 * 8 < cs) v = *(const uint4*)(feat + (int64_t)in * cin + h * CIN_MAX + c8 * 8);
      *(uint4*)&As[r][c8 * 8] = v;
    }
    // W[k][channel][column] -> Wt[column][channel]: 8 columns of one channel per thread and step
    const __half* Wk = W + ((int64_t)k * cin + h * CIN_MAX) * COUT;
    for (int e = tid; e < cs32 * (COUT / 8); e += kBlock) {
      const int c = e / (COUT / 8), j8 = e - c * (COUT / 8);
      uint4 v = zero16;
      if (c < cs) v = *(const uint4*)(Wk + (int64_t)c * COUT + j8 * 8);
      const _Float16* hv = (const _Float16*)&v;
#pragma unroll
      for (int j = 0; j < 8; ++j) Wt[j8 * 8 + j][c] = hv[j];
    }
    __syncthreads();
    for (int c0 = 0; c0 < cs32; c0 += 32) {
      const f16x8 a = *(const f16x8*)&As[wv * 16 + fi][c0 + 8 * fk];
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        const f16x8 b = *(const f16x8*)&Wt[n * 16 + fi][c0 + 8 * fk];
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[n], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int col = n * 16 + fi;
    const float bv = bias ? __half2float(bias[col]) : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t r = row0 + wv * 16 + fk * 4 + j;
      if (r >= num_out) continue;
      float v = acc[n][j] + bv;
      if (residual) v = __half2float(__float2half_rn(v)) + __half2float(residual[r * COUT + col]);
      if (relu) v = fmaxf(v, 0.f);
      out[r * COUT + col] = __float2half_rn(v);
    }
  }
}

// ---- round 5: the same product on a taller tile, with the next offset's loads in flight under the MFMAs.
// The 64-row form above waits for its gathers at every offset (loads -> LDS -> barrier -> MFMA -> barrier), stages the
// weights of an offset once per 64 rows (as many bytes as the rows themselves at 64 channels) and transposes them by
// 2-byte LDS stores. Here a workgroup owns TMH = 128 output rows (wave w: rows 32 w .. 32 w + 31, two 16-row
// fragments against every column fragment), the gathered rows and the weights of offset s + 1 are requested into
// registers before the MFMAs of offset s start, and W[k] is transposed in registers: a thread takes an 8 x 8 block
// (eight 16-byte rows), v_perm_b32 turns it, eight 16-byte LDS stores put it down. Per output row the offsets,
// slices and K chunks come in the same order as above: the same bits.
constexpr int TMH = 128;
__device__ __forceinline__ unsigned u4_word(const uint4& v, int w) { return w == 0 ? v.x : (w == 1 ? v.y : (w == 2 ? v.z : v.w)); }
template <int NC>
__global__ void __launch_bounds__(kBlock)
indice_conv_f16_mfma_tall_kernel(const __half* __restrict__ feat, int cin, const __half* __restrict__ W, int kv,
                                 int centre, const int32_t* __restrict__ table, int64_t num_out,
                                 const __half* __restrict__ bias, int relu, const __half* __restrict__ residual,
                            __half* __restrict__ out) {
  constexpr int COUT = NC * 16;
  constexpr int MT = TMH / 64;                       // 16-row fragments per wave
  constexpr int NA = TMH * (CIN_MAX / 8) / kBlock;   // 16-byte row pieces per thread at 64 channels
  constexpr int NWB = (CIN_MAX / 8) * (COUT / 8);    // 8 x 8 weight blocks of a 64-channel slice
  constexpr int NWT = (NWB + kBlock - 1) / kBlock;   // ... per thread (1)
  __shared__ __align__(16) _Float16 As[TMH][H_LD];
  __shared__ __align__(16) _Float16 Wt[COUT][H_LD];
  __shared__ int Tt[TMH][32];   // the tile's rows of the neighbour table (kv <= 32)
  __shared__ unsigned s_any[TMH / 64];
  __shared__ int steps[32];     // active offsets, accumulation order (centre first for subM)
  __shared__ int nsteps;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 15, fk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TMH;
  for (int e = tid; e < TMH * kv; e += kBlock) {
    int r = e / kv, k = e - r * kv;
    int64_t gr = row0 + r;
    Tt[r][k] = gr < num_out ? table[gr * kv + k] : -1;
  }
  __syncthreads();
  if (tid < TMH) {  // which offsets have a partner in this tile (one bit per offset and wave of rows)
    unsigned m = 0u;
    for (int k = 0; k < kv; ++k)
      if (__ballot(Tt[tid][k] >= 0) != 0) m |= 1u << k;
    if (lane == 0) s_any[wv] = m;
  }
  __syncthreads();
  if (tid == 0) {
    unsigned m = 0u;
    for (int w = 0; w < TMH / 64; ++w) m |= s_any[w];
    int n = 0;
    for (int step = 0; step < kv; ++step) {
      int k = step;
      if (centre >= 0) k = (step == 0) ? centre : (step <= centre ? step - 1 : step);
      if ((m >> k) & 1u) steps[n++] = k;
    }
    nsteps = n;
  }
  __syncthreads();
  // A step stages 64 channels' worth: one offset (or one 64-channel slice of it), or -- up to 32 channels -- TWO
  // offsets side by side (channels [0, 32) and [32, 64) of the LDS rows), so that a 32-channel layer pays the
  // barriers and the staging of a step once per two offsets. The MFMAs of an offset stay its own (K = 32 each).
  const int slices = (cin + CIN_MAX - 1) / CIN_MAX;
  const int cs_one = min(CIN_MAX, cin);               // channels of a step's (first) offset, single-slice layers
  const int G = (slices == 1 && cs_one <= 32) ? 2 : 1;  // offsets per step
  const int ns = slices == 1 ? (nsteps + G - 1) / G : nsteps * slices;
  f32x4 acc[MT][NC];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 ra[NA];
  uint4 rw[NWT][8];
  // geometry of step si: h = slice, cs = channels per offset, ch8 = 16-byte pieces per offset (padded to 32
  // channels), ng = offsets staged
  auto step_geom = [&](int si, int& h, int& cs, int& ch8, int& ng) {
    h = slices == 1 ? 0 : si - (si / slices) * slices;
    cs = min(CIN_MAX, cin - h * CIN_MAX);
    ch8 = ((cs + 31) & ~31) >> 3;
    ng = slices == 1 ? min(G, nsteps - si * G) : 1;
  };
  auto prefetch = [&](int si) {
    int h, cs, ch8, ng;
    step_geom(si, h, cs, ch8, ng);
    const int k0 = steps[slices == 1 ? si * G : si / slices];
    const int k1 = ng > 1 ? steps[si * G + 1] : k0;
    const int npc = ng * ch8;  // pieces per row
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * kBlock;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (e < TMH * npc) {
        const int r = e / npc, c8 = e - r * npc;
        const int g = c8 >= ch8 ? 1 : 0, cl = c8 - g * ch8;
        const int in = Tt[r][g ? k1 : k0];
        if (in >= 0 && cl * 8 < cs) v = *(const uint4*)(feat + (int64_t)in * cin + h * CIN_MAX + cl * 8);
      }
      ra[i] = v;
    }
    // block b = (channel block cb, column block jb): rows cb * 8 .. + 7 of W[k], columns jb * 8 .. + 7
#pragma unroll
    for (int i = 0; i < NWT; ++i) {
      const int b = tid + i * kBlock;
      const int cb = b / (COUT / 8), jb = b - cb * (COUT / 8);
      const int g = cb >= ch8 ? 1 : 0, cbl = cb - g * ch8;
      const __half* Wk = W + ((int64_t)(g ? k1 : k0) * cin + h * CIN_MAX) * COUT;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = cbl * 8 + q;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (b < npc * (COUT / 8) && c < cs) v = *(const uint4*)(Wk + (int64_t)c * COUT + jb * 8);
        rw[i][q] = v;
      }
    }
  };
  if (ns > 0) prefetch(0);
  for (int si = 0; si < ns; ++si) {
    int h, cs, ch8, ng;
    step_geom(si, h, cs, ch8, ng);
    const int npc = ng * ch8;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * kBlock;
      if (e < TMH * npc) { const int r = e / npc, c8 = e - r * npc; *(uint4*)&As[r][c8 * 8] = ra[i]; }
    }
#pragma unroll
    for (int i = 0; i < NWT; ++i) {
      const int b = tid + i * kBlock;
      if (b < npc * (COUT / 8)) {
        const int cb = b / (COUT / 8), jb = b - cb * (COUT / 8);
        // column j of the block = halves j of the eight rows: word (j >> 1), low or high half
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
          unsigned wd[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const unsigned lo = u4_word(rw[i][2 * m], j >> 1), hi = u4_word(rw[i][2 * m + 1], j >> 1);
            wd[m] = __builtin_amdgcn_perm(hi, lo, sel);
          }
          *(uint4*)&Wt[jb * 8 + j][cb * 8] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
        }
      }
    }
    __syncthreads();
    if (si + 1 < ns) prefetch(si + 1);
    for (int c0 = 0; c0 < npc * 8; c0 += 32) {
      f16x8 a[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = *(const f16x8*)&As[wv * (TMH / 4) + m * 16 + fi][c0 + 8 * fk];
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        const f16x8 b = *(const f16x8*)&Wt[n * 16 + fi][c0 + 8 * fk];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b, acc[m][n], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int col = n * 16 + fi;
    const float bv = bias ? __half2float(bias[col]) : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t r = row0 + wv * (TMH / 4) + m * 16 + fk * 4 + j;
        if (r >= num_out) continue;
        float v = acc[m][n][j] + bv;
        if (residual) v = __half2float(__float2half_rn(v)) + __half2float(residual[r * COUT + col]);
        if (relu) v = fmaxf(v, 0.f);
        out[r * COUT + col] = __float2half_rn(v);
      }
    }
  }
}

// ---- round 5, second form (up to 64 input channels: every layer of VoxelBackBone8x): the gathered rows never touch
// LDS. An MFMA A fragment is 16 rows x 32 channels with lane (i, kq) holding channels 8 kq .. 8 kq + 7 of row i --
// exactly one 16-byte piece of a gathered fp16 row, so every lane loads its own piece of its own row's partner
// straight into the fragment register (zeros where the table says "no partner"). What a workgroup shares are the
// weights: KW offsets' W[k], transposed in registers as above, sit in a 28 KB LDS buffer per round (3 offsets at
// 64 x 64, 9 at 32 x 32); the eight waves of a workgroup (256 output rows, 32 per wave) run through a round's offsets
// without meeting -- three offsets' table entries, then their row pieces, are requested together -- and a wave skips
// an offset none of its 32 rows has. Barriers: two per ROUND (9 rounds at 64 x 64) instead of two per offset and 64
// rows; weight bytes per output row: a quarter of the 64-row form's. Per output row the offsets come in the same
// order (centre first for subM) and every accumulator sees its K chunks in the same order: the same bits.
constexpr int kDirWHalves = 14336;  // the weights' LDS buffer (28 KB)
template <int NC, int kDirBlock>
__global__ void __launch_bounds__(kDirBlock)
indice_conv_f16_direct_kernel(const __half* __restrict__ feat, int cin, const __half* __restrict__ W, int kv,
                              int centre, const int32_t* __restrict__ table, int64_t num_out,
                              const __half* __restrict__ bias, int relu, const __half* __restrict__ residual,
                            __half* __restrict__ out,
                              const int32_t* __restrict__ order) {
  constexpr int COUT = NC * 16;
  constexpr int MT = 2;  // 16-row fragments per wave
  constexpr int kDirRows = kDirBlock / 2;
  __shared__ __align__(16) _Float16 Wl[kDirWHalves];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fi = lane & 15, fk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * kDirRows + wv * 32;
  const int cs32 = (cin + 31) & ~31;   // channels as the MFMAs see them (32 or 64)
  const int C0 = cs32 >> 5;
  const int wld = cs32 + 8;            // halves per LDS row of a transposed W[k] (rows start 4 banks apart)
  const int bpo = (cs32 >> 3) * (COUT / 8);  // 8 x 8 blocks per offset
  int KW = min(kv, min(kDirWHalves / (COUT * wld), kDirBlock / bpo));
  if (KW > 9) KW = 9;
  KW -= KW % 3 == 0 ? 0 : (KW > 3 ? KW % 3 : 0);  // whole batches of three offsets where it can be
  const int rounds = (kv + KW - 1) / KW;
  auto kmap = [&](int step) {  // accumulation order: the centre first for subM
    return centre >= 0 ? (step == 0 ? centre : (step <= centre ? step - 1 : step)) : step;
  };
  int64_t rm[MT];  // (positions of `order`, if given: see indice_conv_f32_direct_kernel)
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int64_t pos = row0 + m * 16 + fi;
    rm[m] = pos < num_out ? (order ? (int64_t)order[pos] : pos) : num_out;
    if ((uint64_t)rm[m] >= (uint64_t)num_out) rm[m] = num_out;  // (an entry that is no row: nothing read)
  }
  f32x4 acc[MT][NC];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NC; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 rw[8];
  auto prefetch_w = [&](int rd) {
    const int kw = tid / bpo, b = tid - kw * bpo;
    const int cb = b / (COUT / 8), jb = b - cb * (COUT / 8);
    const int step = rd * KW + kw;
    const bool on = kw < KW && step < kv;
    const __half* Wk = W + (int64_t)kmap(on ? step : 0) * cin * COUT;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = cb * 8 + q;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (on && c < cin) v = *(const uint4*)(Wk + (int64_t)c * COUT + jb * 8);
      rw[q] = v;
    }
  };
  prefetch_w(0);
  for (int rd = 0; rd < rounds; ++rd) {
    {
      const int kw = tid / bpo, b = tid - kw * bpo;
      if (kw < KW) {
        const int cb = b / (COUT / 8), jb = b - cb * (COUT / 8);
        _Float16* Wt = Wl + (size_t)kw * COUT * wld;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned sel = (j & 1) ? 0x07060302u : 0x05040100u;
          unsigned wd[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const unsigned lo = u4_word(rw[2 * m], j >> 1), hi = u4_word(rw[2 * m + 1], j >> 1);
            wd[m] = __builtin_amdgcn_perm(hi, lo, sel);
          }
          *(uint4*)&Wt[(jb * 8 + j) * wld + cb * 8] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
        }
      }
    }
    __syncthreads();
    if (rd + 1 < rounds) prefetch_w(rd + 1);
    const int nk = min(KW, kv - rd * KW);
    for (int kb = 0; kb < nk; kb += 3) {
      int in[3][MT];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const bool on = kb + g < nk;
        const int k = kmap(on ? rd * KW + kb + g : 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) in[g][m] = (on && rm[m] < num_out) ? table[rm[m] * kv + k] : -1;
      }
      f16x8 a[3][MT][2];
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const int ch = c * 32 + 8 * fk;
            if (in[g][m] >= 0 && ch < cin) v = *(const uint4*)(feat + (int64_t)in[g][m] * cin + ch);
            a[g][m][c] = *(const f16x8*)&v;
          }
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        bool on_m[MT];  // a 16-row fragment none of whose rows has this offset adds nothing
        bool any = false;
#pragma unroll
        for (int m = 0; m < MT; ++m) { on_m[m] = __ballot(in[g][m] >= 0) != 0; any = any || on_m[m]; }
        if (!any) continue;
        const _Float16* Wt = Wl + (size_t)(kb + g) * COUT * wld;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (c < C0) {
#pragma unroll
            for (int n = 0; n < NC; ++n) {
              const f16x8 b = *(const f16x8*)&Wt[(n * 16 + fi) * wld + c * 32 + 8 * fk];
#pragma unroll
              for (int m = 0; m < MT; ++m)
                if (on_m[m]) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[g][m][c], b, acc[m][n], 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int col = n * 16 + fi;
    const float bv = bias ? __half2float(bias[col]) : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t pos = row0 + m * 16 + fk * 4 + j;
        if (pos >= num_out) continue;
        const int64_t r = order ? (int64_t)order[pos] : pos;
        if ((uint64_t)r >= (uint64_t)num_out) continue;
        float v = acc[m][n][j] + bv;
        if (residual) v = __half2float(__float2half_rn(v)) + __half2float(residual[r * COUT + col]);
        if (relu) v = fmaxf(v, 0.f);
        out[r * COUT + col] = __float2half_rn(v);
      }
    }
  }
}

// any (Cin, Cout): one thread per output element, fp32 accumulation in the same order, one rounding
__global__ void __launch_bounds__(kBlock)
indice_conv_f16_generic_kernel(const __half* __restrict__ feat, int cin, const __half* __restrict__ W, int kv,
                               int cout, int centre, const int32_t* __restrict__ table, int64_t num_out,
                               const __half* __restrict__ bias, int relu, const __half* __restrict__ residual,
                            __half* __restrict__ out) {
  const int64_t total = num_out * cout;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / cout;
    const int col = (int)(t - r * cout);
    float acc = 0.f;
    for (int step = 0; step < kv; ++step) {
      int k = step;
      if (centre >= 0) k = (step == 0) ? centre : (step <= centre ? step - 1 : step);
      const int in = table[r * kv + k];
      if (in < 0) continue;
      const __half* f = feat + (int64_t)in * cin;
      const __half* w = W + ((int64_t)k * cin) * cout + col;
      for (int c = 0; c < cin; ++c) acc = fmaf(__half2float(f[c]), __half2float(w[(int64_t)c * cout]), acc);
    }
    if (bias) acc = acc + __half2float(bias[col]);
    if (residual) acc = __half2float(__float2half_rn(acc)) + __half2float(residual[t]);
    if (relu) acc = fmaxf(acc, 0.f);
    out[t] = __float2half_rn(acc);
  }
}

// a few input channels (the backbone's first conv: 5 -> 16): one THREAD per output row, its COUT accumulators in
// registers, the filters as fp32 in LDS (every thread of a wave reads the same word: a broadcast). The generic form
// above gives every output ELEMENT a thread, i.e. sixteen threads walk the same 27 table entries and load the same
// partner rows. Same order of operations per element (offsets centre-first, channels ascending, one fmaf each): the
// same bits.
template <int COUT>
__global__ void __launch_bounds__(kBlock)
indice_conv_f16_rows_kernel(const __half* __restrict__ feat, int cin, const __half* __restrict__ W, int kv,
                            int centre, const int32_t* __restrict__ table, int64_t num_out,
                            const __half* __restrict__ bias, int relu, const __half* __restrict__ residual,
                            __half* __restrict__ out) {
  constexpr int kMaxW = 32 * 8 * COUT;  // kv <= 32, cin <= 8
  __shared__ float Wf[kMaxW];
  for (int e = threadIdx.x; e < kv * cin * COUT; e += kBlock) Wf[e] = __half2float(W[e]);
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= num_out) return;
  float acc[COUT];
#pragma unroll
  for (int j = 0; j < COUT; ++j) acc[j] = 0.f;
  // three offsets at a time: their table entries, then their partners' channels, are requested together (a chain of
  // 27 x 2 dependent loads otherwise); the products follow in the offsets' order
  for (int s0 = 0; s0 < kv; s0 += 3) {
    int in[3], kk[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int step = s0 + q;
      int k = step;
      if (centre >= 0) k = (step == 0) ? centre : (step <= centre ? step - 1 : step);
      kk[q] = k;
      in[q] = step < kv ? table[r * kv + k] : -1;
    }
    float fv[3][8];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int c = 0; c < 8; ++c)
        fv[q][c] = (in[q] >= 0 && c < cin) ? __half2float(feat[(int64_t)in[q] * cin + c]) : 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (in[q] < 0) continue;
      const float* w = Wf + kk[q] * cin * COUT;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < cin) {
#pragma unroll
          for (int j = 0; j < COUT; ++j) acc[j] = fmaf(fv[q][c], w[c * COUT + j], acc[j]);
        }
      }
    }
  }
  __half* o = out + r * COUT;
#pragma unroll
  for (int j8 = 0; j8 < COUT / 8; ++j8) {
    __half h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[j8 * 8 + j];
      if (bias) v = v + __half2float(bias[j8 * 8 + j]);
      if (residual) v = __half2float(__float2half_rn(v)) + __half2float(residual[r * COUT + j8 * 8 + j]);
      if (relu) v = fmaxf(v, 0.f);
      h[j] = __float2half_rn(v);
    }
    *(uint4*)(o + j8 * 8) = *(const uint4*)h;
  }
}

__global__ void __launch_bounds__(kBlock)
transpose_filters_f16_kernel(const __half* __restrict__ W, int kv, int cin, int cout, __half* __restrict__ Wt) {
  const int64_t total = (int64_t)kv * cin * cout;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = t / ((int64_t)cin * cout);
    const int r = (int)(t - k * cin * cout);
    const int ci = r / cout, co = r - ci * cout;
    Wt[(k * cout + co) * cin + ci] = W[t];
  }
}

// filter gradient from fp16 rows: the fp32 tile kernel's scheme (per (1024-pair chunk, offset, 32 x 32 tile) outer
// products in fp32 registers, partials in fp32, then the chunks in ascending order), rounded to fp16 at the end
__global__ void __launch_bounds__(kBlock)
filter_grad_partial_f16_kernel(const __half* __restrict__ feat, int cin, const __half* __restrict__ og, int cout,
                               const int32_t* __restrict__ pairs, int64_t pair_stride,
                               const int32_t* __restrict__ num, int side_in, int tiles_j,
                               float* __restrict__ partial, int kv) {
  const int chunk = blockIdx.x, k = blockIdx.y;
  const int ti0 = (blockIdx.z / tiles_j) * kGT, tj0 = (blockIdx.z % tiles_j) * kGT;
  const int n = num[k];
  const int p0 = chunk * kGradChunk;
  if (p0 >= n) return;
  const int p1 = min(n, p0 + kGradChunk);
  const int32_t* pin = pairs + ((int64_t)k * 2 + side_in) * pair_stride;
  const int32_t* pout = pairs + ((int64_t)k * 2 + (1 - side_in)) * pair_stride;
  __shared__ float s_in[kGT][kGT + 1];
  __shared__ float s_og[kGT][kGT + 1];
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
  for (int pb = p0; pb < p1; pb += kGT) {
    for (int q = threadIdx.x; q < kGT * kGT; q += kBlock) {
      const int pr = q >> 5, c = q & 31;
      const int p = pb + pr;
      float vi = 0.f, vo = 0.f;
      if (p < p1) {
        const int i = pin[p], o = pout[p];
        if (ti0 + c < cin) vi = __half2float(feat[(int64_t)i * cin + ti0 + c]);
        if (tj0 + c < cout) vo = __half2float(og[(int64_t)o * cout + tj0 + c]);
      }
      s_in[pr][c] = vi;
      s_og[pr][c] = vo;
    }
    __syncthreads();
#pragma unroll 8
    for (int pr = 0; pr < kGT; ++pr) {
      const float i0 = s_in[pr][2 * ty], i1 = s_in[pr][2 * ty + 1];
      const float o0 = s_og[pr][2 * tx], o1 = s_og[pr][2 * tx + 1];
      a00 = fmaf(i0, o0, a00); a01 = fmaf(i0, o1, a01);
      a10 = fmaf(i1, o0, a10); a11 = fmaf(i1, o1, a11);
    }
    __syncthreads();
  }
  float* dst = partial + ((int64_t)chunk * kv + k) * cin * cout;
  const int ci = ti0 + 2 * ty, cj = tj0 + 2 * tx;
  if (ci < cin && cj < cout) dst[(int64_t)ci * cout + cj] = a00;
  if (ci < cin && cj + 1 < cout) dst[(int64_t)ci * cout + cj + 1] = a01;
  if (ci + 1 < cin && cj < cout) dst[(int64_t)(ci + 1) * cout + cj] = a10;
  if (ci + 1 < cin && cj + 1 < cout) dst[(int64_t)(ci + 1) * cout + cj + 1] = a11;
}
__global__ void __launch_bounds__(kBlock)
filter_grad_reduce_f16_kernel(const float* __restrict__ partial, const int32_t* __restrict__ num, int kv,
                              int cin, int cout, __half* __restrict__ dW) {
  const int64_t per = (int64_t)cin * cout;
  const int64_t total = (int64_t)kv * per;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(t / per);
    const int64_t r = t - (int64_t)k * per;
    const int chunks = (num[k] + kGradChunk - 1) / kGradChunk;
    float acc = 0.f;
    for (int c = 0; c < chunks; ++c) acc = acc + partial[((int64_t)c * kv + k) * per + r];
    dW[t] = __float2half_rn(acc);
  }
}

// max-pool on the fp16 rows themselves (comparisons of fp16 values are exact in fp32)
__global__ void __launch_bounds__(kBlock)
maxpool_fwd_f16_kernel(const __half* __restrict__ feat, int C, const int32_t* __restrict__ table, int kv,
                       int64_t num_out, __half* __restrict__ out) {
  const int64_t total = num_out * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = t / C;
    const int c = (int)(t - o * C);
    float acc = 0.f;
    for (int k = 0; k < kv; ++k) {
      const int i = table[o * kv + k];
      if (i < 0) continue;
      const float v = __half2float(feat[(int64_t)i * C + c]);
      if (acc < v) acc = v;
    }
    out[t] = __float2half_rn(acc);
  }
}
// the reference's half backward adds the selected output gradients in fp16, one offset after the other
// (maxpool.cc:46-62 on at::Half): the same order and the same rounding after every addition
__global__ void __launch_bounds__(kBlock)
maxpool_bwd_f16_kernel(const __half* __restrict__ feat, const __half* __restrict__ outf,
                       const __half* __restrict__ og, int C, const int32_t* __restrict__ table_t, int kv,
                       int64_t num_in, __half* __restrict__ din) {
  const int64_t total = num_in * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / C;
    const int c = (int)(t - i * C);
    const float v = __half2float(feat[t]);
    __half acc = __float2half_rn(0.f);
    for (int k = 0; k < kv; ++k) {
      const int o = table_t[i * kv + k];
      if (o < 0) continue;
      if (__half2float(outf[(int64_t)o * C + c]) == v)
        acc = __float2half_rn(__half2float(acc) + __half2float(og[(int64_t)o * C + c]));
    }
    din[t] = acc;
  }
}

static int g_f16_short_tiles = 0;  // pcs_indice_conv_f16_short_tiles: 1 = the 64-row form of round 4, 2 = the 128-row staged form (cross-checks)
// out[o,:] = fp16( (+bias) relu? ( sum_k feat[table[o,k],:] . W[k] ) ) on an output-major table
static int conv_table_f16(const __half* features, int cin, const __half* filters, int kv, int cout, int centre,
                          const int32_t* table, int64_t num_out, const __half* bias, int relu, __half* out,
                          hipStream_t s, const int32_t* order = nullptr, const __half* residual = nullptr) {
  if (num_out == 0) return PCS_OK;
  const int64_t nb = (num_out + TM - 1) / TM;
  if (nb >= (1LL << 31)) return PCS_ERR_RANGE;
  const bool cout_ok = cout == 16 || cout == 32 || cout == 64 || cout == 128;
  if (cout_ok && kv <= 32 && cin % 16 == 0 && cin <= CIN_MAX && (g_f16_short_tiles == 0 || g_f16_short_tiles == 3)) {
#define PCS_DIRECT(NCV, BLK)                                                                                          \
  indice_conv_f16_direct_kernel<NCV, BLK><<<(int)((num_out + BLK / 2 - 1) / (BLK / 2)), BLK, 0, s>>>(                   \
      features, cin, filters, kv, centre, table, num_out, bias, relu, residual, out, order)
    if (g_f16_short_tiles == 3) {
      switch (cout) {
        case 16: PCS_DIRECT(1, 512); break;
        case 32: PCS_DIRECT(2, 512); break;
        case 64: PCS_DIRECT(4, 512); break;
        default: PCS_DIRECT(8, 512); break;
      }
    } else {
      switch (cout) {
        case 16: PCS_DIRECT(1, 256); break;
        case 32: PCS_DIRECT(2, 256); break;
        case 64: PCS_DIRECT(4, 256); break;
        default: PCS_DIRECT(8, 256); break;
      }
    }
#undef PCS_DIRECT
  } else if (cout_ok && kv <= 32 && cin % 16 == 0 && cin <= 8 * CIN_MAX && g_f16_short_tiles != 1) {
    const int nbt = (int)((num_out + TMH - 1) / TMH);
    switch (cout) {
      case 16: indice_conv_f16_mfma_tall_kernel<1><<<nbt, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, relu, residual, out); break;
      case 32: indice_conv_f16_mfma_tall_kernel<2><<<nbt, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, relu, residual, out); break;
      case 64: indice_conv_f16_mfma_tall_kernel<4><<<nbt, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, relu, residual, out); break;
      default: indice_conv_f16_mfma_tall_kernel<8><<<nbt, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, relu, residual, out); break;
    }
  } else if (cout_ok && kv <= 32 && cin % 16 == 0 && cin <= 8 * CIN_MAX) {
    switch (cout) {
      case 16: indice_conv_f16_mfma_kernel<1><<<(int)nb, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, relu, residual, out); break;
      case 32: indice_conv_f16_mfma_kernel<2><<<(int)nb, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, relu, residual, out); break;
      case 64: indice_conv_f16_mfma_kernel<4><<<(int)nb, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, relu, residual, out); break;
      default: indice_conv_f16_mfma_kernel<8><<<(int)nb, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, relu, residual, out); break;
    }
  } else if (cin <= 8 && kv <= 32 && (cout == 16 || cout == 32) && g_f16_short_tiles != 1) {
    const int nbr = (int)((num_out + kBlock - 1) / kBlock);
    if (cout == 16)
      indice_conv_f16_rows_kernel<16><<<nbr, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, relu, residual, out);
    else
      indice_conv_f16_rows_kernel<32><<<nbr, kBlock, 0, s>>>(features, cin, filters, kv, centre, table, num_out, bias, relu, residual, out);
  } else {
    indice_conv_f16_generic_kernel<<<grid_for(num_out * cout), kBlock, 0, s>>>(features, cin, filters, kv, cout, centre,
                                                                              table, num_out, bias, relu, residual, out);
  }
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}
}  // namespace pcs

extern "C" {

size_t pcs_indice_conv_f16_workspace_bytes(int64_t num_in, int cin, int kv, int cout, int64_t num_out) {
  (void)num_in; (void)cin; (void)cout;
  return pcs_neighbor_table_bytes(num_out, kv);
}

// test hook: 1 = the fp16 convs on round 4's 64-row tiles (the same bits; tests/test_gpu_voxel_conv.py), 0 = the
// 128-row tiles with register prefetch (default)
int pcs_indice_conv_f16_short_tiles(int on) {
  pcs::g_f16_short_tiles = (on >= 0 && on <= 3) ? on : 0;
  return PCS_OK;
}

// features [num_in,cin], filters [kv,cin,cout], bias [cout] or NULL (fused_indice_conv_half), out
// [num_out,cout]: IEEE fp16 bit patterns
int pcs_indice_conv_f16(const void* features, int64_t num_in, int cin, const void* filters, int kv, int cout,
                        const void* bias, const int32_t* pairs, int64_t pair_stride, const int32_t* num,
                        int64_t num_out, int inverse, int subm, void* out, void* ws, size_t ws_bytes,
                        hipStream_t s) {
  using namespace pcs;
  if (!features || !filters || !pairs || !num || !out || !ws || num_in < 0 || num_out < 0 || cin < 1 || cout < 1 ||
      kv < 1)
    return PCS_ERR_ARG;
  if (ws_bytes < pcs_neighbor_table_bytes(num_out, kv)) return PCS_ERR_WORKSPACE;
  if (num_out == 0) return PCS_OK;
  int rc = pcs_build_neighbor_table(pairs, pair_stride, num, kv, inverse, num_out, (int32_t*)ws, s);
  if (rc) return rc;
  const int centre = (subm && (kv & 1)) ? kv / 2 : -1;
  return conv_table_f16((const __half*)features, cin, (const __half*)filters, kv, cout, centre, (const int32_t*)ws,
                        num_out, (const __half*)bias, 0, (__half*)out, s);
}

// the same on a neighbour table the caller keeps (the module path caches it with the rulebook): out =
// fp16(relu?(sum_k feat[table[:,k]] . W[k] + bias)); centre >= 0: the offset accumulated first (subM)
int pcs_indice_conv_table_f16(const void* features, int cin, const void* filters, int kv, int cout, int centre,
                              const int32_t* table, int64_t num_out, const void* bias, int relu, void* out,
                              hipStream_t s) {
  using namespace pcs;
  if (!features || !filters || !table || !out || cin < 1 || cout < 1 || kv < 1 || num_out < 0 || centre >= kv)
    return PCS_ERR_ARG;
  return conv_table_f16((const __half*)features, cin, (const __half*)filters, kv, cout, centre, table, num_out,
                        (const __half*)bias, relu, (__half*)out, s);
}

// ... with the output rows taken in the order of pcs_neighbor_table_order (see pcs_indice_conv_table_ordered_f32)
int pcs_indice_conv_table_ordered_f16(const void* features, int cin, const void* filters, int kv, int cout, int centre,
                                      const int32_t* table, int64_t num_out, const void* bias, int relu, void* out,
                                      const int32_t* order, hipStream_t s) {
  using namespace pcs;
  if (!features || !filters || !table || !out || cin < 1 || cout < 1 || kv < 1 || num_out < 0 || centre >= kv)
    return PCS_ERR_ARG;
  return conv_table_f16((const __half*)features, cin, (const __half*)filters, kv, cout, centre, table, num_out,
                        (const __half*)bias, relu, (__half*)out, s, order);
}

// ... with a residual row: fp16(relu(fp16(conv + bias) + residual[o,:])) -- the conv's own fp16 result and the fp16
// identity row added as the tensor library adds two fp16 rows (in fp32, rounded once), then the ReLU: the bits of
// SparseBasicBlock's separate add (pcs_indice_conv_table_residual_f32 is the fp32 form). residual must not be `out`.
int pcs_indice_conv_table_residual_f16(const void* features, int cin, const void* filters, int kv, int cout, int centre,
                                       const int32_t* table, int64_t num_out, const void* bias, const void* residual,
                                       int relu, void* out, hipStream_t s) {
  using namespace pcs;
  if (!features || !filters || !table || !out || cin < 1 || cout < 1 || kv < 1 || num_out < 0 || centre >= kv ||
      (residual && residual == out))
    return PCS_ERR_ARG;
  return conv_table_f16((const __half*)features, cin, (const __half*)filters, kv, cout, centre, table, num_out,
                        (const __half*)bias, relu, (__half*)out, s, nullptr, (const __half*)residual);
}

// indice_conv_backward_half (src/all.cc:36-37; spconv_ops.h:363-449 on at::Half): in_grad [num_in,cin] and
// filters_grad [kv,cin,cout] in fp16, both from fp32 accumulations rounded once.
size_t pcs_indice_conv_backward_f16_workspace_bytes(int64_t num_in, int64_t pair_stride, int kv, int cin, int cout) {
  if (num_in < 1) num_in = 1;
  if (pair_stride < 1) pair_stride = 1;
  const int64_t chunks = (pair_stride + pcs::kGradChunk - 1) / pcs::kGradChunk;
  size_t b = pcs::align_up((size_t)num_in * kv * sizeof(int32_t));
  b += pcs::align_up((size_t)kv * cin * cout * sizeof(__half));
  b += pcs::align_up((size_t)chunks * kv * cin * cout * sizeof(float));
  return b;
}
int pcs_indice_conv_backward_f16(const void* features, int cin, const void* filters, int kv, int cout,
                                 const void* out_grad, const int32_t* pairs, int64_t pair_stride,
                                 const int32_t* num, int64_t num_in, int64_t num_out, int inverse, int subm,
                                 void* in_grad, void* filters_grad, void* ws, size_t ws_bytes, hipStream_t s) {
  using namespace pcs;
  if (!features || !filters || !out_grad || !pairs || !num || !in_grad || !filters_grad || cin < 1 || cout < 1 ||
      kv < 1 || num_in < 0 || num_out < 0 || pair_stride < 0)
    return PCS_ERR_ARG;
  if (!ws || ws_bytes < pcs_indice_conv_backward_f16_workspace_bytes(num_in, pair_stride, kv, cin, cout))
    return PCS_ERR_WORKSPACE;
  PCS_CHECK_HIP(hipMemsetAsync(filters_grad, 0, (size_t)kv * cin * cout * sizeof(__half), s));
  if (num_in == 0) return PCS_OK;
  if (num_out == 0 || pair_stride == 0) {
    PCS_CHECK_HIP(hipMemsetAsync(in_grad, 0, (size_t)num_in * cin * sizeof(__half), s));
    return PCS_OK;
  }
  char* w = (char*)ws;
  int32_t* table_t = (int32_t*)w;
  w += align_up((size_t)num_in * kv * sizeof(int32_t));
  __half* Wt = (__half*)w;
  w += align_up((size_t)kv * cin * cout * sizeof(__half));
  float* partial = (float*)w;
  // input gradient: the forward conv over the transposed rulebook with W[k]^T
  int rc = pcs_build_neighbor_table(pairs, pair_stride, num, kv, inverse ? 0 : 1, num_in, table_t, s);
  if (rc) return rc;
  transpose_filters_f16_kernel<<<grid_for((int64_t)kv * cin * cout), kBlock, 0, s>>>((const __half*)filters, kv, cin, cout, Wt);
  const int centre = (subm && (kv & 1)) ? kv / 2 : -1;
  rc = conv_table_f16((const __half*)out_grad, cout, Wt, kv, cin, centre, table_t, num_in, nullptr, 0, (__half*)in_grad, s);
  if (rc) return rc;
  // filter gradient
  const int64_t chunks = (pair_stride + kGradChunk - 1) / kGradChunk;
  const int tiles_i = (cin + kGT - 1) / kGT, tiles_j = (cout + kGT - 1) / kGT;
  if (chunks >= 65536 || kv >= 65536) return PCS_ERR_RANGE;
  filter_grad_partial_f16_kernel<<<dim3((unsigned)chunks, kv, tiles_i * tiles_j), kBlock, 0, s>>>(
      (const __half*)features, cin, (const __half*)out_grad, cout, pairs, pair_stride, num, inverse ? 1 : 0, tiles_j,
      partial, kv);
  filter_grad_reduce_f16_kernel<<<grid_for((int64_t)kv * cin * cout), kBlock, 0, s>>>(partial, num, kv, cin, cout,
                                                                                     (__half*)filters_grad);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

size_t pcs_indice_maxpool_f16_workspace_bytes(int64_t num_in, int C, int kv, int64_t num_out) {
  (void)num_in; (void)C;
  return pcs_neighbor_table_bytes(num_out, kv);
}

int pcs_indice_maxpool_f16(const void* features, int64_t num_in, int C, const int32_t* pairs, int64_t pair_stride,
                           const int32_t* num, int kv, int64_t num_out, void* out, void* ws, size_t ws_bytes,
                           hipStream_t s) {
  using namespace pcs;
  if (!features || !pairs || !num || !out || !ws || num_in < 0 || num_out < 0 || C < 1 || kv < 1) return PCS_ERR_ARG;
  if (ws_bytes < pcs_neighbor_table_bytes(num_out, kv)) return PCS_ERR_WORKSPACE;
  if (num_out == 0) return PCS_OK;
  int rc = pcs_build_neighbor_table(pairs, pair_stride, num, kv, 0, num_out, (int32_t*)ws, s);
  if (rc) return rc;
  maxpool_fwd_f16_kernel<<<grid_for(num_out * C), kBlock, 0, s>>>((const __half*)features, C, (const int32_t*)ws, kv,
                                                                 num_out, (__half*)out);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

// indice_maxpool_backward_half (src/all.cc:48-50): features [num_in,C], out_features / out_grad [num_out,C],
// in_grad [num_in,C], all fp16; ws: pcs_neighbor_table_bytes(num_in, kv)
int pcs_indice_maxpool_backward_f16(const void* features, const void* out_features, const void* out_grad, int C,
                                    const int32_t* pairs, int64_t pair_stride, const int32_t* num, int kv,
                                    int64_t num_in, void* in_grad, void* ws, size_t ws_bytes, hipStream_t s) {
  using namespace pcs;
  if (!features || !out_features || !out_grad || !pairs || !num || !in_grad || C < 1 || kv < 1 || num_in < 0)
    return PCS_ERR_ARG;
  if (num_in == 0) return PCS_OK;
  if (!ws || ws_bytes < pcs_neighbor_table_bytes(num_in, kv)) return PCS_ERR_WORKSPACE;
  int rc = pcs_build_neighbor_table(pairs, pair_stride, num, kv, 1, num_in, (int32_t*)ws, s);
  if (rc) return rc;
  maxpool_bwd_f16_kernel<<<grid_for(num_in * C), kBlock, 0, s>>>((const __half*)features, (const __half*)out_features,
                                                                (const __half*)out_grad, C, (const int32_t*)ws, kv,
                                                                num_in, (__half*)in_grad);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

}  // extern "C"
