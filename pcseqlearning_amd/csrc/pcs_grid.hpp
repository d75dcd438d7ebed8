// Fine-cell hash grid shared by the radius-graph engine (pcs_radius_graph.hip) and the fused ICP
// (pcs_icp.hip): reference points are chained (head/next linked lists) into an open-addressing
// table keyed by (reference frame cell, fine x, y, z cell).
#pragma once
#include "pcs_common.hpp"

namespace pcs {

constexpr uint64_t kEmpty = ~0ULL;

// device-side grid description, filled by grid_setup_kernel from the data bounds
struct GridDesc {
  float lo0, vs0;        // reference frame-cell quantisation (graph_utils.py:170-175)
  float lo[3];           // fine-cell origin
  float h, inv_unused;   // fine cell size
  float radius, radius2; // fp32 radius and radius*radius as the reference kernel forms them
  int rings;             // ceil(radius / h)
};

static __global__ void grid_setup_kernel(const float* __restrict__ b_ref, const float* __restrict__ b_qry,
                                  float radius, float cell, GridDesc* __restrict__ g) {
  if (threadIdx.x != 0) return;
  float mn[4], mx[4];
  for (int d = 0; d < 4; ++d) {
    mn[d] = fminf(b_ref[d], b_qry[d]);
    mx[d] = fmaxf(b_ref[4 + d], b_qry[4 + d]);
  }
  float vs0 = (float)(1 - 1e-3);
  g->vs0 = vs0;
  g->lo0 = mn[0] - vs0 * 2.f;  // pc_range_min = all.min(0) - voxel_size*2
  for (int d = 0; d < 3; ++d) g->lo[d] = mn[1 + d] - cell;
  g->h = cell;
  g->radius = radius;
  g->radius2 = radius * radius;
  g->rings = (int)ceilf(radius / cell);
  (void)mx;
}

// reference frame cell: round_half_even((f - lo0) / vs0) + 1
__device__ __forceinline__ int frame_cell(const GridDesc& g, float f) {
  return (int)rintf((f - g.lo0) / g.vs0) + 1;
}

__device__ __forceinline__ uint64_t pack_cell(int fc, int x, int y, int z) {
  return ((uint64_t)(uint32_t)fc << 51) | ((uint64_t)x << 34) | ((uint64_t)y << 17) | (uint64_t)z;
}

constexpr int kCoordMax = 1 << 17;

constexpr int kFrameCells = 4096;

static __global__ void frame_range_init_kernel(float* __restrict__ fr) {
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < kFrameCells; i += blockDim.x * gridDim.x) {
    fr[i] = __builtin_inff();
    fr[kFrameCells + i] = -__builtin_inff();
  }
}

static __global__ void __launch_bounds__(kBlock)
grid_insert_kernel(const float4* __restrict__ ref, int N, const GridDesc* __restrict__ gp,
                   uint64_t* __restrict__ keys, int32_t* __restrict__ head,
                   int32_t* __restrict__ next, uint32_t hmask, float* __restrict__ fr,
                   int32_t* __restrict__ err) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  GridDesc g = *gp;
  float4 p = ref[i];
  int fc = frame_cell(g, p.x);
  int x = (int)floorf((p.y - g.lo[0]) / g.h);
  int y = (int)floorf((p.z - g.lo[1]) / g.h);
  int z = (int)floorf((p.w - g.lo[2]) / g.h);
  if (x < 0 || y < 0 || z < 0 || x >= kCoordMax || y >= kCoordMax || z >= kCoordMax || fc < 0 ||
      fc >= kFrameCells) {
    atomicExch(err, 1);  // extent / cell-size ratio beyond the 17-bit cell coordinates
    next[i] = -1;
    return;
  }
  // range of frame values stored in this frame cell: makes the ring pruning bound exact
  atomic_min_f32(&fr[fc], p.x);
  atomic_max_f32(&fr[kFrameCells + fc], p.x);
  uint64_t key = pack_cell(fc, x, y, z);
  uint32_t slot = mix64(key) & hmask;
  while (true) {
    uint64_t prev = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)kEmpty,
                              (unsigned long long)key);
    if (prev == kEmpty || prev == key) break;
    slot = (slot + 1) & hmask;
  }
  next[i] = atomicExch(&head[slot], i);
}

__device__ __forceinline__ int grid_head(const uint64_t* __restrict__ keys,
                                         const int32_t* __restrict__ head, uint32_t hmask,
                                         uint64_t key) {
  uint32_t slot = mix64(key) & hmask;
  while (true) {
    uint64_t k = keys[slot];
    if (k == key) return head[slot];
    if (k == kEmpty) return -1;
    slot = (slot + 1) & hmask;
  }
}


}  // namespace pcs
