// Sequence loader, device side.
//
// Reference: WaymoDataset.get_lidar / load_data / __getitem__, pcdet/datasets/waymo/
// waymo_dataset.py:334-343, 383-391, 584-630: every sweep's raw rows
// [x, y, z, intensity, elongation, range, rimage_w, rimage_h] are normalised (tanh of the intensity,
// range / 75), moved into the coordinate frame of the last sweep (xyz @ R^T + t with the f64 pose
// product T0^-1 T1, stored back as fp32), tagged with the sweep's sample index, and concatenated;
// collate_batch (pcdet/datasets/dataset.py:222-232) then prepends the batch index.
//
// Here the raw rows of all sweeps of a sequence sit in HBM once and ONE pass writes the four arrays
// the registration path reads. HBM-bound: 40 B read + 40 B written per point.
#include "pcs_common.hpp"

namespace pcs {

// sweep of row i: the last s with offsets[s] <= i (F <= a few hundred: the table sits in LDS)
constexpr int kMaxSweeps = 1024;

__global__ void __launch_bounds__(kBlock)
assemble_sweeps_kernel(const float* __restrict__ raw, const int32_t* __restrict__ seg,
                       const int64_t* __restrict__ offsets, const double* __restrict__ T,
                       const int32_t* __restrict__ sweep_ids, int F, int64_t N, float batch_idx,
                       float4* __restrict__ bxyz, float* __restrict__ feat,
                       int32_t* __restrict__ point_sweep, int32_t* __restrict__ seg_label,
                       int32_t* __restrict__ inst_label) {
  __shared__ int64_t s_off[kMaxSweeps + 1];
  for (int q = threadIdx.x; q <= F; q += blockDim.x) s_off[q] = offsets[q];
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += stride) {
    int lo = 0, hi = F - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (s_off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    const double* t = T + (size_t)lo * 12;
    // two 16-byte loads per row
    const float4 a = ((const float4*)raw)[2 * i], b = ((const float4*)raw)[2 * i + 1];
    const double x = a.x, y = a.y, z = a.z;
    // numpy: (xyz [n,3] f32 -> f64) @ R^T (dgemm, k = 0,1,2 accumulated in order) + t, stored as f32
    const double nx = fma(z, t[2], fma(y, t[1], x * t[0])) + t[3];
    const double ny = fma(z, t[6], fma(y, t[5], x * t[4])) + t[7];
    const double nz = fma(z, t[10], fma(y, t[9], x * t[8])) + t[11];
    bxyz[i] = make_float4(batch_idx, (float)nx, (float)ny, (float)nz);
    feat[3 * i + 0] = (float)tanh((double)a.w);  // np.tanh on fp32 (within 1 ulp of this)
    feat[3 * i + 1] = b.x;
    feat[3 * i + 2] = b.y / 75.f;
    point_sweep[i] = sweep_ids[lo];
    if (seg) {
      const int2 s = ((const int2*)seg)[i];  // [instance, segmentation]
      inst_label[i] = s.x;
      seg_label[i] = s.y;
    }
  }
}

}  // namespace pcs

using namespace pcs;

extern "C" {

int pcs_assemble_sweeps(const float* raw, const int32_t* seg, const int64_t* sweep_offsets,
                        const double* sweep_T, const int32_t* sweep_ids, int num_sweeps, int64_t N,
                        int batch_index, float* point_bxyz, float* point_feat, int32_t* point_sweep,
                        int32_t* segmentation_label, int32_t* instance_label, hipStream_t s) {
  if (N < 0 || num_sweeps < 1 || num_sweeps > kMaxSweeps) return PCS_ERR_ARG;
  if (N == 0) return PCS_OK;
  if (!raw || !sweep_offsets || !sweep_T || !sweep_ids || !point_bxyz || !point_feat || !point_sweep)
    return PCS_ERR_ARG;
  if (seg && (!segmentation_label || !instance_label)) return PCS_ERR_ARG;
  int64_t blocks = (N + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;  // 16 workgroups per CU, grid-stride beyond
  assemble_sweeps_kernel<<<(int)blocks, kBlock, 0, s>>>(raw, seg, sweep_offsets, sweep_T, sweep_ids,
                                                        num_sweeps, N, (float)batch_index,
                                                        (float4*)point_bxyz, point_feat, point_sweep,
                                                        segmentation_label, instance_label);
  PCS_LAUNCH_CHECK();
  return PCS_OK;
}

}  // extern "C"
