// Cell-sorted, wave-cooperative k-nearest radius search (pcs_knn.hip) behind pcs_radius_graph_search for the
// sorted, single-frame-offset case -- the 32-NN graphs of ClusterProposal (cluster_proposal.py:57-83 of the
// reference, graph_utils.py:149-209, torch_hash_kernel.cu:224-409).
#pragma once
#include "pcs_grid.hpp"

namespace pcs {

constexpr int kKnnMaxK = 32;

// one 16-byte table slot per occupied cell of the hashed cloud: its run [start, end) of the cell-sorted points
struct KnnSlot {
  uint64_t key;
  int32_t start, end;
};

struct KnnPlan {
  uint64_t *rkey[2], *qkey[2];  // cell keys (unsorted, sorted) of the hashed / the query cloud
  int32_t *rval[2], *qval[2];   // original indices (iota, sorted)
  float4 *rpts, *qpts;          // the clouds in cell order
  KnnSlot* slot;                // [H]
  uint32_t H;
  int32_t *runflag, *runid, *run_start;  // query runs: [M], [M], [M + 1]
  int32_t* cnt;                 // [8]: 0 runs, 1 work cursor, 2 hard queries, 3 hard cursor
  int32_t* hard;                // [M] queries left to the one-thread search
  void* tmp;
  size_t tmp_bytes;
};

// the part of a radius-graph workspace that the tiled search uses (appended to the engine's own buffers)
void plan_knn(Arena& a, int64_t N, int64_t M, KnnPlan& p);

// ref / query: [N,4] / [M,4] (frame, x, y, z). Writes deg [M], nbr_idx / nbr_d [M, K] (ascending (dist2, index)).
// `grid` has been filled by grid_setup_kernel with cell = knn_cell(radius). err: device word, set on a coordinate
// overflow. Exact, the same edges as the one-thread engine.
int knn_tiled_search(const float4* ref, int64_t N, const float4* query, int64_t M, const GridDesc* grid,
                     int df, int K, const KnnPlan& p, int32_t* err, int32_t* nbr_idx, float* nbr_d,
                     int32_t* deg, hipStream_t s);

// fine cell of the tiled search for a 4-D radius: four rings of cells cover the ball
static inline float knn_cell(float radius) { return radius * 0.25f * 1.0005f; }
constexpr int kKnnRings = 4;

}  // namespace pcs
