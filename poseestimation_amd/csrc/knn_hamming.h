// Internal interface between the C ABI (posehip_api.hip) and the matcher kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace posehip {

// Threads per workgroup of the distance kernel (4 waves).
static constexpr int KNN_BLOCK = 256;
// Packed key = distance << KNN_KEY_SHIFT | train index local to a split.
// distance <= 256 needs 9 bits, which leaves 22 index bits below 0xFFFFFFFF.
static constexpr int KNN_KEY_SHIFT = 22;
static constexpr int64_t KNN_MAX_ROWS_PER_SPLIT = (int64_t)1 << KNN_KEY_SHIFT;

struct KnnPlan {
    int qpl;                // query rows per lane
    int64_t q_blocks;       // grid.x
    int n_splits;           // grid.y: train-row splits
    int64_t rows_per_split; // train rows per split
    size_t part_bytes;      // workspace for per-split keys
};

KnnPlan knn_plan(int64_t nq, int64_t nt, int n_cus);

hipError_t knn_launch(const KnnPlan &p, const void *d_query, int64_t nq, const void *d_train,
                      int64_t nt, int k, int32_t index_base, uint32_t *d_part, int32_t *d_out_idx,
                      int32_t *d_out_dist, uint8_t *d_keep, float ratio, hipStream_t stream);

hipError_t knn_merge_shards_launch(const int32_t *d_cand_idx, const int32_t *d_cand_dist,
                                   int n_shards, int64_t nq, int k, int32_t *d_out_idx,
                                   int32_t *d_out_dist, uint8_t *d_keep, float ratio,
                                   hipStream_t stream);

hipError_t crosscheck_resolve_launch(const int32_t *d_tidx, const int32_t *d_tdist, int64_t nt,
                                     int64_t nq, unsigned long long *d_best, int32_t *d_out_idx,
                                     int32_t *d_out_dist, hipStream_t stream);

} // namespace posehip
