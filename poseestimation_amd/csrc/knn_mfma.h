// Internal interface of the matrix-core distance kernels (knn_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace posehip {

static constexpr int MF_WAVES = 4;                      // waves per workgroup
static constexpr int64_t MF_MAX_ROWS_PER_SPLIT = 32768; // 1024 tiles: the key's index field
// Rows an expanded buffer is readable past its last row: a workgroup's query block
// (up to 4 waves x 8 tiles x 32 rows) and a train tile may run over the end.
static constexpr int64_t MF_PAD_ROWS = 1024;

struct MfmaPlan {
    int q_tiles;            // tiles of 32 queries per wave (2, 4 or 8)
    int64_t q_blocks;       // grid.x: blocks of 128 * q_tiles queries
    int n_splits;           // grid.y: train-row splits
    int64_t rows_per_split; // multiple of 32
    size_t part_bytes;
};

// Bytes of the FP4 image of `rows` descriptors (128 per row, padding included).
size_t knn_mfma_expanded_bytes(int64_t rows);

// 256-bit rows (32 bytes each, packed) -> FP4 image.
hipError_t knn_mfma_expand_launch(const void *d_rows, int64_t rows, void *d_expanded,
                                  hipStream_t stream);

MfmaPlan knn_mfma_plan_splits(int64_t nq, int64_t nt, int q_tiles, int64_t splits);

// Writes part[split][rank][nq] keys (distance << KNN_KEY_SHIFT | row in split), the
// format knn_merge_kernel folds.
hipError_t knn_mfma_partial_launch(const MfmaPlan &p, const void *d_qx, int64_t nq,
                                   const void *d_tx, int64_t nt, int k, uint32_t *d_part,
                                   hipStream_t stream);

} // namespace posehip
