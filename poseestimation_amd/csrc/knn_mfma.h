// Internal interface of the matrix-core distance kernels (knn_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace posehip {

static constexpr int MF_WAVES = 4;                      // waves per workgroup
static constexpr int64_t MF_MAX_ROWS_PER_SPLIT = 32768; // 1024 tiles: the key's index field
// Rows an FP4 image is readable past its last row: the train tiles loaded ahead (three
// tiles of 32 rows) run over the end.
static constexpr int64_t MF_PAD_ROWS = 256;

struct MfmaPlan {
    int q_tiles;            // tiles of 32 queries per wave (2 or 4)
    int64_t q_blocks;       // grid.x: blocks of 128 * q_tiles queries
    int n_splits;           // grid.y: train-row splits
    int64_t rows_per_split; // multiple of 32 (of the largest frame, for a batch)
    size_t part_bytes;
};

// Train rows per split of a problem with nt train rows cut into n_splits (multiples of
// the 32-row tile).
__host__ __device__ inline int64_t mfma_rows_per_split(int64_t nt, int n_splits)
{
    int64_t rows = (nt + n_splits - 1) / n_splits;
    rows = (rows + 31) & ~(int64_t)31;
    return rows < 32 ? 32 : rows;
}

// Whether the matrix-core path is the faster one for a problem / a batch of this size
// (below, the two extra launches and the FP4 images cost more than they save).
bool knn_mfma_pays(int64_t nq, int64_t nt);
bool knn_mfma_pays_frames(int n_frames, int64_t max_nq, int64_t max_nt);

// Bytes of the FP4 image of `rows` descriptors (128 per row, padding included).
size_t knn_mfma_expanded_bytes(int64_t rows);

// 256-bit rows (32 bytes each, packed) -> FP4 image.
hipError_t knn_mfma_expand_launch(const void *d_rows, int64_t rows, void *d_expanded,
                                  hipStream_t stream);

MfmaPlan knn_mfma_plan(int64_t nq, int64_t nt, int n_cus);
MfmaPlan knn_mfma_plan_splits(int64_t nq, int64_t nt, int q_tiles, int64_t splits);
MfmaPlan knn_mfma_plan_frames(int n_frames, int64_t max_nq, int64_t max_nt, int64_t total_nq,
                              int n_cus);

// Same contract as knn_launch: FP4 image of the train set (d_tx, a workspace of
// knn_mfma_expanded_bytes(nt)), the distance kernel, the merge of the splits.  d_train ==
// nullptr: d_tx already holds the image (a resident train set), no pre-pass.  events, when
// given (five of them), bracket the FP4 pre-pass [4, 0], the distance kernel [0, 1] and the
// merge [1, 2] (empty with a single split: the distance kernel then writes the k-NN rows
// itself); [2, 3] is empty.
hipError_t knn_mfma_launch(const MfmaPlan &p, const void *d_query, int64_t nq, const void *d_train,
                           int64_t nt, int k, int32_t index_base, void *d_tx,
                           uint32_t *d_part, int32_t *d_out_idx, int32_t *d_out_dist,
                           uint8_t *d_keep, float ratio, hipStream_t stream,
                           hipEvent_t *events = nullptr);

// Same contract as knn_frames_launch; q_end is the end of the row range the query table
// spans (q_end rows of part / outputs are addressed), [t_begin, t_end) the train table's.
hipError_t knn_mfma_frames_launch(const MfmaPlan &p, const void *d_query, const int32_t *d_q_off,
                                  int64_t q_end, const void *d_train,
                                  const int32_t *d_t_off, int64_t t_begin, int64_t t_end,
                                  int n_frames, int k, void *d_tx, uint32_t *d_part,
                                  int32_t *d_out_idx, int32_t *d_out_dist, uint8_t *d_keep,
                                  float ratio, hipStream_t stream, hipEvent_t *events = nullptr,
                                  bool uniform_frames = false);

// Sustained rate of the distance kernel's MFMA alone (v_mfma_scale_f32_32x32x64_f8f6f4 on FP4
// operands of +-4 at random, four independent chains, two waves per SIMD on every CU), in
// FLOP per second.  d_scratch: n_cus * 2 * 256 floats.  Blocks until measured.
hipError_t mfma_fp4_peak(int n_cus, int iters, float *d_scratch, hipStream_t stream,
                         double *flop_per_s);

} // namespace posehip
