// Internal interface between the C ABI (posehip_api.hip) and the matcher kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace posehip {

// Most waves per workgroup of the distance kernel.
static constexpr int KNN_MAX_WAVES = 16;
// Packed key = distance << KNN_KEY_SHIFT | train index local to a split.
// distance <= 256 needs 9 bits, which leaves 22 index bits below 0xFFFFFFFF.
static constexpr int KNN_KEY_SHIFT = 22;
static constexpr int64_t KNN_MAX_ROWS_PER_SPLIT = (int64_t)1 << KNN_KEY_SHIFT;

struct KnnPlan {
    int waves;              // waves per workgroup (all on the same 64 queries)
    bool lazy;              // threshold-gated key update
    int64_t q_blocks;       // grid.x: blocks of 64 queries
    int n_splits;           // grid.y: train-row splits
    int64_t rows_per_split; // train rows per workgroup
    int rows_per_wave;      // train rows per wave inside a split
    size_t part_bytes;      // workspace for per-split keys
};

KnnPlan knn_plan(int64_t nq, int64_t nt, int n_cus);
KnnPlan knn_plan_custom(int64_t nq, int64_t nt, int waves, bool lazy, int64_t max_blocks);
KnnPlan knn_plan_splits(int64_t nq, int64_t nt, int waves, bool lazy, int64_t splits);

hipError_t knn_launch(const KnnPlan &p, const void *d_query, int64_t nq, const void *d_train,
                      int64_t nt, int k, int32_t index_base, uint32_t *d_part, int32_t *d_out_idx,
                      int32_t *d_out_dist, uint8_t *d_keep, float ratio, hipStream_t stream,
                      hipEvent_t *events = nullptr);

// A batch of independent matching problems in one launch: frame f matches query
// rows [q_off[f], q_off[f+1]) against train rows [t_off[f], t_off[f+1]) (device
// arrays of n_frames + 1 ascending offsets).  Outputs are indexed by query row over
// the whole batch; train indices are local to the frame.  max_nq / max_nt bound
// the frames' sizes (the launch grid and the split count come from them).
// uniform_frames: all frames hold the same numbers of rows (enables the XCD-contiguous
// workgroup order, see the kernel).
KnnPlan knn_plan_frames(int n_frames, int64_t max_nq, int64_t max_nt, int64_t total_nq, int n_cus);
hipError_t knn_frames_launch(const KnnPlan &p, const void *d_query, const int32_t *d_q_off,
                             const void *d_train, const int32_t *d_t_off, int n_frames,
                             int64_t total_nq, int k, uint32_t *d_part, int32_t *d_out_idx,
                             int32_t *d_out_dist, uint8_t *d_keep, float ratio,
                             hipStream_t stream, hipEvent_t *events = nullptr,
                             bool uniform_frames = false);

// Folds the per-split keys part[split][rank][nq] (distance << KNN_KEY_SHIFT | row in
// split) into the k-NN rows.
hipError_t knn_merge_launch(const uint32_t *d_part, int64_t nq, int n_splits,
                            int64_t rows_per_split, int k, int32_t index_base,
                            int32_t *d_out_idx, int32_t *d_out_dist, uint8_t *d_keep, float ratio,
                            hipStream_t stream);

// The same for a batch of frames of at most max_nq query rows each.  waves > 0: the
// splits of knn_frames_launch with that many waves; waves < 0: the matrix-core kernel's.
hipError_t knn_merge_frames_launch(const uint32_t *d_part, const int32_t *d_q_off,
                                   const int32_t *d_t_off, int64_t total_nq, int64_t max_nq,
                                   int n_frames, int n_splits, int waves, int k,
                                   int32_t *d_out_idx, int32_t *d_out_dist, uint8_t *d_keep,
                                   float ratio, hipStream_t stream);

// Train rows per split of a frame with nt train rows under plan p.
int64_t frame_rows_per_split(int64_t nt, const KnnPlan &p);

// Sustained rate of the inner loop's instruction mix (v_xor_b32 with an SGPR
// operand + accumulating v_bcnt_u32_b32) with 8 waves per SIMD on every CU:
// the measured denominator of the matcher's roofline.  Blocks until done.
hipError_t valu_popcount_peak(int n_cus, int iters, uint32_t *d_scratch, hipStream_t stream,
                              double *lane_ops_per_s);

hipError_t knn_merge_shards_launch(const int32_t *d_cand_idx, const int32_t *d_cand_dist,
                                   int n_shards, int64_t nq, int k, int32_t *d_out_idx,
                                   int32_t *d_out_dist, uint8_t *d_keep, float ratio,
                                   hipStream_t stream);

// Cross-check resolve.  d_sidx / d_sdist: every query row's nearest train row and its distance;
// d_tidx: every train row's nearest query row.  Query q keeps (d_sidx[q] + index_base,
// d_sdist[q]) iff d_tidx[d_sidx[q]] == q, else (-1, INT_MAX).  packed (a shard of a larger
// train set, whose result still has to meet the other shards'): every query keeps its
// candidate, the index as 2 * (d_sidx[q] + index_base) + (1 iff d_tidx[d_sidx[q]] == q).
hipError_t crosscheck_resolve_launch(const int32_t *d_sidx, const int32_t *d_sdist,
                                     const int32_t *d_tidx, int64_t nq, int32_t *d_out_idx,
                                     int32_t *d_out_dist, hipStream_t stream,
                                     int32_t index_base = 0, bool packed = false);

// The same for a batch of frames: d_sidx / d_sdist indexed by query row of the whole buffer
// (frame-local train indices), d_tidx by train row of the whole buffer (frame-local query
// indices); outputs by query row (frame-local train indices).
hipError_t crosscheck_resolve_frames_launch(const int32_t *d_sidx, const int32_t *d_sdist,
                                            const int32_t *d_tidx, const int32_t *d_q_off,
                                            const int32_t *d_t_off, int n_frames, int64_t max_nq,
                                            int32_t *d_out_idx, int32_t *d_out_dist,
                                            hipStream_t stream);

// Unpacks merged shard candidates of the cross-check in place: idx = 2 * index + flag
// becomes index where the flag is set, (-1, INT_MAX) where it is not.
hipError_t crosscheck_unpack_launch(int32_t *d_idx, int32_t *d_dist, int64_t nq,
                                    hipStream_t stream);

} // namespace posehip
