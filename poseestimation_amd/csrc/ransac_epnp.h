// Internal interface between the C ABI (posehip_api.hip) and the RANSAC-EPnP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace posehip {

struct CameraK {
    double fx, fy, cx, cy;
};

// n_problems x hyp_per_problem hypotheses: EPnP on rows d_subsets[5h..5h+5) of the
// hypothesis' problem -> d_models[6h..] = rvec|tvec and the inlier count over that problem's
// rows -> d_counts[h].  Problem p owns rows [d_begins[p], d_ends[p]) (for n_problems + 1
// ascending offsets pass d_ends = d_begins + 1); d_begins == nullptr means one problem over
// rows 0..n.  thr2 = (float)(reprojectionError^2).
// overlap (optional): a second stream and two events with which the scoring of one
// GPU-full of hypotheses runs underneath the next one's solves.
struct RansacOverlap {
    hipStream_t side;
    hipEvent_t solved, scored;
};
hipError_t ransac_hypotheses_launch(const float *d_obj, const float *d_img, int n, CameraK K,
                                    const int32_t *d_subsets, const int32_t *d_begins,
                                    const int32_t *d_ends, int n_problems, int hyp_per_problem,
                                    float thr2, double *d_models, int32_t *d_counts, int n_cus,
                                    hipStream_t stream, const RansacOverlap *overlap = nullptr,
                                    hipEvent_t *events = nullptr, double *d_front = nullptr);
// d_front, when given (ransac_front_workspace_bytes(n_cus) bytes), lets a batch of more than
// 64 hypotheses per CU take the two-kernel solve (decomposition at two waves per SIMD, then
// the rest), one GPU-full of hypotheses per round.
size_t ransac_front_workspace_bytes(int n_cus);
// Which solve a large batch takes: 0 / 1 the single kernel (the default: it is the fastest),
// 2 two kernels (second half at one wave per SIMD), 3 two kernels with the second half at
// two waves per SIMD as well.  Bit-identical models either way.
void ransac_set_solve_variant(int variant);
int ransac_solve_variant();
// events, when given (five of them), are recorded on `stream` around the first round's solve
// kernel [0, 1], the last round's solve kernel [2, 3] (the same launch when the call is one
// round) and the last round's scoring kernel [3, 4].

// Sustained fp64 VALU rates in lane-operations per second with 8 waves per SIMD on every
// CU: v_fma_f64 alone, and the v_mul_f64 + v_add_f64 mix of code built without
// contraction.  d_scratch: n_cus * 8 * 256 doubles.  Blocks until measured.
hipError_t valu_fp64_peak(int n_cus, int iters, double *d_scratch, hipStream_t stream,
                          double *fma_per_s, double *mul_add_per_s);

// EPnP on rows d_idx[0..n) (d_idx may be null: rows 0..n) -> d_model = rvec|tvec.
hipError_t epnp_solve_launch(const float *d_obj, const float *d_img, const int32_t *d_idx, int n,
                             CameraK K, double *d_model, hipStream_t stream);

// The end of solvePnPRansac for a batch in one launch (one workgroup of three waves per
// problem, a wave per EPnP candidate): problem b owns
// rows [d_begins[b], d_ends[b]); its inliers under d_best_models[6b..] (every row where
// d_use_all_rows[b] != 0) are listed in ascending order, relative to d_begins[b], at
// d_idx + d_begins[b] (device scratch the kernel reads back; as many entries as rows overall)
// and d_idx_out + d_begins[b]; d_n_inliers[b] = the count; EPnP on the listed rows ->
// d_models_out[6b..] = rvec|tvec (untouched if fewer than 4 rows are listed).
// d_begins == nullptr: one problem over rows 0..one.rows with the model one.best, both
// passed by value (d_ends, d_best_models and d_use_all_rows are then unused).
struct RefitOne {
    double best[6];
    int rows;
};
// With `replay` (single problem only: d_begins == nullptr) the kernel first chooses the model
// itself, by OpenCV's sequential selection rule over the device-resident inlier counts of the
// call's hypotheses (strictly larger count wins, every win shrinks the iteration budget by
// RANSACUpdateNumIters), so that a lone solvePnPRansac needs no host round trip between
// scoring and refit.  iter_table[g] = log(1 - (g / rows)^5) as the HOST's libm computes it
// (NaN where the host's rule returns 0), num_log = log(max(1 - confidence, DBL_MIN)): the
// device only multiplies, compares, divides and rounds, each bit-identical to the host's.
// out[0] = iterations the sequential loop would have run, out[1] = index of the chosen
// hypothesis or -1 (then d_n_inliers[0] = 0 and nothing else is written).
struct RefitReplay {
    const int32_t *counts = nullptr; // [total] inlier counts, in hypothesis order
    const double *models = nullptr;  // [total][6]
    const double *iter_table = nullptr; // [rows + 1]
    double num_log = 0;
    int total = 0;
    int32_t *out = nullptr;
};
hipError_t epnp_refit_batched_launch(const float *d_obj, const float *d_img,
                                     const int32_t *d_begins, const int32_t *d_ends,
                                     const double *d_best_models, const uint8_t *d_use_all_rows,
                                     const RefitOne &one, int n_problems, CameraK K, float thr2,
                                     int32_t *d_idx,
                                     int32_t *d_idx_out, int32_t *d_n_inliers,
                                     double *d_models_out, hipStream_t stream,
                                     const RefitReplay *replay = nullptr);

} // namespace posehip
