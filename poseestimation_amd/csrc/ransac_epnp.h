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
                                    hipStream_t stream, const RansacOverlap *overlap = nullptr);

// One byte per row: 1 iff the row is an inlier of d_model (rvec|tvec).
hipError_t inlier_mask_launch(const float *d_obj, const float *d_img, int n, CameraK K,
                              const double *d_model, float thr2, uint8_t *d_mask,
                              hipStream_t stream);

// EPnP on rows d_idx[0..n) (d_idx may be null: rows 0..n) -> d_model = rvec|tvec.
hipError_t epnp_solve_launch(const float *d_obj, const float *d_img, const int32_t *d_idx, int n,
                             CameraK K, double *d_model, hipStream_t stream);

// The two steps above for a batch: problem p owns rows [d_begins[p], d_ends[p]) (at most
// max_n of them) and the model d_models[6p..]; d_mask is indexed by row.
hipError_t inlier_mask_batched_launch(const float *d_obj, const float *d_img,
                                      const int32_t *d_begins, const int32_t *d_ends,
                                      int n_problems, int max_n, CameraK K, const double *d_models,
                                      float thr2, uint8_t *d_mask, hipStream_t stream);

// Problem b refits on rows d_idx[d_idx_begins[b] .. + d_idx_counts[b]), relative to its first
// row d_row_begins[b] -> d_models[6b..].
hipError_t epnp_solve_batched_launch(const float *d_obj, const float *d_img,
                                     const int32_t *d_row_begins, const int32_t *d_idx,
                                     const int32_t *d_idx_begins, const int32_t *d_idx_counts,
                                     int n_problems, CameraK K, double *d_models,
                                     hipStream_t stream);

} // namespace posehip
