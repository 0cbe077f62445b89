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
// rows -> d_counts[h].  Problem p owns rows [d_offsets[p], d_offsets[p+1]); d_offsets ==
// nullptr means one problem over rows 0..n.  thr2 = (float)(reprojectionError^2).
hipError_t ransac_hypotheses_launch(const float *d_obj, const float *d_img, int n, CameraK K,
                                    const int32_t *d_subsets, const int32_t *d_offsets,
                                    int n_problems, int hyp_per_problem, float thr2,
                                    double *d_models, int32_t *d_counts, int n_cus,
                                    hipStream_t stream);

// One byte per row: 1 iff the row is an inlier of d_model (rvec|tvec).
hipError_t inlier_mask_launch(const float *d_obj, const float *d_img, int n, CameraK K,
                              const double *d_model, float thr2, uint8_t *d_mask,
                              hipStream_t stream);

// EPnP on rows d_idx[0..n) (d_idx may be null: rows 0..n) -> d_model = rvec|tvec.
hipError_t epnp_solve_launch(const float *d_obj, const float *d_img, const int32_t *d_idx, int n,
                             CameraK K, double *d_model, hipStream_t stream);

} // namespace posehip
