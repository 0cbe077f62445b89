// Checks the fp32 first pass of the RANSAC scoring kernel against the exact evaluation it
// stands in for: the kernel is compiled with SCORE_STATS, which makes it evaluate
// inlier_of()'s error for EVERY (row, model) beside the fp32 verdict and count
//   verdicts, uncertain verdicts, certain verdicts that differ from the exact one (must be 0),
//   batches of 64 rows sent to the exact path, and the largest |err32 - err| / B seen among
//   the certain verdicts (must stay below 1; the constants leave it far below).
// Models: EPnP hypotheses of scenes with outliers (good and bad poses alike), then random
// poses at every depth (behind the camera, at almost no depth, far away), tiny and huge
// rotations, poses that put rows exactly on the threshold circle.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DSCORE_STATS -Iposeestimation_amd/csrc tools/score_filter_check.hip -o tools/score_filter_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../poseestimation_amd/csrc/ransac_epnp.hip"
using namespace posehip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

static int report(const char *what, double seconds = 0)
{
    unsigned long long st[8];
    if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_score_stats), sizeof(st)) != hipSuccess) return 1;
    float worst;
    const unsigned bits = (unsigned)st[5];
    memcpy(&worst, &bits, 4);
    printf("{\"models\": \"%s\", \"verdicts\": %llu, \"uncertain\": %llu, \"uncertain_share\": %.3g, "
           "\"wrong_certain_verdicts\": %llu, \"batches\": %llu, \"batches_on_exact_path\": %llu, "
           "\"exact_path_share\": %.3g, \"max_error_over_bound\": %.4g}\n",
           what, st[0], st[1], st[0] ? (double)st[1] / st[0] : 0.0, st[2], st[4], st[3],
           st[4] ? (double)st[3] / st[4] : 0.0, worst);
    (void)seconds;
    memset(st, 0, sizeof(st));
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_score_stats), st, sizeof(st)) != hipSuccess) return 1;
    return st[2] != 0;
}

int main()
{
    const int n = 2000, H = 8192;
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<double> U(-1, 1);
    std::normal_distribution<double> N(0, 1);
    const double f = 300.0 / std::tan(22.5 * M_PI / 180), ang = 0.6;
    std::vector<float> obj(3 * n), img(2 * n);
    for (int i = 0; i < n; i++) {
        double X = U(rng), Y = 1.2934 * U(rng), Z = 0.41149 * U(rng);
        double x = std::cos(ang) * X + std::sin(ang) * Z, y = Y, z = -std::sin(ang) * X + std::cos(ang) * Z + 6;
        obj[3 * i] = X; obj[3 * i + 1] = Y; obj[3 * i + 2] = Z;
        bool out = (rng() % 10) < 3;
        img[2 * i] = out ? 400 + 400 * U(rng) : x / z * f + 400 + N(rng);
        img[2 * i + 1] = out ? 300 + 300 * U(rng) : y / z * f + 300 + N(rng);
    }
    std::vector<int32_t> sub(5 * H);
    for (int h = 0; h < H; h++)
        for (int k = 0; k < 5; k++) sub[5 * h + k] = (int)(rng() % n);
    float *d_obj, *d_img; int32_t *d_sub, *d_counts; double *d_models;
    CK(hipMalloc(&d_obj, obj.size() * 4)); CK(hipMalloc(&d_img, img.size() * 4)); CK(hipMalloc(&d_sub, sub.size() * 4));
    CK(hipMalloc(&d_counts, H * 4)); CK(hipMalloc(&d_models, H * 48));
    CK(hipMemcpy(d_obj, obj.data(), obj.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sub, sub.data(), sub.size() * 4, hipMemcpyHostToDevice));
    CameraK K{f, f, 400, 300};
    int bad = 0;
    // 1. the hypotheses of the scene itself (solve + score through the library's launcher)
    CK(ransac_hypotheses_launch(d_obj, d_img, n, K, d_sub, nullptr, nullptr, 1, H, 64.f, d_models, d_counts, 256, 0));
    CK(hipDeviceSynchronize());
    bad += report("EPnP hypotheses of a scene with 30 % outliers");
    // 2. random poses
    std::vector<double> m(6 * (size_t)H);
    auto score = [&](const char *what) -> int {
        if (hipMemcpy(d_models, m.data(), m.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return 1;
        ransac_score_kernel<<<(H + SCORE_HPB - 1) / SCORE_HPB, 64 * SCORE_WAVES>>>(
            d_obj, d_img, n, K, d_models, nullptr, nullptr, H, 0, H, 64.f, d_counts);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        return report(what);
    };
    for (int h = 0; h < H; h++) { // near the true pose: most rows are inliers or near the circle
        m[6 * h] = 0.05 * N(rng); m[6 * h + 1] = ang + 0.05 * N(rng); m[6 * h + 2] = 0.05 * N(rng);
        m[6 * h + 3] = 0.1 * N(rng); m[6 * h + 4] = 0.1 * N(rng); m[6 * h + 5] = 6 + 0.3 * N(rng);
    }
    bad += score("poses around the true one");
    for (int h = 0; h < H; h++) { // anything: behind the camera, through the box, far away
        for (int k = 0; k < 3; k++) m[6 * h + k] = 3.2 * U(rng);
        m[6 * h + 3] = 10 * U(rng); m[6 * h + 4] = 10 * U(rng); m[6 * h + 5] = 12 * U(rng);
    }
    bad += score("random poses, depth in [-12, 12]");
    for (int h = 0; h < H; h++) { // the camera inside the box: rows at almost no depth
        for (int k = 0; k < 3; k++) m[6 * h + k] = 3.2 * U(rng);
        m[6 * h + 3] = 0.3 * U(rng); m[6 * h + 4] = 0.3 * U(rng); m[6 * h + 5] = 0.05 * U(rng);
    }
    bad += score("camera inside the box");
    for (int h = 0; h < H; h++) { // huge and tiny magnitudes, NaN and infinity
        const int kind = h % 8;
        const double big = kind == 0 ? 1e6 : kind == 1 ? 1e-9 : kind == 2 ? 1e30 : kind == 3 ? 1.0
                         : kind == 4 ? 3e38 : kind == 5 ? 1e-40 : kind == 6 ? 1e18 : 1e300;
        for (int k = 0; k < 3; k++) m[6 * h + k] = (kind == 2 ? 1e3 : 1.0) * U(rng);
        for (int k = 3; k < 6; k++) m[6 * h + k] = big * U(rng);
        if (h % 97 == 0) m[6 * h + 5] = NAN;
        if (h % 101 == 0) m[6 * h + 3] = INFINITY;
    }
    bad += score("huge, tiny and non-finite translations");
    printf("%s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
