// Times ph_ransac_epnp_hypotheses_batched_dev (P problems x H hypotheses x n correspondences,
// 30 % outliers) through the C ABI; standalone so that rocprofv3 can wrap it.
// build: hipcc -O2 tools/pnp_time.cpp -Iinclude -Lposeestimation_amd -lposehip -Wl,-rpath,'$ORIGIN/../poseestimation_amd' -o tools/pnp_time_bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "posehip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
#define PH(x) do { ph_status st_ = (x); if (st_ != PH_OK) { printf("posehip error %d: %s\n", st_, ph_last_error_string()); exit(1); } } while (0)
int main(int argc, char **argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 16, H = argc > 2 ? atoi(argv[2]) : 4096, n = argc > 3 ? atoi(argv[3]) : 2000;
    PH(ph_init(0));
    std::mt19937_64 rng(3);
    std::uniform_real_distribution<double> U(-1, 1);
    std::normal_distribution<double> N(0, 1);
    const double f = 300.0 / std::tan(22.5 * M_PI / 180), K[9] = {f, 0, 400, 0, f, 300, 0, 0, 1};
    std::vector<float> obj((size_t)3 * n * P), img((size_t)2 * n * P);
    std::vector<int32_t> off(P + 1), sub((size_t)5 * H * P);
    for (int p = 0; p <= P; p++) off[p] = p * n;
    for (int p = 0; p < P; p++) {
        const double ang = 0.4 + 0.02 * p;
        for (int i = 0; i < n; i++) {
            const size_t r = (size_t)p * n + i;
            const double X = U(rng), Y = 1.2934 * U(rng), Z = 0.41149 * U(rng);
            const double x = std::cos(ang) * X + std::sin(ang) * Z, y = Y, z = -std::sin(ang) * X + std::cos(ang) * Z + 6;
            obj[3 * r] = (float)X; obj[3 * r + 1] = (float)Y; obj[3 * r + 2] = (float)Z;
            const bool out = (rng() % 10) < 3;
            img[2 * r] = (float)(out ? 400 + 400 * U(rng) : x / z * f + 400 + N(rng));
            img[2 * r + 1] = (float)(out ? 300 + 300 * U(rng) : y / z * f + 300 + N(rng));
        }
        PH(ph_ransac_subsets(0xFFFFFFFFFFFFFFFFull, n, H, sub.data() + (size_t)5 * H * p));
    }
    float *d_obj, *d_img; int32_t *d_off, *d_sub, *d_counts; double *d_models;
    CK(hipMalloc(&d_obj, obj.size() * 4)); CK(hipMalloc(&d_img, img.size() * 4)); CK(hipMalloc(&d_off, off.size() * 4));
    CK(hipMalloc(&d_sub, sub.size() * 4)); CK(hipMalloc(&d_counts, (size_t)H * P * 4)); CK(hipMalloc(&d_models, (size_t)H * P * 48));
    CK(hipMemcpy(d_obj, obj.data(), obj.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_sub, sub.data(), sub.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&] { PH(ph_ransac_epnp_hypotheses_batched_dev(d_obj, d_img, d_off, P, K, d_sub, H, 8.0f, d_models, d_counts, st)); };
    for (int w = 0; w < 2; w++) run();
    CK(hipStreamSynchronize(st));
    std::vector<float> ts;
    for (int r = 0; r < 8; r++) {
        CK(hipEventRecord(e0, st)); run(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    std::vector<int32_t> c((size_t)H * P); CK(hipMemcpy(c.data(), d_counts, c.size() * 4, hipMemcpyDeviceToHost));
    printf("{\"problems\": %d, \"hyp_per_problem\": %d, \"points\": %d, \"ms_median\": %.4f, \"hyps_per_s\": %.4g, \"max_count\": %d}\n",
           P, H, n, ts[ts.size() / 2], (double)P * H / (ts[ts.size() / 2] * 1e-3), *std::max_element(c.begin(), c.end()));
    ph_shutdown();
    return 0;
}
