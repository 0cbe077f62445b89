// Per-phase cycle counts of one RANSAC-EPnP hypothesis (wave 0), for tuning.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp -DPNP_STAMPS -Iposeestimation_amd/csrc tools/pnp_stamps.hip -o tools/pnp_stamps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <random>
#include <vector>
__device__ long long g_pnp_stamps[32];
#include "../poseestimation_amd/csrc/ransac_epnp.hip"
using namespace posehip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
static int print_hyp_stamps()
{
    long long st[32];
    if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_pnp_stamps), sizeof(st)) != hipSuccess) return 1;
    // hypothesis kernels: the candidates' Gauss-Newton and pose steps run side by side in the lanes of a group
    const char *names[] = {"", "control points", "basis inverse", "MtM", "jacobi 12x12", "L_6x10", "betas1", "betas2", "betas3", "gn 1|2", "gn 3", "Rt round 1", "Rt round 2", "-", "-", "rodrigues"};
    for (int k = 1; k <= 15; k++) if (st[k] == 0) st[k] = st[k - 1]; // stamps this mode does not write
    for (int k = 1; k <= 15; k++) printf("%-16s %9lld cycles\n", names[k], st[k] - st[k - 1]);
    printf("total            %9lld cycles\n", st[15] - st[0]);
    return 0;
}
int main(int argc, char **argv)
{
    const int n = 2000, H = argc > 1 ? atoi(argv[1]) : 4096;
    std::mt19937_64 rng(3);
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
    for (int h = 0; h < H; h++) for (int k = 0; k < 5; k++) sub[5 * h + k] = (int)(rng() % n);
    float *d_obj, *d_img; int32_t *d_sub, *d_counts; double *d_models;
    CK(hipMalloc(&d_obj, obj.size() * 4)); CK(hipMalloc(&d_img, img.size() * 4)); CK(hipMalloc(&d_sub, sub.size() * 4));
    CK(hipMalloc(&d_counts, H * 4)); CK(hipMalloc(&d_models, H * 48));
    CK(hipMemcpy(d_obj, obj.data(), obj.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sub, sub.data(), sub.size() * 4, hipMemcpyHostToDevice));
    CameraK K{f, f, 400, 300};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (argc > 1 && !strcmp(argv[1], "refit")) { // refit [n]: per-phase cycles of the one-wave refit kernel
        const int m = argc > 2 ? atoi(argv[2]) : 354;
        std::vector<int32_t> id(m);
        for (int i = 0; i < m; i++) id[i] = i;
        int32_t *d_id; double *d_model;
        CK(hipMalloc(&d_id, m * 4)); CK(hipMalloc(&d_model, 48));
        CK(hipMemcpy(d_id, id.data(), m * 4, hipMemcpyHostToDevice));
        for (int r = 0; r < 3; r++) {
            CK(hipEventRecord(e0));
            CK(epnp_solve_launch(d_obj, d_img, d_id, m, K, d_model, 0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("refit on %d points: %.1f us\n", m, ms * 1e3);
        }
        long long st[32];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_pnp_stamps), sizeof(st)));
        // wave 0 of the three: the common part and candidate 1 (waves 1 and 2 run candidates 2 and 3 meanwhile)
        const char *names[] = {"", "control points", "basis inverse", "MtM", "jacobi 12x12", "L_6x10", "betas1", "gn1", "Rt1"};
        for (int k = 1; k <= 8; k++) printf("%-16s %9lld cycles\n", names[k], st[k] - st[k - 1]);
        printf("total (wave 0)   %9lld cycles\n", st[8] - st[0]);
        return 0;
    }
    if (argc > 3) { // H HPW LDS_KB: solve kernel only, workgroup LDS padded to LDS_KB (spreads the waves)
        const int hpw = atoi(argv[2]);
        const size_t lds = (size_t)atoi(argv[3]) * 1024;
        for (int r = 0; r < 3; r++) {
            CK(hipEventRecord(e0));
            if (hpw == 4) {
                CK(hipFuncSetAttribute((const void *)ransac_solve_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ransac_solve_kernel<4><<<(H + 3) / 4, 64, lds, 0>>>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models);
            } else if (hpw == 8) {
                CK(hipFuncSetAttribute((const void *)ransac_solve_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ransac_solve_kernel<8><<<(H + 7) / 8, 64, lds, 0>>>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models);
            } else {
                CK(hipFuncSetAttribute((const void *)ransac_solve_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ransac_solve_kernel<16><<<(H + 15) / 16, 64, lds, 0>>>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models);
            }
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("H=%d HPW=%d LDS=%zu KB solve only: %.3f ms -> %.3g hyps/s\n", H, hpw, lds / 1024, ms, H / (ms * 1e-3));
        }
        return 0;
    }
    if (argc > 2) { // force hypotheses per workgroup: solve kernel only
        const int hpw = atoi(argv[2]);
        for (int r = 0; r < 3; r++) {
            CK(hipEventRecord(e0));
            if (hpw == 5) CK(launch_solve3<4>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models, 0)); // 4 per workgroup of three waves
            else if (hpw == 3) CK(launch_solve3<8>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models, 0)); // 8 per workgroup of three waves
            else if (hpw == 1) CK(launch_solve<1>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models, 0));
            else if (hpw == 2) CK(launch_solve<2>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models, 0));
            else if (hpw == 4) CK(launch_solve<4>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models, 0));
            else if (hpw == 8) CK(launch_solve<8>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models, 0));
            else if (hpw == 16) CK(launch_solve<16>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models, 0));
            else if (hpw == 32) CK(launch_solve<32>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models, 0));
            else CK(launch_solve<64>(d_obj, d_img, K, d_sub, nullptr, H, 0, H, d_models, 0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("H=%d HPW=%d solve only: %.3f ms -> %.3g hyps/s\n", H, hpw, ms, H / (ms * 1e-3));
        }
        return print_hyp_stamps();
    }
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(e0));
        CK(ransac_hypotheses_launch(d_obj, d_img, n, K, d_sub, nullptr, nullptr, 1, H, 64.f, d_models, d_counts, 256, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("H=%d: %.3f ms -> %.3g hyps/s\n", H, ms, H / (ms * 1e-3));
    }
    if (print_hyp_stamps()) return 1;
    std::vector<int32_t> c(H); CK(hipMemcpy(c.data(), d_counts, H * 4, hipMemcpyDeviceToHost));
    int mx = 0; for (int v : c) mx = v > mx ? v : mx;
    printf("max count %d\n", mx);
    return 0;
}
