// How fast the 12 x 12 one-sided Jacobi decomposition (65 % of a RANSAC-EPnP hypothesis) runs
// on its own: the library's routine (two passes interleaved, ~500 VGPRs, one wave per SIMD)
// against the plain routine capped at 256 VGPRs with two waves of 16 matrices per SIMD.
// Same matrices in LDS either way (128 per CU).  Answers whether a separate Jacobi kernel
// with a leaner register footprint would pay.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp -Iposeestimation_amd/csrc tools/jacobi_rate.hip -o tools/jacobi_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
#include <algorithm>
#include <cmath>
#include "../poseestimation_amd/csrc/ransac_epnp.hip"
using namespace posehip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int HPW, int VARIANT> __device__ __forceinline__ void body(const double *in, double *out, int n)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int h = blockIdx.x * HPW + lane;
    if (lane >= HPW || h >= n)
        return;
    const EpnpLds<HPW> S((lds_double *)lds + lane);
    for (int e = 0; e < 144; e++)
        S.ut[e] = in[(size_t)h * 144 + e];
    if (VARIANT == 0)
        jacobi_svd_12<HPW>(S.ut, S.w);
    else
        jacobi_svd<HPW, false, 12, 12>(S.ut, 12, S.w, S.ut, 12, 12, 12);
    double acc = 0;
    for (int e = 0; e < 12; e++)
        acc += S.w[e] + S.ut[11 * 12 + e];
    out[h] = acc;
}

template <int HPW> __global__ __launch_bounds__(64) void k_lib(const double *in, double *out, int n) { body<HPW, 0>(in, out, n); }
template <int HPW> __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lean(const double *in, double *out, int n) { body<HPW, 1>(in, out, n); }
template <int HPW> __global__ __launch_bounds__(64) void k_plain(const double *in, double *out, int n) { body<HPW, 1>(in, out, n); }
template <int HPW> __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lib_lean(const double *in, double *out, int n) { body<HPW, 0>(in, out, n); }

// The sweeps with the matrix in REGISTERS, one lane per matrix, 64 matrices per wave: the cyclic
// pair order fully unrolled so that every row index is a constant (288 VGPRs for the matrix,
// 24 for the norms), no LDS at all.  The register file holds 4 x 64 matrices per CU where LDS
// holds 128.  out[h] = sum of the squared row norms after the sweeps (= the squared Frobenius
// norm of the input, which rotations keep) + the largest |dot| left between two rows relative
// to their norms (~1e-15 when converged).
template <int I, int J>
__device__ __forceinline__ void reg_pair(double (&A)[12][12], double (&W)[12], double eps, bool &changed)
{
    double p = 0;
#pragma unroll
    for (int k = 0; k < 12; k++)
        p += A[I][k] * A[J][k];
    double a = W[I], b = W[J];
    const bool rotate = !(fabs(p) <= eps * sqrt(a * b));
    if (__builtin_amdgcn_ballot_w64(rotate) == 0)
        return;
    if (rotate) {
        p *= 2;
        const double beta = a - b, gamma = cv_hypot(p, beta);
        double c, s;
        jacobi_rotation(p, beta, gamma, c, s);
        a = b = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const double t0 = c * A[I][k] + s * A[J][k];
            const double t1 = -s * A[I][k] + c * A[J][k];
            A[I][k] = t0;
            A[J][k] = t1;
            a += t0 * t0;
            b += t1 * t1;
        }
        W[I] = a;
        W[J] = b;
        changed = true;
    }
}
template <int I, int J>
__device__ __forceinline__ void reg_sweep(double (&A)[12][12], double (&W)[12], double eps, bool &changed)
{
    reg_pair<I, J>(A, W, eps, changed);
    if constexpr (J + 1 < 12)
        reg_sweep<I, J + 1>(A, W, eps, changed);
    else if constexpr (I + 2 < 12)
        reg_sweep<I + 1, I + 2>(A, W, eps, changed);
}
__global__ __launch_bounds__(64) void k_reg(const double *in, double *out, int n)
{
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= n)
        return;
    double A[12][12], W[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            A[i][k] = in[(size_t)h * 144 + i * 12 + k];
            sd += A[i][k] * A[i][k];
        }
        W[i] = sd;
    }
    const double eps = DBL_EPSILON * 10;
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
        reg_sweep<0, 1>(A, W, eps, changed);
        if (__builtin_amdgcn_ballot_w64(changed) == 0)
            break;
    }
    double acc = 0, worst = 0;
#pragma unroll
    for (int i = 0; i < 12; i++)
        acc += W[i];
#pragma unroll
    for (int i = 0; i < 11; i++) {
        double p = 0;
#pragma unroll
        for (int k = 0; k < 12; k++)
            p += A[i][k] * A[i + 1][k];
        const double rel = fabs(p) / sqrt(W[i] * W[i + 1] + DBL_MIN);
        worst = rel > worst && W[i] > 1e-6 * acc && W[i + 1] > 1e-6 * acc ? rel : worst;
    }
    out[h] = acc;
    out[n + h] = worst;
}

template <typename Kern> static int run(const char *name, Kern kern, int hpw, const double *d_in, double *d_out, int n, std::vector<double> &res)
{
    const size_t lds = (size_t)hpw * LDS_DOUBLES * sizeof(double);
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(e0));
        kern<<<(n + hpw - 1) / hpw, 64, lds, 0>>>(d_in, d_out, n);
        CK(hipGetLastError());
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    res.resize(n);
    CK(hipMemcpy(res.data(), d_out, (size_t)n * 8, hipMemcpyDeviceToHost));
    printf("%-28s HPW=%2d: %.3f ms for %d matrices = %.3e /s\n", name, hpw, best, n, n / (best * 1e-3));
    fflush(stdout);
    return 0;
}

int main()
{
    const int n = 32768;
    std::mt19937_64 rng(5);
    std::normal_distribution<double> N(0, 1);
    std::vector<double> in((size_t)n * 144);
    for (int h = 0; h < n; h++) { // MtM of a random 10 x 12 matrix with a spread of scales
        double M[10][12];
        for (auto &row : M) for (int k = 0; k < 12; k++) row[k] = N(rng) * (k % 3 == 2 ? 1.0 : 300.0);
        for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) {
            double s = 0; for (int r = 0; r < 10; r++) s += M[r][i] * M[r][j];
            in[(size_t)h * 144 + i * 12 + j] = s;
        }
    }
    double *d_in, *d_out;
    CK(hipMalloc(&d_in, in.size() * 8)); CK(hipMalloc(&d_out, (size_t)n * 8));
    CK(hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> a, b, c, d, e;
    if (run("library routine", k_lib<32>, 32, d_in, d_out, n, a)) return 1;
    if (run("library routine", k_lib<16>, 16, d_in, d_out, n, e)) return 1;
    if (run("plain routine", k_plain<32>, 32, d_in, d_out, n, b)) return 1;
    if (run("plain, <=256 VGPRs, 2/SIMD", k_lean<16>, 16, d_in, d_out, n, c)) return 1;
    if (run("plain, 4 waves/SIMD", k_plain<8>, 8, d_in, d_out, n, c)) return 1;
    if (run("plain, 2 waves/SIMD", k_plain<16>, 16, d_in, d_out, n, c)) return 1;
    if (run("plain, 8 waves/SIMD", k_plain<4>, 4, d_in, d_out, n, c)) return 1;
    if (run("library, <=256 VGPRs, 2/SIMD", k_lib_lean<16>, 16, d_in, d_out, n, d)) return 1;
    { // register-resident sweeps: 64 matrices per wave, twice the matrices so that every SIMD has a wave
        const int n2 = 2 * n;
        double *d_in2, *d_out2;
        CK(hipMalloc(&d_in2, (size_t)n2 * 144 * 8)); CK(hipMalloc(&d_out2, (size_t)n2 * 2 * 8));
        CK(hipMemcpy(d_in2, in.data(), in.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_in2 + in.size(), in.data(), in.size() * 8, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int m : {n, n2}) {
            float best = 1e30f;
            for (int r = 0; r < 4; r++) {
                CK(hipEventRecord(e0));
                k_reg<<<(m + 63) / 64, 64, 0, 0>>>(d_in2, d_out2, m);
                CK(hipGetLastError());
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0 && ms < best) best = ms;
            }
            std::vector<double> o((size_t)m * 2);
            CK(hipMemcpy(o.data(), d_out2, o.size() * 8, hipMemcpyDeviceToHost));
            double frob_err = 0, worst = 0;
            for (int h = 0; h < m; h++) {
                double f = 0;
                for (int e2 = 0; e2 < 144; e2++) { const double v = in[(size_t)(h % n) * 144 + e2]; f += v * v; }
                frob_err = std::max(frob_err, std::fabs(o[h] - f) / f);
                worst = std::max(worst, o[(size_t)m + h]);
            }
            printf("%-28s HPW=64: %.3f ms for %d matrices = %.3e /s  (norm drift %.1e, residual dot %.1e)\n",
                   "matrix in registers", best, m, m / (best * 1e-3), frob_err, worst);
        }
    }
    int bad = 0;
    for (int h = 0; h < n; h++) bad += (a[h] != b[h]) + (a[h] != c[h]) + (a[h] != d[h]) + (a[h] != e[h]);
    printf("results identical across variants: %s\n", bad ? "NO" : "yes");
    return 0;
}
