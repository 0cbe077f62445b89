// fp64 VALU issue rate on gfx950: dependent and independent chains of v_fma_f64 /
// v_mul_f64 + v_add_f64 at 1..8 waves per SIMD (the RANSAC-EPnP solve is made of these).
// build: hipcc -O3 --offload-arch=gfx950 tools/fp64_rate.hip -o tools/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int CHAINS, bool FMA>
__global__ __launch_bounds__(64) void rate_kernel(double *out, double a, double b, int iters)
{
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = a + c + threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (FMA) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
                else { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b)); }
            }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int CHAINS, bool FMA> int run(double *d, int waves_per_simd, int lanes)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * 4 * waves_per_simd, iters = 2000;
    (void)lanes;
    rate_kernel<CHAINS, FMA><<<blocks, 64>>>(d, 1.0000001, 1e-9, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    rate_kernel<CHAINS, FMA><<<blocks, 64>>>(d, 1.0000001, 1e-9, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_wave = (double)iters * 16 * CHAINS * (FMA ? 1 : 2);
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("{\"op\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"cycles_per_instr_per_wave\": %.2f, \"cycles_per_instr_per_simd\": %.2f}\n",
           FMA ? "fma" : "mul+add", CHAINS, waves_per_simd, ms, cycles / instr_per_wave, cycles / (instr_per_wave * waves_per_simd));
    return 0;
}

int main()
{
    double *d; CK(hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(double)));
    for (int w : {1, 2, 4, 8}) {
        if (run<1, true>(d, w, 64)) return 1;
        if (run<4, true>(d, w, 64)) return 1;
        if (run<1, false>(d, w, 64)) return 1;
        if (run<4, false>(d, w, 64)) return 1;
    }
    return 0;
}
