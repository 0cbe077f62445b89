// Measures the sustained VALU issue rate of the exact instruction mix of the
// Hamming inner loop (v_xor_b32 with an SGPR operand + accumulating
// v_bcnt_u32_b32) on every CU at once.  Its result is the denominator of the
// matcher's roofline: evals/s peak = lane-ops/s / 16.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// 8 independent chains per lane; each step = 1 xor + 1 bcnt-accumulate.
template <int MODE>
__global__ __launch_bounds__(256) void mix_kernel(uint32_t *out, uint32_t s0, uint32_t s1, int iters)
{
    uint32_t q0 = threadIdx.x * 2654435761u, q1 = q0 ^ 0x9E3779B9u, q2 = q0 * 3, q3 = q0 * 5;
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) { // xor + bcnt
                uint32_t x0, x1, x2, x3, x4, x5, x6, x7;
                asm volatile(
                    "v_xor_b32 %8, %16, %20\n v_xor_b32 %9, %17, %21\n v_xor_b32 %10, %16, %22\n v_xor_b32 %11, %17, %23\n"
                    "v_xor_b32 %12, %16, %21\n v_xor_b32 %13, %17, %20\n v_xor_b32 %14, %16, %23\n v_xor_b32 %15, %17, %22\n"
                    "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n v_bcnt_u32_b32 %2, %10, %2\n v_bcnt_u32_b32 %3, %11, %3\n"
                    "v_bcnt_u32_b32 %4, %12, %4\n v_bcnt_u32_b32 %5, %13, %5\n v_bcnt_u32_b32 %6, %14, %6\n v_bcnt_u32_b32 %7, %15, %7\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                      "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
                    : "s"(s0), "s"(s1), "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q0), "v"(q1));
            } else if (MODE == 1) { // bcnt only
                asm volatile(
                    "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n v_bcnt_u32_b32 %2, %10, %2\n v_bcnt_u32_b32 %3, %11, %3\n"
                    "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %9, %5\n v_bcnt_u32_b32 %6, %10, %6\n v_bcnt_u32_b32 %7, %11, %7\n"
                    "v_bcnt_u32_b32 %0, %9, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %11, %2\n v_bcnt_u32_b32 %3, %10, %3\n"
                    "v_bcnt_u32_b32 %4, %9, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %11, %6\n v_bcnt_u32_b32 %7, %10, %7\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "v"(q0), "v"(q1), "v"(q2), "v"(q3));
            } else { // xor only (reference full-rate op)
                asm volatile(
                    "v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %9, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %9, %3\n"
                    "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %9, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %9, %7\n"
                    "v_xor_b32 %0, %9, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %9, %2\n v_xor_b32 %3, %8, %3\n"
                    "v_xor_b32 %4, %9, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %9, %6\n v_xor_b32 %7, %8, %7\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "s"(s0), "s"(s1));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> double run(int waves_per_simd, int iters, uint32_t *d_out, int n_cus)
{
    // 256 threads = 4 waves = one per SIMD; waves_per_simd workgroups per CU.
    const int blocks = n_cus * waves_per_simd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    mix_kernel<MODE><<<blocks, 256>>>(d_out, 0x12345678u, 0x9abcdef0u, iters / 8);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        mix_kernel<MODE><<<blocks, 256>>>(d_out, 0x12345678u, 0x9abcdef0u, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double ops = (double)blocks * 256.0 * (double)iters * 8 * 16; // lane-ops
    return ops / (ts[2] * 1e-3);
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int n_cus = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.name, p.gcnArchName, n_cus, p.clockRate / 1000);
    uint32_t *d_out; CK(hipMalloc(&d_out, (size_t)n_cus * 8 * 256 * 4));
    const int iters = 20000;
    for (int w : {1, 2, 4, 8}) {
        double a = run<0>(w, iters, d_out, n_cus);
        double b = run<1>(w, iters, d_out, n_cus);
        double c = run<2>(w, iters, d_out, n_cus);
        printf("{\"waves_per_simd\": %d, \"xor_bcnt_Tlaneops\": %.3f, \"bcnt_Tlaneops\": %.3f, \"xor_Tlaneops\": %.3f, "
               "\"xor_bcnt_lanes_per_clk_per_cu_at_2p4GHz\": %.1f, \"evals_per_s_at_16_ops\": %.4g}\n",
               w, a / 1e12, b / 1e12, c / 1e12, a / n_cus / 2.4e9, a / 16);
    }
    CK(hipFree(d_out));
    return 0;
}
