// Issue-rate microbenchmark for the matrix-core matcher's instruction mix on gfx950:
// cycles per loop iteration of 16 x v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 operands, four
// chains of four) with NV VALU instructions (v_max3_i32 on MFMA-independent registers)
// spread between them, at 1, 2 and 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_rate.hip -o tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int NV, int SCALE>
__global__ __launch_bounds__(256) void rate_kernel(const uint4 *in, float *out, int iters, unsigned long long *cycles)
{
    const uint4 v = in[threadIdx.x];
    i32x8 a = {(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
    i32x8 b = {(int)v.w, (int)v.z, (int)v.y, (int)v.x, 0, 0, 0, 0};
    f32x16 c0, acc[4];
    for (int i = 0; i < 16; i++)
        c0[i] = (float)(i + threadIdx.x);
    for (int j = 0; j < 4; j++)
        acc[j] = c0;
    int m[8];
    for (int i = 0; i < 8; i++)
        m[i] = threadIdx.x * (i + 3);
    constexpr int SA = SCALE ? 0x89898989 : 0, SB = SCALE ? 0x7F7F7F7F : 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int s = 0; s < 4; s++)
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 4, 4, 0, SA, 0, SB);
        }
#pragma unroll
        for (int i = 0; i < NV; i++)
            m[i & 7] = max(max(m[i & 7], m[(i + 1) & 7]), m[(i + 3) & 7]);
#pragma unroll
        for (int g = 0; g < 16; g++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (NV + 15) / 16, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 16; i++)
            s += acc[j][i];
    int ms = 0;
    for (int i = 0; i < 8; i++)
        ms ^= m[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)ms;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *cycles = t1 - t0;
}

template <int NV, int SCALE> static void run(int blocks_per_cu, int n_cus, const uint4 *in, float *out, unsigned long long *dcyc)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    rate_kernel<NV, SCALE><<<n_cus * blocks_per_cu, 256>>>(in, out, iters, dcyc);
    CK(hipEventRecord(e0));
    rate_kernel<NV, SCALE><<<n_cus * blocks_per_cu, 256>>>(in, out, iters, dcyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long cyc;
    CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
    // memtime counts at 100 MHz on this part: report wall-clock-derived figures too
    const double per_iter_ns = ms * 1e6 / iters;
    printf("valu/iter=%3d scale=%d waves/SIMD=%d: %.1f ns per iteration per wave-slot (%.1f ns per MFMA-slot), "
           "counter %.1f ticks/iter, %.2f PFLOP/s\n",
           NV, SCALE, blocks_per_cu, per_iter_ns, per_iter_ns / 16, (double)cyc / iters,
           (double)n_cus * blocks_per_cu * 4 * iters * 16 * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e15);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    uint4 *in;
    float *out;
    unsigned long long *dcyc;
    CK(hipMalloc(&in, 256 * 16));
    {
        unsigned char h[256 * 16];
        for (int i = 0; i < 256 * 16; i++)
            h[i] = (unsigned char)(0x66 | ((rand() & 1) << 3) | ((rand() & 1) << 7));
        CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&out, (size_t)n_cus * 3 * 256 * 4));
    CK(hipMalloc(&dcyc, 8));
    for (int w = 1; w <= 3; w++) {
        run<0, 1>(w, n_cus, in, out, dcyc);
        run<0, 0>(w, n_cus, in, out, dcyc);
        run<48, 1>(w, n_cus, in, out, dcyc);
        run<64, 1>(w, n_cus, in, out, dcyc);
        run<80, 1>(w, n_cus, in, out, dcyc);
        run<96, 1>(w, n_cus, in, out, dcyc);
        run<96, 0>(w, n_cus, in, out, dcyc);
        run<112, 1>(w, n_cus, in, out, dcyc);
    }
    return 0;
}
