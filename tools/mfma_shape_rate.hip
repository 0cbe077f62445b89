// A/B of the two block-scaled FP4 MFMA shapes for the matcher's loop on gfx950, by WALL time
// and with the shader clock beside it (the chip can hold a different clock on either shape):
//   shape 32: 16 x v_mfma_scale_f32_32x32x64_f8f6f4  per iteration (4 chains of 4)
//   shape 16: 32 x v_mfma_scale_f32_16x16x128_f8f6f4 per iteration (16 chains of 2)
// Both are the same work per wave (128 queries x 32 train rows x 256 bits) with the same NV
// VALU instructions (v_max3_i32 on MFMA-independent registers) spread between the MFMAs,
// FP4 operands of +-4 at random, 1..3 waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_shape_rate.hip -o tools/mfma_shape_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// FMT: operand format code of the instruction (4 = FP4 E2M1, 3 = FP6 E3M2); SCALED: the
// block-scaled form with E8M0 scales 2^10 / 2^0, or (0) the form without scale operands.
template <int NV, int SHAPE, int FMT = 4, int SCALED = 1>
__global__ __launch_bounds__(256) void rate_kernel(const uint4 *in, float *out, int iters,
                                                   unsigned long long *stamps)
{
    constexpr int SA = SCALED ? 0x89898989 : 0, SB = SCALED ? 0x7F7F7F7F : 0;
    // 8 distinct B operands (query tiles) and 2 A operands, as in the kernel
    i32x8 a[4], b[8];
    for (int i = 0; i < 4; i++) {
        const uint4 v = in[(threadIdx.x + 64 * i) & 1023], u = in[(threadIdx.x + 64 * i + 500) & 1023];
        a[i] = i32x8{(int)v.x, (int)v.y, (int)v.z, (int)v.w, FMT == 4 ? 0 : (int)u.x, FMT == 4 ? 0 : (int)u.y, 0, 0};
    }
    for (int i = 0; i < 8; i++) {
        const uint4 v = in[(threadIdx.x * 3 + 64 * i + 7) & 1023], u = in[(threadIdx.x * 3 + 64 * i + 300) & 1023];
        b[i] = i32x8{(int)v.x, (int)v.y, (int)v.z, (int)v.w, FMT == 4 ? 0 : (int)u.x, FMT == 4 ? 0 : (int)u.y, 0, 0};
    }
    int m[8];
    for (int i = 0; i < 8; i++)
        m[i] = threadIdx.x * (i + 3);
    float s = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SHAPE == 32) {
        f32x16 acc[4];
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 16; i++)
                acc[j][i] = (float)(i + threadIdx.x);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[k], b[(2 * j + k) & 7], acc[j], FMT, FMT, 0, SA, 0, SB);
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
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 16; i++)
                s += acc[j][i];
    } else {
        f32x4 acc[16];
        for (int j = 0; j < 16; j++)
            for (int i = 0; i < 4; i++)
                acc[j][i] = (float)(i + threadIdx.x);
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
#pragma unroll
                for (int k = 0; k < 2; k++)
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[k + 2 * (j >> 3)], b[j & 7], acc[j], FMT, FMT, 0, SA, 0, SB);
            }
#pragma unroll
            for (int i = 0; i < NV; i++)
                m[i & 7] = max(max(m[i & 7], m[(i + 1) & 7]), m[(i + 3) & 7]);
#pragma unroll
            for (int g = 0; g < 32; g++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (NV + 31) / 32, 0);
            }
        }
        for (int j = 0; j < 16; j++)
            for (int i = 0; i < 4; i++)
                s += acc[j][i];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    int ms = 0;
    for (int i = 0; i < 8; i++)
        ms ^= m[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)ms;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = w1 - w0;
    }
}

static int cmp(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : x > y;
}

template <int NV, int SHAPE, int FMT = 4, int SCALED = 1>
static void run(int blocks_per_cu, int n_cus, const uint4 *in, float *out, unsigned long long *d_st)
{
    const int iters = 4000, blocks = n_cus * blocks_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 40; w++) // settle the clock under this load
        rate_kernel<NV, SHAPE, FMT, SCALED><<<blocks, 256>>>(in, out, iters, d_st);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int w = 0; w < reps; w++)
        rate_kernel<NV, SHAPE, FMT, SCALED><<<blocks, 256>>>(in, out, iters, d_st);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    unsigned long long *st = (unsigned long long *)malloc((size_t)blocks * 16);
    CK(hipMemcpy(st, d_st, (size_t)blocks * 16, hipMemcpyDeviceToHost));
    double *ghz = (double *)malloc(sizeof(double) * blocks);
    double cyc = 0;
    for (int i = 0; i < blocks; i++) {
        ghz[i] = (double)st[2 * i] / (double)st[2 * i + 1] * 0.1;
        cyc += (double)st[2 * i] / iters / blocks;
    }
    qsort(ghz, blocks, sizeof(double), cmp);
    const double flop = (double)blocks * 4 * iters * 2.0 * 128 * 32 * 256; // per launch
    printf("{\"shape\": %d, \"format\": \"%s\", \"scaled\": %d, \"valu_per_iter\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"PFLOPs\": %.3f, "
           "\"clock_GHz_median\": %.3f, \"cycles_per_iter\": %.1f}\n",
           SHAPE, FMT == 4 ? "fp4" : "fp6_e3m2", SCALED, NV, blocks_per_cu, ms, flop / (ms * 1e-3) / 1e15, ghz[blocks / 2], cyc);
    fflush(stdout);
    free(st);
    free(ghz);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    uint4 *in;
    float *out;
    unsigned long long *d_st;
    CK(hipMalloc(&in, 1024 * 16));
    {
        unsigned char h[1024 * 16];
        srand(7);
        for (int i = 0; i < 1024 * 16; i++)
            h[i] = (unsigned char)(0x66 | ((rand() & 1) << 3) | ((rand() & 1) << 7));
        CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    }
    // FP6 E3M2 operands of +-28 at random: 6-bit elements 0x1F | sign << 5, packed
    uint4 *in6;
    CK(hipMalloc(&in6, 1024 * 16));
    {
        unsigned char h[1024 * 16] = {0};
        for (int e = 0; e < 1024 * 16 * 8 / 6; e++) {
            const unsigned v = 0x1F | ((rand() & 1) << 5);
            for (int bit = 0; bit < 6; bit++)
                if (v >> bit & 1)
                    h[(e * 6 + bit) >> 3] |= (unsigned char)(1u << ((e * 6 + bit) & 7));
        }
        CK(hipMemcpy(in6, h, sizeof(h), hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&out, (size_t)n_cus * 3 * 256 * 4));
    CK(hipMalloc(&d_st, (size_t)n_cus * 3 * 16));
    for (int w = 1; w <= 3; w++) {
        run<0, 32>(w, n_cus, in, out, d_st);
        run<0, 16>(w, n_cus, in, out, d_st);
        run<64, 32>(w, n_cus, in, out, d_st);
        run<64, 16>(w, n_cus, in, out, d_st);
        run<96, 32>(w, n_cus, in, out, d_st);
        run<96, 16>(w, n_cus, in, out, d_st);
        // without the scale operands
        run<0, 32, 4, 0>(w, n_cus, in, out, d_st);
        run<0, 16, 4, 0>(w, n_cus, in, out, d_st);
        run<96, 32, 4, 0>(w, n_cus, in, out, d_st);
        run<96, 16, 4, 0>(w, n_cus, in, out, d_st);
        // FP6 E3M2 operands (+-28), with and without scale
        run<0, 32, 3, 0>(w, n_cus, in6, out, d_st);
        run<0, 16, 3, 0>(w, n_cus, in6, out, d_st);
        run<96, 32, 3, 0>(w, n_cus, in6, out, d_st);
        run<96, 16, 3, 0>(w, n_cus, in6, out, d_st);
        run<96, 32, 3, 1>(w, n_cus, in6, out, d_st);
        run<96, 16, 3, 1>(w, n_cus, in6, out, d_st);
    }
    return 0;
}
