// A/B of the two ways to feed wave-uniform train rows to the Hamming inner loop
// (one query row per lane in both):
//   sgpr : scalar loads (s_load_dwordx16) into SGPRs, used as the scalar operand
//          of v_xor_b32 -- what the library does;
//   lds  : the workgroup stages a train tile in LDS with coalesced 32-byte
//          per-lane loads, every lane then reads each row with two broadcast
//          ds_read_b128.
// Same key update (eager, k = 2), no train splits (nq large enough to fill the
// chip), results compared word for word.
// build: hipcc -O3 --offload-arch=gfx950 tools/knn_variants.hip -o tools/knn_variants
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) { uint32_t r; asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc)); return r; }
__device__ __forceinline__ uint32_t bcnt0(uint32_t x) { uint32_t r; asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t med3(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t ham(const uint32_t (&q)[8], uint4 a, uint4 b)
{
    uint32_t d = bcnt0(q[0] ^ a.x);
    d = bcnt_acc(q[1] ^ a.y, d); d = bcnt_acc(q[2] ^ a.z, d); d = bcnt_acc(q[3] ^ a.w, d);
    d = bcnt_acc(q[4] ^ b.x, d); d = bcnt_acc(q[5] ^ b.y, d); d = bcnt_acc(q[6] ^ b.z, d); d = bcnt_acc(q[7] ^ b.w, d);
    return d;
}
__device__ __forceinline__ void upd(uint32_t &m1, uint32_t &m2, uint32_t d, uint32_t j)
{
    const uint32_t key = (d << 22) + j;
    m2 = med3(m1, m2, key);
    m1 = min(m1, key);
}

__global__ __launch_bounds__(256) void k_sgpr(const uint4 *__restrict__ q, const uint4 *__restrict__ t, int nq, int nt, uint2 *__restrict__ out)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    uint32_t qq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qi < nq) { uint4 a = q[2 * qi], b = q[2 * qi + 1]; qq[0] = a.x; qq[1] = a.y; qq[2] = a.z; qq[3] = a.w; qq[4] = b.x; qq[5] = b.y; qq[6] = b.z; qq[7] = b.w; }
    uint32_t m1 = ~0u, m2 = ~0u;
    for (int j = 0; j + 4 <= nt; j += 4) {
        uint4 r[8];
#pragma unroll
        for (int u = 0; u < 8; u++) r[u] = t[2 * j + u]; // uniform address -> scalar loads
#pragma unroll
        for (int u = 0; u < 4; u++) upd(m1, m2, ham(qq, r[2 * u], r[2 * u + 1]), j + u);
    }
    if (qi < nq) out[qi] = make_uint2(m1, m2);
}

template <int TILE>
__global__ __launch_bounds__(256) void k_lds(const uint4 *__restrict__ q, const uint4 *__restrict__ t, int nq, int nt, uint2 *__restrict__ out)
{
    __shared__ uint4 tile[2][TILE * 2];
    const int qi = blockIdx.x * 256 + threadIdx.x;
    uint32_t qq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qi < nq) { uint4 a = q[2 * qi], b = q[2 * qi + 1]; qq[0] = a.x; qq[1] = a.y; qq[2] = a.z; qq[3] = a.w; qq[4] = b.x; qq[5] = b.y; qq[6] = b.z; qq[7] = b.w; }
    uint32_t m1 = ~0u, m2 = ~0u;
    const int n_tiles = nt / TILE;
    // stage tile 0: TILE rows = 2*TILE uint4, each lane copies one 32-byte row (two 16-byte loads)
    for (int e = threadIdx.x; e < TILE; e += 256) { tile[0][2 * e] = t[2 * e]; tile[0][2 * e + 1] = t[2 * e + 1]; }
    __syncthreads();
    for (int tt = 0; tt < n_tiles; tt++) {
        const int cur = tt & 1;
        if (tt + 1 < n_tiles)
            for (int e = threadIdx.x; e < TILE; e += 256) {
                const size_t g = (size_t)(tt + 1) * TILE + e;
                tile[cur ^ 1][2 * e] = t[2 * g];
                tile[cur ^ 1][2 * e + 1] = t[2 * g + 1];
            }
#pragma unroll 4
        for (int j = 0; j < TILE; j++) {
            const uint4 a = tile[cur][2 * j], b = tile[cur][2 * j + 1]; // same address in every lane: broadcast
            upd(m1, m2, ham(qq, a, b), tt * TILE + j);
        }
        __syncthreads();
    }
    if (qi < nq) out[qi] = make_uint2(m1, m2);
}

template <typename F> double time_it(F f, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char **argv)
{
    const int nq = argc > 1 ? atoi(argv[1]) : 262144, nt = argc > 2 ? atoi(argv[2]) : 16384; // nt multiple of 256
    std::mt19937_64 rng(7);
    std::vector<uint64_t> hq((size_t)nq * 4), ht((size_t)nt * 4);
    for (auto &v : hq) v = rng();
    for (auto &v : ht) v = rng();
    uint4 *dq, *dt; uint2 *o1, *o2;
    CK(hipMalloc(&dq, (size_t)nq * 32)); CK(hipMalloc(&dt, (size_t)nt * 32)); CK(hipMalloc(&o1, (size_t)nq * 8)); CK(hipMalloc(&o2, (size_t)nq * 8));
    CK(hipMemcpy(dq, hq.data(), (size_t)nq * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, ht.data(), (size_t)nt * 32, hipMemcpyHostToDevice));
    const int blocks = (nq + 255) / 256;
    const double ms_s = time_it([&] { k_sgpr<<<blocks, 256>>>(dq, dt, nq, nt, o1); }, 7);
    const double ms_l = time_it([&] { k_lds<256><<<blocks, 256>>>(dq, dt, nq, nt, o2); }, 7);
    const double ms_l2 = time_it([&] { k_lds<1024><<<blocks, 256>>>(dq, dt, nq, nt, o2); }, 7);
    std::vector<uint2> a(nq), b(nq);
    CK(hipMemcpy(a.data(), o1, (size_t)nq * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o2, (size_t)nq * 8, hipMemcpyDeviceToHost));
    size_t bad = 0; for (int i = 0; i < nq; i++) bad += a[i].x != b[i].x || a[i].y != b[i].y;
    const double ev = (double)nq * nt;
    printf("{\"nq\": %d, \"nt\": %d, \"sgpr_ms\": %.4f, \"sgpr_Tevals\": %.3f, \"lds256_ms\": %.4f, \"lds256_Tevals\": %.3f, \"lds1024_ms\": %.4f, \"lds1024_Tevals\": %.3f, \"mismatches\": %zu}\n",
           nq, nt, ms_s, ev / ms_s / 1e9, ms_l, ev / ms_l / 1e9, ms_l2, ev / ms_l2 / 1e9, bad);
    return 0;
}
