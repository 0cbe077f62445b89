// Matrix-core k-NN probe: exactness against a CPU brute force on tie-heavy data, then
// timing.  Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -x hip tools/mfma_probe.cpp \
//         -mllvm -amdgpu-mfma-vgpr-form poseestimation_amd/csrc/knn_mfma.hip \
//         poseestimation_amd/csrc/knn_hamming.hip -o tools/mfma_probe
// Run: tools/mfma_probe [nq nt frames splits q_tiles [reps]]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../poseestimation_amd/csrc/knn_hamming.h"
#include "../poseestimation_amd/csrc/knn_mfma.h"

using namespace posehip;

#ifdef POSEHIP_MF_CLOCKS
namespace posehip {
extern __device__ unsigned long long mf_clock_probe[2];
}
#endif

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));     \
            exit(2);                                                                              \
        }                                                                                         \
    } while (0)

static int ham(const uint32_t *a, const uint32_t *b)
{
    int d = 0;
    for (int i = 0; i < 8; i++)
        d += __builtin_popcount(a[i] ^ b[i]);
    return d;
}

// rows drawn from a small pool with a few flipped bits: many equal distances
static std::vector<uint32_t> make_rows(int64_t n, int pool, int flips, uint32_t seed)
{
    std::mt19937 g(seed);
    std::vector<uint32_t> base((size_t)pool * 8), out((size_t)n * 8);
    for (auto &w : base)
        w = g();
    for (int64_t i = 0; i < n; i++) {
        const int p = (int)(g() % (uint32_t)pool);
        for (int j = 0; j < 8; j++)
            out[i * 8 + j] = base[(size_t)p * 8 + j];
        for (int f = 0; f < flips; f++) {
            const uint32_t bit = g() & 255;
            out[i * 8 + (bit >> 5)] ^= 1u << (bit & 31);
        }
    }
    return out;
}

static int check(int64_t nq, int64_t nt, int64_t splits, int q_tiles, int k, int pool, int flips)
{
    auto q = make_rows(nq, pool, flips, 1 + (uint32_t)nq), t = make_rows(nt, pool, flips, 7 + (uint32_t)nt);
    if (pool == 1 && nt > 3) { // complement rows: distance 256
        for (int j = 0; j < 8; j++)
            t[8 * 3 + j] = ~q[j];
    }
    void *dq, *dt, *dqx, *dtx;
    uint32_t *dpart;
    int32_t *didx, *ddist;
    CK(hipMalloc(&dq, nq * 32));
    CK(hipMalloc(&dt, nt * 32));
    CK(hipMalloc(&dqx, knn_mfma_expanded_bytes(nq)));
    CK(hipMalloc(&dtx, knn_mfma_expanded_bytes(nt)));
    CK(hipMalloc(&didx, nq * k * 4));
    CK(hipMalloc(&ddist, nq * k * 4));
    CK(hipMemset(dqx, 0xFF, knn_mfma_expanded_bytes(nq)));
    CK(hipMemset(dtx, 0xFF, knn_mfma_expanded_bytes(nt)));
    CK(hipMemcpy(dq, q.data(), nq * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, t.data(), nt * 32, hipMemcpyHostToDevice));
    MfmaPlan p = knn_mfma_plan_splits(nq, nt, q_tiles, splits);
    CK(hipMalloc(&dpart, p.part_bytes));
    CK(knn_mfma_launch(p, dq, nq, dt, nt, k, 0, dtx, dpart, didx, ddist, nullptr, 0.f, 0));
    CK(hipDeviceSynchronize());
    std::vector<int32_t> idx(nq * k), dist(nq * k);
    CK(hipMemcpy(idx.data(), didx, nq * k * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(dist.data(), ddist, nq * k * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int64_t i = 0; i < nq; i++) {
        uint64_t e[2] = {~0ull, ~0ull}; // expected: two smallest (dist, idx)
        for (int64_t j = 0; j < nt; j++) {
            const uint64_t key = ((uint64_t)ham(&q[i * 8], &t[j * 8]) << 32) | (uint64_t)j;
            if (key < e[0]) {
                e[1] = e[0];
                e[0] = key;
            } else if (key < e[1])
                e[1] = key;
        }
        for (int r = 0; r < k; r++) {
            const int32_t wi = e[r] == ~0ull ? -1 : (int32_t)(e[r] & 0xFFFFFFFF);
            const int32_t wd = e[r] == ~0ull ? 0x7FFFFFFF : (int32_t)(e[r] >> 32);
            if (idx[i * k + r] != wi || dist[i * k + r] != wd) {
                if (bad < 5)
                    printf("  MISMATCH q=%lld rank %d: got (%d,%d) want (%d,%d)\n", (long long)i, r,
                           dist[i * k + r], idx[i * k + r], wd, wi);
                bad++;
            }
        }
    }
    printf("check nq=%lld nt=%lld splits=%d rows/split=%lld q_tiles=%d k=%d pool=%d flips=%d: %s (%d bad)\n",
           (long long)nq, (long long)nt, p.n_splits, (long long)p.rows_per_split, q_tiles, k, pool,
           flips, bad ? "FAIL" : "ok", bad);
    fflush(stdout);
    CK(hipFree(dq));
    CK(hipFree(dt));
    CK(hipFree(dqx));
    CK(hipFree(dtx));
    CK(hipFree(dpart));
    CK(hipFree(didx));
    CK(hipFree(ddist));
    return bad;
}

// F frames of nq x nt in one launch (F == 1: the single-problem kernel)
static void timeit(int64_t nq, int64_t nt, int F, int64_t splits, int q_tiles, int reps)
{
    auto q = make_rows(nq * F, 1 << 20, 0, 3), t = make_rows(nt * F, 1 << 20, 0, 5);
    void *dq, *dt, *dqx, *dtx;
    uint32_t *dpart;
    int32_t *didx, *ddist, *doff;
    CK(hipMalloc(&dq, nq * F * 32));
    CK(hipMalloc(&dt, nt * F * 32));
    CK(hipMalloc(&dqx, knn_mfma_expanded_bytes(nq * F)));
    CK(hipMalloc(&dtx, knn_mfma_expanded_bytes(nt * F)));
    CK(hipMalloc(&didx, nq * F * 2 * 4));
    CK(hipMalloc(&ddist, nq * F * 2 * 4));
    CK(hipMemcpy(dq, q.data(), nq * F * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, t.data(), nt * F * 32, hipMemcpyHostToDevice));
    std::vector<int32_t> off(2 * (F + 1));
    for (int f = 0; f <= F; f++) {
        off[f] = (int32_t)(f * nq);
        off[F + 1 + f] = (int32_t)(f * nt);
    }
    CK(hipMalloc(&doff, off.size() * 4));
    CK(hipMemcpy(doff, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    MfmaPlan p = knn_mfma_plan_splits(nq, nt, q_tiles, splits);
    p.part_bytes *= F;
    CK(hipMalloc(&dpart, p.part_bytes));
    hipEvent_t ev[5];
    for (auto &e : ev)
        CK(hipEventCreate(&e));
    float best[3] = {1e30f, 1e30f, 1e30f}, best_total = 1e30f;
    for (int r = 0; r < reps + 2; r++) {
        if (F == 1)
            CK(knn_mfma_launch(p, dq, nq, dt, nt, 2, 0, dtx, dpart, didx, ddist, nullptr, 0.f, 0, ev));
        else
            CK(knn_mfma_frames_launch(p, dq, doff, nq * F, dt, doff + F + 1, 0, nt * F, F, 2,
                                      dtx, dpart, didx, ddist, nullptr, 0.f, 0, ev, true));
        CK(hipEventSynchronize(ev[3]));
        float m[3], tot;
        CK(hipEventElapsedTime(&m[0], ev[4], ev[0]));
        CK(hipEventElapsedTime(&m[1], ev[0], ev[1]));
        CK(hipEventElapsedTime(&m[2], ev[1], ev[2]));
        CK(hipEventElapsedTime(&tot, ev[4], ev[2]));
        if (r >= 2) {
            for (int i = 0; i < 3; i++)
                best[i] = std::min(best[i], m[i]);
            best_total = std::min(best_total, tot);
        }
    }
    const double evals = (double)nq * (double)nt * F;
    printf("time F=%d nq=%lld nt=%lld splits=%d rows/split=%lld q_tiles=%d grid=%lldx%dx%d: prepass %.1f us, "
           "kernel %.1f us (%.3e evals/s, %.2f PFLOP/s fp4), merge %.1f us, all %.1f us (%.3e evals/s)\n",
           F, (long long)nq, (long long)nt, p.n_splits, (long long)p.rows_per_split, q_tiles,
           (long long)p.q_blocks, p.n_splits, F, best[0] * 1e3, best[1] * 1e3,
           evals / (best[1] * 1e-3), evals * 512 / (best[1] * 1e-3) / 1e15, best[2] * 1e3,
           best_total * 1e3, evals / (best_total * 1e-3));
#ifdef POSEHIP_MF_CLOCKS
    if (F > 3) {
        unsigned long long c[2];
        CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(posehip::mf_clock_probe), sizeof(c)));
        const int tiles = (int)((p.rows_per_split + 31) / 32);
        printf("  one workgroup: %llu shader cycles in %.2f us (%.0f MHz), %d tiles -> %.0f cycles per tile\n",
               c[0], c[1] / 100.0, c[0] / (c[1] / 100.0), tiles, (double)c[0] / tiles);
    }
#endif
    fflush(stdout);
    CK(hipFree(dq));
    CK(hipFree(dt));
    CK(hipFree(dqx));
    CK(hipFree(dtx));
    CK(hipFree(dpart));
    CK(hipFree(didx));
    CK(hipFree(ddist));
    CK(hipFree(doff));
}

int main(int argc, char **argv)
{
    if (argc >= 6) { // nq nt F splits q_tiles [reps]
        timeit(atoll(argv[1]), atoll(argv[2]), atoi(argv[3]), atoll(argv[4]), atoi(argv[5]),
               argc > 6 ? atoi(argv[6]) : 5);
        return 0;
    }
    int bad = 0;
    for (int qt : {4, 2})
        for (int k : {2, 1}) {
            bad += check(700, 3001, 3, qt, k, 16, 3);
            bad += check(33, 40, 1, qt, k, 4, 1);
            bad += check(1, 1, 1, qt, k, 1, 0);
            bad += check(513, 31, 1, qt, k, 1, 0);
            bad += check(300, 2000, 1, qt, k, 1, 0); // all distances equal except row 3
            bad += check(1500, 5000, 7, qt, k, 1 << 20, 0);
            bad += check(200, 64, 1, qt, k, 2, 1);
            bad += check(200, 65, 1, qt, k, 2, 1);
            bad += check(200, 97, 2, qt, k, 2, 1);
        }
    bad += check(64, 40000, 1, 4, 2, 8, 2); // two splits, the longer one close to 1024 tiles
    printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
    if (bad)
        return 1;
    for (int qt : {4, 2})
        for (int splits : {2, 3, 4, 5, 6, 8})
            timeit(10000, 10000, 16, splits, qt, 5);
    return 0;
}
