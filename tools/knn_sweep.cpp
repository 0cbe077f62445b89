// Sweeps launch plans of the Hamming k-NN kernel (internal interface) and prints
// one JSON line per plan.  Tuning aid; not part of the product library.
// build: hipcc -O2 -std=c++17 tools/knn_sweep.cpp poseestimation_amd/csrc/knn_hamming.o -Iposeestimation_amd/csrc -o tools/knn_sweep
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "knn_hamming.h"

using namespace posehip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char **argv)
{
    long nq = argc > 1 ? atol(argv[1]) : 10000, nt = argc > 2 ? atol(argv[2]) : 10000;
    int reps = argc > 3 ? atoi(argv[3]) : 30;
    std::mt19937_64 rng(7);
    std::vector<uint64_t> hq(nq * 4), ht(nt * 4);
    for (auto &v : hq) v = rng();
    for (auto &v : ht) v = rng();
    void *dq, *dt; int32_t *di, *dd, *di0, *dd0; uint8_t *dk; uint32_t *part;
    CK(hipMalloc(&dq, nq * 32)); CK(hipMalloc(&dt, nt * 32));
    CK(hipMalloc(&di, nq * 8)); CK(hipMalloc(&dd, nq * 8)); CK(hipMalloc(&dk, nq));
    CK(hipMalloc(&di0, nq * 8)); CK(hipMalloc(&dd0, nq * 8));
    size_t part_cap = (size_t)1 << 30;
    CK(hipMalloc(&part, part_cap));
    CK(hipMemcpy(dq, hq.data(), nq * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, ht.data(), nt * 32, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // reference result with the simplest plan
    KnnPlan ref = knn_plan_custom(nq, nt, 1, false, 1);
    CK(knn_launch(ref, dq, nq, dt, nt, 2, 0, part, di0, dd0, dk, 0.75f, st));
    CK(hipStreamSynchronize(st));
    std::vector<int32_t> r_i(nq * 2), r_d(nq * 2), g_i(nq * 2), g_d(nq * 2);
    CK(hipMemcpy(r_i.data(), di0, nq * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r_d.data(), dd0, nq * 8, hipMemcpyDeviceToHost));
    struct V { int waves; bool lazy; int per_simd; };
    std::vector<V> vs;
    for (int waves : {2, 4, 8})
        for (int lazy : {0, 1})
            for (int waves_per_simd : {4, 8, 12, 16, 24, 32, 48})
                vs.push_back({waves, lazy != 0, waves_per_simd});
    for (auto v : vs) {
        const long max_blocks = (long)256 * 4 * v.per_simd / v.waves; // resident workgroups
        if (max_blocks < 1) continue;
        KnnPlan p = knn_plan_custom(nq, nt, v.waves, v.lazy, max_blocks);
        if (p.part_bytes > part_cap) continue;
        for (int w = 0; w < 2; w++) CK(knn_launch(p, dq, nq, dt, nt, 2, 0, part, di, dd, dk, 0.75f, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(g_i.data(), di, nq * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(g_d.data(), dd, nq * 8, hipMemcpyDeviceToHost));
        const bool ok = g_i == r_i && g_d == r_d;
        std::vector<float> ts;
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0, st));
            CK(knn_launch(p, dq, nq, dt, nt, 2, 0, part, di, dd, dk, 0.75f, st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("{\"nq\": %ld, \"nt\": %ld, \"waves\": %d, \"lazy\": %d, \"waves_per_simd\": %d, \"q_blocks\": %ld, \"splits\": %d, \"rows\": %ld, \"rows_per_wave\": %d, \"ok\": %d, \"ms_med\": %.4f, \"ms_min\": %.4f, \"Tevals\": %.3f}\n",
               nq, nt, v.waves, (int)v.lazy, v.per_simd, (long)p.q_blocks, p.n_splits, (long)p.rows_per_split, p.rows_per_wave, (int)ok,
               ts[ts.size() / 2], ts[0], (double)nq * nt / (ts[ts.size() / 2] * 1e-3) / 1e12);
        fflush(stdout);
    }
    return 0;
}
