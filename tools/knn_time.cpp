// Times ph_knn_hamming256_dev (device-resident inputs) with HIP events.
// build: hipcc -O2 tools/knn_time.cpp -Iinclude -Lposeestimation_amd -lposehip -Wl,-rpath,'$ORIGIN/../poseestimation_amd' -o tools/knn_time
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "posehip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
#define PH(x) do { ph_status st_ = (x); if (st_ != PH_OK) { printf("posehip error %d: %s\n", st_, ph_last_error_string()); exit(1); } } while (0)

int main(int argc, char **argv)
{
    PH(ph_init(0));
    struct Sz { long nq, nt; int reps; };
    std::vector<Sz> sizes = {{500, 500, 50}, {2000, 50000, 50}, {4096, 50000, 30}, {10000, 10000, 50}, {30000, 10000, 30}, {50000, 20000, 20}, {100000, 10000, 20}, {100000, 50000, 10}, {262144, 16384, 10}, {1000000, 50000, 3}};
    if (argc == 3) sizes = {{atol(argv[1]), atol(argv[2]), 10}};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (argc == 4) { // nq nt F: F independent nq x nt frames in ONE batched launch
        const long nq = atol(argv[1]), nt = atol(argv[2]); const int F = atoi(argv[3]);
        std::mt19937_64 rng(7);
        std::vector<uint64_t> hq(nq * F * 4), ht(nt * F * 4);
        for (auto &v : hq) v = rng();
        for (auto &v : ht) v = rng();
        std::vector<int32_t> qo(F + 1), to(F + 1);
        for (int f = 0; f <= F; f++) { qo[f] = (int32_t)(f * nq); to[f] = (int32_t)(f * nt); }
        void *dq, *dt; int32_t *di, *dd; uint8_t *dk;
        CK(hipMalloc(&dq, hq.size() * 8)); CK(hipMalloc(&dt, ht.size() * 8));
        CK(hipMalloc(&di, nq * F * 8)); CK(hipMalloc(&dd, nq * F * 8)); CK(hipMalloc(&dk, nq * F));
        CK(hipMemcpy(dq, hq.data(), hq.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dt, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
        for (int w = 0; w < 3; w++) PH(ph_knn_hamming256_frames_dev(dq, qo.data(), dt, to.data(), F, 2, di, dd, dk, 0.75f, st));
        CK(hipStreamSynchronize(st));
        std::vector<float> ts;
        for (int r = 0; r < 10; r++) {
            CK(hipEventRecord(e0, st));
            PH(ph_knn_hamming256_frames_dev(dq, qo.data(), dt, to.data(), F, 2, di, dd, dk, 0.75f, st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        printf("{\"frames\": %d, \"nq\": %ld, \"nt\": %ld, \"ms_median\": %.4f, \"ms_min\": %.4f, \"evals_per_s_median\": %.4g}\n",
               F, nq, nt, ts[ts.size() / 2], ts[0], (double)F * nq * nt / (ts[ts.size() / 2] * 1e-3));
        ph_shutdown();
        return 0;
    }
    for (auto s : sizes) {
        std::mt19937_64 rng(7);
        std::vector<uint64_t> hq(s.nq * 4), ht(s.nt * 4);
        for (auto &v : hq) v = rng();
        for (auto &v : ht) v = rng();
        void *dq, *dt; int32_t *di, *dd; uint8_t *dk;
        CK(hipMalloc(&dq, s.nq * 32)); CK(hipMalloc(&dt, s.nt * 32));
        CK(hipMalloc(&di, s.nq * 8)); CK(hipMalloc(&dd, s.nq * 8)); CK(hipMalloc(&dk, s.nq));
        CK(hipMemcpy(dq, hq.data(), s.nq * 32, hipMemcpyHostToDevice));
        CK(hipMemcpy(dt, ht.data(), s.nt * 32, hipMemcpyHostToDevice));
        for (int w = 0; w < 3; w++) PH(ph_knn_hamming256_dev(dq, s.nq, dt, s.nt, 2, 0, di, dd, dk, 0.75f, st));
        CK(hipStreamSynchronize(st));
        std::vector<float> ts;
        for (int r = 0; r < s.reps; r++) {
            CK(hipEventRecord(e0, st));
            PH(ph_knn_hamming256_dev(dq, s.nq, dt, s.nt, 2, 0, di, dd, dk, 0.75f, st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        double med = ts[ts.size() / 2], mn = ts[0];
        printf("{\"nq\": %ld, \"nt\": %ld, \"ms_median\": %.4f, \"ms_min\": %.4f, \"evals_per_s_median\": %.4g, \"evals_per_s_best\": %.4g}\n",
               s.nq, s.nt, med, mn, (double)s.nq * s.nt / (med * 1e-3), (double)s.nq * s.nt / (mn * 1e-3));
        CK(hipFree(dq)); CK(hipFree(dt)); CK(hipFree(di)); CK(hipFree(dd)); CK(hipFree(dk));
    }
    ph_shutdown();
    return 0;
}
