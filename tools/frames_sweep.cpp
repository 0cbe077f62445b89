// Sweeps launch plans (waves per workgroup, splits, lazy) of the batched-frames k-NN
// launch and prints one JSON line per plan.  Tuning aid; not part of the product library.
// build: hipcc -O2 -std=c++17 tools/frames_sweep.cpp -Iposeestimation_amd/csrc -Lposeestimation_amd -lposehip -Wl,-rpath,'$ORIGIN/../poseestimation_amd' -o tools/frames_sweep
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
    int F = argc > 3 ? atoi(argv[3]) : 16, reps = 10;
    std::mt19937_64 rng(7);
    std::vector<uint64_t> hq(nq * F * 4), ht(nt * F * 4);
    for (auto &v : hq) v = rng();
    for (auto &v : ht) v = rng();
    std::vector<int32_t> qo(F + 1), to(F + 1);
    for (int f = 0; f <= F; f++) { qo[f] = (int32_t)(f * nq); to[f] = (int32_t)(f * nt); }
    void *dq, *dt; int32_t *di, *dd, *di0, *dd0, *dqo, *dto; uint8_t *dk; uint32_t *part;
    CK(hipMalloc(&dq, hq.size() * 8)); CK(hipMalloc(&dt, ht.size() * 8));
    CK(hipMalloc(&di, nq * F * 8)); CK(hipMalloc(&dd, nq * F * 8)); CK(hipMalloc(&dk, nq * F));
    CK(hipMalloc(&di0, nq * F * 8)); CK(hipMalloc(&dd0, nq * F * 8));
    CK(hipMalloc(&dqo, (F + 1) * 4)); CK(hipMalloc(&dto, (F + 1) * 4));
    const size_t part_cap = (size_t)1 << 30;
    CK(hipMalloc(&part, part_cap));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dqo, qo.data(), (F + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dto, to.data(), (F + 1) * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long total = nq * F;
    auto make = [&](int waves, bool lazy, long splits) {
        KnnPlan p = knn_plan_splits(nq, nt, waves, lazy, splits);
        p.part_bytes = (size_t)p.n_splits * 2 * (size_t)total * sizeof(uint32_t);
        return p;
    };
    KnnPlan ref = knn_plan_frames(F, nq, nt, total, 256);
    CK(knn_frames_launch(ref, dq, dqo, dt, dto, F, total, 2, part, di0, dd0, dk, 0.75f, st, nullptr, true));
    CK(hipStreamSynchronize(st));
    std::vector<int32_t> r_i(total * 2), r_d(total * 2), g_i(total * 2), g_d(total * 2);
    CK(hipMemcpy(r_i.data(), di0, total * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r_d.data(), dd0, total * 8, hipMemcpyDeviceToHost));
    printf("{\"library_plan\": {\"waves\": %d, \"splits\": %d, \"lazy\": %d}}\n", ref.waves, ref.n_splits, (int)ref.lazy);
    for (int waves : {1, 2, 4, 8})
        for (int lazy : {0, 1})
            for (long splits : {1, 2, 3, 4, 5, 6, 8, 10, 13, 16, 20}) {
                for (int xcd : {1, 0}) {
                    KnnPlan p = make(waves, lazy, splits);
                    if (p.n_splits != splits || p.part_bytes > part_cap) continue;
                    for (int w = 0; w < 2; w++) CK(knn_frames_launch(p, dq, dqo, dt, dto, F, total, 2, part, di, dd, dk, 0.75f, st, nullptr, xcd != 0));
                    CK(hipStreamSynchronize(st));
                    CK(hipMemcpy(g_i.data(), di, total * 8, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(g_d.data(), dd, total * 8, hipMemcpyDeviceToHost));
                    const bool ok = g_i == r_i && g_d == r_d;
                    std::vector<float> ts;
                    for (int r = 0; r < reps; r++) {
                        CK(hipEventRecord(e0, st));
                        CK(knn_frames_launch(p, dq, dqo, dt, dto, F, total, 2, part, di, dd, dk, 0.75f, st, nullptr, xcd != 0));
                        CK(hipEventRecord(e1, st));
                        CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
                    }
                    std::sort(ts.begin(), ts.end());
                    printf("{\"waves\": %d, \"lazy\": %d, \"splits\": %ld, \"xcd\": %d, \"rows_per_wave\": %d, \"ok\": %d, \"ms_med\": %.4f, \"Tevals\": %.3f}\n",
                           waves, lazy, splits, xcd, p.rows_per_wave, (int)ok, ts[ts.size() / 2], (double)total * nt / (ts[ts.size() / 2] * 1e-3) / 1e12);
                    fflush(stdout);
                }
            }
    return 0;
}
