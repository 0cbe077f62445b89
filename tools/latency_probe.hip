// Host-side latency building blocks on this box: launch + synchronise, pinned copy, flag polling.
// build: hipcc -O2 --offload-arch=gfx950 tools/latency_probe.hip -Iinclude -Lposeestimation_amd -lposehip -Wl,-rpath,'$ORIGIN/../poseestimation_amd' -o tools/latency_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "posehip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
__global__ void empty_kernel() {}
__global__ void flag_kernel(volatile unsigned *flag, unsigned v) { *flag = v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> double med(F f, int reps = 200)
{
    std::vector<double> t;
    for (int i = 0; i < reps; i++) { double a = now(); f(); t.push_back(now() - a); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2] * 1e6;
}
int main()
{
    if (ph_init(0) != PH_OK) return 1;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    void *pin, *dev; CK(hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault)); CK(hipMalloc(&dev, 1 << 20));
    unsigned *flag_h, *flag_d; CK(hipHostMalloc((void **)&flag_h, 64, hipHostMallocDefault)); CK(hipHostGetDevicePointer((void **)&flag_d, flag_h, 0));
    *flag_h = 0;
    for (int i = 0; i < 10; i++) { empty_kernel<<<1, 64, 0, st>>>(); CK(hipStreamSynchronize(st)); }
    printf("{\"empty_kernel_plus_sync_us\": %.1f,\n", med([&] { empty_kernel<<<1, 64, 0, st>>>(); (void)hipStreamSynchronize(st); }));
    printf(" \"five_empty_kernels_plus_sync_us\": %.1f,\n", med([&] { for (int k = 0; k < 5; k++) empty_kernel<<<1, 64, 0, st>>>(); (void)hipStreamSynchronize(st); }));
    printf(" \"copy_32KB_plus_sync_us\": %.1f,\n", med([&] { (void)hipMemcpyAsync(dev, pin, 32768, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }));
    printf(" \"copy_32KB_kernel_sync_us\": %.1f,\n", med([&] { (void)hipMemcpyAsync(dev, pin, 32768, hipMemcpyHostToDevice, st); empty_kernel<<<1, 64, 0, st>>>(); (void)hipStreamSynchronize(st); }));
    unsigned seq = 0;
    printf(" \"kernel_plus_flag_poll_us\": %.1f,\n", med([&] { seq++; empty_kernel<<<1, 64, 0, st>>>(); flag_kernel<<<1, 1, 0, st>>>(flag_d, seq); while (*(volatile unsigned *)flag_h != seq) {} }));
    printf(" \"five_kernels_plus_flag_poll_us\": %.1f,\n", med([&] { seq++; for (int k = 0; k < 5; k++) empty_kernel<<<1, 64, 0, st>>>(); flag_kernel<<<1, 1, 0, st>>>(flag_d, seq); while (*(volatile unsigned *)flag_h != seq) {} }));
    CK(hipStreamSynchronize(st));
    std::mt19937_64 rng(1);
    std::vector<uint64_t> q(500 * 4), t(500 * 4);
    for (auto &v : q) v = rng();
    for (auto &v : t) v = rng();
    std::vector<ph_dmatch> out(500); int n = 0;
    for (int i = 0; i < 10; i++) ph_match_crosscheck_hamming256((const uint8_t *)q.data(), 500, 32, (const uint8_t *)t.data(), 500, 32, out.data(), &n);
    printf(" \"crosscheck_500x500_host_call_us\": %.1f,\n", med([&] { ph_match_crosscheck_hamming256((const uint8_t *)q.data(), 500, 32, (const uint8_t *)t.data(), 500, 32, out.data(), &n); }));
    printf(" \"match_500x500_host_call_us\": %.1f}\n", med([&] { ph_match_hamming256((const uint8_t *)q.data(), 500, 32, (const uint8_t *)t.data(), 500, 32, out.data(), &n); }));
    {   // the benchmark's frame through the host entry point: PCIe both ways included
        const int n = 10000;
        std::vector<uint64_t> bq((size_t)n * 4), bt((size_t)n * 4);
        for (auto &v : bq) v = rng();
        for (auto &v : bt) v = rng();
        std::vector<ph_dmatch> bo(n); int bn = 0;
        for (int i = 0; i < 5; i++) ph_knn2_ratio_hamming256((const uint8_t *)bq.data(), n, 32, (const uint8_t *)bt.data(), n, 32, 0.75f, bo.data(), &bn);
        const double us = med([&] { ph_knn2_ratio_hamming256((const uint8_t *)bq.data(), n, 32, (const uint8_t *)bt.data(), n, 32, 0.75f, bo.data(), &bn); }, 100);
        printf("{\"knn2_ratio_10k_x_10k_host_call_us\": %.1f, \"evals_per_s_pcie_inclusive\": %.4g}\n", us, 1e8 / (us * 1e-6));
    }
    ph_shutdown();
    return 0;
}
