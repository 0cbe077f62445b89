// Counts every allocating HIP call made while ph_knn_hamming256_noalloc_dev runs.
//
// The program DEFINES hipMalloc, hipFree and the other allocator entry points of the HIP
// runtime itself (exported with -rdynamic): the dynamic linker resolves a shared library's
// calls to a symbol the executable defines to the executable's definition, so every such
// call that libposehip.so makes lands here, is counted, and is forwarded to the runtime's
// own function (dlsym RTLD_NEXT).  A free-memory reading before and after a call cannot see
// a pooled or sub-page allocation; a call count can.
//
// Sequence: allocate the test's own buffers; ph_init; one ALLOCATING call of
// ph_knn_hamming256_dev as the positive control (the counters must move: the interposition
// works); then, per distance kernel, two calls of the allocation-free entry point, the second
// smaller, with the counters at zero before and required to be zero after, each compared bit
// for bit with ph_knn_hamming256_dev on the same buffers.  Prints one JSON line; exit status 0
// iff everything holds.
//
// build: g++ -std=c++17 -O2 -rdynamic -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude
//            tests/noalloc_probe.cpp -Lposeestimation_amd -lposehip -L/opt/rocm/lib -lamdhip64 -ldl
//            -Wl,-rpath,'$ORIGIN/../poseestimation_amd' -Wl,-rpath,/opt/rocm/lib -o tests/noalloc_probe
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "posehip.h"

static std::atomic<long> g_allocs{0}, g_frees{0};

template <typename Fn> static Fn real(const char *name)
{
    void *p = dlsym(RTLD_NEXT, name);
    if (!p) {
        std::fprintf(stderr, "noalloc_probe: the HIP runtime has no %s\n", name);
        std::abort();
    }
    return reinterpret_cast<Fn>(p);
}

extern "C" {
hipError_t hipMalloc(void **p, size_t n)
{
    static auto f = real<hipError_t (*)(void **, size_t)>("hipMalloc");
    g_allocs++;
    return f(p, n);
}
hipError_t hipFree(void *p)
{
    static auto f = real<hipError_t (*)(void *)>("hipFree");
    g_frees++;
    return f(p);
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned int flags)
{
    static auto f = real<hipError_t (*)(void **, size_t, unsigned int)>("hipHostMalloc");
    g_allocs++;
    return f(p, n, flags);
}
hipError_t hipHostFree(void *p)
{
    static auto f = real<hipError_t (*)(void *)>("hipHostFree");
    g_frees++;
    return f(p);
}
hipError_t hipMallocAsync(void **p, size_t n, hipStream_t s)
{
    static auto f = real<hipError_t (*)(void **, size_t, hipStream_t)>("hipMallocAsync");
    g_allocs++;
    return f(p, n, s);
}
hipError_t hipFreeAsync(void *p, hipStream_t s)
{
    static auto f = real<hipError_t (*)(void *, hipStream_t)>("hipFreeAsync");
    g_frees++;
    return f(p, s);
}
hipError_t hipMallocManaged(void **p, size_t n, unsigned int flags)
{
    static auto f = real<hipError_t (*)(void **, size_t, unsigned int)>("hipMallocManaged");
    g_allocs++;
    return f(p, n, flags);
}
hipError_t hipExtMallocWithFlags(void **p, size_t n, unsigned int flags)
{
    static auto f = real<hipError_t (*)(void **, size_t, unsigned int)>("hipExtMallocWithFlags");
    g_allocs++;
    return f(p, n, flags);
}
hipError_t hipMallocPitch(void **p, size_t *pitch, size_t w, size_t h)
{
    static auto f = real<hipError_t (*)(void **, size_t *, size_t, size_t)>("hipMallocPitch");
    g_allocs++;
    return f(p, pitch, w, h);
}
hipError_t hipHostRegister(void *p, size_t n, unsigned int flags)
{
    static auto f = real<hipError_t (*)(void *, size_t, unsigned int)>("hipHostRegister");
    g_allocs++;
    return f(p, n, flags);
}
} // extern "C"

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                          \
            return 2;                                                                             \
        }                                                                                         \
    } while (0)
#define PH(x)                                                                                     \
    do {                                                                                          \
        ph_status s_ = (x);                                                                       \
        if (s_ != PH_OK) {                                                                        \
            std::fprintf(stderr, "%s: status %d: %s\n", #x, (int)s_, ph_last_error_string());     \
            return 2;                                                                             \
        }                                                                                         \
    } while (0)

int main()
{
    const int64_t max_q = 9000, max_t = 12000;
    std::mt19937 rng(5);
    std::vector<uint32_t> hq((size_t)max_q * 8), ht((size_t)max_t * 8);
    for (auto &w : hq)
        w = rng();
    for (auto &w : ht)
        w = rng();
    void *d_q, *d_t;
    int32_t *d_idx[2], *d_dist[2];
    uint8_t *d_keep[2];
    CK(hipMalloc(&d_q, hq.size() * 4));
    CK(hipMalloc(&d_t, ht.size() * 4));
    for (int i = 0; i < 2; i++) {
        CK(hipMalloc((void **)&d_idx[i], (size_t)max_q * 8));
        CK(hipMalloc((void **)&d_dist[i], (size_t)max_q * 8));
        CK(hipMalloc((void **)&d_keep[i], (size_t)max_q));
    }
    CK(hipMemcpy(d_q, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_t, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    PH(ph_init(0));

    // positive control: the allocating entry point grows the library's workspace
    const long a0 = g_allocs.load();
    PH(ph_knn_hamming256_dev(d_q, max_q, d_t, max_t, 2, 0, d_idx[0], d_dist[0], d_keep[0], 0.8f, nullptr));
    PH(ph_stream_synchronize(nullptr));
    const long control = g_allocs.load() - a0;

    struct Case {
        int mode;
        int64_t nq[2], nt[2];
    } cases[] = {{1, {1500, 257}, {3001, 999}}, {0, {9000, 6100}, {12000, 5003}}, {2, {700, 129}, {2051, 300}}};
    long allocs_seen = 0, frees_seen = 0;
    bool equal = true;
    int calls = 0;
    std::vector<int32_t> a, b;
    std::vector<uint8_t> ka, kb;
    for (const Case &c : cases) {
        PH(ph_knn_set_path(c.mode));
        for (int call = 0; call < 2; call++) {
            const int64_t nq = c.nq[call], nt = c.nt[call];
            // reference first (may allocate), then the entry point under test with the counters at zero
            PH(ph_knn_hamming256_dev(d_q, nq, d_t, nt, 2, 100 * call, d_idx[0], d_dist[0], d_keep[0], 0.8f, nullptr));
            PH(ph_stream_synchronize(nullptr));
            const int path_ref = ph_knn_last_path();
            g_allocs = 0;
            g_frees = 0;
            PH(ph_knn_hamming256_noalloc_dev(d_q, nq, d_t, nt, 2, 100 * call, d_idx[1], d_dist[1], d_keep[1], 0.8f,
                                             nullptr));
            PH(ph_stream_synchronize(nullptr));
            allocs_seen += g_allocs.load();
            frees_seen += g_frees.load();
            equal = equal && ph_knn_last_path() == path_ref;
            a.resize((size_t)nq * 2);
            b.resize((size_t)nq * 2);
            ka.resize((size_t)nq);
            kb.resize((size_t)nq);
            for (int which = 0; which < 2; which++) { // idx, then dist
                CK(hipMemcpy(a.data(), which ? d_dist[0] : d_idx[0], (size_t)nq * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(b.data(), which ? d_dist[1] : d_idx[1], (size_t)nq * 8, hipMemcpyDeviceToHost));
                equal = equal && std::memcmp(a.data(), b.data(), (size_t)nq * 8) == 0;
            }
            CK(hipMemcpy(ka.data(), d_keep[0], (size_t)nq, hipMemcpyDeviceToHost));
            CK(hipMemcpy(kb.data(), d_keep[1], (size_t)nq, hipMemcpyDeviceToHost));
            equal = equal && std::memcmp(ka.data(), kb.data(), (size_t)nq) == 0;
            calls++;
        }
    }
    PH(ph_knn_set_path(0));
    ph_shutdown();
    const bool ok = control > 0 && allocs_seen == 0 && frees_seen == 0 && equal && calls == 6;
    std::printf("{\"allocating_calls_of_the_control\": %ld, \"noalloc_calls\": %d, \"allocations_during_them\": %ld, "
                "\"frees_during_them\": %ld, \"equal_to_allocating_entry_point\": %s, \"ok\": %s}\n",
                control, calls, allocs_seen, frees_seen, equal ? "true" : "false", ok ? "true" : "false");
    return ok ? 0 : 1;
}
