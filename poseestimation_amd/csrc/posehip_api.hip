// C ABI of libposehip.so (declared in include/posehip.h): context, workspace,
// argument checks and host<->device staging around the matcher kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "knn_hamming.h"
#include "posehip.h"

namespace {

using namespace posehip;

thread_local char g_err[512] = "";

ph_status fail(ph_status st, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return st;
}

#define PH_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(PH_ERR_HIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,     \
                        __LINE__);                                                                \
    } while (0)

// A device buffer that only ever grows; freed at ph_shutdown.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess)
                return e;
        }
        size_t want = std::max(bytes, (size_t)1 << 20);
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess)
            cap = want;
        else
            p = nullptr;
        return e;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Context {
    bool ready = false;
    int device = -1;
    int n_cus = 256;
    hipStream_t stream = nullptr;
    DevBuf part;   // per-split keys of the distance kernel
    DevBuf part2;  // same, for the swapped pass of the cross-check
    DevBuf best;   // cross-check per-query 64-bit keys
    DevBuf tmp_i;  // cross-check per-train nearest query index
    DevBuf tmp_d;  // cross-check per-train nearest query distance
    DevBuf in_q, in_t, out_i, out_d, out_k; // staging for the host entry points
    bool profiling = false;                 // record events around the distance kernel
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    bool ev_valid = false;
};

Context g_ctx;
std::mutex g_mu;

ph_status ensure_ready()
{
    if (!g_ctx.ready)
        return fail(PH_ERR_NOT_INITIALIZED, "ph_init() has not been called");
    hipError_t e = hipSetDevice(g_ctx.device); // the current device is per-thread state
    if (e != hipSuccess)
        return fail(PH_ERR_HIP, "hipSetDevice(%d): %s", g_ctx.device, hipGetErrorString(e));
    return PH_OK;
}

hipStream_t pick_stream(void *s) { return s ? (hipStream_t)s : g_ctx.stream; }

ph_status check_rows(const void *p, int64_t n, const char *what)
{
    if (n < 0)
        return fail(PH_ERR_INVALID_ARG, "%s: negative row count", what);
    if (n > 0 && p == nullptr)
        return fail(PH_ERR_INVALID_ARG, "%s: null pointer with %lld rows", what, (long long)n);
    if (n > 0x7FFFFFFF)
        return fail(PH_ERR_TOO_LARGE, "%s: %lld rows exceed the 32-bit index range", what,
                    (long long)n);
    return PH_OK;
}

// Fills outputs for the degenerate case nt == 0 (every slot "no match").
__global__ void fill_empty_kernel(int32_t *idx, int32_t *dist, uint8_t *keep, int64_t nq, int k)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq)
        return;
    for (int r = 0; r < k; r++) {
        idx[i * k + r] = PH_NO_MATCH_IDX;
        dist[i * k + r] = PH_NO_MATCH_DIST;
    }
    if (keep)
        keep[i] = 0;
}

// k-NN on device buffers; the caller holds g_mu and has called ensure_ready().
ph_status knn_dev_locked(const void *d_q, int64_t nq, const void *d_t, int64_t nt, int k,
                         int32_t base, int32_t *d_idx, int32_t *d_dist, uint8_t *d_keep,
                         float ratio, DevBuf &part, hipStream_t st)
{
    if (nq == 0)
        return PH_OK;
    if (nt == 0) {
        fill_empty_kernel<<<(unsigned)((nq + 255) / 256), 256, 0, st>>>(d_idx, d_dist, d_keep, nq,
                                                                        k);
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    if (((uintptr_t)d_q & 15) || ((uintptr_t)d_t & 15))
        return fail(PH_ERR_INVALID_ARG, "descriptor buffers must be 16-byte aligned");
    if ((int64_t)base + nt - 1 > 0x7FFFFFFF)
        return fail(PH_ERR_TOO_LARGE, "train_index_base + nt exceeds the 32-bit index range");
    KnnPlan plan = knn_plan(nq, nt, g_ctx.n_cus);
    if (plan.rows_per_split * (int64_t)plan.n_splits < nt)
        return fail(PH_ERR_TOO_LARGE, "train set of %lld rows is too large for one call", (long long)nt);
    PH_HIP(part.reserve(plan.part_bytes));
    hipEvent_t *ev = g_ctx.profiling ? g_ctx.ev : nullptr;
    PH_HIP(knn_launch(plan, d_q, nq, d_t, nt, k, base, (uint32_t *)part.p, d_idx, d_dist, d_keep,
                      ratio, st, ev));
    g_ctx.ev_valid = ev != nullptr;
    return PH_OK;
}

ph_status crosscheck_dev_locked(const void *d_q, int64_t nq, const void *d_t, int64_t nt,
                                int32_t *d_idx, int32_t *d_dist, hipStream_t st)
{
    if (nq == 0)
        return PH_OK;
    if (nt == 0) {
        fill_empty_kernel<<<(unsigned)((nq + 255) / 256), 256, 0, st>>>(d_idx, d_dist, nullptr,
                                                                        nq, 1);
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    PH_HIP(g_ctx.tmp_i.reserve((size_t)nt * sizeof(int32_t)));
    PH_HIP(g_ctx.tmp_d.reserve((size_t)nt * sizeof(int32_t)));
    PH_HIP(g_ctx.best.reserve((size_t)nq * sizeof(unsigned long long)));
    // Roles swapped: every train row finds its nearest query row.
    ph_status s = knn_dev_locked(d_t, nt, d_q, nq, 1, 0, (int32_t *)g_ctx.tmp_i.p,
                                 (int32_t *)g_ctx.tmp_d.p, nullptr, 0.f, g_ctx.part2, st);
    if (s != PH_OK)
        return s;
    PH_HIP(crosscheck_resolve_launch((const int32_t *)g_ctx.tmp_i.p, (const int32_t *)g_ctx.tmp_d.p,
                                     nt, nq, (unsigned long long *)g_ctx.best.p, d_idx, d_dist,
                                     st));
    return PH_OK;
}

// Copies `n` descriptor rows with arbitrary stride into a packed device buffer.
ph_status upload_rows(DevBuf &buf, const uint8_t *src, int64_t n, size_t step, hipStream_t st)
{
    if (n == 0)
        return PH_OK;
    PH_HIP(buf.reserve((size_t)n * 32));
    if (step == 32)
        PH_HIP(hipMemcpyAsync(buf.p, src, (size_t)n * 32, hipMemcpyHostToDevice, st));
    else
        PH_HIP(hipMemcpy2DAsync(buf.p, 32, src, step, 32, (size_t)n, hipMemcpyHostToDevice, st));
    return PH_OK;
}

ph_status check_host_args(const uint8_t *q, int nq, size_t q_step, const uint8_t *t, int nt,
                          size_t t_step)
{
    ph_status s = check_rows(q, nq, "query");
    if (s != PH_OK)
        return s;
    s = check_rows(t, nt, "train");
    if (s != PH_OK)
        return s;
    if ((nq > 0 && q_step < 32) || (nt > 0 && t_step < 32))
        return fail(PH_ERR_INVALID_ARG, "row stride below 32 bytes");
    return PH_OK;
}

} // namespace

extern "C" {

ph_status ph_init(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx.ready) {
        if (g_ctx.device == device)
            return PH_OK;
        return fail(PH_ERR_INVALID_ARG, "already bound to device %d; call ph_shutdown() first",
                    g_ctx.device);
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(PH_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fail(PH_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, n - 1);
    PH_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PH_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PH_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                    device, prop.gcnArchName);
    g_ctx.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    PH_HIP(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
    g_ctx.device = device;
    g_ctx.ready = true;
    return PH_OK;
}

void ph_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ctx.ready)
        return;
    (void)hipSetDevice(g_ctx.device);
    (void)hipStreamSynchronize(g_ctx.stream);
    DevBuf *all[] = {&g_ctx.part, &g_ctx.part2, &g_ctx.best, &g_ctx.tmp_i, &g_ctx.tmp_d,
                     &g_ctx.in_q, &g_ctx.in_t,  &g_ctx.out_i, &g_ctx.out_d, &g_ctx.out_k};
    for (DevBuf *b : all)
        b->release();
    (void)hipStreamDestroy(g_ctx.stream);
    for (hipEvent_t e : g_ctx.ev)
        if (e)
            (void)hipEventDestroy(e);
    g_ctx = Context();
}

const char *ph_last_error_string(void) { return g_err; }

int ph_device(void) { return g_ctx.ready ? g_ctx.device : -1; }

ph_status ph_stream_synchronize(void *stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    ph_status s = ensure_ready();
    if (s != PH_OK)
        return s;
    PH_HIP(hipStreamSynchronize(pick_stream(stream)));
    return PH_OK;
}

ph_status ph_set_profiling(int enable)
{
    std::lock_guard<std::mutex> lk(g_mu);
    ph_status s = ensure_ready();
    if (s != PH_OK)
        return s;
    if (enable && !g_ctx.ev[0])
        for (hipEvent_t &e : g_ctx.ev)
            PH_HIP(hipEventCreate(&e));
    g_ctx.profiling = enable != 0;
    g_ctx.ev_valid = false;
    return PH_OK;
}

ph_status ph_knn_last_timing(float *distance_kernel_ms, float *merge_kernel_ms)
{
    std::lock_guard<std::mutex> lk(g_mu);
    ph_status s = ensure_ready();
    if (s != PH_OK)
        return s;
    if (!g_ctx.ev_valid)
        return fail(PH_ERR_INVALID_ARG, "no profiled k-NN call (ph_set_profiling(1) first)");
    PH_HIP(hipEventSynchronize(g_ctx.ev[2]));
    if (distance_kernel_ms)
        PH_HIP(hipEventElapsedTime(distance_kernel_ms, g_ctx.ev[0], g_ctx.ev[1]));
    if (merge_kernel_ms)
        PH_HIP(hipEventElapsedTime(merge_kernel_ms, g_ctx.ev[1], g_ctx.ev[2]));
    return PH_OK;
}

ph_status ph_valu_popcount_peak(int iters, double *lane_ops_per_s)
{
    std::lock_guard<std::mutex> lk(g_mu);
    ph_status s = ensure_ready();
    if (s != PH_OK)
        return s;
    if (iters < 1 || !lane_ops_per_s)
        return fail(PH_ERR_INVALID_ARG, "iters must be >= 1 and the output non-null");
    PH_HIP(g_ctx.tmp_i.reserve((size_t)g_ctx.n_cus * 8 * 256 * sizeof(uint32_t)));
    PH_HIP(valu_popcount_peak(g_ctx.n_cus, iters, (uint32_t *)g_ctx.tmp_i.p, g_ctx.stream,
                              lane_ops_per_s));
    return PH_OK;
}

ph_status ph_knn_hamming256_dev(const void *d_query, int64_t nq, const void *d_train, int64_t nt,
                                int k, int32_t train_index_base, int32_t *d_out_idx,
                                int32_t *d_out_dist, uint8_t *d_keep, float ratio, void *stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    ph_status s = ensure_ready();
    if (s != PH_OK)
        return s;
    if (k != 1 && k != 2)
        return fail(PH_ERR_INVALID_ARG, "k must be 1 or 2 (got %d)", k);
    if (d_keep && k != 2)
        return fail(PH_ERR_INVALID_ARG, "the ratio test needs k == 2");
    if (train_index_base < 0)
        return fail(PH_ERR_INVALID_ARG, "negative train_index_base");
    if ((s = check_rows(d_query, nq, "query")) != PH_OK)
        return s;
    if ((s = check_rows(d_train, nt, "train")) != PH_OK)
        return s;
    if (nq > 0 && (!d_out_idx || !d_out_dist))
        return fail(PH_ERR_INVALID_ARG, "null output pointer");
    return knn_dev_locked(d_query, nq, d_train, nt, k, train_index_base, d_out_idx, d_out_dist,
                          d_keep, ratio, g_ctx.part, pick_stream(stream));
}

ph_status ph_knn_merge_dev(const int32_t *d_cand_idx, const int32_t *d_cand_dist, int n_shards,
                           int64_t nq, int k, int32_t *d_out_idx, int32_t *d_out_dist,
                           uint8_t *d_keep, float ratio, void *stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    ph_status s = ensure_ready();
    if (s != PH_OK)
        return s;
    if (k != 1 && k != 2)
        return fail(PH_ERR_INVALID_ARG, "k must be 1 or 2 (got %d)", k);
    if (d_keep && k != 2)
        return fail(PH_ERR_INVALID_ARG, "the ratio test needs k == 2");
    if (n_shards < 1 || nq < 0)
        return fail(PH_ERR_INVALID_ARG, "n_shards must be >= 1 and nq >= 0");
    if (nq == 0)
        return PH_OK;
    if (!d_cand_idx || !d_cand_dist || !d_out_idx || !d_out_dist)
        return fail(PH_ERR_INVALID_ARG, "null pointer");
    PH_HIP(knn_merge_shards_launch(d_cand_idx, d_cand_dist, n_shards, nq, k, d_out_idx, d_out_dist,
                                   d_keep, ratio, pick_stream(stream)));
    return PH_OK;
}

ph_status ph_match_crosscheck_hamming256_dev(const void *d_query, int64_t nq, const void *d_train,
                                             int64_t nt, int32_t *d_out_idx, int32_t *d_out_dist,
                                             void *stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    ph_status s = ensure_ready();
    if (s != PH_OK)
        return s;
    if ((s = check_rows(d_query, nq, "query")) != PH_OK)
        return s;
    if ((s = check_rows(d_train, nt, "train")) != PH_OK)
        return s;
    if (nq > 0 && (!d_out_idx || !d_out_dist))
        return fail(PH_ERR_INVALID_ARG, "null output pointer");
    return crosscheck_dev_locked(d_query, nq, d_train, nt, d_out_idx, d_out_dist,
                                 pick_stream(stream));
}

ph_status ph_knn_hamming256(const uint8_t *query, int nq, size_t q_step, const uint8_t *train,
                            int nt, size_t t_step, int k, int32_t *out_idx, int32_t *out_dist)
{
    std::lock_guard<std::mutex> lk(g_mu);
    ph_status s = ensure_ready();
    if (s != PH_OK)
        return s;
    if (k != 1 && k != 2)
        return fail(PH_ERR_INVALID_ARG, "k must be 1 or 2 (got %d)", k);
    if ((s = check_host_args(query, nq, q_step, train, nt, t_step)) != PH_OK)
        return s;
    if (nq == 0)
        return PH_OK;
    if (!out_idx || !out_dist)
        return fail(PH_ERR_INVALID_ARG, "null output pointer");
    hipStream_t st = g_ctx.stream;
    if ((s = upload_rows(g_ctx.in_q, query, nq, q_step, st)) != PH_OK)
        return s;
    if ((s = upload_rows(g_ctx.in_t, train, nt, t_step, st)) != PH_OK)
        return s;
    const size_t ob = (size_t)nq * k * sizeof(int32_t);
    PH_HIP(g_ctx.out_i.reserve(ob));
    PH_HIP(g_ctx.out_d.reserve(ob));
    s = knn_dev_locked(g_ctx.in_q.p, nq, g_ctx.in_t.p, nt, k, 0, (int32_t *)g_ctx.out_i.p,
                       (int32_t *)g_ctx.out_d.p, nullptr, 0.f, g_ctx.part, st);
    if (s != PH_OK)
        return s;
    PH_HIP(hipMemcpyAsync(out_idx, g_ctx.out_i.p, ob, hipMemcpyDeviceToHost, st));
    PH_HIP(hipMemcpyAsync(out_dist, g_ctx.out_d.p, ob, hipMemcpyDeviceToHost, st));
    PH_HIP(hipStreamSynchronize(st));
    return PH_OK;
}

ph_status ph_match_hamming256(const uint8_t *query, int nq, size_t q_step, const uint8_t *train,
                              int nt, size_t t_step, ph_dmatch *out, int *n_out)
{
    if (!n_out)
        return fail(PH_ERR_INVALID_ARG, "n_out is null");
    *n_out = 0;
    if (nq > 0 && nt > 0 && !out)
        return fail(PH_ERR_INVALID_ARG, "null output pointer");
    std::vector<int32_t> idx((size_t)std::max(nq, 0)), dist((size_t)std::max(nq, 0));
    ph_status s = ph_knn_hamming256(query, nq, q_step, train, nt, t_step, 1, idx.data(),
                                    dist.data());
    if (s != PH_OK)
        return s;
    int n = 0;
    for (int i = 0; i < nq; i++) {
        if (idx[i] < 0)
            continue;
        out[n].queryIdx = i;
        out[n].trainIdx = idx[i];
        out[n].imgIdx = 0;
        out[n].distance = (float)dist[i];
        n++;
    }
    *n_out = n;
    return PH_OK;
}

ph_status ph_match_crosscheck_hamming256(const uint8_t *query, int nq, size_t q_step,
                                         const uint8_t *train, int nt, size_t t_step,
                                         ph_dmatch *out, int *n_out)
{
    if (!n_out)
        return fail(PH_ERR_INVALID_ARG, "n_out is null");
    *n_out = 0;
    std::vector<int32_t> idx, dist;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        ph_status s = ensure_ready();
        if (s != PH_OK)
            return s;
        if ((s = check_host_args(query, nq, q_step, train, nt, t_step)) != PH_OK)
            return s;
        if (nq == 0 || nt == 0)
            return PH_OK;
        if (!out)
            return fail(PH_ERR_INVALID_ARG, "null output pointer");
        hipStream_t st = g_ctx.stream;
        if ((s = upload_rows(g_ctx.in_q, query, nq, q_step, st)) != PH_OK)
            return s;
        if ((s = upload_rows(g_ctx.in_t, train, nt, t_step, st)) != PH_OK)
            return s;
        const size_t ob = (size_t)nq * sizeof(int32_t);
        PH_HIP(g_ctx.out_i.reserve(ob));
        PH_HIP(g_ctx.out_d.reserve(ob));
        s = crosscheck_dev_locked(g_ctx.in_q.p, nq, g_ctx.in_t.p, nt, (int32_t *)g_ctx.out_i.p,
                                  (int32_t *)g_ctx.out_d.p, st);
        if (s != PH_OK)
            return s;
        idx.resize((size_t)nq);
        dist.resize((size_t)nq);
        PH_HIP(hipMemcpyAsync(idx.data(), g_ctx.out_i.p, ob, hipMemcpyDeviceToHost, st));
        PH_HIP(hipMemcpyAsync(dist.data(), g_ctx.out_d.p, ob, hipMemcpyDeviceToHost, st));
        PH_HIP(hipStreamSynchronize(st));
    }
    int n = 0;
    for (int i = 0; i < nq; i++) {
        if (idx[i] < 0)
            continue;
        out[n].queryIdx = i;
        out[n].trainIdx = idx[i];
        out[n].imgIdx = 0;
        out[n].distance = (float)dist[i];
        n++;
    }
    *n_out = n;
    return PH_OK;
}

ph_status ph_knn2_ratio_hamming256(const uint8_t *query, int nq, size_t q_step,
                                   const uint8_t *train, int nt, size_t t_step, float ratio,
                                   ph_dmatch *out, int *n_out)
{
    if (!n_out)
        return fail(PH_ERR_INVALID_ARG, "n_out is null");
    *n_out = 0;
    std::vector<int32_t> idx, dist;
    std::vector<uint8_t> keep;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        ph_status s = ensure_ready();
        if (s != PH_OK)
            return s;
        if ((s = check_host_args(query, nq, q_step, train, nt, t_step)) != PH_OK)
            return s;
        if (nq == 0 || nt == 0)
            return PH_OK;
        if (!out)
            return fail(PH_ERR_INVALID_ARG, "null output pointer");
        hipStream_t st = g_ctx.stream;
        if ((s = upload_rows(g_ctx.in_q, query, nq, q_step, st)) != PH_OK)
            return s;
        if ((s = upload_rows(g_ctx.in_t, train, nt, t_step, st)) != PH_OK)
            return s;
        const size_t ob = (size_t)nq * 2 * sizeof(int32_t);
        PH_HIP(g_ctx.out_i.reserve(ob));
        PH_HIP(g_ctx.out_d.reserve(ob));
        PH_HIP(g_ctx.out_k.reserve((size_t)nq));
        s = knn_dev_locked(g_ctx.in_q.p, nq, g_ctx.in_t.p, nt, 2, 0, (int32_t *)g_ctx.out_i.p,
                           (int32_t *)g_ctx.out_d.p, (uint8_t *)g_ctx.out_k.p, ratio, g_ctx.part,
                           st);
        if (s != PH_OK)
            return s;
        idx.resize((size_t)nq * 2);
        dist.resize((size_t)nq * 2);
        keep.resize((size_t)nq);
        PH_HIP(hipMemcpyAsync(idx.data(), g_ctx.out_i.p, ob, hipMemcpyDeviceToHost, st));
        PH_HIP(hipMemcpyAsync(dist.data(), g_ctx.out_d.p, ob, hipMemcpyDeviceToHost, st));
        PH_HIP(hipMemcpyAsync(keep.data(), g_ctx.out_k.p, (size_t)nq, hipMemcpyDeviceToHost, st));
        PH_HIP(hipStreamSynchronize(st));
    }
    int n = 0;
    for (int i = 0; i < nq; i++) {
        if (!keep[i])
            continue;
        out[n].queryIdx = i;
        out[n].trainIdx = idx[2 * (size_t)i];
        out[n].imgIdx = 0;
        out[n].distance = (float)dist[2 * (size_t)i];
        n++;
    }
    *n_out = n;
    return PH_OK;
}

ph_status ph_min_dist_filter(const ph_dmatch *m, int n, ph_dmatch *out, int *n_out)
{
    if (!n_out)
        return fail(PH_ERR_INVALID_ARG, "n_out is null");
    *n_out = 0;
    if (n < 0 || (n > 0 && (!m || !out)))
        return fail(PH_ERR_INVALID_ARG, "bad match list");
    // src/cereal_box.cpp:345-356: min_dist starts at 10000; the comparison is in double.
    double min_dist = 10000;
    for (int i = 0; i < n; i++)
        min_dist = std::min(min_dist, (double)m[i].distance);
    const double thr = std::max(2 * min_dist, 30.0);
    int k = 0;
    for (int i = 0; i < n; i++)
        if ((double)m[i].distance <= thr)
            out[k++] = m[i];
    *n_out = k;
    return PH_OK;
}

} // extern "C"
