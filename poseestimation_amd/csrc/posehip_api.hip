// C ABI of libposehip.so (declared in include/posehip.h): context, workspace,
// argument checks and host<->device staging around the matcher kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cfloat>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "knn_hamming.h"
#include "knn_mfma.h"
#include "ransac_epnp.h"
#include "posehip.h"

// Scratch of ph_knn_hamming256_noalloc_dev: the FP4 image of the train rows and the
// per-split keys of one call.  Every GPU the library runs on holds its own copy at its
// own address; `scratch` itself is only the host-side stand-in of the symbol, the
// per-device address comes from hipGetSymbolAddress in ph_init (g_noalloc_ptr).
static __device__ __align__(256) unsigned char scratch[PH_NOALLOC_SCRATCH_BYTES];

// A train set resident in HBM (ph_db_upload); opaque to the C ABI.  It belongs to the
// context that uploaded it (and dies with it) and serves calls of any context on its GPU.
struct ph_db {
    void *d = nullptr;
    void *x = nullptr; // FP4 image of the rows, built by the first call that takes the matrix cores
    std::mutex build_mu; // ... once, whichever context's call comes first
    int64_t rows = 0;
    int device = -1;
    struct ph_ctx *owner = nullptr;
};

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
        // grow by at least half: a stream of slowly growing requests (keypoint counts
        // creeping upward) reallocates O(log) times; hipFree synchronises the whole device
        const size_t want = std::max(std::max(bytes, cap + cap / 2), (size_t)1 << 20);
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess)
                return e;
        }
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

// Pinned, device-visible host memory that only ever grows.  Kernels write small
// results straight into it (no device-to-host copy), and host inputs are gathered in
// it so that one asynchronous copy moves them.
struct PinBuf {
    void *p = nullptr;  // host address
    void *dp = nullptr; // the same memory as the device sees it
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        const size_t want = std::max(std::max(bytes, cap + cap / 2), (size_t)1 << 16);
        release();
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        e = hipHostGetDevicePointer(&dp, p, 0);
        if (e != hipSuccess) {
            (void)hipHostFree(p);
            p = dp = nullptr;
            return e;
        }
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (p)
            (void)hipHostFree(p);
        p = dp = nullptr;
        cap = 0;
    }
};

// ---- RANSAC host logic ------------------------------------------------------------
// cv::RNG: multiply-with-carry, state = lo32 * 4164903690 + hi32.
struct CvRng {
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffu) {}
    uint32_t next()
    {
        state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + a); }
};

// RANSACPointSetRegistrator::getSubset for the PnP callback (checkSubset always
// accepts): five distinct row indices, redrawing a slot while it repeats an
// earlier one.  The draws do not depend on any model, so the whole sequence of
// subsets can be generated before a single hypothesis is solved.
void draw_subsets(CvRng &rng, int count, int n_subsets, int32_t *out)
{
    for (int s = 0; s < n_subsets; s++) {
        int32_t *idx = out + (size_t)s * 5;
        for (int i = 0; i < 5; i++) {
            int v;
            bool dup;
            do {
                v = rng.uniform(0, count);
                dup = false;
                for (int j = 0; j < i; j++)
                    dup |= idx[j] == v;
            } while (dup);
            idx[i] = v;
        }
    }
}

void draw_subsets(uint64_t rng_state, int count, int n_subsets, int32_t *out)
{
    CvRng rng(rng_state);
    draw_subsets(rng, count, n_subsets, out);
}

// The sample sequence depends only on (RNG seed, number of correspondences), so it
// is kept across calls: a video's frames mostly repeat a few hundred distinct
// correspondence counts, and a repeated count costs no RNG work at all.
struct SubsetCache {
    struct Entry {
        CvRng rng;
        std::vector<int32_t> subs; // 5 row indices per sample, in drawing order
    };
    uint64_t seed = 0;
    size_t bytes = 0;
    std::unordered_map<int, Entry> by_count;

    // The first n_subsets samples for `count` correspondences (valid until the next call).
    const int32_t *get(uint64_t rng_state, int count, int n_subsets)
    {
        if (rng_state != seed || bytes > ((size_t)256 << 20)) {
            by_count.clear();
            bytes = 0;
            seed = rng_state;
        }
        auto it = by_count.find(count);
        if (it == by_count.end())
            it = by_count.emplace(count, Entry{CvRng(rng_state), {}}).first;
        Entry &e = it->second;
        const size_t have = e.subs.size() / 5;
        if (have < (size_t)n_subsets) {
            // grow geometrically: a budget that shrinks and grows again costs nothing
            const size_t want = std::max((size_t)n_subsets, have * 2);
            e.subs.resize(want * 5);
            draw_subsets(e.rng, count, (int)(want - have), e.subs.data() + have * 5);
            bytes += (want - have) * 5 * sizeof(int32_t);
        }
        return e.subs.data();
    }
};

// RANSACUpdateNumIters, in two parts: the two logarithms (the only libm calls) and the rule on
// them.  The lone-call path hands the logarithms to the GPU as a table over every possible
// inlier count and lets the refit kernel apply the rule (replay_update_iters in
// ransac_epnp.hip: the same product, comparison, quotient and rounding).
double iters_num_log(double p)
{
    p = std::max(p, 0.);
    p = std::min(p, 1.);
    return std::log(std::max(1. - p, DBL_MIN));
}

// log(1 - (1 - ep)^model_points), or NaN where the rule returns 0 (no outliers left).
double iters_denom_log(double ep, int model_points)
{
    ep = std::max(ep, 0.);
    ep = std::min(ep, 1.);
    const double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN)
        return std::nan("");
    return std::log(denom);
}

int update_num_iters(double p, double ep, int model_points, int max_iters)
{
    const double num = iters_num_log(p), denom = iters_denom_log(ep, model_points);
    if (denom != denom)
        return 0;
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

// iters_denom_log for every inlier count 0..n of a problem of n correspondences, kept across
// calls per n (a video's frames repeat a few hundred distinct counts; the first call with a
// new n pays n pow + log evaluations, about 50 ns each).
struct IterTableCache {
    size_t bytes = 0;
    std::unordered_map<int, std::vector<double>> by_n;
    const double *get(int n, int model_points)
    {
        if (bytes > ((size_t)64 << 20)) {
            by_n.clear();
            bytes = 0;
        }
        auto it = by_n.find(n);
        if (it == by_n.end()) {
            std::vector<double> t((size_t)n + 1);
            for (int g = 0; g <= n; g++)
                t[(size_t)g] = iters_denom_log((double)(n - g) / n, model_points);
            bytes += t.size() * sizeof(double);
            it = by_n.emplace(n, std::move(t)).first;
        }
        return it->second.data();
    }
};

} // namespace

// One binding of the library to a GPU: stream, workspace, staging buffers, settings and the
// lock that serialises the calls made through it.  ph_init creates the process-wide default
// context; ph_ctx_create makes further ones (ph_ctx in the C ABI).
struct ph_ctx {
    std::mutex mu;              // held for the duration of every call through this context
    std::atomic<int> users{0};  // calls that have picked this context (ph_ctx_destroy waits for 0)
    SubsetCache subsets;        // RANSAC sample sequences, per (seed, correspondence count)
    IterTableCache iter_tables; // RANSAC budget-rule logarithms, per correspondence count
    bool ready = false;
    int device = -1;
    int n_cus = 256;
    hipStream_t stream = nullptr;
    DevBuf part;   // per-split keys of the distance kernel
    DevBuf part2;  // same, for the swapped pass of the cross-check
    DevBuf xt;     // FP4 image of the train rows (matrix-core path)
    DevBuf best;   // cross-check per-query nearest train index | distance
    DevBuf tmp_i;  // cross-check per-train nearest query index
    DevBuf tmp_d;  // cross-check per-train nearest query distance
    DevBuf d_in;            // host entry points: device copy of the gathered inputs
    PinBuf h_in, h_out;     // host entry points: pinned gather buffer / zero-copy results
    DevBuf pnp_idx;         // RANSAC-EPnP: inlier rows of the refit
    DevBuf pnp_front;       // RANSAC-EPnP: hand-over between the two solve kernels (one round)
    DevBuf pnp_hyp;         // RANSAC-EPnP, lone call: models | counts of its hypotheses
    // Frame-offset tables of batched calls travel through a small ring of pinned
    // slots (one asynchronous copy each; a slot is reused once its call has finished).
    struct OffsetSlot {
        PinBuf h;
        DevBuf d;
        hipEvent_t done = nullptr;
    };
    OffsetSlot off_ring[4];
    int off_next = 0;
    int off_cached = -1; // slot whose tables equal the last call's (re-used without a copy)
    int off_cached_frames = 0;
    // The workspace above is shared by all calls: the stream that used it last.
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;
    hipEvent_t scratch_ev = nullptr;
    posehip::RansacOverlap pnp_overlap = {nullptr, nullptr, nullptr}; // side stream of batched RANSAC
    bool profiling = false;                 // record events around the distance kernel
    int knn_path = 0;                       // 0: by size, 1: popcount kernel only, 2: matrix cores only
    int last_path = 0;                      // distance kernel of the last k-NN call: 0 VALU, 1 MFMA
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // [4]: ahead of the FP4 pre-pass
    bool ev_valid = false;
    hipEvent_t pnp_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // RANSAC solve / score
    bool pnp_ev_valid = false;
};

namespace {

using Context = ph_ctx;

// Registry of live contexts and resident train sets.  g_reg_mu is held only for look-ups and
// insertions, never across a GPU call; lock order: g_reg_mu before a context's mu.
std::mutex g_reg_mu;
std::unordered_set<Context *> g_live;
Context *g_default = nullptr;              // the context of ph_init
thread_local Context *t_current = nullptr; // ph_ctx_make_current on this thread; null = default
int g_pending_knn_path = 0;                // ph_knn_set_path before ph_init
// Device address of `scratch` on every GPU ph_init has bound so far (each GPU has its own
// copy of the array); looked up once per device with that device current.
constexpr int MAX_DEVICES = 64;
unsigned char *g_noalloc_ptr[MAX_DEVICES] = {};
std::unordered_set<ph_db *> g_dbs; // live resident train sets (freed with their context)

bool db_live(const ph_db *db)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return db && g_dbs.count(const_cast<ph_db *>(db));
}


// Every entry point that needs a GPU starts with one of these: it picks the calling thread's
// context (the one made current with ph_ctx_make_current, else the default one), takes its
// lock for the duration of the call and makes its device current (the current device is
// per-thread state in HIP).
struct Call {
    Context *c = nullptr;
    ph_status st = PH_OK;
    Call()
    {
        {
            std::lock_guard<std::mutex> lk(g_reg_mu);
            if (t_current && !g_live.count(t_current)) {
                st = fail(PH_ERR_INVALID_ARG, "this thread's current context has been destroyed");
                return;
            }
            Context *want = t_current ? t_current : g_default;
            if (!want) {
                st = fail(PH_ERR_NOT_INITIALIZED, "ph_init() has not been called");
                return;
            }
            c = want;
            c->users.fetch_add(1);
        }
        c->mu.lock();
        hipError_t e = hipSetDevice(c->device);
        if (e != hipSuccess)
            st = fail(PH_ERR_HIP, "hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
    }
    ~Call()
    {
        if (c) {
            c->mu.unlock();
            c->users.fetch_sub(1);
        }
    }
    Call(const Call &) = delete;
    Call &operator=(const Call &) = delete;
};

hipStream_t pick_stream(Context &c, void *s) { return s ? (hipStream_t)s : c.stream; }

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

// Work queued on another stream may still be using the shared workspace: make `st`
// wait for it (same stream: stream order already does).
ph_status scratch_guard(Context &c, hipStream_t st)
{
    if (c.scratch_used && c.scratch_stream != st) {
        hipError_t e = hipSuccess;
        if (!c.scratch_ev)
            e = hipEventCreateWithFlags(&c.scratch_ev, hipEventDisableTiming);
        if (e == hipSuccess)
            e = hipEventRecord(c.scratch_ev, c.scratch_stream);
        if (e == hipSuccess)
            e = hipStreamWaitEvent(st, c.scratch_ev, 0);
        if (e != hipSuccess) { // e.g. the caller destroyed the earlier stream
            (void)hipGetLastError();
            PH_HIP(hipDeviceSynchronize());
        }
    }
    c.scratch_stream = st;
    c.scratch_used = true;
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

// k-NN on device buffers; the caller holds the context's lock (a Call).
// db: the resident set d_t belongs to, if any (its FP4 image is built once and kept).
ph_status knn_dev_locked(Context &c, const void *d_q, int64_t nq, const void *d_t, int64_t nt, int k,
                         int32_t base, int32_t *d_idx, int32_t *d_dist, uint8_t *d_keep,
                         float ratio, DevBuf &part, hipStream_t st, ph_db *db = nullptr)
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
    hipEvent_t *ev = c.profiling ? c.ev : nullptr;
    if (c.knn_path == 2 || (c.knn_path == 0 && knn_mfma_pays(nq, nt))) {
        const MfmaPlan mp = knn_mfma_plan(nq, nt, c.n_cus);
        ph_status gs = scratch_guard(c, st);
        if (gs != PH_OK)
            return gs;
        PH_HIP(part.reserve(mp.part_bytes));
        void *image = nullptr;
        const void *rows_to_expand = d_t;
        if (db) {
            std::lock_guard<std::mutex> build(db->build_mu);
            if (!db->x) { // once per resident set; later calls on any stream find it complete
                PH_HIP(hipMalloc(&db->x, knn_mfma_expanded_bytes(nt)));
                hipError_t e = knn_mfma_expand_launch(d_t, nt, db->x, st);
                if (e == hipSuccess)
                    e = hipStreamSynchronize(st);
                if (e != hipSuccess) {
                    (void)hipFree(db->x);
                    db->x = nullptr;
                    return fail(PH_ERR_HIP, "building the FP4 image of the train set: %s",
                                hipGetErrorString(e));
                }
            }
            image = db->x;
            rows_to_expand = nullptr;
        } else {
            PH_HIP(c.xt.reserve(knn_mfma_expanded_bytes(nt)));
            image = c.xt.p;
        }
        PH_HIP(knn_mfma_launch(mp, d_q, nq, rows_to_expand, nt, k, base, image,
                               (uint32_t *)part.p, d_idx, d_dist, d_keep, ratio, st, ev));
        c.ev_valid = ev != nullptr;
        c.last_path = 1;
        return PH_OK;
    }
    KnnPlan plan = knn_plan(nq, nt, c.n_cus);
    if (plan.rows_per_split * (int64_t)plan.n_splits < nt)
        return fail(PH_ERR_TOO_LARGE, "train set of %lld rows is too large for one call", (long long)nt);
    ph_status gs = scratch_guard(c, st);
    if (gs != PH_OK)
        return gs;
    PH_HIP(part.reserve(plan.part_bytes));
    c.last_path = 0;
    PH_HIP(knn_launch(plan, d_q, nq, d_t, nt, k, base, (uint32_t *)part.p, d_idx, d_dist, d_keep,
                      ratio, st, ev));
    c.ev_valid = ev != nullptr;
    return PH_OK;
}

// Cross-check match on device buffers (the caller holds the context's lock): both 1-NN searches, then the
// gather that keeps a query's nearest train row iff the query is that row's nearest query.
// packed: the result is one shard's candidate list for ph_knn_merge_dev (see
// crosscheck_resolve_launch).
ph_status crosscheck_dev_locked(Context &c, const void *d_q, int64_t nq, const void *d_t, int64_t nt,
                                int32_t *d_idx, int32_t *d_dist, hipStream_t st, int32_t base = 0,
                                bool packed = false)
{
    if (nq == 0)
        return PH_OK;
    if (nt == 0) {
        fill_empty_kernel<<<(unsigned)((nq + 255) / 256), 256, 0, st>>>(d_idx, d_dist, nullptr,
                                                                        nq, 1);
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    ph_status gs = scratch_guard(c, st);
    if (gs != PH_OK)
        return gs;
    PH_HIP(c.tmp_i.reserve((size_t)nt * sizeof(int32_t)));
    PH_HIP(c.tmp_d.reserve((size_t)nt * sizeof(int32_t)));
    PH_HIP(c.best.reserve((size_t)nq * 2 * sizeof(int32_t)));
    int32_t *sidx = (int32_t *)c.best.p, *sdist = sidx + nq;
    // every query row's nearest train row ...
    ph_status s = knn_dev_locked(c, d_q, nq, d_t, nt, 1, 0, sidx, sdist, nullptr, 0.f, c.part, st);
    if (s != PH_OK)
        return s;
    // ... and, roles swapped, every train row's nearest query row
    s = knn_dev_locked(c, d_t, nt, d_q, nq, 1, 0, (int32_t *)c.tmp_i.p, (int32_t *)c.tmp_d.p,
                       nullptr, 0.f, c.part2, st);
    if (s != PH_OK)
        return s;
    PH_HIP(crosscheck_resolve_launch(sidx, sdist, (const int32_t *)c.tmp_i.p, nq, d_idx, d_dist,
                                     st, base, packed));
    return PH_OK;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// The host entry points queue copies and kernels that read and write the pinned staging
// buffers.  When one of them fails half-way, what it has queued must finish before the
// next call repacks those buffers: unless disarmed (the success path has synchronised
// already), the destructor waits for the library stream.
struct DrainOnError {
    Context &c;
    bool armed = true;
    ~DrainOnError()
    {
        if (armed && c.stream)
            (void)hipStreamSynchronize(c.stream);
    }
};

void pack_rows(uint8_t *dst, const uint8_t *src, int64_t n, size_t step)
{
    if (n <= 0)
        return;
    if (step == 32) {
        memcpy(dst, src, (size_t)n * 32);
        return;
    }
    for (int64_t i = 0; i < n; i++)
        memcpy(dst + i * 32, src + (size_t)i * step, 32);
}

// Host descriptors -> device: both sets are packed into the pinned gather buffer and
// moved with ONE asynchronous copy.  On return d_q / d_t point into c.d_in.
ph_status upload_descriptors(Context &c, const uint8_t *q, int64_t nq, size_t q_step, const uint8_t *t,
                             int64_t nt, size_t t_step, hipStream_t st, const void **d_q,
                             const void **d_t)
{
    const size_t t_off = align256((size_t)nq * 32), total = t_off + (size_t)nt * 32;
    PH_HIP(c.h_in.reserve(total));
    PH_HIP(c.d_in.reserve(total));
    pack_rows((uint8_t *)c.h_in.p, q, nq, q_step);
    pack_rows((uint8_t *)c.h_in.p + t_off, t, nt, t_step);
    PH_HIP(hipMemcpyAsync(c.d_in.p, c.h_in.p, total, hipMemcpyHostToDevice, st));
    *d_q = c.d_in.p;
    *d_t = (const uint8_t *)c.d_in.p + t_off;
    return PH_OK;
}

// One side (query or train) of a batch of frames: device rows, HOST offset table, the
// row range the table spans and the longest frame.
struct FrameSide {
    const void *d_rows;
    const int32_t *offsets;
    int64_t begin, end, max_rows;
};

// Device copies of a batch's tables, in the order of knn_frames_locked's (q, t) arguments,
// and the ring slot they travel in (its `done` event is recorded behind the k-NN launch; a
// caller that queues more kernels reading the tables records it again behind those).
struct FrameTables {
    const int32_t *d_q_off = nullptr, *d_t_off = nullptr;
    Context::OffsetSlot *slot = nullptr;
};

ph_status knn_frames_locked(Context &c, const FrameSide &q, const FrameSide &t, int n_frames, bool uniform,
                            int k, DevBuf &part, int32_t *d_out_idx, int32_t *d_out_dist,
                            uint8_t *d_keep, float ratio, hipStream_t st, hipEvent_t *ev,
                            FrameTables *tables = nullptr);

// Results of the host matcher entry points, written by the kernels straight into the
// pinned result buffer: idx[nq*k] | dist[nq*k] | keep[nq].
struct HostResult {
    int32_t *idx, *dist;
    uint8_t *keep;
};

enum class MatchMode { Knn, CrossCheck, Ratio };

// The caller holds the context's lock, has validated the arguments and nq > 0.  Returns after the
// stream has been synchronised, with `r` pointing at host-readable results.
ph_status host_match_locked(Context &c, const uint8_t *q, int nq, size_t q_step, const uint8_t *t, int nt,
                            size_t t_step, MatchMode mode, int k, float ratio, HostResult &r)
{
    hipStream_t st = c.stream;
    const size_t n_out = (size_t)nq * k;
    const size_t dist_off = align256(n_out * 4), keep_off = dist_off + align256(n_out * 4);
    PH_HIP(c.h_out.reserve(keep_off + (size_t)nq));
    uint8_t *hb = (uint8_t *)c.h_out.p, *db = (uint8_t *)c.h_out.dp;
    int32_t *d_idx = (int32_t *)db, *d_dist = (int32_t *)(db + dist_off);
    uint8_t *d_keep = db + keep_off;
    ph_status s;
    // The cross-check's two searches (every query's nearest train row, every train row's
    // nearest query) are two independent problems; when the two sets are of similar, moderate
    // size (a frame against a frame, the reference's case) they go down as a batch of two
    // frames in ONE launch.  A batch is planned for its longest frame on either side, so very
    // unequal or very large sets keep the two separate searches.
    const int64_t longer = std::max(nq, nt), shorter = std::min(nq, nt);
    const bool as_two_frames = mode == MatchMode::CrossCheck && nt > 0 && longer <= 65536 &&
                               longer <= 8 * shorter;
    if (as_two_frames) {
        // The rows go up as q | t | q, so that the query side of the batch is the buffer
        // from row 0 (frames q, t) and its train side the same buffer from row nq (t, q).
        const size_t rows = (size_t)nq + nt, total = (rows + nq) * 32;
        PH_HIP(c.h_in.reserve(total));
        PH_HIP(c.d_in.reserve(total));
        uint8_t *h = (uint8_t *)c.h_in.p;
        pack_rows(h, q, nq, q_step);
        pack_rows(h + (size_t)nq * 32, t, nt, t_step);
        memcpy(h + rows * 32, h, (size_t)nq * 32);
        PH_HIP(hipMemcpyAsync(c.d_in.p, h, total, hipMemcpyHostToDevice, st));
        if ((s = scratch_guard(c, st)) != PH_OK)
            return s;
        PH_HIP(c.best.reserve(rows * 2 * sizeof(int32_t)));
        int32_t *near_idx = (int32_t *)c.best.p, *near_dist = near_idx + rows;
        const int32_t q_off[3] = {0, nq, (int32_t)rows}, t_off[3] = {0, nt, (int32_t)rows};
        FrameSide qs = {c.d_in.p, q_off, 0, (int64_t)rows, longer};
        FrameSide ts = {(const uint8_t *)c.d_in.p + (size_t)nq * 32, t_off, 0, (int64_t)rows,
                        longer};
        if ((s = knn_frames_locked(c, qs, ts, 2, nq == nt, 1, c.part, near_idx, near_dist, nullptr,
                                   0.f, st, nullptr)) != PH_OK)
            return s;
        PH_HIP(crosscheck_resolve_launch(near_idx, near_dist, near_idx + nq, nq, d_idx, d_dist, st));
    } else {
        const void *d_q = nullptr, *d_t = nullptr;
        if ((s = upload_descriptors(c, q, nq, q_step, t, nt, t_step, st, &d_q, &d_t)) != PH_OK)
            return s;
        if (mode == MatchMode::CrossCheck)
            s = crosscheck_dev_locked(c, d_q, nq, d_t, nt, d_idx, d_dist, st);
        else
            s = knn_dev_locked(c, d_q, nq, d_t, nt, k, 0, d_idx, d_dist,
                               mode == MatchMode::Ratio ? d_keep : nullptr, ratio, c.part, st);
    }
    if (s != PH_OK)
        return s;
    PH_HIP(hipStreamSynchronize(st));
    r.idx = (int32_t *)hb;
    r.dist = (int32_t *)(hb + dist_off);
    r.keep = hb + keep_off;
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

// ---- batches of frames -----------------------------------------------------------
struct FrameBatch {
    int64_t max_nq = 0, max_nt = 0;
    bool uniform = true; // every frame has max_nq query rows and max_nt train rows
    int64_t q_begin = 0, q_end = 0, t_begin = 0, t_end = 0;
    const int32_t *d_q_off = nullptr, *d_t_off = nullptr; // device copies of the tables
};

ph_status check_frames(const int32_t *q_offsets, const int32_t *t_offsets, int n_frames,
                       FrameBatch &fb)
{
    if (n_frames < 0 || n_frames > 65535)
        return fail(PH_ERR_INVALID_ARG, "n_frames must be in 0..65535 (got %d)", n_frames);
    if (n_frames == 0)
        return PH_OK;
    if (!q_offsets || !t_offsets)
        return fail(PH_ERR_INVALID_ARG, "null offset table");
    if (q_offsets[0] < 0 || t_offsets[0] < 0)
        return fail(PH_ERR_INVALID_ARG, "negative offset");
    for (int f = 0; f < n_frames; f++) {
        const int64_t a = (int64_t)q_offsets[f + 1] - q_offsets[f];
        const int64_t b = (int64_t)t_offsets[f + 1] - t_offsets[f];
        if (a < 0 || b < 0)
            return fail(PH_ERR_INVALID_ARG, "offsets of frame %d are not ascending", f);
        fb.max_nq = std::max(fb.max_nq, a);
        fb.max_nt = std::max(fb.max_nt, b);
    }
    for (int f = 0; f < n_frames; f++)
        if (q_offsets[f + 1] - q_offsets[f] != fb.max_nq || t_offsets[f + 1] - t_offsets[f] != fb.max_nt)
            fb.uniform = false;
    fb.q_begin = q_offsets[0];
    fb.q_end = q_offsets[n_frames];
    fb.t_begin = t_offsets[0];
    fb.t_end = t_offsets[n_frames];
    return PH_OK;
}

// Offset tables: host -> a pinned ring slot -> device, in stream order.  The caller
// records slot->done on the stream behind the last kernel that reads the tables.
ph_status upload_frame_tables(Context &c, const int32_t *q_offsets, const int32_t *t_offsets, int n_frames,
                              hipStream_t st, FrameBatch &fb, Context::OffsetSlot *&slot)
{
    const size_t tab_bytes = (size_t)(n_frames + 1) * sizeof(int32_t);
    if (c.off_cached >= 0 && c.off_cached_frames == n_frames) {
        // the same batch layout as the last call (a steady stream of equal batches): the
        // device copy is still valid and later work on it is ordered by the stream
        Context::OffsetSlot &cached = c.off_ring[c.off_cached];
        if (memcmp(cached.h.p, q_offsets, tab_bytes) == 0 &&
            memcmp((const uint8_t *)cached.h.p + tab_bytes, t_offsets, tab_bytes) == 0) {
            slot = &cached;
            fb.d_q_off = (const int32_t *)cached.d.p;
            fb.d_t_off = fb.d_q_off + (n_frames + 1);
            return PH_OK;
        }
    }
    c.off_cached = -1;
    slot = &c.off_ring[c.off_next];
    const int slot_index = c.off_next;
    c.off_next = (c.off_next + 1) % 4;
    if (!slot->done)
        PH_HIP(hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
    else
        PH_HIP(hipEventSynchronize(slot->done));
    const size_t tab = (size_t)(n_frames + 1) * sizeof(int32_t);
    PH_HIP(slot->h.reserve(2 * tab));
    PH_HIP(slot->d.reserve(2 * tab));
    memcpy(slot->h.p, q_offsets, tab);
    memcpy((uint8_t *)slot->h.p + tab, t_offsets, tab);
    PH_HIP(hipMemcpyAsync(slot->d.p, slot->h.p, 2 * tab, hipMemcpyHostToDevice, st));
    fb.d_q_off = (const int32_t *)slot->d.p;
    fb.d_t_off = fb.d_q_off + (n_frames + 1);
    c.off_cached = slot_index;
    c.off_cached_frames = n_frames;
    return PH_OK;
}

// k-NN of every frame's q rows among its t rows, into outputs indexed by q row; the
// caller holds the context's lock.  Picks the matrix-core or the popcount kernel by size.
ph_status knn_frames_locked(Context &c, const FrameSide &q, const FrameSide &t, int n_frames, bool uniform,
                            int k, DevBuf &part, int32_t *d_out_idx, int32_t *d_out_dist,
                            uint8_t *d_keep, float ratio, hipStream_t st, hipEvent_t *ev,
                            FrameTables *tables)
{
    ph_status s;
    FrameBatch fb; // carries the device tables out of upload_frame_tables
    Context::OffsetSlot *slot = nullptr;
    const bool mfma = c.knn_path == 2 ||
                      (c.knn_path == 0 && knn_mfma_pays_frames(n_frames, q.max_rows, t.max_rows));
    MfmaPlan mp;
    KnnPlan plan;
    if (mfma) {
        mp = knn_mfma_plan_frames(n_frames, q.max_rows, t.max_rows, q.end, c.n_cus);
    } else {
        plan = knn_plan_frames(n_frames, q.max_rows, t.max_rows, q.end, c.n_cus);
        if (frame_rows_per_split(t.max_rows, plan) * (int64_t)plan.n_splits < t.max_rows)
            return fail(PH_ERR_TOO_LARGE, "set of %lld rows is too large for one call",
                        (long long)t.max_rows);
    }
    if ((s = scratch_guard(c, st)) != PH_OK)
        return s;
    PH_HIP(part.reserve(mfma ? mp.part_bytes : plan.part_bytes));
    if (mfma) {
        PH_HIP(c.xt.reserve(knn_mfma_expanded_bytes(t.end - t.begin)));
    }
    if ((s = upload_frame_tables(c, q.offsets, t.offsets, n_frames, st, fb, slot)) != PH_OK)
        return s;
    if (mfma)
        PH_HIP(knn_mfma_frames_launch(mp, q.d_rows, fb.d_q_off, q.end, t.d_rows,
                                      fb.d_t_off, t.begin, t.end, n_frames, k, c.xt.p,
                                      (uint32_t *)part.p, d_out_idx, d_out_dist, d_keep, ratio,
                                      st, ev, uniform));
    else
        PH_HIP(knn_frames_launch(plan, q.d_rows, fb.d_q_off, t.d_rows, fb.d_t_off, n_frames, q.end,
                                 k, (uint32_t *)part.p, d_out_idx, d_out_dist, d_keep, ratio, st,
                                 ev, uniform));
    PH_HIP(hipEventRecord(slot->done, st));
    c.last_path = mfma ? 1 : 0;
    c.ev_valid = ev != nullptr; // the events, if any, belong to this call's kernels
    if (tables)
        *tables = FrameTables{fb.d_q_off, fb.d_t_off, slot};
    return PH_OK;
}


ph_status check_pnp_args(const void *obj, const void *img, int n, const double *K)
{
    if (n < 0 || (n > 0 && (!obj || !img)) || !K)
        return fail(PH_ERR_INVALID_ARG, "bad correspondence arrays or camera matrix");
    if (!(K[0] != 0.0) || !(K[4] != 0.0))
        return fail(PH_ERR_INVALID_ARG, "camera matrix has a zero focal length");
    return PH_OK;
}

CameraK camera_of(const double *K) { return CameraK{K[0], K[4], K[2], K[5]}; }

// The hand-over workspace of the two-kernel solve (ph_ransac_set_solve_path(2 | 3) only), for
// calls large enough to take it (more than 64 hypotheses per CU).  It is shared by all
// calls, like the matcher's.
ph_status pnp_front_workspace(Context &c, int64_t n_hyp, hipStream_t st, double *&front)
{
    front = nullptr;
    if (ransac_solve_variant() < 2 || n_hyp <= (int64_t)c.n_cus * 64)
        return PH_OK;
    ph_status s = scratch_guard(c, st);
    if (s != PH_OK)
        return s;
    PH_HIP(c.pnp_front.reserve(ransac_front_workspace_bytes(c.n_cus)));
    front = (double *)c.pnp_front.p;
    return PH_OK;
}

// Releases everything a context holds on its GPU; the caller has made sure that no call is
// in flight and none can start (the context is out of the registry).
void context_teardown(Context &c)
{
    (void)hipSetDevice(c.device);
    (void)hipDeviceSynchronize(); // callers' streams may still hold work that uses the workspace
    DevBuf *all[] = {&c.part,  &c.part2,   &c.best, &c.tmp_i,     &c.tmp_d,
                     &c.d_in,  &c.pnp_idx, &c.xt,   &c.pnp_front, &c.pnp_hyp};
    for (DevBuf *b : all)
        b->release();
    c.h_in.release();
    c.h_out.release();
    for (Context::OffsetSlot &slot : c.off_ring) {
        slot.h.release();
        slot.d.release();
        if (slot.done)
            (void)hipEventDestroy(slot.done);
    }
    if (c.scratch_ev)
        (void)hipEventDestroy(c.scratch_ev);
    if (c.stream)
        (void)hipStreamDestroy(c.stream);
    if (c.pnp_overlap.side)
        (void)hipStreamDestroy(c.pnp_overlap.side);
    if (c.pnp_overlap.solved)
        (void)hipEventDestroy(c.pnp_overlap.solved);
    if (c.pnp_overlap.scored)
        (void)hipEventDestroy(c.pnp_overlap.scored);
    for (hipEvent_t e : c.ev)
        if (e)
            (void)hipEventDestroy(e);
    for (hipEvent_t e : c.pnp_ev)
        if (e)
            (void)hipEventDestroy(e);
}

// A context that has just been taken out of g_live: wait for the calls that had picked it,
// free the train sets it owns and everything else, delete it.
void context_destroy_unregistered(Context *c)
{
    while (c->users.load() > 0)
        std::this_thread::yield();
    std::vector<ph_db *> mine;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (ph_db *db : g_dbs)
            if (db->owner == c)
                mine.push_back(db);
        for (ph_db *db : mine)
            g_dbs.erase(db);
    }
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize(); // queued work of any context on this GPU may still read them
    for (ph_db *db : mine) {
        (void)hipFree(db->d);
        (void)hipFree(db->x);
        delete db;
    }
    context_teardown(*c);
    delete c;
}

// A new context on `device`, registered as live.
ph_status context_create(int device, Context *&out)
{
    out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(PH_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fail(PH_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, n - 1);
    if (device >= MAX_DEVICES)
        return fail(PH_ERR_INVALID_ARG, "device %d beyond the %d this library tracks", device,
                    MAX_DEVICES);
    PH_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PH_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PH_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                    device, prop.gcnArchName);
    {
        // this device's copy of `scratch` (the device is current): looked up once per device
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (!g_noalloc_ptr[device]) {
            void *p = nullptr;
            PH_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            g_noalloc_ptr[device] = (unsigned char *)p;
        }
    }
    Context *c = new Context();
    c->device = device;
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (err == hipSuccess)
        err = hipStreamCreateWithFlags(&c->pnp_overlap.side, hipStreamNonBlocking);
    if (err == hipSuccess)
        err = hipEventCreateWithFlags(&c->pnp_overlap.solved, hipEventDisableTiming);
    if (err == hipSuccess)
        err = hipEventCreateWithFlags(&c->pnp_overlap.scored, hipEventDisableTiming);
    if (err != hipSuccess) {
        context_teardown(*c);
        delete c;
        return fail(PH_ERR_HIP, "creating the streams of a context on device %d: %s", device,
                    hipGetErrorString(err));
    }
    c->ready = true;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_live.insert(c);
    out = c;
    return PH_OK;
}

} // namespace

extern "C" {

ph_status ph_init(int device)
{
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (g_default) {
            if (g_default->device == device)
                return PH_OK;
            return fail(PH_ERR_INVALID_ARG,
                        "already bound to device %d; call ph_shutdown() first, or bind further "
                        "GPUs with ph_ctx_create()", g_default->device);
        }
    }
    Context *c = nullptr;
    ph_status s = context_create(device, c);
    if (s != PH_OK)
        return s;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (g_default) { // another thread's ph_init got there first
        Context *mine = c;
        g_live.erase(mine);
        // (no call can have picked `mine`: it was never current anywhere)
        context_teardown(*mine);
        delete mine;
        return g_default->device == device
                   ? PH_OK
                   : fail(PH_ERR_INVALID_ARG, "already bound to device %d", g_default->device);
    }
    c->knn_path = g_pending_knn_path;
    if (const char *path = getenv("POSEHIP_KNN_PATH")) { // same values as ph_knn_set_path
        if (!strcmp(path, "popcount") || !strcmp(path, "1"))
            c->knn_path = 1;
        else if (!strcmp(path, "matrix") || !strcmp(path, "2"))
            c->knn_path = 2;
    }
    g_default = c;
    return PH_OK;
}

void ph_shutdown(void)
{
    Context *c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        c = g_default;
        g_default = nullptr;
        g_pending_knn_path = 0;
        if (c)
            g_live.erase(c);
    }
    if (!c)
        return;
    context_destroy_unregistered(c);
    ransac_set_solve_variant(0); // a measurement setting, like the k-NN path: does not outlive the binding
}

ph_status ph_ctx_create(int device, ph_ctx **ctx)
{
    if (!ctx)
        return fail(PH_ERR_INVALID_ARG, "ctx is null");
    *ctx = nullptr;
    Context *c = nullptr;
    ph_status s = context_create(device, c);
    if (s != PH_OK)
        return s;
    if (const char *path = getenv("POSEHIP_KNN_PATH")) {
        if (!strcmp(path, "popcount") || !strcmp(path, "1"))
            c->knn_path = 1;
        else if (!strcmp(path, "matrix") || !strcmp(path, "2"))
            c->knn_path = 2;
    }
    *ctx = c;
    return PH_OK;
}

ph_status ph_ctx_destroy(ph_ctx *ctx)
{
    if (!ctx)
        return PH_OK;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (!g_live.count(ctx))
            return fail(PH_ERR_INVALID_ARG, "not a live context");
        if (ctx == g_default)
            return fail(PH_ERR_INVALID_ARG, "the default context is released by ph_shutdown()");
        g_live.erase(ctx); // from here on no new call can pick it
    }
    if (t_current == ctx)
        t_current = nullptr;
    context_destroy_unregistered(ctx);
    return PH_OK;
}

ph_status ph_ctx_make_current(ph_ctx *ctx)
{
    if (ctx) {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (!g_live.count(ctx))
            return fail(PH_ERR_INVALID_ARG, "not a live context");
    }
    t_current = ctx;
    return PH_OK;
}

ph_ctx *ph_ctx_current(void) { return t_current; }

int ph_ctx_device(const ph_ctx *ctx)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return ctx && g_live.count(const_cast<ph_ctx *>(ctx)) ? ctx->device : -1;
}

const char *ph_last_error_string(void) { return g_err; }

int ph_device(void)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (t_current) // a context that has been destroyed under this thread serves no call: -1
        return g_live.count(t_current) ? t_current->device : -1;
    return g_default ? g_default->device : -1;
}

ph_status ph_stream_synchronize(void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    PH_HIP(hipStreamSynchronize(pick_stream(c, stream)));
    return PH_OK;
}

ph_status ph_set_profiling(int enable)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (enable && !c.ev[0])
        for (hipEvent_t &e : c.ev)
            PH_HIP(hipEventCreate(&e));
    if (enable && !c.pnp_ev[0])
        for (hipEvent_t &e : c.pnp_ev)
            PH_HIP(hipEventCreate(&e));
    c.profiling = enable != 0;
    c.ev_valid = false;
    c.pnp_ev_valid = false;
    return PH_OK;
}

ph_status ph_knn_set_path(int mode)
{
    if (mode < 0 || mode > 2)
        return fail(PH_ERR_INVALID_ARG, "k-NN path must be 0 (by size), 1 (popcount) or 2 (matrix cores)");
    Context *c = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        c = t_current && g_live.count(t_current) ? t_current : g_default;
        if (!c) { // before ph_init: kept for the default context
            g_pending_knn_path = mode;
            return PH_OK;
        }
        c->users.fetch_add(1);
    }
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->knn_path = mode;
    }
    c->users.fetch_sub(1);
    return PH_OK;
}

int ph_knn_last_path(void)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    Context *c = t_current && g_live.count(t_current) ? t_current : g_default;
    return c ? c->last_path : 0;
}

ph_status ph_knn_last_timing_overhead(float *empty_interval_ms)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (!c.ev_valid || !empty_interval_ms)
        return fail(PH_ERR_INVALID_ARG, "no profiled k-NN call (ph_set_profiling(1) first)");
    PH_HIP(hipEventSynchronize(c.ev[3]));
    PH_HIP(hipEventElapsedTime(empty_interval_ms, c.ev[2], c.ev[3]));
    return PH_OK;
}

ph_status ph_knn_last_prepass_timing(float *prepass_ms)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (!c.ev_valid || !prepass_ms)
        return fail(PH_ERR_INVALID_ARG, "no profiled k-NN call (ph_set_profiling(1) first)");
    *prepass_ms = 0.f;
    if (c.last_path == 1) {
        PH_HIP(hipEventSynchronize(c.ev[0]));
        PH_HIP(hipEventElapsedTime(prepass_ms, c.ev[4], c.ev[0]));
    }
    return PH_OK;
}

ph_status ph_knn_last_timing(float *distance_kernel_ms, float *merge_kernel_ms)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (!c.ev_valid)
        return fail(PH_ERR_INVALID_ARG, "no profiled k-NN call (ph_set_profiling(1) first)");
    PH_HIP(hipEventSynchronize(c.ev[2]));
    if (distance_kernel_ms)
        PH_HIP(hipEventElapsedTime(distance_kernel_ms, c.ev[0], c.ev[1]));
    if (merge_kernel_ms)
        PH_HIP(hipEventElapsedTime(merge_kernel_ms, c.ev[1], c.ev[2]));
    return PH_OK;
}

ph_status ph_valu_popcount_peak(int iters, double *lane_ops_per_s)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (iters < 1 || !lane_ops_per_s)
        return fail(PH_ERR_INVALID_ARG, "iters must be >= 1 and the output non-null");
    if ((s = scratch_guard(c, c.stream)) != PH_OK) // tmp_i is cross-check workspace as well
        return s;
    PH_HIP(c.tmp_i.reserve((size_t)c.n_cus * 8 * 256 * sizeof(uint32_t)));
    PH_HIP(valu_popcount_peak(c.n_cus, iters, (uint32_t *)c.tmp_i.p, c.stream,
                              lane_ops_per_s));
    return PH_OK;
}

ph_status ph_valu_fp64_peak(int iters, double *fma_lane_ops_per_s,
                            double *mul_add_lane_ops_per_s)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (iters < 1 || !fma_lane_ops_per_s || !mul_add_lane_ops_per_s)
        return fail(PH_ERR_INVALID_ARG, "iters must be >= 1 and the outputs non-null");
    if ((s = scratch_guard(c, c.stream)) != PH_OK) // tmp_d is cross-check workspace as well
        return s;
    PH_HIP(c.tmp_d.reserve((size_t)c.n_cus * 8 * 256 * sizeof(double)));
    PH_HIP(valu_fp64_peak(c.n_cus, iters, (double *)c.tmp_d.p, c.stream,
                          fma_lane_ops_per_s, mul_add_lane_ops_per_s));
    return PH_OK;
}

ph_status ph_mfma_fp4_peak(int iters, double *flop_per_s)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    if (iters < 1 || !flop_per_s)
        return fail(PH_ERR_INVALID_ARG, "iters must be >= 1 and the output non-null");
    if ((s = scratch_guard(c, c.stream)) != PH_OK) // tmp_i is cross-check workspace as well
        return s;
    PH_HIP(c.tmp_i.reserve((size_t)c.n_cus * 2 * 256 * sizeof(float)));
    PH_HIP(mfma_fp4_peak(c.n_cus, iters, (float *)c.tmp_i.p, c.stream, flop_per_s));
    return PH_OK;
}

ph_status ph_ransac_set_solve_path(int mode)
{
    if (mode < 0 || mode > 3)
        return fail(PH_ERR_INVALID_ARG, "solve path must be 0 (default), 1, 2 or 3");
    ransac_set_solve_variant(mode);
    return PH_OK;
}

ph_status ph_ransac_last_timing(float *first_solve_ms, float *last_solve_ms, float *last_score_ms)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (!c.pnp_ev_valid)
        return fail(PH_ERR_INVALID_ARG,
                    "no profiled RANSAC hypotheses call (ph_set_profiling(1) first)");
    PH_HIP(hipEventSynchronize(c.pnp_ev[4]));
    if (first_solve_ms)
        PH_HIP(hipEventElapsedTime(first_solve_ms, c.pnp_ev[0], c.pnp_ev[1]));
    if (last_solve_ms)
        PH_HIP(hipEventElapsedTime(last_solve_ms, c.pnp_ev[2], c.pnp_ev[3]));
    if (last_score_ms)
        PH_HIP(hipEventElapsedTime(last_score_ms, c.pnp_ev[3], c.pnp_ev[4]));
    return PH_OK;
}

ph_status ph_knn_hamming256_dev(const void *d_query, int64_t nq, const void *d_train, int64_t nt,
                                int k, int32_t train_index_base, int32_t *d_out_idx,
                                int32_t *d_out_dist, uint8_t *d_keep, float ratio, void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
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
    return knn_dev_locked(c, d_query, nq, d_train, nt, k, train_index_base, d_out_idx, d_out_dist,
                          d_keep, ratio, c.part, pick_stream(c, stream));
}

ph_status ph_knn_hamming256_noalloc_dev(const void *d_query, int64_t nq, const void *d_train,
                                        int64_t nt, int k, int32_t train_index_base,
                                        int32_t *d_out_idx, int32_t *d_out_dist, uint8_t *d_keep,
                                        float ratio, void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
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
    // the device address of this GPU's copy of `scratch` (never the array's name)
    unsigned char *ws = g_noalloc_ptr[c.device];
    if (!ws)
        return fail(PH_ERR_NOT_INITIALIZED, "no scratch address for device %d", c.device);
    hipStream_t st = pick_stream(c, stream);
    if (nq == 0)
        return PH_OK;
    if (nt == 0) {
        fill_empty_kernel<<<(unsigned)((nq + 255) / 256), 256, 0, st>>>(d_out_idx, d_out_dist,
                                                                        d_keep, nq, k);
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    if (((uintptr_t)d_query & 15) || ((uintptr_t)d_train & 15))
        return fail(PH_ERR_INVALID_ARG, "descriptor buffers must be 16-byte aligned");
    if ((int64_t)train_index_base + nt - 1 > 0x7FFFFFFF)
        return fail(PH_ERR_TOO_LARGE, "train_index_base + nt exceeds the 32-bit index range");
    // Same kernels and plans as ph_knn_hamming256_dev; every word of scratch a call reads it
    // has written before (the pre-pass writes the image, the distance kernel every key the
    // merge folds), so nothing has to start at zero and the last call's contents do not matter.
    if (c.knn_path == 2 || (c.knn_path == 0 && knn_mfma_pays(nq, nt))) {
        const MfmaPlan mp = knn_mfma_plan(nq, nt, c.n_cus);
        const size_t image = align256(knn_mfma_expanded_bytes(nt)), need = image + mp.part_bytes;
        if (need > PH_NOALLOC_SCRATCH_BYTES)
            return fail(PH_ERR_TOO_LARGE,
                        "%lld x %lld needs %zu bytes of scratch; the allocation-free entry point "
                        "holds %zu (use ph_knn_hamming256_dev)",
                        (long long)nq, (long long)nt, need, (size_t)PH_NOALLOC_SCRATCH_BYTES);
        PH_HIP(knn_mfma_launch(mp, d_query, nq, d_train, nt, k, train_index_base, ws,
                               (uint32_t *)(ws + image), d_out_idx, d_out_dist, d_keep, ratio, st,
                               nullptr));
        c.last_path = 1;
        c.ev_valid = false; // no events recorded: an earlier call's must not be read as this one's
        return PH_OK;
    }
    const KnnPlan plan = knn_plan(nq, nt, c.n_cus);
    if (plan.rows_per_split * (int64_t)plan.n_splits < nt)
        return fail(PH_ERR_TOO_LARGE, "train set of %lld rows is too large for one call",
                    (long long)nt);
    if (plan.part_bytes > PH_NOALLOC_SCRATCH_BYTES)
        return fail(PH_ERR_TOO_LARGE,
                    "%lld x %lld needs %zu bytes of scratch; the allocation-free entry point "
                    "holds %zu (use ph_knn_hamming256_dev)",
                    (long long)nq, (long long)nt, plan.part_bytes,
                    (size_t)PH_NOALLOC_SCRATCH_BYTES);
    PH_HIP(knn_launch(plan, d_query, nq, d_train, nt, k, train_index_base, (uint32_t *)ws,
                      d_out_idx, d_out_dist, d_keep, ratio, st, nullptr));
    c.last_path = 0;
    c.ev_valid = false;
    return PH_OK;
}

ph_status ph_knn_hamming256_frames_dev(const void *d_query, const int32_t *q_offsets,
                                       const void *d_train, const int32_t *t_offsets,
                                       int n_frames, int k, int32_t *d_out_idx,
                                       int32_t *d_out_dist, uint8_t *d_keep, float ratio,
                                       void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (k != 1 && k != 2)
        return fail(PH_ERR_INVALID_ARG, "k must be 1 or 2 (got %d)", k);
    if (d_keep && k != 2)
        return fail(PH_ERR_INVALID_ARG, "the ratio test needs k == 2");
    FrameBatch fb;
    if ((s = check_frames(q_offsets, t_offsets, n_frames, fb)) != PH_OK)
        return s;
    if (fb.q_end == fb.q_begin)
        return PH_OK;
    if (!d_query || !d_out_idx || !d_out_dist || (fb.max_nt > 0 && !d_train))
        return fail(PH_ERR_INVALID_ARG, "null pointer");
    hipStream_t st = pick_stream(c, stream);
    if (fb.max_nt == 0) {
        const int64_t rows = fb.q_end - fb.q_begin;
        fill_empty_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, st>>>(
            d_out_idx + fb.q_begin * k, d_out_dist + fb.q_begin * k,
            d_keep ? d_keep + fb.q_begin : nullptr, rows, k);
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    if (((uintptr_t)d_query & 15) || ((uintptr_t)d_train & 15))
        return fail(PH_ERR_INVALID_ARG, "descriptor buffers must be 16-byte aligned");
    hipEvent_t *ev = c.profiling ? c.ev : nullptr;
    FrameSide q = {d_query, q_offsets, fb.q_begin, fb.q_end, fb.max_nq};
    FrameSide t = {d_train, t_offsets, fb.t_begin, fb.t_end, fb.max_nt};
    if ((s = knn_frames_locked(c, q, t, n_frames, fb.uniform, k, c.part, d_out_idx, d_out_dist,
                               d_keep, ratio, st, ev)) != PH_OK)
        return s;
    return PH_OK;
}

ph_status ph_match_crosscheck_hamming256_frames_dev(const void *d_query, const int32_t *q_offsets,
                                                    const void *d_train, const int32_t *t_offsets,
                                                    int n_frames, int32_t *d_out_idx,
                                                    int32_t *d_out_dist, void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    FrameBatch fb;
    if ((s = check_frames(q_offsets, t_offsets, n_frames, fb)) != PH_OK)
        return s;
    if (fb.q_end == fb.q_begin)
        return PH_OK;
    if (!d_query || !d_out_idx || !d_out_dist || (fb.max_nt > 0 && !d_train))
        return fail(PH_ERR_INVALID_ARG, "null pointer");
    hipStream_t st = pick_stream(c, stream);
    if (fb.max_nt == 0) {
        const int64_t rows = fb.q_end - fb.q_begin;
        fill_empty_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, st>>>(
            d_out_idx + fb.q_begin, d_out_dist + fb.q_begin, nullptr, rows, 1);
        PH_HIP(hipGetLastError());
        return PH_OK;
    }
    if (((uintptr_t)d_query & 15) || ((uintptr_t)d_train & 15))
        return fail(PH_ERR_INVALID_ARG, "descriptor buffers must be 16-byte aligned");
    if ((s = scratch_guard(c, st)) != PH_OK)
        return s;
    PH_HIP(c.tmp_i.reserve((size_t)fb.t_end * sizeof(int32_t)));
    PH_HIP(c.tmp_d.reserve((size_t)fb.t_end * sizeof(int32_t)));
    PH_HIP(c.best.reserve((size_t)fb.q_end * 2 * sizeof(int32_t)));
    int32_t *sidx = (int32_t *)c.best.p, *sdist = sidx + fb.q_end;
    FrameTables tables;
    FrameSide q = {d_query, q_offsets, fb.q_begin, fb.q_end, fb.max_nq};
    FrameSide t = {d_train, t_offsets, fb.t_begin, fb.t_end, fb.max_nt};
    // every query row's nearest train row of its own frame ...
    if ((s = knn_frames_locked(c, q, t, n_frames, fb.uniform, 1, c.part, sidx, sdist, nullptr, 0.f,
                               st, nullptr)) != PH_OK)
        return s;
    // ... and, roles swapped, every train row's nearest query row (so the device tables
    // come back as tables.d_q_off = train offsets, d_t_off = query offsets)
    if ((s = knn_frames_locked(c, t, q, n_frames, fb.uniform, 1, c.part2,
                               (int32_t *)c.tmp_i.p, (int32_t *)c.tmp_d.p, nullptr, 0.f,
                               st, nullptr, &tables)) != PH_OK)
        return s;
    // (the tables' ring slot stays busy until the resolve kernel below has run: same
    // stream, and the next call that reuses the slot waits for its event)
    PH_HIP(crosscheck_resolve_frames_launch(sidx, sdist, (const int32_t *)c.tmp_i.p,
                                            tables.d_t_off, tables.d_q_off, n_frames, fb.max_nq,
                                            d_out_idx, d_out_dist, st));
    PH_HIP(hipEventRecord(tables.slot->done, st));
    return PH_OK;
}

ph_status ph_knn_merge_dev(const int32_t *d_cand_idx, const int32_t *d_cand_dist, int n_shards,
                           int64_t nq, int k, int32_t *d_out_idx, int32_t *d_out_dist,
                           uint8_t *d_keep, float ratio, void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
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
                                   d_keep, ratio, pick_stream(c, stream)));
    return PH_OK;
}

ph_status ph_match_crosscheck_hamming256_dev(const void *d_query, int64_t nq, const void *d_train,
                                             int64_t nt, int32_t *d_out_idx, int32_t *d_out_dist,
                                             void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if ((s = check_rows(d_query, nq, "query")) != PH_OK)
        return s;
    if ((s = check_rows(d_train, nt, "train")) != PH_OK)
        return s;
    if (nq > 0 && (!d_out_idx || !d_out_dist))
        return fail(PH_ERR_INVALID_ARG, "null output pointer");
    return crosscheck_dev_locked(c, d_query, nq, d_train, nt, d_out_idx, d_out_dist,
                                 pick_stream(c, stream));
}

ph_status ph_match_crosscheck_hamming256_shard_dev(const void *d_query, int64_t nq,
                                                   const void *d_train_shard, int64_t nt,
                                                   int32_t train_index_base, int32_t *d_out_idx,
                                                   int32_t *d_out_dist, void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if ((s = check_rows(d_query, nq, "query")) != PH_OK)
        return s;
    if ((s = check_rows(d_train_shard, nt, "train")) != PH_OK)
        return s;
    if (nq > 0 && (!d_out_idx || !d_out_dist))
        return fail(PH_ERR_INVALID_ARG, "null output pointer");
    if (train_index_base < 0)
        return fail(PH_ERR_INVALID_ARG, "negative train_index_base");
    if (nt > 0 && (int64_t)train_index_base + nt - 1 > 0x3FFFFFFF)
        return fail(PH_ERR_TOO_LARGE, "train_index_base + nt exceeds 2^30 rows (the packed "
                                      "candidates carry a flag below the index)");
    return crosscheck_dev_locked(c, d_query, nq, d_train_shard, nt, d_out_idx, d_out_dist,
                                 pick_stream(c, stream), train_index_base, true);
}

ph_status ph_match_crosscheck_unpack_dev(int32_t *d_idx, int32_t *d_dist, int64_t nq, void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (nq < 0 || (nq > 0 && (!d_idx || !d_dist)))
        return fail(PH_ERR_INVALID_ARG, "bad candidate arrays");
    PH_HIP(crosscheck_unpack_launch(d_idx, d_dist, nq, pick_stream(c, stream)));
    return PH_OK;
}

ph_status ph_knn_hamming256(const uint8_t *query, int nq, size_t q_step, const uint8_t *train,
                            int nt, size_t t_step, int k, int32_t *out_idx, int32_t *out_dist)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (k != 1 && k != 2)
        return fail(PH_ERR_INVALID_ARG, "k must be 1 or 2 (got %d)", k);
    if ((s = check_host_args(query, nq, q_step, train, nt, t_step)) != PH_OK)
        return s;
    if (nq == 0)
        return PH_OK;
    if (!out_idx || !out_dist)
        return fail(PH_ERR_INVALID_ARG, "null output pointer");
    HostResult r;
    DrainOnError drain{c};
    if ((s = host_match_locked(c, query, nq, q_step, train, nt, t_step, MatchMode::Knn, k, 0.f, r)) !=
        PH_OK)
        return s;
    drain.armed = false;
    memcpy(out_idx, r.idx, (size_t)nq * k * sizeof(int32_t));
    memcpy(out_dist, r.dist, (size_t)nq * k * sizeof(int32_t));
    return PH_OK;
}

namespace {
// Shared tail of the three match() flavours: one DMatch per kept query, ascending.
ph_status host_match_list(const uint8_t *query, int nq, size_t q_step, const uint8_t *train, int nt,
                          size_t t_step, MatchMode mode, float ratio, ph_dmatch *out, int *n_out)
{
    if (!n_out)
        return fail(PH_ERR_INVALID_ARG, "n_out is null");
    *n_out = 0;
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if ((s = check_host_args(query, nq, q_step, train, nt, t_step)) != PH_OK)
        return s;
    if (nq == 0 || nt == 0)
        return PH_OK; // OpenCV clears the match list
    if (!out)
        return fail(PH_ERR_INVALID_ARG, "null output pointer");
    const int k = mode == MatchMode::Ratio ? 2 : 1;
    HostResult r;
    DrainOnError drain{c};
    if ((s = host_match_locked(c, query, nq, q_step, train, nt, t_step, mode, k, ratio, r)) != PH_OK)
        return s;
    drain.armed = false;
    int n = 0;
    for (int i = 0; i < nq; i++) {
        const bool kept = mode == MatchMode::Ratio ? r.keep[i] != 0 : r.idx[(size_t)i * k] >= 0;
        if (!kept)
            continue;
        out[n].queryIdx = i;
        out[n].trainIdx = r.idx[(size_t)i * k];
        out[n].imgIdx = 0;
        out[n].distance = (float)r.dist[(size_t)i * k];
        n++;
    }
    *n_out = n;
    return PH_OK;
}
} // namespace

ph_status ph_match_hamming256(const uint8_t *query, int nq, size_t q_step, const uint8_t *train,
                              int nt, size_t t_step, ph_dmatch *out, int *n_out)
{
    return host_match_list(query, nq, q_step, train, nt, t_step, MatchMode::Knn, 0.f, out, n_out);
}

ph_status ph_match_crosscheck_hamming256(const uint8_t *query, int nq, size_t q_step,
                                         const uint8_t *train, int nt, size_t t_step,
                                         ph_dmatch *out, int *n_out)
{
    return host_match_list(query, nq, q_step, train, nt, t_step, MatchMode::CrossCheck, 0.f, out,
                           n_out);
}

ph_status ph_knn2_ratio_hamming256(const uint8_t *query, int nq, size_t q_step,
                                   const uint8_t *train, int nt, size_t t_step, float ratio,
                                   ph_dmatch *out, int *n_out)
{
    return host_match_list(query, nq, q_step, train, nt, t_step, MatchMode::Ratio, ratio, out,
                           n_out);
}

ph_status ph_ratio_filter(const int32_t *idx, const int32_t *dist, int nq, float ratio,
                          ph_dmatch *out, int *n_out)
{
    if (!n_out)
        return fail(PH_ERR_INVALID_ARG, "n_out is null");
    *n_out = 0;
    if (nq < 0 || (nq > 0 && (!idx || !dist || !out)))
        return fail(PH_ERR_INVALID_ARG, "bad k-NN arrays");
    int n = 0;
    for (int i = 0; i < nq; i++) {
        const int32_t i1 = idx[2 * (size_t)i], i2 = idx[2 * (size_t)i + 1];
        if (i1 < 0 || i2 < 0)
            continue; // fewer than two neighbours: nothing to compare with
        if ((float)dist[2 * (size_t)i] < ratio * (float)dist[2 * (size_t)i + 1]) {
            out[n].queryIdx = i;
            out[n].trainIdx = i1;
            out[n].imgIdx = 0;
            out[n].distance = (float)dist[2 * (size_t)i];
            n++;
        }
    }
    *n_out = n;
    return PH_OK;
}

ph_status ph_db_upload(const uint8_t *train, int64_t nt, size_t t_step, ph_db **db)
{
    if (db)
        *db = nullptr;
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (!db)
        return fail(PH_ERR_INVALID_ARG, "db is null");
    if ((s = check_rows(train, nt, "train")) != PH_OK)
        return s;
    if (nt > 0 && t_step < 32)
        return fail(PH_ERR_INVALID_ARG, "row stride below 32 bytes");
    ph_db *h = new ph_db();
    h->rows = nt;
    h->device = c.device;
    h->owner = &c;
    if (nt > 0) {
        hipError_t e = hipMalloc(&h->d, (size_t)nt * 32);
        if (e != hipSuccess) {
            delete h;
            return fail(PH_ERR_HIP, "hipMalloc of %lld train rows: %s", (long long)nt,
                        hipGetErrorString(e));
        }
        // rows are packed through the pinned gather buffer, 16 MiB at a time
        const int64_t chunk_rows = (16 << 20) / 32;
        e = c.h_in.reserve((size_t)std::min(nt, chunk_rows) * 32);
        for (int64_t r0 = 0; e == hipSuccess && r0 < nt; r0 += chunk_rows) {
            const int64_t m = std::min(chunk_rows, nt - r0);
            pack_rows((uint8_t *)c.h_in.p, train + (size_t)r0 * t_step, m, t_step);
            e = hipMemcpyAsync((uint8_t *)h->d + (size_t)r0 * 32, c.h_in.p, (size_t)m * 32,
                               hipMemcpyHostToDevice, c.stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(c.stream); // the gather buffer is reused
        }
        if (e != hipSuccess) {
            (void)hipFree(h->d);
            delete h;
            return fail(PH_ERR_HIP, "uploading the train set: %s", hipGetErrorString(e));
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_dbs.insert(h);
    }
    *db = h;
    return PH_OK;
}

ph_status ph_db_free(ph_db *db)
{
    if (!db)
        return PH_OK;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (!g_dbs.count(db)) // also: freed already with its context
            return fail(PH_ERR_INVALID_ARG, "not a live train-set handle");
        g_dbs.erase(db); // from here on no new call accepts it
    }
    hipError_t e = hipSetDevice(db->device);
    if (e == hipSuccess)
        e = hipDeviceSynchronize(); // queued work of any context on this GPU may still read it
    (void)hipFree(db->d);
    (void)hipFree(db->x);
    delete db;
    if (e != hipSuccess)
        return fail(PH_ERR_HIP, "ph_db_free: %s", hipGetErrorString(e));
    return PH_OK;
}

int64_t ph_db_rows(const ph_db *db)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return db && g_dbs.count(const_cast<ph_db *>(db)) ? db->rows : -1;
}

const void *ph_db_device_ptr(const ph_db *db)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    return db && g_dbs.count(const_cast<ph_db *>(db)) ? db->d : nullptr;
}

ph_status ph_db_knn_hamming256(const ph_db *db, const uint8_t *query, int64_t nq, size_t q_step,
                               int k, int32_t *out_idx, int32_t *out_dist, uint8_t *keep,
                               float ratio)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (!db_live(db))
        return fail(PH_ERR_INVALID_ARG, "not a live train-set handle");
    if (db->device != c.device)
        return fail(PH_ERR_INVALID_ARG, "the train set lives on device %d, this context on %d",
                    db->device, c.device);
    if (k != 1 && k != 2)
        return fail(PH_ERR_INVALID_ARG, "k must be 1 or 2 (got %d)", k);
    if (keep && k != 2)
        return fail(PH_ERR_INVALID_ARG, "the ratio test needs k == 2");
    if ((s = check_rows(query, nq, "query")) != PH_OK)
        return s;
    if (nq > 0 && q_step < 32)
        return fail(PH_ERR_INVALID_ARG, "row stride below 32 bytes");
    if (nq == 0)
        return PH_OK;
    if (!out_idx || !out_dist)
        return fail(PH_ERR_INVALID_ARG, "null output pointer");
    hipStream_t st = c.stream;
    const void *d_q = nullptr, *d_unused = nullptr;
    DrainOnError drain{c};
    if ((s = upload_descriptors(c, query, nq, q_step, nullptr, 0, 32, st, &d_q, &d_unused)) != PH_OK)
        return s;
    const size_t n_out = (size_t)nq * k;
    const size_t dist_off = align256(n_out * 4), keep_off = dist_off + align256(n_out * 4);
    PH_HIP(c.h_out.reserve(keep_off + (size_t)nq));
    uint8_t *hb = (uint8_t *)c.h_out.p, *dbuf = (uint8_t *)c.h_out.dp;
    if ((s = knn_dev_locked(c, d_q, nq, db->d, db->rows, k, 0, (int32_t *)dbuf,
                            (int32_t *)(dbuf + dist_off), keep ? dbuf + keep_off : nullptr, ratio,
                            c.part, st, const_cast<ph_db *>(db))) != PH_OK)
        return s;
    PH_HIP(hipStreamSynchronize(st));
    drain.armed = false;
    memcpy(out_idx, hb, n_out * sizeof(int32_t));
    memcpy(out_dist, hb + dist_off, n_out * sizeof(int32_t));
    if (keep)
        memcpy(keep, hb + keep_off, (size_t)nq);
    return PH_OK;
}

ph_status ph_db_knn_hamming256_dev(const ph_db *db, const void *d_query, int64_t nq, int k,
                                   int32_t *d_out_idx, int32_t *d_out_dist, uint8_t *d_keep,
                                   float ratio, void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (!db_live(db))
        return fail(PH_ERR_INVALID_ARG, "not a live train-set handle");
    if (db->device != c.device)
        return fail(PH_ERR_INVALID_ARG, "the train set lives on device %d, this context on %d",
                    db->device, c.device);
    if (k != 1 && k != 2)
        return fail(PH_ERR_INVALID_ARG, "k must be 1 or 2 (got %d)", k);
    if (d_keep && k != 2)
        return fail(PH_ERR_INVALID_ARG, "the ratio test needs k == 2");
    if ((s = check_rows(d_query, nq, "query")) != PH_OK)
        return s;
    if (nq > 0 && (!d_out_idx || !d_out_dist))
        return fail(PH_ERR_INVALID_ARG, "null output pointer");
    return knn_dev_locked(c, d_query, nq, db->d, db->rows, k, 0, d_out_idx, d_out_dist, d_keep, ratio,
                          c.part, pick_stream(c, stream), const_cast<ph_db *>(db));
}

ph_status ph_ransac_subsets(uint64_t rng_state, int n_points, int n_subsets, int32_t *out)
{
    if (n_points < 5 || n_subsets < 0 || (n_subsets > 0 && !out))
        return fail(PH_ERR_INVALID_ARG, "need >= 5 points and a non-null output");
    draw_subsets(rng_state, n_points, n_subsets, out);
    return PH_OK;
}

ph_status ph_ransac_epnp_hypotheses_dev(const float *d_obj, const float *d_img, int n,
                                        const double *K, const int32_t *d_subsets, int n_hyp,
                                        float reproj_err, double *d_models, int32_t *d_counts,
                                        void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if ((s = check_pnp_args(d_obj, d_img, n, K)) != PH_OK)
        return s;
    if (n < 5 || n_hyp < 0 || (n_hyp > 0 && (!d_subsets || !d_models || !d_counts)))
        return fail(PH_ERR_INVALID_ARG, "need >= 5 correspondences and non-null buffers");
    const double thr = reproj_err;
    hipEvent_t *ev = c.profiling && n_hyp > 0 ? c.pnp_ev : nullptr;
    double *front = nullptr;
    if ((s = pnp_front_workspace(c, n_hyp, pick_stream(c, stream), front)) != PH_OK)
        return s;
    PH_HIP(ransac_hypotheses_launch(d_obj, d_img, n, camera_of(K), d_subsets, nullptr, nullptr, 1,
                                    n_hyp, (float)(thr * thr), d_models, d_counts, c.n_cus,
                                    pick_stream(c, stream), &c.pnp_overlap, ev, front));
    c.pnp_ev_valid = ev != nullptr;
    return PH_OK;
}

ph_status ph_ransac_epnp_hypotheses_batched_dev(const float *d_obj, const float *d_img,
                                                const int32_t *d_point_offsets, int n_problems,
                                                const double *K, const int32_t *d_subsets,
                                                int hyp_per_problem, float reproj_err,
                                                double *d_models, int32_t *d_counts, void *stream)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if ((s = check_pnp_args(d_obj, d_img, 1, K)) != PH_OK)
        return s;
    if (n_problems < 0 || hyp_per_problem < 0)
        return fail(PH_ERR_INVALID_ARG, "negative problem or hypothesis count");
    if ((int64_t)n_problems * hyp_per_problem > 0x7FFFFFFF)
        return fail(PH_ERR_TOO_LARGE, "more than 2^31-1 hypotheses in one batch");
    if (n_problems == 0 || hyp_per_problem == 0)
        return PH_OK;
    if (!d_point_offsets || !d_subsets || !d_models || !d_counts)
        return fail(PH_ERR_INVALID_ARG, "null pointer");
    const double thr = reproj_err;
    // (n_problems * hyp_per_problem > 0 here: an empty batch has returned above, so the
    // launch below records every event it is given)
    hipEvent_t *ev = c.profiling ? c.pnp_ev : nullptr;
    double *front = nullptr;
    if ((s = pnp_front_workspace(c, (int64_t)n_problems * hyp_per_problem, pick_stream(c, stream),
                                 front)) != PH_OK)
        return s;
    PH_HIP(ransac_hypotheses_launch(d_obj, d_img, 0, camera_of(K), d_subsets, d_point_offsets,
                                    d_point_offsets + 1, n_problems, hyp_per_problem,
                                    (float)(thr * thr), d_models, d_counts, c.n_cus,
                                    pick_stream(c, stream), &c.pnp_overlap, ev, front));
    c.pnp_ev_valid = ev != nullptr;
    return PH_OK;
}

namespace {
// Host correspondences (and, for RANSAC, the subset table) -> device with one copy.
// Layout of c.d_in: obj[n*3] f32 | img[n*2] f32 | subsets[n_sub*5] i32.
struct PnpDev {
    const float *obj, *img;
    const int32_t *subsets;
};

ph_status upload_pnp(Context &c, const float *obj, const float *img, int n, const int32_t *subsets,
                     size_t n_sub, hipStream_t st, PnpDev &d)
{
    const size_t img_off = align256((size_t)n * 12), sub_off = img_off + align256((size_t)n * 8);
    const size_t total = sub_off + n_sub * 5 * sizeof(int32_t);
    PH_HIP(c.h_in.reserve(total));
    PH_HIP(c.d_in.reserve(total));
    uint8_t *h = (uint8_t *)c.h_in.p;
    memcpy(h, obj, (size_t)n * 12);
    memcpy(h + img_off, img, (size_t)n * 8);
    if (n_sub)
        memcpy(h + sub_off, subsets, n_sub * 5 * sizeof(int32_t));
    PH_HIP(hipMemcpyAsync(c.d_in.p, h, total, hipMemcpyHostToDevice, st));
    const uint8_t *db = (const uint8_t *)c.d_in.p;
    d.obj = (const float *)db;
    d.img = (const float *)(db + img_off);
    d.subsets = (const int32_t *)(db + sub_off);
    return PH_OK;
}
} // namespace

ph_status ph_solvepnp_epnp(const float *obj, const float *img, int n, const double *K,
                           double *rvec, double *tvec)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if ((s = check_pnp_args(obj, img, n, K)) != PH_OK)
        return s;
    if (n < 4 || !rvec || !tvec)
        return fail(PH_ERR_INVALID_ARG, "EPnP needs >= 4 correspondences and output pointers");
    hipStream_t st = c.stream;
    PnpDev d;
    DrainOnError drain{c};
    if ((s = upload_pnp(c, obj, img, n, nullptr, 0, st, d)) != PH_OK)
        return s;
    PH_HIP(c.h_out.reserve(6 * sizeof(double)));
    PH_HIP(epnp_solve_launch(d.obj, d.img, nullptr, n, camera_of(K), (double *)c.h_out.dp, st));
    PH_HIP(hipStreamSynchronize(st));
    drain.armed = false;
    const double *m = (const double *)c.h_out.p;
    memcpy(rvec, m, 3 * sizeof(double));
    memcpy(tvec, m + 3, 3 * sizeof(double));
    return PH_OK;
}

ph_status ph_solvepnp_ransac_epnp(const float *obj, const float *img, int n, const double *K,
                                  int iterations_count, float reproj_err, double confidence,
                                  uint64_t rng_state, double *rvec, double *tvec,
                                  int32_t *inliers, int *n_inliers, int *found,
                                  int *iterations_run)
{
    if (found)
        *found = 0;
    if (n_inliers)
        *n_inliers = 0;
    if (iterations_run)
        *iterations_run = 0;
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if ((s = check_pnp_args(obj, img, n, K)) != PH_OK)
        return s;
    if (!rvec || !tvec || !found)
        return fail(PH_ERR_INVALID_ARG, "rvec, tvec and found must be non-null");
    if (!(confidence > 0 && confidence < 1))
        return fail(PH_ERR_INVALID_ARG, "confidence must lie in (0, 1)"); // CV_Assert in run()
    if (n < 5)
        // OpenCV asserts npoints >= 4 and switches to P3P at exactly 4: not this path
        return fail(PH_ERR_INVALID_ARG, "RANSAC-EPnP needs >= 5 correspondences (got %d)", n);
    const int model_points = 5;
    const CameraK cam = camera_of(K);
    const double thr = reproj_err;
    const float thr2 = (float)(thr * thr);
    hipStream_t st = c.stream;
    DrainOnError drain{c}; // stays armed: one more wait on an idle stream costs nothing here

    if (n == model_points) { // run(): count == modelPoints -> one solve, every point an inlier
        PnpDev d;
        if ((s = upload_pnp(c, obj, img, n, nullptr, 0, st, d)) != PH_OK)
            return s;
        PH_HIP(c.h_out.reserve(6 * sizeof(double)));
        PH_HIP(epnp_solve_launch(d.obj, d.img, nullptr, n, cam, (double *)c.h_out.dp, st));
        PH_HIP(hipStreamSynchronize(st));
        const double *m = (const double *)c.h_out.p;
        memcpy(rvec, m, 3 * sizeof(double));
        memcpy(tvec, m + 3, 3 * sizeof(double));
        if (inliers)
            for (int i = 0; i < n; i++)
                inliers[i] = i;
        if (n_inliers)
            *n_inliers = n;
        *found = 1;
        return PH_OK;
    }

    // A budget of one chunk on a problem of moderate size (the reference's per-frame call is
    // 100 iterations on a few hundred pairs) runs WITHOUT a host round trip: solve and score
    // every hypothesis of the budget, then one more launch that replays OpenCV's sequential
    // selection rule over the device-resident counts (strictly larger count wins, every win
    // shrinks the budget), lists the chosen model's inliers and refits on them.  One copy in,
    // three launches, one synchronisation.  The rule's logarithms come from this host (a
    // table over every possible inlier count, cached per problem size), so the iteration
    // count and the chosen hypothesis are those of the host replay below, bit for bit.
    constexpr int FUSED_MAX_HYPOTHESES = 4096, FUSED_MAX_ROWS = 4096;
    if (std::max(iterations_count, 1) <= FUSED_MAX_HYPOTHESES && n <= FUSED_MAX_ROWS) {
        const int total = std::max(iterations_count, 1);
        const int32_t *subs = c.subsets.get(rng_state, n, total);
        const double *table = c.iter_tables.get(n, model_points);
        // pinned input area: obj | img | subsets | table, moved with one copy
        const size_t img_off = align256((size_t)n * 12), sub_off = img_off + align256((size_t)n * 8);
        const size_t tab_off = sub_off + align256((size_t)total * 5 * sizeof(int32_t));
        const size_t in_total = tab_off + ((size_t)n + 1) * sizeof(double);
        PH_HIP(c.h_in.reserve(in_total));
        PH_HIP(c.d_in.reserve(in_total));
        uint8_t *h = (uint8_t *)c.h_in.p;
        memcpy(h, obj, (size_t)n * 12);
        memcpy(h + img_off, img, (size_t)n * 8);
        memcpy(h + sub_off, subs, (size_t)total * 5 * sizeof(int32_t));
        memcpy(h + tab_off, table, ((size_t)n + 1) * sizeof(double));
        PH_HIP(hipMemcpyAsync(c.d_in.p, h, in_total, hipMemcpyHostToDevice, st));
        const uint8_t *din = (const uint8_t *)c.d_in.p;
        const float *d_obj = (const float *)din, *d_img = (const float *)(din + img_off);
        // device-resident hypotheses: models | counts
        const size_t cnt_off = align256((size_t)total * 6 * sizeof(double));
        PH_HIP(c.pnp_hyp.reserve(cnt_off + (size_t)total * sizeof(int32_t)));
        double *d_models = (double *)c.pnp_hyp.p;
        int32_t *d_counts = (int32_t *)((uint8_t *)c.pnp_hyp.p + cnt_off);
        PH_HIP(c.pnp_idx.reserve((size_t)n * sizeof(int32_t)));
        // pinned results, written by the last kernel: refit model | inlier count | iterations,
        // chosen hypothesis | inlier list
        const size_t ninl_off = 256, rep_off = 512, list_off = 768;
        PH_HIP(c.h_out.reserve(list_off + (size_t)n * sizeof(int32_t)));
        uint8_t *hb = (uint8_t *)c.h_out.p, *db = (uint8_t *)c.h_out.dp;
        int32_t *h_rep = (int32_t *)(hb + rep_off);
        h_rep[0] = 0;
        h_rep[1] = -2; // overwritten by the kernel
        PH_HIP(ransac_hypotheses_launch(d_obj, d_img, n, cam, (const int32_t *)(din + sub_off),
                                        nullptr, nullptr, 1, total, thr2, d_models, d_counts,
                                        c.n_cus, st));
        RefitOne one{};
        one.rows = n;
        RefitReplay rp;
        rp.counts = d_counts;
        rp.models = d_models;
        rp.iter_table = (const double *)(din + tab_off);
        rp.num_log = iters_num_log(confidence);
        rp.total = total;
        rp.out = (int32_t *)(db + rep_off);
        PH_HIP(epnp_refit_batched_launch(d_obj, d_img, nullptr, nullptr, nullptr, nullptr, one, 1,
                                         cam, thr2, (int32_t *)c.pnp_idx.p,
                                         (int32_t *)(db + list_off), (int32_t *)(db + ninl_off),
                                         (double *)db, st, &rp));
        PH_HIP(hipStreamSynchronize(st));
        if (h_rep[1] == -2)
            return fail(PH_ERR_HIP, "the refit kernel did not report its selection");
        if (iterations_run)
            *iterations_run = h_rep[0];
        if (h_rep[1] < 0) { // no hypothesis with 5 or more inliers
            *found = 0;
            return PH_OK;
        }
        const int k = *(const int32_t *)(hb + ninl_off);
        if (k < 4) { // cannot happen with a chosen model of >= 5 inliers
            *found = 0;
            return PH_OK;
        }
        const double *m6 = (const double *)hb;
        memcpy(rvec, m6, 3 * sizeof(double));
        memcpy(tvec, m6 + 3, 3 * sizeof(double));
        if (inliers)
            memcpy(inliers, hb + list_off, (size_t)k * sizeof(int32_t));
        if (n_inliers)
            *n_inliers = k;
        *found = 1;
        return PH_OK;
    }

    // Larger budgets and problems: the subset sequence is model-independent, so all hypotheses
    // of the budget are solved and scored in parallel, in chunks; OpenCV's sequential rule
    // (strictly larger count wins, budget shrinks on every win) is then replayed
    // over the counts, and chunks beyond the shrunken budget are never launched.
    int niters = std::max(iterations_count, 1);
    const int total = niters;
    // samples are drawn and uploaded one chunk at a time: a budget the adaptive rule cuts
    // after a few hundred iterations never pays for the samples of its nominal count
    const int chunk = 4096, cmax = std::min(total, chunk);
    PnpDev d;
    if ((s = upload_pnp(c, obj, img, n, c.subsets.get(rng_state, n, cmax), (size_t)cmax, st, d)) !=
        PH_OK)
        return s;
    int32_t *h_sub = (int32_t *)((uint8_t *)c.h_in.p +
                                 ((const uint8_t *)d.subsets - (const uint8_t *)c.d_in.p));
    // pinned results, written by the kernels: models | counts | refit model | inlier count |
    // inlier list
    const size_t counts_off = align256((size_t)cmax * 6 * sizeof(double));
    const size_t best_off = counts_off + align256((size_t)cmax * sizeof(int32_t));
    const size_t refit_off = best_off, ninl_off = refit_off + 256, list_off = ninl_off + 256;
    PH_HIP(c.h_out.reserve(list_off + (size_t)n * sizeof(int32_t)));
    uint8_t *hb = (uint8_t *)c.h_out.p, *db = (uint8_t *)c.h_out.dp;
    const double *models = (const double *)hb;
    const int32_t *counts = (const int32_t *)(hb + counts_off);
    int max_good = 0, iter = 0;
    double best[6] = {0, 0, 0, 0, 0, 0};
    for (int base = 0; base < total && base < niters; base += chunk) {
        const int m = std::min(chunk, total - base);
        if (base > 0) { // the previous chunk has been waited for: its samples can be replaced
            memcpy(h_sub, c.subsets.get(rng_state, n, base + m) + (size_t)base * 5,
                   (size_t)m * 5 * sizeof(int32_t));
            PH_HIP(hipMemcpyAsync((void *)d.subsets, h_sub, (size_t)m * 5 * sizeof(int32_t),
                                  hipMemcpyHostToDevice, st));
        }
        PH_HIP(ransac_hypotheses_launch(d.obj, d.img, n, cam, d.subsets, nullptr,
                                        nullptr, 1, m, thr2, (double *)db,
                                        (int32_t *)(db + counts_off), c.n_cus, st));
        PH_HIP(hipStreamSynchronize(st));
        for (iter = base; iter < base + m && iter < niters; iter++) {
            const int good = counts[iter - base];
            if (good > std::max(max_good, model_points - 1)) {
                memcpy(best, &models[(size_t)(iter - base) * 6], sizeof(best));
                max_good = good;
                niters = update_num_iters(confidence, (double)(n - good) / n, model_points, niters);
            }
        }
    }
    if (iterations_run)
        *iterations_run = iter;
    if (max_good <= 0) {
        *found = 0;
        return PH_OK;
    }
    // inliers of the best model and one EPnP solve over them, in one launch
    RefitOne one;
    memcpy(one.best, best, sizeof(best));
    one.rows = n;
    PH_HIP(c.pnp_idx.reserve((size_t)n * sizeof(int32_t)));
    PH_HIP(epnp_refit_batched_launch(d.obj, d.img, nullptr, nullptr, nullptr, nullptr, one, 1, cam,
                                     thr2, (int32_t *)c.pnp_idx.p, (int32_t *)(db + list_off),
                                     (int32_t *)(db + ninl_off), (double *)(db + refit_off), st));
    PH_HIP(hipStreamSynchronize(st));
    const int k = *(const int32_t *)(hb + ninl_off);
    if (k < 4) { // cannot happen with max_good >= 5; solvePnP would throw and RANSAC report failure
        memcpy(rvec, best, 3 * sizeof(double));
        memcpy(tvec, best + 3, 3 * sizeof(double));
        *found = 0;
        return PH_OK;
    }
    const double *m6 = (const double *)(hb + refit_off);
    memcpy(rvec, m6, 3 * sizeof(double));
    memcpy(tvec, m6 + 3, 3 * sizeof(double));
    if (inliers)
        memcpy(inliers, hb + list_off, (size_t)k * sizeof(int32_t));
    if (n_inliers)
        *n_inliers = k;
    *found = 1;
    return PH_OK;
}

ph_status ph_solvepnp_ransac_epnp_batched(const float *obj, const float *img,
                                          const int32_t *point_offsets, int n_problems,
                                          const double *K, int iterations_count, float reproj_err,
                                          double confidence, uint64_t rng_state, double *rvecs,
                                          double *tvecs, int32_t *inliers, int32_t *n_inliers,
                                          int32_t *found, int32_t *iterations_run)
{
    Call call;
    if (call.st != PH_OK)
        return call.st;
    Context &c = *call.c;
    ph_status s = PH_OK;
    (void)s;
    if (n_problems < 0 || n_problems > 65535)
        return fail(PH_ERR_INVALID_ARG, "n_problems must be in 0..65535 (got %d)", n_problems);
    if (n_problems == 0)
        return PH_OK;
    if (!point_offsets || !rvecs || !tvecs || !found)
        return fail(PH_ERR_INVALID_ARG, "point_offsets, rvecs, tvecs and found must be non-null");
    if (point_offsets[0] < 0)
        return fail(PH_ERR_INVALID_ARG, "negative offset");
    for (int p = 0; p < n_problems; p++)
        if (point_offsets[p + 1] < point_offsets[p])
            return fail(PH_ERR_INVALID_ARG, "offsets of problem %d are not ascending", p);
    const int row_hi = point_offsets[n_problems];
    if ((s = check_pnp_args(obj, img, row_hi, K)) != PH_OK)
        return s;
    if (!(confidence > 0 && confidence < 1))
        return fail(PH_ERR_INVALID_ARG, "confidence must lie in (0, 1)");
    for (int p = 0; p < n_problems; p++) {
        found[p] = 0;
        if (n_inliers)
            n_inliers[p] = 0;
        if (iterations_run)
            iterations_run[p] = 0;
    }
    const int model_points = 5;
    const CameraK cam = camera_of(K);
    const double thr = reproj_err;
    const float thr2 = (float)(thr * thr);
    hipStream_t st = c.stream;
    DrainOnError drain{c}; // stays armed: one more wait on an idle stream costs nothing here

    struct Prob {
        int n, niters, iter = 0, max_good = 0;
        double best[6] = {0, 0, 0, 0, 0, 0};
        bool live;
    };
    std::vector<Prob> pr;
    pr.reserve((size_t)n_problems);
    for (int p = 0; p < n_problems; p++) {
        Prob q{};
        q.n = point_offsets[p + 1] - point_offsets[p];
        q.niters = std::max(iterations_count, 1);
        q.live = q.n > model_points; // fewer: no pose; exactly five: one solve, no sampling
        pr.push_back(q);
    }

    // Hypotheses per round: enough to fill the GPU, shared by the live problems.
    const int hyp_target = c.n_cus * 128;
    const size_t cap_hyps = (size_t)std::max(hyp_target, n_problems * 32);
    // pinned input area: obj | img staging, then small tables the kernels read in place
    const size_t rows = (size_t)row_hi;
    const size_t img_off = align256(rows * 12), tab_off = img_off + align256(rows * 8);
    const size_t np4 = align256((size_t)n_problems * sizeof(int32_t));
    const size_t beg_off = tab_off, end_off = beg_off + np4, all_off = end_off + np4;
    const size_t best_off = all_off + np4;
    const size_t sub_off = best_off + align256((size_t)n_problems * 6 * sizeof(double));
    const size_t in_total = sub_off + align256(cap_hyps * 5 * sizeof(int32_t));
    PH_HIP(c.h_in.reserve(in_total));
    PH_HIP(c.d_in.reserve(tab_off));
    uint8_t *hi = (uint8_t *)c.h_in.p, *di = (uint8_t *)c.h_in.dp;
    memcpy(hi, obj, rows * 12);
    memcpy(hi + img_off, img, rows * 8);
    PH_HIP(hipMemcpyAsync(c.d_in.p, hi, tab_off, hipMemcpyHostToDevice, st));
    const float *d_obj = (const float *)c.d_in.p;
    const float *d_img = (const float *)((const uint8_t *)c.d_in.p + img_off);
    // pinned result area, written by the kernels: models | counts | inlier lists (one slot
    // per row) | inlier counts | refit models
    const size_t cnt_off = align256(cap_hyps * 6 * sizeof(double));
    const size_t list_off = cnt_off + align256(cap_hyps * sizeof(int32_t));
    const size_t ninl_off = list_off + align256(rows * sizeof(int32_t));
    const size_t refit_off = ninl_off + np4;
    PH_HIP(c.h_out.reserve(refit_off + (size_t)n_problems * 6 * sizeof(double)));
    uint8_t *ho = (uint8_t *)c.h_out.p, *dout = (uint8_t *)c.h_out.dp;
    const double *models = (const double *)ho;
    const int32_t *counts = (const int32_t *)(ho + cnt_off);
    int32_t *h_beg = (int32_t *)(hi + beg_off), *h_end = (int32_t *)(hi + end_off);
    int32_t *h_sub = (int32_t *)(hi + sub_off);

    std::vector<int> active;
    for (;;) {
        active.clear();
        int max_left = 0;
        for (int p = 0; p < n_problems; p++)
            if (pr[p].live && pr[p].iter < pr[p].niters) {
                active.push_back(p);
                max_left = std::max(max_left, pr[p].niters - pr[p].iter);
            }
        if (active.empty())
            break;
        const int na = (int)active.size();
        int chunk = std::min(std::max(hyp_target / na, 32), 4096);
        chunk = std::min(chunk, max_left);
        // every live problem gets `chunk` slots; slots beyond its budget repeat a valid
        // sample and are ignored by the replay
        for (int a = 0; a < na; a++) {
            Prob &q = pr[active[a]];
            h_beg[a] = point_offsets[active[a]];
            h_end[a] = point_offsets[active[a] + 1];
            const int m = std::min(chunk, q.niters - q.iter);
            int32_t *sub = h_sub + (size_t)a * chunk * 5;
            memcpy(sub, c.subsets.get(rng_state, q.n, q.iter + m) + (size_t)q.iter * 5,
                   (size_t)m * 5 * sizeof(int32_t));
            for (int j = m; j < chunk; j++)
                for (int e = 0; e < 5; e++)
                    sub[(size_t)j * 5 + e] = e;
        }
        PH_HIP(ransac_hypotheses_launch(d_obj, d_img, 0, cam, (const int32_t *)(di + sub_off),
                                        (const int32_t *)(di + beg_off),
                                        (const int32_t *)(di + end_off), na, chunk, thr2,
                                        (double *)dout, (int32_t *)(dout + cnt_off), c.n_cus,
                                        st));
        PH_HIP(hipStreamSynchronize(st));
        for (int a = 0; a < na; a++) {
            Prob &q = pr[active[a]];
            const size_t h0 = (size_t)a * chunk;
            for (int j = 0; j < chunk && q.iter < q.niters; j++, q.iter++) {
                const int good = counts[h0 + j];
                if (good > std::max(q.max_good, model_points - 1)) {
                    memcpy(q.best, &models[(h0 + j) * 6], sizeof(q.best));
                    q.max_good = good;
                    q.niters = update_num_iters(confidence, (double)(q.n - good) / q.n,
                                                model_points, q.niters);
                }
            }
        }
    }

    // the final step in one launch: problems with a model list its inliers and refit on
    // them, problems of exactly five rows solve on all their rows
    std::vector<int> refit;
    double *h_best = (double *)(hi + best_off);
    uint8_t *h_all = hi + all_off;
    for (int p = 0; p < n_problems; p++) {
        if (iterations_run)
            iterations_run[p] = pr[p].iter;
        const bool has_model = pr[p].live && pr[p].max_good > 0;
        if (!has_model && pr[p].n != model_points)
            continue;
        const int b = (int)refit.size();
        h_beg[b] = point_offsets[p];
        h_end[b] = point_offsets[p + 1];
        h_all[b] = has_model ? 0 : 1;
        memcpy(h_best + (size_t)b * 6, pr[p].best, sizeof(pr[p].best));
        refit.push_back(p);
    }
    if (refit.empty())
        return PH_OK;
    PH_HIP(c.pnp_idx.reserve(rows * sizeof(int32_t)));
    PH_HIP(epnp_refit_batched_launch(d_obj, d_img, (const int32_t *)(di + beg_off),
                                     (const int32_t *)(di + end_off),
                                     (const double *)(di + best_off), di + all_off, RefitOne{},
                                     (int)refit.size(), cam, thr2, (int32_t *)c.pnp_idx.p,
                                     (int32_t *)(dout + list_off), (int32_t *)(dout + ninl_off),
                                     (double *)(dout + refit_off), st));
    PH_HIP(hipStreamSynchronize(st));
    const double *rm = (const double *)(ho + refit_off);
    const int32_t *lists = (const int32_t *)(ho + list_off), *ninl = (const int32_t *)(ho + ninl_off);
    for (size_t b = 0; b < refit.size(); b++) {
        const int p = refit[b], k = ninl[b];
        if (k < 4) // cannot happen with a model of >= 5 inliers
            continue;
        memcpy(rvecs + (size_t)p * 3, rm + b * 6, 3 * sizeof(double));
        memcpy(tvecs + (size_t)p * 3, rm + b * 6 + 3, 3 * sizeof(double));
        if (inliers)
            memcpy(inliers + point_offsets[p], lists + point_offsets[p], (size_t)k * sizeof(int32_t));
        if (n_inliers)
            n_inliers[p] = k;
        found[p] = 1;
    }
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
