// Brute-force Hamming-256 k-NN (k <= 2) for gfx950 (MI355X).
//
// Arithmetic replaced: cv::BFMatcher(NORM_HAMMING)::knnMatch as called at
// src/cereal_box.cpp:344 and src/orb_feature_matching/matcher.py:21 of the
// reference (OpenCV batchDistance: popcount(a XOR b) over 32 bytes, K-best
// list with strict '<' insertion => lowest train index wins ties).
//
// Mapping.  One query row per lane (QPL rows per lane), held in 8*QPL VGPRs
// for the whole kernel.  Train rows are wave-uniform: they are fetched with
// scalar loads (s_load_dwordx8) into SGPRs and used directly as the scalar
// operand of v_xor_b32, so the inner loop carries no vector-memory or LDS
// instruction at all.  Per (query, train) pair the VALU executes
//     8 x v_xor_b32 + 8 x v_bcnt_u32_b32 (accumulating)        = 16 ops
//   + 1 x v_lshl_add_u32 (key = dist << 22 | local train index)
//   + 1 x v_min_u32 + 1 x v_med3_u32 (running best / second best) =  3 ops
// Keys order by (distance, train index), so min / median-of-three reproduce
// the strict-'<' insertion rule exactly.
//
// A launch is a grid (query blocks) x (train splits); each workgroup writes
// its two best keys per query for its split, and knn_merge_kernel folds the
// splits in ascending train order.  Splits exist only to fill 256 CUs when the
// query count alone cannot.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knn_hamming.h"

namespace posehip {

static constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

struct Row {
    uint4 lo, hi;
};

// popcount(x) + acc in one VALU instruction.  Written as asm because the compiler
// otherwise splits the 8-term sum into v_bcnt(x, 0) plus a tree of v_add3_u32.
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

__device__ __forceinline__ uint32_t bcnt0(uint32_t x)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ uint32_t hamming256(const uint32_t (&q)[8], const Row &t)
{
    uint32_t d = bcnt0(q[0] ^ t.lo.x);
    d = bcnt_acc(q[1] ^ t.lo.y, d);
    d = bcnt_acc(q[2] ^ t.lo.z, d);
    d = bcnt_acc(q[3] ^ t.lo.w, d);
    d = bcnt_acc(q[4] ^ t.hi.x, d);
    d = bcnt_acc(q[5] ^ t.hi.y, d);
    d = bcnt_acc(q[6] ^ t.hi.z, d);
    d = bcnt_acc(q[7] ^ t.hi.w, d);
    return d;
}

// key = dist << KNN_KEY_SHIFT | j with j wave-uniform: one v_lshl_add_u32 with
// an SGPR addend.
__device__ __forceinline__ uint32_t make_key(uint32_t d, uint32_t j)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(d), "n"(KNN_KEY_SHIFT), "s"(j));
    return r;
}

__device__ __forceinline__ void top2_update(uint32_t &m1, uint32_t &m2, uint32_t key)
{
    m2 = med3_u32(m1, m2, key);
    m1 = min(m1, key);
}

template <int UNROLL> struct RowGroup {
    Row r[UNROLL];
};

template <int UNROLL>
__device__ __forceinline__ void load_group(RowGroup<UNROLL> &g, const Row *__restrict__ rows)
{
#pragma unroll
    for (int u = 0; u < UNROLL; u++)
        g.r[u] = rows[u];
}

template <int QPL, int UNROLL>
__device__ __forceinline__ void eval_group(const RowGroup<UNROLL> &g, const uint32_t (&q)[QPL][8],
                                           uint32_t (&m1)[QPL], uint32_t (&m2)[QPL], uint32_t j)
{
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
        for (int r = 0; r < QPL; r++) {
            const uint32_t d = hamming256(q[r], g.r[u]);
            top2_update(m1[r], m2[r], make_key(d, j + u));
        }
    }
}

// part layout: [split][rank 0/1][nq] packed keys (dist << KEY_SHIFT | local index).
template <int QPL, int UNROLL>
__global__ __launch_bounds__(KNN_BLOCK) void knn_partial_kernel(const uint4 *__restrict__ query,
                                                                const uint4 *__restrict__ train,
                                                                int64_t nq, int64_t nt,
                                                                int64_t rows_per_split,
                                                                uint32_t *__restrict__ part)
{
    const int64_t qbase = (int64_t)blockIdx.x * (KNN_BLOCK * QPL) + threadIdx.x;
    const int split = blockIdx.y;
    const int64_t t0 = (int64_t)split * rows_per_split;
    const int64_t t1 = min(nt, t0 + rows_per_split);
    const int n = (int)(t1 - t0);

    uint32_t q[QPL][8];
    uint32_t m1[QPL], m2[QPL];
#pragma unroll
    for (int r = 0; r < QPL; r++) {
        const int64_t qi = qbase + (int64_t)r * KNN_BLOCK;
        uint4 a = make_uint4(0, 0, 0, 0), b = a;
        if (qi < nq) {
            a = query[2 * qi];
            b = query[2 * qi + 1];
        }
        q[r][0] = a.x; q[r][1] = a.y; q[r][2] = a.z; q[r][3] = a.w;
        q[r][4] = b.x; q[r][5] = b.y; q[r][6] = b.z; q[r][7] = b.w;
        m1[r] = KEY_NONE;
        m2[r] = KEY_NONE;
    }

    // Scalar loads return out of order, so the only usable wait is "all of
    // them".  Two SGPR row groups alternate: the group for step g+1 is
    // requested right after the wait that retires group g and flies while
    // group g is evaluated.
    const Row *__restrict__ rows = reinterpret_cast<const Row *>(train) + t0;
    const int n_groups = n / UNROLL;
    RowGroup<UNROLL> ga, gb;
    if (n_groups > 0)
        load_group(ga, rows);
    int g = 0;
    for (; g + 1 < n_groups; g += 2) {
        load_group(gb, rows + (g + 1) * UNROLL);
        eval_group<QPL, UNROLL>(ga, q, m1, m2, (uint32_t)(g * UNROLL));
        if (g + 2 < n_groups)
            load_group(ga, rows + (g + 2) * UNROLL);
        eval_group<QPL, UNROLL>(gb, q, m1, m2, (uint32_t)((g + 1) * UNROLL));
    }
    if (g < n_groups)
        eval_group<QPL, UNROLL>(ga, q, m1, m2, (uint32_t)(g * UNROLL));
    for (int j = n_groups * UNROLL; j < n; j++) {
        RowGroup<1> t;
        load_group(t, rows + j);
        eval_group<QPL, 1>(t, q, m1, m2, (uint32_t)j);
    }

    uint32_t *p1 = part + ((int64_t)split * 2) * nq;
    uint32_t *p2 = p1 + nq;
#pragma unroll
    for (int r = 0; r < QPL; r++) {
        const int64_t qi = qbase + (int64_t)r * KNN_BLOCK;
        if (qi < nq) {
            p1[qi] = m1[r];
            p2[qi] = m2[r];
        }
    }
}

// Folds the per-split keys in ascending split order into the final k-NN rows.
// A key from a later split never displaces an equal distance from an earlier
// one (strict '<'), which is the lowest-train-index rule across splits.
__global__ __launch_bounds__(256) void knn_merge_kernel(const uint32_t *__restrict__ part,
                                                        int64_t nq, int n_splits,
                                                        int64_t rows_per_split, int k,
                                                        int32_t index_base,
                                                        int32_t *__restrict__ out_idx,
                                                        int32_t *__restrict__ out_dist,
                                                        uint8_t *__restrict__ keep, float ratio)
{
    const int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq)
        return;
    // 64-bit keys: dist << 32 | global index; ~0 = none.
    uint64_t b1 = ~0ull, b2 = ~0ull;
    for (int s = 0; s < n_splits; s++) {
        const uint32_t *p = part + ((int64_t)s * 2) * nq;
        const uint32_t k1 = p[qi], k2 = p[nq + qi];
        const uint64_t base = (uint64_t)((int64_t)s * rows_per_split);
        const uint32_t mask = (1u << KNN_KEY_SHIFT) - 1;
        const uint64_t g1 =
            k1 == KEY_NONE ? ~0ull : ((uint64_t)(k1 >> KNN_KEY_SHIFT) << 32) | (base + (k1 & mask));
        const uint64_t g2 =
            k2 == KEY_NONE ? ~0ull : ((uint64_t)(k2 >> KNN_KEY_SHIFT) << 32) | (base + (k2 & mask));
        // merge sorted pairs (b1<=b2), (g1<=g2)
        const uint64_t lo = min(b1, g1);
        const uint64_t hi = max(b1, g1);
        b2 = min(hi, min(b2, g2));
        b1 = lo;
    }
    const int32_t i1 = b1 == ~0ull ? -1 : (int32_t)(uint32_t)b1 + index_base;
    const int32_t d1 = b1 == ~0ull ? 0x7FFFFFFF : (int32_t)(b1 >> 32);
    if (k == 1) {
        out_idx[qi] = i1;
        out_dist[qi] = d1;
        return;
    }
    const int32_t i2 = b2 == ~0ull ? -1 : (int32_t)(uint32_t)b2 + index_base;
    const int32_t d2 = b2 == ~0ull ? 0x7FFFFFFF : (int32_t)(b2 >> 32);
    out_idx[2 * qi] = i1;
    out_idx[2 * qi + 1] = i2;
    out_dist[2 * qi] = d1;
    out_dist[2 * qi + 1] = d2;
    if (keep)
        keep[qi] = (i1 >= 0 && i2 >= 0 && (float)d1 < ratio * (float)d2) ? 1 : 0;
}

// Merge of all-gathered per-shard candidate rows ([shard][nq][k], global
// indices, shards in ascending index order).
__global__ __launch_bounds__(256) void knn_merge_shards_kernel(
    const int32_t *__restrict__ cand_idx, const int32_t *__restrict__ cand_dist, int n_shards,
    int64_t nq, int k, int32_t *__restrict__ out_idx, int32_t *__restrict__ out_dist,
    uint8_t *__restrict__ keep, float ratio)
{
    const int64_t qi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq)
        return;
    uint64_t b1 = ~0ull, b2 = ~0ull;
    for (int s = 0; s < n_shards; s++) {
        const int64_t o = ((int64_t)s * nq + qi) * k;
        for (int r = 0; r < k; r++) {
            const int32_t ix = cand_idx[o + r];
            if (ix < 0)
                continue;
            const uint64_t g = ((uint64_t)(uint32_t)cand_dist[o + r] << 32) | (uint32_t)ix;
            if (g < b1) {
                b2 = b1;
                b1 = g;
            } else if (g < b2) {
                b2 = g;
            }
        }
    }
    const int32_t i1 = b1 == ~0ull ? -1 : (int32_t)(uint32_t)b1;
    const int32_t d1 = b1 == ~0ull ? 0x7FFFFFFF : (int32_t)(b1 >> 32);
    if (k == 1) {
        out_idx[qi] = i1;
        out_dist[qi] = d1;
        return;
    }
    const int32_t i2 = b2 == ~0ull ? -1 : (int32_t)(uint32_t)b2;
    const int32_t d2 = b2 == ~0ull ? 0x7FFFFFFF : (int32_t)(b2 >> 32);
    out_idx[2 * qi] = i1;
    out_idx[2 * qi + 1] = i2;
    out_dist[2 * qi] = d1;
    out_dist[2 * qi + 1] = d2;
    if (keep)
        keep[qi] = (i1 >= 0 && i2 >= 0 && (float)d1 < ratio * (float)d2) ? 1 : 0;
}

// ---- cross-check resolve -----------------------------------------------------
// Step 3 of BFMatcher's cross-check: for train i ascending,
//   q = tidx[i]; if (tdist[i] < dist[q]) { dist[q] = tdist[i]; nidx[q] = i; }
// i.e. per query the minimum of (tdist, i) over the train rows that chose it:
// an order-independent 64-bit atomic min.
__global__ __launch_bounds__(256) void cc_init_kernel(unsigned long long *__restrict__ best,
                                                      int64_t nq)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nq)
        best[i] = ~0ull;
}

__global__ __launch_bounds__(256) void cc_scatter_kernel(const int32_t *__restrict__ tidx,
                                                         const int32_t *__restrict__ tdist,
                                                         int64_t nt,
                                                         unsigned long long *__restrict__ best)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nt)
        return;
    const int32_t q = tidx[i];
    if (q < 0)
        return;
    const unsigned long long key = ((unsigned long long)(uint32_t)tdist[i] << 32) | (uint32_t)i;
    atomicMin(&best[q], key);
}

__global__ __launch_bounds__(256) void cc_finalize_kernel(
    const unsigned long long *__restrict__ best, int64_t nq, int32_t *__restrict__ out_idx,
    int32_t *__restrict__ out_dist)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq)
        return;
    const unsigned long long b = best[i];
    out_idx[i] = b == ~0ull ? -1 : (int32_t)(uint32_t)b;
    out_dist[i] = b == ~0ull ? 0x7FFFFFFF : (int32_t)(b >> 32);
}

// ---- launchers ---------------------------------------------------------------

KnnPlan knn_plan(int64_t nq, int64_t nt, int n_cus)
{
    KnnPlan p;
    p.qpl = 1;
    const int64_t rows_per_block = (int64_t)KNN_BLOCK * p.qpl;
    p.q_blocks = (nq + rows_per_block - 1) / rows_per_block;
    // Aim at >= 8 workgroups of 4 waves per CU; never split below 256 train
    // rows per workgroup, never above the key's local-index range.
    const int64_t want_blocks = (int64_t)n_cus * 8;
    int64_t splits = (want_blocks + p.q_blocks - 1) / p.q_blocks;
    const int64_t max_splits = (nt + 255) / 256;
    if (splits > max_splits)
        splits = max_splits;
    if (splits < 1)
        splits = 1;
    int64_t rows = (nt + splits - 1) / splits;
    if (rows > KNN_MAX_ROWS_PER_SPLIT)
        rows = KNN_MAX_ROWS_PER_SPLIT;
    if (rows < 1)
        rows = 1;
    p.rows_per_split = rows;
    p.n_splits = (int)((nt + rows - 1) / rows);
    if (p.n_splits < 1)
        p.n_splits = 1;
    p.part_bytes = (size_t)p.n_splits * 2 * (size_t)nq * sizeof(uint32_t);
    return p;
}

hipError_t knn_launch(const KnnPlan &p, const void *d_query, int64_t nq, const void *d_train,
                      int64_t nt, int k, int32_t index_base, uint32_t *d_part, int32_t *d_out_idx,
                      int32_t *d_out_dist, uint8_t *d_keep, float ratio, hipStream_t stream)
{
    dim3 grid((unsigned)p.q_blocks, (unsigned)p.n_splits);
    knn_partial_kernel<1, 4><<<grid, KNN_BLOCK, 0, stream>>>(
        (const uint4 *)d_query, (const uint4 *)d_train, nq, nt, p.rows_per_split, d_part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    const unsigned mb = (unsigned)((nq + 255) / 256);
    knn_merge_kernel<<<mb, 256, 0, stream>>>(d_part, nq, p.n_splits, p.rows_per_split, k,
                                             index_base, d_out_idx, d_out_dist, d_keep, ratio);
    return hipGetLastError();
}

hipError_t knn_merge_shards_launch(const int32_t *d_cand_idx, const int32_t *d_cand_dist,
                                   int n_shards, int64_t nq, int k, int32_t *d_out_idx,
                                   int32_t *d_out_dist, uint8_t *d_keep, float ratio,
                                   hipStream_t stream)
{
    const unsigned mb = (unsigned)((nq + 255) / 256);
    knn_merge_shards_kernel<<<mb, 256, 0, stream>>>(d_cand_idx, d_cand_dist, n_shards, nq, k,
                                                    d_out_idx, d_out_dist, d_keep, ratio);
    return hipGetLastError();
}

hipError_t crosscheck_resolve_launch(const int32_t *d_tidx, const int32_t *d_tdist, int64_t nt,
                                     int64_t nq, unsigned long long *d_best, int32_t *d_out_idx,
                                     int32_t *d_out_dist, hipStream_t stream)
{
    cc_init_kernel<<<(unsigned)((nq + 255) / 256), 256, 0, stream>>>(d_best, nq);
    cc_scatter_kernel<<<(unsigned)((nt + 255) / 256), 256, 0, stream>>>(d_tidx, d_tdist, nt,
                                                                         d_best);
    cc_finalize_kernel<<<(unsigned)((nq + 255) / 256), 256, 0, stream>>>(d_best, nq, d_out_idx,
                                                                          d_out_dist);
    return hipGetLastError();
}

} // namespace posehip
