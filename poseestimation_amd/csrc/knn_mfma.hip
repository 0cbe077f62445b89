// Brute-force Hamming-256 k-NN (k <= 2) on the gfx950 matrix cores.
//
// Same arithmetic as knn_hamming.hip (cv::BFMatcher(NORM_HAMMING)::knnMatch as called at
// src/cereal_box.cpp:344 and src/orb_feature_matching/matcher.py:21 of the reference:
// popcount(a XOR b) over 32 bytes, strict '<' insertion => lowest train index wins ties),
// computed as a matrix product.  Every descriptor bit becomes one FP4 (E2M1) element,
// -4 for a set bit and +4 for a clear one, so that for a query row q and a train row t
//     dot(q, t) = 16 * (256 - 2 * hamming(q, t))
// exactly (256 products of +-16 summed in fp32).  v_mfma_scale_f32_32x32x64_f8f6f4 forms a
// 32 (train rows) x 32 (queries) tile of such dot products over 64 bits per instruction,
// four instructions per tile; the block scale 2^10 and the accumulator input
//     C[row] = 2^22 + 31 - row_in_tile
// turn the accumulator itself into an ordering key,
//     key = (256 - hamming) * 2^15 + (31 - row_in_tile)            (an integer < 2^24),
// whose MAXIMUM over a tile is the nearest train row, lowest row first among equals.  The
// running best / second best of a lane carry an index field relative to the tile being
// processed: 32 is added to both before every tile (two VALU ops), which keeps earlier
// tiles ahead of later ones, and the sixteen keys of the tile are folded in with
// v_max3 / v_med3 (1.25 ops per key).  The VALU therefore spends 22 instructions per
// 16 keys where the popcount kernel spends 292.
//
// A workgroup is 4 waves; wave w holds QT tiles of 32 queries as B fragments in registers
// for the whole kernel, all four waves stream the same train tiles, staged through LDS in
// fragment order (one 16-byte piece per thread and tile, loaded a tile ahead).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knn_hamming.h"
#include "knn_mfma.h"

namespace posehip {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr float MF_NONE = -65536.0f;   // stays negative under 1024 additions of 32
static constexpr int MF_SCALE_A = 0x89898989; // E8M0 2^10 in every byte
static constexpr int MF_SCALE_B = 0x7F7F7F7F; // E8M0 2^0

// ---- bit -> FP4 expansion -------------------------------------------------------
// 8 bits -> 8 nibbles: bit i moves to bit 4 i, then nibble = 0x6 (+4.0) | bit << 3 (sign).
__device__ __forceinline__ uint32_t fp4_from_bits8(uint32_t b)
{
    uint32_t x = (b | (b << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return (x << 3) | 0x66666666u;
}

// One thread per 32 descriptor bits: dword c of row i -> 16 bytes at piece c of the
// expanded row (128 bytes per row).  K order is the same for every row, which is all the
// dot product needs.
__global__ __launch_bounds__(256) void fp4_expand_kernel(const uint32_t *__restrict__ rows,
                                                         int64_t n_dwords,
                                                         uint4 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_dwords)
        return;
    const uint32_t w = rows[i];
    out[i] = make_uint4(fp4_from_bits8(w & 0xFF), fp4_from_bits8((w >> 8) & 0xFF),
                        fp4_from_bits8((w >> 16) & 0xFF), fp4_from_bits8(w >> 24));
}

// ---- top-2 by maximum -------------------------------------------------------------
// Keys are non-negative floats holding integers, "none" is negative: their bit patterns
// order like the values under a signed integer compare, which gives v_max3_i32 without
// any float canonicalisation; the median uses v_med3_f32.
__device__ __forceinline__ float imax3(float a, float b, float c)
{
    return __int_as_float(max(max(__float_as_int(a), __float_as_int(b)), __float_as_int(c)));
}

template <int K>
__device__ __forceinline__ void top2_max4(float &m1, float &m2, float k0, float k1, float k2,
                                          float k3)
{
    const float a1 = imax3(m1, k0, k1);
    const float a2 = imax3(a1, k2, k3);
    if constexpr (K == 2) {
        const float b1 = __builtin_amdgcn_fmed3f(m1, k0, k1);
        const float b2 = __builtin_amdgcn_fmed3f(a1, k2, k3);
        m2 = imax3(m2, b1, b2);
    }
    m1 = a2;
}

__device__ __forceinline__ i32x8 frag(const uint4 v)
{
    return i32x8{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
}

// The four dependent MFMAs of one 32 x 32 tile of keys.
__device__ __forceinline__ f32x16 mfma_chain(const i32x8 (&a)[4], const i32x8 (&b)[4],
                                             const f32x16 &c0)
{
    f32x16 acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[0], c0, 4, 4, 0,
                                                                 MF_SCALE_A, 0, MF_SCALE_B);
#pragma unroll
    for (int s = 1; s < 4; s++)
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s], b[s], acc, 4, 4, 0,
                                                              MF_SCALE_A, 0, MF_SCALE_B);
    return acc;
}

template <int K, bool MASKED>
__device__ __forceinline__ void fold_tile(f32x16 acc, float &m1, float &m2, int valid, int h)
{
    if constexpr (MASKED) {
#pragma unroll
        for (int i = 0; i < 16; i++)
            if ((i & 3) + 8 * (i >> 2) + 4 * h >= valid)
                acc[i] = MF_NONE;
    }
    m1 += 32.0f;
    if constexpr (K == 2)
        m2 += 32.0f;
#pragma unroll
    for (int g = 0; g < 4; g++)
        top2_max4<K>(m1, m2, acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
}

// One train tile against the wave's QT query tiles.  MASKED: rows at or beyond `valid`
// (the tail of the last tile) are taken out before the fold.  The MFMAs of query tile
// j + 1 are issued ahead of the fold of tile j (two accumulator sets), and the scheduler
// is told to keep that interleaving: one MFMA, then a share of the fold's VALU work.
template <int K, int QT, bool MASKED>
__device__ __forceinline__ void mfma_tile(const i32x8 (&a)[4], const i32x8 (&b)[QT][4],
                                          const f32x16 &c0, float (&m1)[QT], float (&m2)[QT],
                                          int valid, int h)
{
    f32x16 cur = mfma_chain(a, b[0], c0);
#pragma unroll
    for (int j = 0; j < QT; j++) {
        f32x16 nxt;
        if (j + 1 < QT)
            nxt = mfma_chain(a, b[j + 1], c0);
        fold_tile<K, MASKED>(cur, m1[j], m2[j], valid, h);
        if (!MASKED && j + 1 < QT) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, K == 2 ? 6 : 3, 0); // VALU
            }
        }
        if (j + 1 < QT)
            cur = nxt;
    }
}

// qx / tx: expanded rows of ONE matching problem (8 pieces of 16 bytes per row); both
// buffers must be readable for MF_PAD_ROWS rows past their last row.  Output as in
// knn_partial_body: part[split][rank][part_stride] keys local to the split.
template <int K, int QT>
__device__ __forceinline__ void knn_mfma_body(const uint4 *__restrict__ qx,
                                              const uint4 *__restrict__ tx, int64_t nq, int64_t nt,
                                              int64_t rows_per_split, uint32_t *__restrict__ part,
                                              int64_t part_row0, int64_t part_stride,
                                              unsigned q_block, unsigned split)
{
    __shared__ uint4 lds[2][256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int64_t t0 = (int64_t)split * rows_per_split;
    const int n = (int)max((int64_t)0, min(nt - t0, rows_per_split));
    const int n_tiles = (n + 31) >> 5;
    const int64_t qbase = ((int64_t)q_block * MF_WAVES + wave) * (QT * 32);

    i32x8 b[QT][4];
#pragma unroll
    for (int j = 0; j < QT; j++)
#pragma unroll
        for (int s = 0; s < 4; s++)
            b[j][s] = frag(qx[(qbase + j * 32 + r) * 8 + 2 * s + h]);

    // accumulator register i of a lane in half h is train row (i & 3) + 8 (i >> 2) + 4 h
    f32x16 c0;
#pragma unroll
    for (int i = 0; i < 16; i++)
        c0[i] = 4194304.0f + 31.0f - (float)((i & 3) + 8 * (i >> 2) + 4 * h);

    float m1[QT], m2[QT];
#pragma unroll
    for (int j = 0; j < QT; j++)
        m1[j] = m2[j] = MF_NONE;

    // thread (wave w, lane l) stages piece 2 w + (l >> 5) of tile row l & 31 into slot
    // [w][l]: k-step s of the tile is then the 64 consecutive slots [s][0..63].
    const uint4 *__restrict__ src = tx + (t0 + r) * 8 + 2 * wave + h;
    uint4 stage = make_uint4(0, 0, 0, 0);
    if (n_tiles > 0)
        stage = src[0];
    for (int t = 0; t < n_tiles; t++) {
        lds[t & 1][tid] = stage;
        if (t + 1 < n_tiles)
            stage = src[(int64_t)(t + 1) * 256];
        __syncthreads();
        i32x8 a[4];
#pragma unroll
        for (int s = 0; s < 4; s++)
            a[s] = frag(lds[t & 1][s * 64 + lane]);
        const int valid = n - t * 32;
        if (valid >= 32)
            mfma_tile<K, QT, false>(a, b, c0, m1, m2, 32, h);
        else
            mfma_tile<K, QT, true>(a, b, c0, m1, m2, valid, h);
    }

    // the two lane halves hold disjoint rows of the same tiles for the same queries
#pragma unroll
    for (int j = 0; j < QT; j++) {
        const float o1 = __shfl_xor(m1[j], 32), o2 = __shfl_xor(m2[j], 32);
        const float hi = fmaxf(m1[j], o1), lo = fminf(m1[j], o1);
        m2[j] = fmaxf(lo, fmaxf(m2[j], o2));
        m1[j] = hi;
    }
    if (h != 0)
        return;
    const int top = 32 * (n_tiles - 1) + 31;
#pragma unroll
    for (int j = 0; j < QT; j++) {
        const int64_t qi = qbase + j * 32 + r;
        if (qi >= nq)
            continue;
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        if (m1[j] >= 0.0f) {
            const uint32_t u = (uint32_t)m1[j];
            k1 = ((256u - (u >> 15)) << KNN_KEY_SHIFT) | (uint32_t)(top - (int)(u & 32767u));
        }
        if (K == 2 && m2[j] >= 0.0f) {
            const uint32_t u = (uint32_t)m2[j];
            k2 = ((256u - (u >> 15)) << KNN_KEY_SHIFT) | (uint32_t)(top - (int)(u & 32767u));
        }
        uint32_t *p1 = part + ((int64_t)split * 2) * part_stride + part_row0;
        p1[qi] = k1;
        p1[part_stride + qi] = k2;
    }
}

template <int K, int QT>
__global__ __launch_bounds__(MF_WAVES * 64) void knn_mfma_kernel(
    const uint4 *__restrict__ qx, const uint4 *__restrict__ tx, int64_t nq, int64_t nt,
    int64_t rows_per_split, uint32_t *__restrict__ part)
{
    // XCD-contiguous order of the (split, query block) space, as in knn_partial_kernel
    const unsigned X = gridDim.x, N = X * gridDim.y;
    const unsigned L = blockIdx.x + X * blockIdx.y;
    const unsigned base = N >> 3, rem = N & 7, c = L & 7;
    const unsigned nl = c * base + min(c, rem) + (L >> 3);
    knn_mfma_body<K, QT>(qx, tx, nq, nt, rows_per_split, part, 0, nq, nl % X, nl / X);
}

// ---- launchers ---------------------------------------------------------------------

size_t knn_mfma_expanded_bytes(int64_t rows)
{
    return (size_t)(rows + MF_PAD_ROWS) * 128;
}

hipError_t knn_mfma_expand_launch(const void *d_rows, int64_t rows, void *d_expanded,
                                  hipStream_t stream)
{
    const int64_t n = rows * 8;
    if (n <= 0)
        return hipSuccess;
    fp4_expand_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(
        (const uint32_t *)d_rows, n, (uint4 *)d_expanded);
    return hipGetLastError();
}

MfmaPlan knn_mfma_plan_splits(int64_t nq, int64_t nt, int q_tiles, int64_t splits)
{
    MfmaPlan p;
    p.q_tiles = q_tiles;
    p.q_blocks = (nq + 32 * q_tiles * MF_WAVES - 1) / (32 * q_tiles * MF_WAVES);
    if (splits < 1)
        splits = 1;
    int64_t rows = (nt + splits - 1) / splits;
    rows = (rows + 31) & ~(int64_t)31;
    if (rows < 32)
        rows = 32;
    if (rows > MF_MAX_ROWS_PER_SPLIT)
        rows = MF_MAX_ROWS_PER_SPLIT;
    p.rows_per_split = rows;
    p.n_splits = (int)((nt + rows - 1) / rows);
    if (p.n_splits < 1)
        p.n_splits = 1;
    p.part_bytes = (size_t)p.n_splits * 2 * (size_t)nq * sizeof(uint32_t);
    return p;
}

hipError_t knn_mfma_partial_launch(const MfmaPlan &p, const void *d_qx, int64_t nq,
                                   const void *d_tx, int64_t nt, int k, uint32_t *d_part,
                                   hipStream_t stream)
{
    if (p.n_splits < 1 || p.n_splits > 65535 || p.rows_per_split > MF_MAX_ROWS_PER_SPLIT ||
        (p.rows_per_split & 31) != 0 || (uint64_t)p.q_blocks * (uint64_t)p.n_splits >= (1ull << 31))
        return hipErrorInvalidValue;
    dim3 grid((unsigned)p.q_blocks, (unsigned)p.n_splits);
    const uint4 *q4 = (const uint4 *)d_qx, *t4 = (const uint4 *)d_tx;
#define POSEHIP_MF_LAUNCH(KK, QQ)                                                               \
    knn_mfma_kernel<KK, QQ><<<grid, MF_WAVES * 64, 0, stream>>>(q4, t4, nq, nt,                \
                                                                p.rows_per_split, d_part)
    if (p.q_tiles == 4 && k == 2)
        POSEHIP_MF_LAUNCH(2, 4);
    else if (p.q_tiles == 4)
        POSEHIP_MF_LAUNCH(1, 4);
    else if (p.q_tiles == 2 && k == 2)
        POSEHIP_MF_LAUNCH(2, 2);
    else if (p.q_tiles == 2)
        POSEHIP_MF_LAUNCH(1, 2);
    else if (p.q_tiles == 8 && k == 2)
        POSEHIP_MF_LAUNCH(2, 8);
    else if (p.q_tiles == 8)
        POSEHIP_MF_LAUNCH(1, 8);
    else
        return hipErrorInvalidValue;
#undef POSEHIP_MF_LAUNCH
    return hipGetLastError();
}

} // namespace posehip
