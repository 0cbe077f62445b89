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
//     C[row] = 2^22 + 31 - row_in_tile     (+ 32 in even tiles, see below)
// turn the accumulator itself into an ordering key,
//     key = (256 - hamming) * 2^15 + (31 - row_in_tile)            (an integer < 2^24),
// whose MAXIMUM over a tile is the nearest train row, lowest row first among equals.  The
// running best / second best of a lane carry an index field relative to the PAIR of tiles
// being processed (even tiles use C + 32, so a pair spans 64 index steps): 64 is added to
// both before every pair (two VALU ops), which keeps earlier tiles ahead of later ones,
// and the sixteen keys of a tile are folded in with v_max3 / v_med3 (1.25 ops per key).
// The VALU therefore spends 21 instructions per 16 keys where the popcount kernel spends
// 292.
//
// A workgroup is 4 waves; wave w expands its QT tiles of 32 queries to FP4 once and holds
// them as B fragments in registers for the whole kernel; the train rows are expanded by a
// pre-pass (128 bytes per row), all four waves stream the same train tiles, which an LDS-DMA ring
// (global_load_lds, no VGPRs) stages in fragment order three tiles ahead.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

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
__device__ __forceinline__ uint4 fp4_from_bits32(uint32_t w)
{
    return make_uint4(fp4_from_bits8(w & 0xFF), fp4_from_bits8((w >> 8) & 0xFF),
                      fp4_from_bits8((w >> 16) & 0xFF), fp4_from_bits8(w >> 24));
}

__global__ __launch_bounds__(256) void fp4_expand_kernel(const uint32_t *__restrict__ rows,
                                                         int64_t n_dwords,
                                                         uint4 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_dwords)
        out[i] = fp4_from_bits32(rows[i]);
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

// Folds the 16 keys of one (train tile, query tile) into the lane's running pair.  AGE: the
// keys belong to an even train tile, the first of its pair: the running keys first move one
// pair of tiles (64 index steps) back.
template <int K, bool MASKED, bool AGE>
__device__ __forceinline__ void fold_tile(f32x16 acc, float &m1, float &m2, int valid, int h)
{
    if constexpr (MASKED) {
#pragma unroll
        for (int i = 0; i < 16; i++)
            if ((i & 3) + 8 * (i >> 2) + 4 * h >= valid)
                acc[i] = MF_NONE;
    }
    if constexpr (AGE) {
        m1 += 64.0f;
        if constexpr (K == 2)
            m2 += 64.0f;
    }
#pragma unroll
    for (int g = 0; g < 4; g++)
        top2_max4<K>(m1, m2, acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
}

// One train tile against the wave's QT query tiles, software-pipelined with a lag of
// two: while the four MFMAs of query tile j run, the VALU folds the keys of query tile
// j - 2, so that no fold starts within the latency of the MFMA that produced its keys.
// The keys of the tile's last two query tiles stay in pend0 / pend1 (older first) and are
// folded under the first MFMAs of the NEXT train tile; on entry they hold the previous
// train tile's.  LAST: this is the split's last tile, whose rows at or beyond `valid`
// are taken out before the fold (the pending keys of the tile before are complete).
// RELOAD: once every MFMA is issued, fetch the next tile's fragments.  The scheduler is
// told to keep one MFMA per five or six VALU instructions, which is what the MFMA pipe
// (32 cycles each) can hide.
// Scheduling pattern of one chain of four MFMAs with the fold that runs beside it (AGE:
// that fold has two more VALU instructions).
template <int K, bool AGE> __device__ __forceinline__ void sched_chain()
{
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, K == 2 ? (AGE ? 6 : 5) : (AGE ? 3 : 2), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, K == 2 ? (AGE ? 6 : 5) : 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, K == 2 ? 5 : 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, K == 2 ? 5 : 2, 0);
}

template <int K, int QT, bool LAST, bool RELOAD, int PARITY>
__device__ __forceinline__ void mfma_tile(i32x8 (&a)[4], const i32x8 (&b)[QT][4],
                                          const f32x16 &c, float (&m1)[QT], float (&m2)[QT],
                                          f32x16 &pend0, f32x16 &pend1, int valid, int h,
                                          const uint4 *next_frags)
{
    static_assert(QT >= 2, "the fold lags two query tiles behind");
#pragma unroll
    for (int j = 0; j < QT; j++) {
        const f32x16 acc = mfma_chain(a, b[j], c);
        const int jf = (j + QT - 2) % QT; // the query tile pend0 belongs to
        // j < 2: keys of the previous train tile (other parity, complete); else this tile's
        if (j < 2)
            fold_tile<K, false, PARITY == 1>(pend0, m1[jf], m2[jf], valid, h);
        else
            fold_tile<K, LAST, PARITY == 0>(pend0, m1[jf], m2[jf], valid, h);
        pend0 = pend1;
        pend1 = acc;
    }
    if constexpr (RELOAD) {
#pragma unroll
        for (int s = 0; s < 4; s++)
            a[s] = frag(next_frags[s * 64]);
    }
    if constexpr (!LAST) {
        sched_chain<K, PARITY == 1>();
        sched_chain<K, PARITY == 1>();
#pragma unroll
        for (int j = 2; j < QT; j++)
            sched_chain<K, PARITY == 0>();
    }
}

// The folds still pending after a split's last tile (both belong to that tile).
template <int K, int QT, int PARITY>
__device__ __forceinline__ void fold_pending(float (&m1)[QT], float (&m2)[QT], const f32x16 &pend0,
                                             const f32x16 &pend1, int valid, int h)
{
    fold_tile<K, true, PARITY == 0>(pend0, m1[QT - 2], m2[QT - 2], valid, h);
    fold_tile<K, true, PARITY == 0>(pend1, m1[QT - 1], m2[QT - 1], valid, h);
}

// 16 bytes per lane from global memory straight into LDS (no VGPR in between): the wave's
// 64 pieces land at lds_wave_base + 16 * lane.
__device__ __forceinline__ void load_to_lds(const uint4 *g, uint4 *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)g,
        (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// q_raw: the nq >= 1 query rows of ONE matching problem as they are (32 bytes each); tx:
// the FP4 image of its train rows (8 pieces of 16 bytes per row), readable for
// MF_PAD_ROWS rows past the last one.  Output as in
// knn_partial_body: part[split][rank][part_stride] keys local to the split.
//
// With a single split there is nothing to merge: out.idx != nullptr makes the kernel write
// the final k-NN rows itself (row part_row0 + i of the outputs) instead of keys.
struct DirectOut {
    int32_t *idx, *dist; // nullptr: write per-split keys to `part`
    uint8_t *keep;
    float ratio;
    int32_t index_base;
};

template <int K, int QT>
__device__ __forceinline__ void knn_mfma_body(const uint4 *__restrict__ q_raw,
                                              const uint4 *__restrict__ tx, int64_t nq, int64_t nt,
                                              int64_t rows_per_split, uint32_t *__restrict__ part,
                                              int64_t part_row0, int64_t part_stride,
                                              unsigned q_block, unsigned split, const DirectOut &out)
{
    __shared__ uint4 lds[4][256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r = lane & 31;
    const int64_t t0 = (int64_t)split * rows_per_split;
    const int n = (int)max((int64_t)0, min(nt - t0, rows_per_split));
    const int n_tiles = (n + 31) >> 5;
    const int64_t qbase = ((int64_t)q_block * MF_WAVES + wave) * (QT * 32);

    // LDS-DMA ring of four slots: thread (wave w, lane l) stages piece 2 w + (l >> 5) of tile
    // row l & 31 into slot [w][l], so that k-step s of a tile is the 64 consecutive slots
    // [s][0..63].  Every wave issues ONE load per tile, for tiles 0 .. n_tiles + 2 (the reads
    // past the split stay inside the buffer's padding), so the count of loads in flight is the
    // same at every wait.  The first four go out before anything else, under the query loads.
    const uint4 *__restrict__ src = tx + (t0 + r) * 8 + 2 * wave + h;
    if (n_tiles > 0) {
#pragma unroll
        for (int t = 0; t < 4; t++)
            load_to_lds(src + (int64_t)t * 256, &lds[t][wave * 64]);
    }

    // B fragments, expanded here from the raw rows (two uint4 per row): lane half h of
    // k-step s holds piece 2 s + h, the FP4 image of dword 2 s + h of the row.  Rows past
    // the problem's last are clamped (their results are never written).
    i32x8 b[QT][4];
#pragma unroll
    for (int j = 0; j < QT; j++) {
        const int64_t row = min(qbase + j * 32 + r, nq - 1);
        const uint4 lo = q_raw[2 * row], hi = q_raw[2 * row + 1];
        const uint32_t w[4] = {h ? lo.y : lo.x, h ? lo.w : lo.z, h ? hi.y : hi.x, h ? hi.w : hi.z};
#pragma unroll
        for (int s = 0; s < 4; s++)
            b[j][s] = frag(fp4_from_bits32(w[s]));
    }

    // accumulator register i of a lane in half h is train row (i & 3) + 8 (i >> 2) + 4 h
    // (even tiles carry 32 more in the index field than the odd tile of their pair)
    f32x16 c0, c1;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        c1[i] = 4194304.0f + 31.0f - (float)((i & 3) + 8 * (i >> 2) + 4 * h);
        c0[i] = c1[i] + 32.0f;
    }

    float m1[QT], m2[QT];
#pragma unroll
    for (int j = 0; j < QT; j++)
        m1[j] = m2[j] = MF_NONE;
    f32x16 pend0, pend1; // keys awaiting their fold; "none" before the first tile
#pragma unroll
    for (int i = 0; i < 16; i++)
        pend0[i] = pend1[i] = MF_NONE;

    if (n_tiles > 0) {
        // (tiles 0..3 were requested before the queries; the query loads behind them have
        // returned, so have they)
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        i32x8 a[4];
#pragma unroll
        for (int s = 0; s < 4; s++)
            a[s] = frag(lds[0][s * 64 + lane]);
        // Iteration t: wait for the wave's own piece of tile t + 1 (two younger loads stay in
        // flight), barrier -- tile t + 1 is now complete and every wave has taken its
        // fragments of tile t, whose slot the load of tile t + 4 may overwrite.
        auto step = [&](int t, auto parity) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            load_to_lds(src + (int64_t)(t + 4) * 256, &lds[t & 3][wave * 64]);
            mfma_tile<K, QT, false, true, decltype(parity)::value>(
                a, b, decltype(parity)::value ? c1 : c0, m1, m2, pend0, pend1, 32, h,
                &lds[(t + 1) & 3][lane]);
        };
        int t = 0;
        for (; t + 2 < n_tiles; t += 2) {
            step(t, std::integral_constant<int, 0>());
            step(t + 1, std::integral_constant<int, 1>());
        }
        // one or two tiles are left; the last one (its fragments are already in registers
        // when its turn comes) may be partial
        const int valid = n - 32 * (n_tiles - 1);
        if (t + 1 < n_tiles) {
            step(t, std::integral_constant<int, 0>());
            mfma_tile<K, QT, true, false, 1>(a, b, c1, m1, m2, pend0, pend1, valid, h, nullptr);
            fold_pending<K, QT, 1>(m1, m2, pend0, pend1, valid, h);
        } else {
            mfma_tile<K, QT, true, false, 0>(a, b, c0, m1, m2, pend0, pend1, valid, h, nullptr);
            fold_pending<K, QT, 0>(m1, m2, pend0, pend1, valid, h);
        }
        // no load may still be writing LDS when the workgroup gives it back
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    const int top = 64 * ((n_tiles - 1) >> 1) + 63; // index field 0 = last row of the last pair
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
        if (out.idx == nullptr) {
            uint32_t *p1 = part + ((int64_t)split * 2) * part_stride + part_row0;
            p1[qi] = k1;
            p1[part_stride + qi] = k2;
            continue;
        }
        // a single split: these ARE the k-NN rows (same formats as knn_merge_body writes)
        const uint32_t mask = (1u << KNN_KEY_SHIFT) - 1;
        const int64_t o = part_row0 + qi;
        const bool v1 = k1 != 0xFFFFFFFFu, v2 = k2 != 0xFFFFFFFFu;
        const int32_t i1 = v1 ? (int32_t)(k1 & mask) + out.index_base : -1;
        const int32_t d1 = v1 ? (int32_t)(k1 >> KNN_KEY_SHIFT) : 0x7FFFFFFF;
        if (K == 1) {
            out.idx[o] = i1;
            out.dist[o] = d1;
            continue;
        }
        const int32_t i2 = v2 ? (int32_t)(k2 & mask) + out.index_base : -1;
        const int32_t d2 = v2 ? (int32_t)(k2 >> KNN_KEY_SHIFT) : 0x7FFFFFFF;
        *reinterpret_cast<int2 *>(out.idx + 2 * o) = make_int2(i1, i2);
        *reinterpret_cast<int2 *>(out.dist + 2 * o) = make_int2(d1, d2);
        if (out.keep)
            out.keep[o] = (v1 && v2 && (float)d1 < out.ratio * (float)d2) ? 1 : 0;
    }
}

template <int K, int QT>
__global__ __launch_bounds__(MF_WAVES * 64) void knn_mfma_kernel(
    const uint4 *__restrict__ q_raw, const uint4 *__restrict__ tx, int64_t nq, int64_t nt,
    int64_t rows_per_split, uint32_t *__restrict__ part, DirectOut out)
{
    // XCD-contiguous order of the (split, query block) space, as in knn_partial_kernel
    const unsigned X = gridDim.x, N = X * gridDim.y;
    const unsigned L = blockIdx.x + X * blockIdx.y;
    const unsigned base = N >> 3, rem = N & 7, c = L & 7;
    const unsigned nl = c * base + min(c, rem) + (L >> 3);
    knn_mfma_body<K, QT>(q_raw, tx, nq, nt, rows_per_split, part, 0, nq, nl % X, nl / X, out);
}

// A batch of independent matching problems (grid.z = frame), as knn_partial_frames_kernel:
// frame f matches query rows [q_off[f], q_off[f+1]) against train rows [t_off[f],
// t_off[f+1]).  q_raw: the query buffer itself; tx: the FP4 image of the train rows from
// t_row0 on.
#ifdef POSEHIP_MF_CLOCKS
// measurement build (tools/mfma_probe): shader cycles and 100 MHz wall ticks one workgroup spent
__device__ unsigned long long mf_clock_probe[2];
#endif

template <int K, int QT>
__global__ __launch_bounds__(MF_WAVES * 64) void knn_mfma_frames_kernel(
    const uint4 *__restrict__ q_raw, const uint4 *__restrict__ tx,
    const int32_t *__restrict__ q_off, const int32_t *__restrict__ t_off, int64_t t_row0,
    int64_t total_nq, uint32_t *__restrict__ part, int xcd_chunks, DirectOut out)
{
    unsigned bx = blockIdx.x, by = blockIdx.y, f = blockIdx.z;
    if (xcd_chunks) {
        const unsigned X = gridDim.x, Y = gridDim.y, N = X * Y * gridDim.z;
        const unsigned L = bx + X * (by + Y * f);
        const unsigned base = N >> 3, rem = N & 7, c = L & 7;
        const unsigned nl = c * base + min(c, rem) + (L >> 3); // bijective for any N
        bx = nl % X;
        const unsigned t = nl / X;
        by = t % Y;
        f = t / Y;
    }
    const int64_t q0 = q_off[f], nq = q_off[f + 1] - q0;
    const int64_t t0 = t_off[f], nt = t_off[f + 1] - t0;
    if ((int64_t)bx * (MF_WAVES * QT * 32) >= nq)
        return;
#ifdef POSEHIP_MF_CLOCKS
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#endif
    knn_mfma_body<K, QT>(q_raw + 2 * q0, tx + 8 * (t0 - t_row0), nq, nt,
                         mfma_rows_per_split(nt, (int)gridDim.y), part, q0, total_nq, bx, by, out);
#ifdef POSEHIP_MF_CLOCKS
    if (blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 3 && threadIdx.x == 0) {
        mf_clock_probe[0] = __builtin_readcyclecounter() - c0;
        mf_clock_probe[1] = wall_clock64() - w0;
    }
#endif
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
    const int64_t min_splits = (nt + MF_MAX_ROWS_PER_SPLIT - 1) / MF_MAX_ROWS_PER_SPLIT;
    if (splits < min_splits)
        splits = min_splits;
    if (splits < 1)
        splits = 1;
    p.rows_per_split = mfma_rows_per_split(nt, (int)splits);
    p.n_splits = (int)((nt + p.rows_per_split - 1) / p.rows_per_split);
    if (p.n_splits < 1)
        p.n_splits = 1;
    p.part_bytes = (size_t)p.n_splits * 2 * (size_t)nq * sizeof(uint32_t);
    return p;
}

// Splits: the (query block x split) grid should give every CU several workgroups (two are
// resident per CU) while a split keeps at least 16 tiles, whose MFMAs hide the start-up
// and the final key hand-over.
static int64_t mfma_pick_splits(int64_t q_blocks_total, int64_t nt, int n_cus)
{
    int64_t splits = (5 * (int64_t)n_cus + q_blocks_total - 1) / q_blocks_total;
    const int64_t by_rows = nt / 512;
    if (splits > by_rows)
        splits = by_rows;
    return splits < 1 ? 1 : splits;
}

MfmaPlan knn_mfma_plan(int64_t nq, int64_t nt, int n_cus)
{
    const int qt = nq <= 2048 ? 2 : 4;
    const int64_t q_blocks = (nq + 32 * qt * MF_WAVES - 1) / (32 * qt * MF_WAVES);
    return knn_mfma_plan_splits(nq, nt, qt, mfma_pick_splits(q_blocks, nt, n_cus));
}

MfmaPlan knn_mfma_plan_frames(int n_frames, int64_t max_nq, int64_t max_nt, int64_t total_nq,
                              int n_cus)
{
    const int qt = max_nq <= 256 ? 2 : 4;
    const int64_t q_blocks = (max_nq + 32 * qt * MF_WAVES - 1) / (32 * qt * MF_WAVES);
    MfmaPlan p = knn_mfma_plan_splits(max_nq, max_nt, qt,
                                      mfma_pick_splits(q_blocks * n_frames, max_nt, n_cus));
    p.part_bytes = (size_t)p.n_splits * 2 * (size_t)total_nq * sizeof(uint32_t);
    return p;
}

// Measured on MI355X (tools/path_crossover.py, whole calls on resident inputs): the
// pre-pass and the extra launches cost about 5 us, which the matrix cores win back from
// about 3e7 pairs on (5000 x 5000; 2.1x at 10k x 10k, 5x at 200k x 60k); query sets too
// short to fill the 128-query minimum of a workgroup stay with the popcount kernel.
bool knn_mfma_pays(int64_t nq, int64_t nt)
{
    return nq >= 256 && nt >= 256 && nq * nt >= 30000000;
}

bool knn_mfma_pays_frames(int n_frames, int64_t max_nq, int64_t max_nt)
{
    return max_nq >= 256 && max_nt >= 256 && (int64_t)n_frames * max_nq * max_nt >= 40000000;
}

static hipError_t record(hipEvent_t *events, int i, hipStream_t stream)
{
    return events ? hipEventRecord(events[i], stream) : hipSuccess;
}

hipError_t knn_mfma_launch(const MfmaPlan &p, const void *d_query, int64_t nq, const void *d_train,
                           int64_t nt, int k, int32_t index_base, void *d_tx,
                           uint32_t *d_part, int32_t *d_out_idx, int32_t *d_out_dist,
                           uint8_t *d_keep, float ratio, hipStream_t stream, hipEvent_t *events)
{
    if (p.n_splits < 1 || p.n_splits > 65535 || p.rows_per_split > MF_MAX_ROWS_PER_SPLIT ||
        (p.rows_per_split & 31) != 0 || p.rows_per_split * (int64_t)p.n_splits < nt ||
        (uint64_t)p.q_blocks * (uint64_t)p.n_splits >= (1ull << 31))
        return hipErrorInvalidValue;
    hipError_t e;
    if ((e = record(events, 4, stream)) != hipSuccess)
        return e;
    if (d_train && (e = knn_mfma_expand_launch(d_train, nt, d_tx, stream)) != hipSuccess)
        return e;
    if ((e = record(events, 0, stream)) != hipSuccess)
        return e;
    dim3 grid((unsigned)p.q_blocks, (unsigned)p.n_splits);
    const uint4 *q4 = (const uint4 *)d_query, *t4 = (const uint4 *)d_tx;
    const bool direct = p.n_splits == 1;
    const DirectOut out = {direct ? d_out_idx : nullptr, d_out_dist, d_keep, ratio, index_base};
#define POSEHIP_MF_LAUNCH(KK, QQ)                                                               \
    knn_mfma_kernel<KK, QQ><<<grid, MF_WAVES * 64, 0, stream>>>(q4, t4, nq, nt,                \
                                                                p.rows_per_split, d_part, out)
    if (p.q_tiles == 4 && k == 2)
        POSEHIP_MF_LAUNCH(2, 4);
    else if (p.q_tiles == 4)
        POSEHIP_MF_LAUNCH(1, 4);
    else if (p.q_tiles == 2 && k == 2)
        POSEHIP_MF_LAUNCH(2, 2);
    else if (p.q_tiles == 2)
        POSEHIP_MF_LAUNCH(1, 2);
    else
        return hipErrorInvalidValue;
#undef POSEHIP_MF_LAUNCH
    if ((e = hipGetLastError()) != hipSuccess)
        return e;
    if ((e = record(events, 1, stream)) != hipSuccess)
        return e;
    if (!direct && (e = knn_merge_launch(d_part, nq, p.n_splits, p.rows_per_split, k, index_base,
                                         d_out_idx, d_out_dist, d_keep, ratio, stream)) != hipSuccess)
        return e;
    if ((e = record(events, 2, stream)) != hipSuccess)
        return e;
    return record(events, 3, stream);
}

hipError_t knn_mfma_frames_launch(const MfmaPlan &p, const void *d_query, const int32_t *d_q_off,
                                  int64_t q_end, const void *d_train,
                                  const int32_t *d_t_off, int64_t t_begin, int64_t t_end,
                                  int n_frames, int k, void *d_tx, uint32_t *d_part,
                                  int32_t *d_out_idx, int32_t *d_out_dist, uint8_t *d_keep,
                                  float ratio, hipStream_t stream, hipEvent_t *events,
                                  bool uniform_frames)
{
    if (p.n_splits < 1 || p.n_splits > 65535 || n_frames < 1 || n_frames > 65535 ||
        p.q_blocks < 1 || p.q_blocks > 0x7FFFFFFF)
        return hipErrorInvalidValue;
    const int xcd = uniform_frames && n_frames >= 8 &&
                            (uint64_t)p.q_blocks * (uint64_t)p.n_splits * (uint64_t)n_frames < (1ull << 31)
                        ? 1 : 0;
    hipError_t e;
    if ((e = record(events, 4, stream)) != hipSuccess)
        return e;
    if ((e = knn_mfma_expand_launch((const uint8_t *)d_train + t_begin * 32, t_end - t_begin, d_tx,
                                    stream)) != hipSuccess)
        return e;
    if ((e = record(events, 0, stream)) != hipSuccess)
        return e;
    dim3 grid((unsigned)p.q_blocks, (unsigned)p.n_splits, (unsigned)n_frames);
    const uint4 *q4 = (const uint4 *)d_query, *t4 = (const uint4 *)d_tx;
    const bool direct = p.n_splits == 1;
    const DirectOut out = {direct ? d_out_idx : nullptr, d_out_dist, d_keep, ratio, 0};
#define POSEHIP_MF_LAUNCH(KK, QQ)                                                               \
    knn_mfma_frames_kernel<KK, QQ><<<grid, MF_WAVES * 64, 0, stream>>>(                         \
        q4, t4, d_q_off, d_t_off, t_begin, q_end, d_part, xcd, out)
    if (p.q_tiles == 4 && k == 2)
        POSEHIP_MF_LAUNCH(2, 4);
    else if (p.q_tiles == 4)
        POSEHIP_MF_LAUNCH(1, 4);
    else if (p.q_tiles == 2 && k == 2)
        POSEHIP_MF_LAUNCH(2, 2);
    else if (p.q_tiles == 2)
        POSEHIP_MF_LAUNCH(1, 2);
    else
        return hipErrorInvalidValue;
#undef POSEHIP_MF_LAUNCH
    if ((e = hipGetLastError()) != hipSuccess)
        return e;
    if ((e = record(events, 1, stream)) != hipSuccess)
        return e;
    const int64_t max_nq = p.q_blocks * (32 * p.q_tiles * MF_WAVES);
    if (!direct &&
        (e = knn_merge_frames_launch(d_part, d_q_off, d_t_off, q_end, max_nq, n_frames, p.n_splits,
                                     -1, k, d_out_idx, d_out_dist, d_keep, ratio, stream)) !=
            hipSuccess)
        return e;
    if ((e = record(events, 2, stream)) != hipSuccess)
        return e;
    return record(events, 3, stream);
}

// ---- FP4 MFMA peak ----------------------------------------------------------------
// Nothing but the distance kernel's MFMA: four independent chains of four
// v_mfma_scale_f32_32x32x64_f8f6f4 per iteration (the same scales, FP4 operands of +-4 at
// random, held in registers), two waves per SIMD on every CU as in the distance kernel.
// What it sustains is the ceiling the chip gives this instruction at the clock it holds
// under that load; the dense-FP4 spec figure assumes the nominal clock.
__global__ __launch_bounds__(256) void mfma_peak_kernel(float *out, unsigned seed, int iters)
{
    // operand nibbles 0x6 (+4) / 0xE (-4), sign bits from a hash of the thread id
    unsigned x = (threadIdx.x + 1u) * 2654435761u ^ seed;
    i32x8 a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        unsigned w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            x ^= x << 13;
            x ^= x >> 17;
            x ^= x << 5;
            w[i] = 0x66666666u | (x & 0x88888888u);
        }
        a[s] = i32x8{(int)w[0], (int)w[1], (int)w[2], (int)w[3], 0, 0, 0, 0};
        b[s] = i32x8{(int)w[4], (int)w[5], (int)w[6], (int)w[7], 0, 0, 0, 0};
    }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 16; i++)
            acc[j][i] = (float)(i + j);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int j = 0; j < 4; j++) // consecutive MFMAs belong to different chains
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                    a[s], b[(s + j) & 3], acc[j], 4, 4, 0, MF_SCALE_A, 0, MF_SCALE_B);
    }
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 16; i++)
            r += acc[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

hipError_t mfma_fp4_peak(int n_cus, int iters, float *d_scratch, hipStream_t stream,
                         double *flop_per_s)
{
    const int blocks = n_cus * 2; // 2 workgroups x 4 waves per CU = 2 waves per SIMD
    hipEvent_t e0, e1;
    hipError_t e;
    if ((e = hipEventCreate(&e0)) != hipSuccess)
        return e;
    if ((e = hipEventCreate(&e1)) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return e;
    }
    float best = 0.f;
    for (int r = 0; r < 8; r++) { // the first passes settle the clock under this load
        (void)hipEventRecord(e0, stream);
        mfma_peak_kernel<<<blocks, 256, 0, stream>>>(d_scratch, 0x9e3779b9u + r, iters);
        (void)hipEventRecord(e1, stream);
        if ((e = hipEventSynchronize(e1)) != hipSuccess)
            break;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 4 && (best == 0.f || ms < best))
            best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess)
        return e;
    // one MFMA = 32 x 32 x 64 multiply-adds
    *flop_per_s = (double)blocks * 4.0 * (double)iters * 16.0 * 2.0 * 32 * 32 * 64 / (best * 1e-3);
    return hipGetLastError();
}

} // namespace posehip
