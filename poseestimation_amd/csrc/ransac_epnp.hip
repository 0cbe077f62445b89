// RANSAC-EPnP hypothesis generation and inlier scoring for gfx950 (MI355X).
//
// Arithmetic replaced: cv::solvePnPRansac(..., SOLVEPNP_EPNP), the pose stage of
// the reference's EPnP branch (reference docs/index.md:75,120; TODO at
// src/cereal_box.cpp:292-295).  Per RANSAC iteration OpenCV solves EPnP on a
// 5-point sample (four control points, 12 x 12 null-space problem by one-sided
// Jacobi SVD, three beta initialisations each refined by 5 Gauss-Newton steps,
// pose from the best one), converts R to a rotation vector, and counts the
// points whose float squared reprojection error is <= threshold^2.
//
// Mapping.  The EPnP solve is a sequential fp64 computation: every operation is in
// the order of a plain sequential statement of the algorithm (built with
// -ffp-contract=off) so that poses agree with a CPU statement to the last bits.
// A hypothesis keeps its small matrices (12 x 12 null-space problem, 6 x 10
// system, ...) in LDS, interleaved over the hypotheses of a wave (element e of
// hypothesis h at word e*HPW + h, so a wave access is conflict-free), 1248 bytes
// per hypothesis: LDS holds 128 hypotheses per CU, i.e. 32 or fewer per wave, so
// a hypothesis is served by a GROUP OF 2..8 LANES.  The lanes of a group run the
// same instructions on the same LDS copy and split the work the sequential order
// leaves independent - the Jacobi pairs (i, j) with equal i + j, which touch
// different rows, and the three candidates' Gauss-Newton and pose steps - each
// piece with the operations and operands it has in the sequential order.
// Per-point quantities (barycentric coordinates, camera-frame points) are
// recomputed where they are used instead of stored, so the same routine refits
// on any number of inliers in O(1) memory.
// Where few problems are in flight (a lone small call; the refit on the inliers) a
// problem gets a workgroup of three waves, one per candidate, each on its own LDS copy.
// Scoring is a second kernel with TWO HYPOTHESES PER WAVEFRONT at a time: the 64 lanes
// stride over the correspondences, a packed-fp32 first pass with a rigorous error bound
// settles almost every (row, model) verdict and the exact fp64 evaluation runs only for
// the batches it cannot settle; the inlier count is reduced with ballot + popcount.
#include <hip/hip_runtime.h>

#include <atomic>
#include <float.h>
#include <stdint.h>

#include "ransac_epnp.h"

namespace posehip {

namespace {

// Cycle stamps for the profiling tool (tools/pnp_stamps.hip defines PNP_STAMPS
// and the array before including this file); compiled out of the library.
#ifdef PNP_STAMPS
#define PNP_STAMP(k)                                                                              \
    do {                                                                                          \
        if (blockIdx.x == 0 && threadIdx.x == 0)                                                  \
            g_pnp_stamps[k] = clock64();                                                          \
    } while (0)
#else
#define PNP_STAMP(k)                                                                              \
    do {                                                                                          \
    } while (0)
#endif

// ---- cv::RNG (multiply-with-carry), used by the SVD's zero-row completion ----
struct Rng {
    uint64_t state;
    __device__ uint32_t next()
    {
        state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
        return (uint32_t)state;
    }
};

// hypot as OpenCV's JacobiSVD computes it: lapack.cpp has a function template hypot of its own
// in namespace cv, which the unqualified call in JacobiSVDImpl_ resolves to (it hides the C
// library's): the larger magnitude times sqrt(1 + ratio^2).  Written with selects, so that
// it is straight-line code; every lane performs exactly the division, product, sum, square
// root and product of its own case:
//   a = |a|, b = |b|;  a > b: b /= a, a * sqrt(1 + b*b);  else b > 0: a /= b, b * sqrt(1 + a*a);
//   else 0.
__device__ __forceinline__ double cv_hypot(double a, double b)
{
    a = fabs(a);
    b = fabs(b);
    const bool a_larger = a > b;
    const double big = a_larger ? a : b, small = a_larger ? b : a;
    const double q = small / big;
    const double h = big * sqrt(1 + q * q);
    return (a_larger || b > 0) ? h : 0.0;
}

// Rotation (c, s) of a Jacobi pair:
//   beta < 0:  delta = (gamma - beta) * 0.5;  s = sqrt(delta / gamma);  c = p / (gamma * s * 2)
//   else:      c = sqrt((gamma + beta) / (gamma * 2));                  s = p / (gamma * c * 2)
// written with selects around ONE square root and two divisions: the lanes of a wave
// split about evenly on the sign of beta, and as a branch both sides (a square root and
// two divisions each) would be issued for every rotation.  Each lane performs exactly
// the operations of its own case, so the results are unchanged.
__device__ __forceinline__ void jacobi_rotation(double p, double beta, double gamma, double &c,
                                                double &s)
{
    const bool neg = beta < 0;
    const double num = neg ? (gamma - beta) * 0.5 : gamma + beta;
    const double den = neg ? gamma : gamma * 2;
    const double r = sqrt(num / den);
    const double o = p / (gamma * r * 2);
    c = neg ? o : r;
    s = neg ? r : o;
}

// Pointer into a lane's interleaved LDS arrays: consecutive elements of one
// lane are ST words apart (ST = hypotheses per workgroup).
// The pointer is typed as LDS: left generic, the compiler loses track of the address
// space in the larger instantiations and falls back to flat loads, which cost several
// times an LDS read.
typedef __attribute__((address_space(3))) double lds_double;
template <int ST> struct LP {
    lds_double *p;
    __device__ lds_double &operator[](int e) const { return p[e * ST]; }
    __device__ lds_double &operator*() const { return *p; }
    __device__ LP operator+(int e) const { return LP{p + e * ST}; }
    __device__ LP &operator+=(int e)
    {
        p += e * ST;
        return *this;
    }
    __device__ LP &operator++(int)
    {
        p += ST;
        return *this;
    }
};

// Orders the LDS traffic of the lanes of one wave: what any lane wrote before this point
// is what every lane reads after it (the LDS serves a wave's instructions in order; this
// keeps the compiler from moving accesses across the point).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One pair of a fixed-size decomposition (rows i and j of At, length M; rows of Vt, length
// N), exactly as JacobiSVDImpl_ states it.  True if the pair was rotated.
template <int ST, bool WITH_V, int M, int N>
__device__ __forceinline__ bool jacobi_small_pair(LP<ST> At, LP<ST> W, LP<ST> Vt, int i, int j,
                                                  double eps)
{
    const LP<ST> Ai = At + i * M, Aj = At + j * M;
    double a = W[i], p = 0, b = W[j], c, s;
    double ai[M], aj[M];
#pragma unroll
    for (int k = 0; k < M; k++) {
        ai[k] = Ai[k];
        aj[k] = Aj[k];
    }
#pragma unroll
    for (int k = 0; k < M; k++)
        p += ai[k] * aj[k];
    if (fabs(p) <= eps * sqrt(a * b))
        return false;
    p *= 2;
    const double beta = a - b, gamma = cv_hypot(p, beta);
    jacobi_rotation(p, beta, gamma, c, s);
    a = b = 0;
#pragma unroll
    for (int k = 0; k < M; k++) {
        const double t0 = c * ai[k] + s * aj[k];
        const double t1 = -s * ai[k] + c * aj[k];
        Ai[k] = t0;
        Aj[k] = t1;
        a += t0 * t0;
        b += t1 * t1;
    }
    W[i] = a;
    W[j] = b;
    if (WITH_V) {
        const LP<ST> Vi = Vt + i * N, Vj = Vt + j * N;
        double vi[N], vj[N];
#pragma unroll
        for (int k = 0; k < N; k++) {
            vi[k] = Vi[k];
            vj[k] = Vj[k];
        }
#pragma unroll
        for (int k = 0; k < N; k++) {
            Vi[k] = c * vi[k] + s * vj[k];
            Vj[k] = -s * vi[k] + c * vj[k];
        }
    }
    return true;
}

// jacobi_small_pair by TWO lanes: both read the rows, test and compute the rotation; the
// first then rotates and stores row i of At (with its norm) and of Vt, the second
// (second == true) row j.  A row's elements and norm are computed by the operations of
// jacobi_small_pair in their order, so the result is bit-identical.
template <int ST, bool WITH_V, int M, int N>
__device__ __forceinline__ bool jacobi_small_pair_half(LP<ST> At, LP<ST> W, LP<ST> Vt, int i, int j,
                                                       double eps, bool second)
{
    const LP<ST> Ai = At + i * M, Aj = At + j * M;
    double a = W[i], p = 0, b = W[j], c, s;
    double ai[M], aj[M];
#pragma unroll
    for (int k = 0; k < M; k++) {
        ai[k] = Ai[k];
        aj[k] = Aj[k];
    }
    double vi[N], vj[N];
    if (WITH_V) {
        const LP<ST> Vi = Vt + i * N, Vj = Vt + j * N;
#pragma unroll
        for (int k = 0; k < N; k++) {
            vi[k] = Vi[k];
            vj[k] = Vj[k];
        }
    }
#pragma unroll
    for (int k = 0; k < M; k++)
        p += ai[k] * aj[k];
    if (fabs(p) <= eps * sqrt(a * b))
        return false;
    p *= 2;
    const double beta = a - b, gamma = cv_hypot(p, beta);
    jacobi_rotation(p, beta, gamma, c, s);
    const double cv = second ? -s : c, cj = second ? c : s;
    const LP<ST> row = second ? Aj : Ai;
    double norm2 = 0;
#pragma unroll
    for (int k = 0; k < M; k++) {
        const double t = cv * ai[k] + cj * aj[k];
        row[k] = t;
        norm2 += t * t;
    }
    W[second ? j : i] = norm2;
    if (WITH_V) {
        const LP<ST> vrow = Vt + (second ? j : i) * N;
#pragma unroll
        for (int k = 0; k < N; k++)
            vrow[k] = cv * vi[k] + cj * vj[k];
    }
    return true;
}

// Lanes of a wave that share one problem's LDS copy (see PoseMode::POSE_HELPED): lane
// `first + stride * g` is member g of `count`.  count == 1: the lane is on its own.
struct LaneGroup {
    int member = 0, first = 0, stride = 64, count = 1;
    __device__ bool any(bool x) const // over the members of this lane's group
    {
        unsigned long long mask = 0;
        for (int m = 0; m < count; m++)
            mask |= 1ull << (first + m * stride);
        return (__builtin_amdgcn_ballot_w64(x) & mask) != 0;
    }
};

// One-sided Jacobi SVD on the n rows of At (length m each); rotations
// accumulate in Vt (n x n).  On exit the rows of At are the left singular
// vectors, W descending.  Same sweep order, thresholds and zero-row completion
// as OpenCV's JacobiSVDImpl_<double>.  W is an n-entry work array.  WITH_V =
// false skips the right-vector accumulation (it never feeds back into At or W).
// M, N > 0 fix the sizes at compile time (astep == M, vstep == N): the rows of a pair are
// then read into registers in one batch and the loops over their elements unroll, instead
// of one LDS round trip per element.  Same operations in the same order either way.
// grp.count > 1 (fixed sizes only): the members of the group, which all make this call
// on the same arrays, split the sweeps: the pairs (i, j) with i + j - 1 == T touch
// different rows and depend only on pairs of earlier T, so a sweep runs as 2N - 3 steps
// whose pairs go to different members, each pair with the operations of the cyclic order.
template <int ST, bool WITH_V, int M = 0, int N = 0>
__device__ void jacobi_svd(LP<ST> At, int astep, LP<ST> W, LP<ST> Vt, int vstep, int m, int n,
                           LaneGroup grp = LaneGroup())
{
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
    const int max_iter = m > 30 ? m : 30;
    double c, s;
    if constexpr (M > 0) {
        m = M;
        n = N;
        astep = M;
        vstep = N;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) {
            const double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sd;
        if (WITH_V) {
            for (int k = 0; k < n; k++)
                Vt[i * vstep + k] = 0;
            Vt[i * vstep + i] = 1;
        }
    }
    bool swept = false;
    if constexpr (M > 0 && N > 3) {
        if (grp.count > 1) {
            wave_sync();
            for (int iter = 0; iter < max_iter; iter++) {
                bool changed = false;
                // four or more members: two lanes per pair (a step of these sizes has at
                // most two pairs)
                const bool halves = grp.count >= 4;
                const int per_round = halves ? grp.count >> 1 : grp.count;
                const int slot = halves ? grp.member >> 1 : grp.member;
                for (int T = 0; T <= 2 * N - 4; T++) {
                    const int lo = T > N - 2 ? T - (N - 2) : 0, hi = T >> 1;
                    for (int i0 = lo; i0 <= hi; i0 += per_round) {
                        const int i = i0 + slot;
                        if (i <= hi)
                            changed |= halves ? jacobi_small_pair_half<ST, WITH_V, M, N>(
                                                    At, W, Vt, i, T + 1 - i, eps, grp.member & 1)
                                              : jacobi_small_pair<ST, WITH_V, M, N>(At, W, Vt, i,
                                                                                    T + 1 - i, eps);
                        wave_sync();
                    }
                }
                if (!grp.any(changed))
                    break;
            }
            swept = true;
        }
    }
    for (int iter = 0; iter < max_iter && !swept; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                if constexpr (M > 0) {
                    changed |= jacobi_small_pair<ST, WITH_V, M, N>(At, W, Vt, i, j, eps);
                    continue;
                }
                const LP<ST> Ai = At + i * astep, Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++)
                    p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b))
                    continue;
                p *= 2;
                const double beta = a - b, gamma = cv_hypot(p, beta);
                jacobi_rotation(p, beta, gamma, c, s);
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    const double t0 = c * Ai[k] + s * Aj[k];
                    const double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
                if (WITH_V) {
                    const LP<ST> Vi = Vt + i * vstep, Vj = Vt + j * vstep;
                    for (int k = 0; k < n; k++) {
                        const double t0 = c * Vi[k] + s * Vj[k];
                        const double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (!changed)
            break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) {
            const double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++)
            if (W[j] < W[k])
                j = k;
        if (i != j) {
            double t = W[i];
            W[i] = W[j];
            W[j] = t;
            for (int k = 0; k < m; k++) {
                t = At[i * astep + k];
                At[i * astep + k] = At[j * astep + k];
                At[j * astep + k] = t;
            }
            if (WITH_V)
                for (int k = 0; k < n; k++) {
                    t = Vt[i * vstep + k];
                    Vt[i * vstep + k] = Vt[j * vstep + k];
                    Vt[j * vstep + k] = t;
                }
        }
    }
    Rng rng{0x12345678};
    for (int i = 0; i < n; i++) {
        double sd = W[i];
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / m;
            for (int k = 0; k < m; k++)
                At[i * astep + k] = (rng.next() & 256) != 0 ? val0 : -val0;
            for (int iter = 0; iter < 2; iter++)
                for (int j = 0; j < i; j++) {
                    sd = 0;
                    for (int k = 0; k < m; k++)
                        sd += At[i * astep + k] * At[j * astep + k];
                    double asum = 0;
                    for (int k = 0; k < m; k++) {
                        const double t = At[i * astep + k] - sd * At[j * astep + k];
                        At[i * astep + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (int k = 0; k < m; k++)
                        At[i * astep + k] *= asum;
                }
            sd = 0;
            for (int k = 0; k < m; k++) {
                const double t = At[i * astep + k];
                sd += t * t;
            }
            sd = sqrt(sd);
        }
        s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < m; k++)
            At[i * astep + k] *= s;
    }
}

// Forces a value to be computed where the statement stands.
#define KEEP(x) asm volatile("" : "+v"(x))

// ---- building blocks of the 12 x 12 decomposition ---------------------------------
// Row i of a pass (i, i+1), (i, i+2), ... lives in registers with its squared norm.
template <int N> struct JacobiRow {
    double v[N], w;
    bool dirty;
    template <int ST> __device__ __forceinline__ void load(LP<ST> row, double norm2)
    {
#pragma unroll
        for (int k = 0; k < N; k++)
            v[k] = row[k];
        w = norm2;
        dirty = false;
    }
    template <int ST> __device__ __forceinline__ void store(LP<ST> row, lds_double &norm2) const
    {
        if (!dirty)
            return;
#pragma unroll
        for (int k = 0; k < N; k++)
            row[k] = v[k];
        norm2 = w;
    }
};

// One pair (row in registers, row j in LDS) exactly as JacobiSVDImpl_ states it.
template <int ST, int N>
__device__ __forceinline__ void jacobi_pair(LP<ST> At, LP<ST> W, JacobiRow<N> &r, int j, double eps,
                                            bool &changed)
{
    const LP<ST> Aj = At + j * N;
    double aj[N];
#pragma unroll
    for (int k = 0; k < N; k++)
        aj[k] = Aj[k];
    double a = r.w, p = 0, b = W[j];
#pragma unroll
    for (int k = 0; k < N; k++)
        p += r.v[k] * aj[k];
    if (fabs(p) <= eps * sqrt(a * b))
        return;
    p *= 2;
    const double beta = a - b, gamma = cv_hypot(p, beta);
    double c, s;
    jacobi_rotation(p, beta, gamma, c, s);
    a = b = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double t0 = c * r.v[k] + s * aj[k];
        const double t1 = -s * r.v[k] + c * aj[k];
        r.v[k] = t0;
        Aj[k] = t1;
        a += t0 * t0;
        b += t1 * t1;
    }
    r.w = a;
    W[j] = b;
    changed = true;
    r.dirty = true;
}

// The arithmetic of one pair without control flow, so that two of them can be issued
// interleaved: every quantity is computed as in jacobi_pair and `rotate` says whether the
// pair is past the convergence test.
template <int N> struct PairMath {
    double aj[N], nv[N], nj[N], na, nb, p, a, b;
    bool rotate;
};

// Partner row, dot product and convergence test.
template <int ST, int N>
__device__ __forceinline__ void jacobi_pair_test(LP<ST> At, LP<ST> W, const JacobiRow<N> &r, int j,
                                                 double eps, PairMath<N> &m)
{
    const LP<ST> Aj = At + j * N;
#pragma unroll
    for (int k = 0; k < N; k++)
        m.aj[k] = Aj[k];
    m.a = r.w;
    m.b = W[j];
    double p = 0;
#pragma unroll
    for (int k = 0; k < N; k++)
        p += r.v[k] * m.aj[k];
    m.p = p;
    m.rotate = !(fabs(p) <= eps * sqrt(m.a * m.b));
}

// The rotation (c, s) of a pair past the convergence test.
template <int N>
__device__ __forceinline__ void jacobi_pair_rotation(PairMath<N> &m, double &c, double &s)
{
    const double p = m.p * 2;
    const double beta = m.a - m.b;
    const double gamma = cv_hypot(p, beta);
    jacobi_rotation(p, beta, gamma, c, s);
}

// Rotation and rotated rows.
template <int N>
__device__ __forceinline__ void jacobi_pair_math(const JacobiRow<N> &r, PairMath<N> &m)
{
    double c, s;
    jacobi_pair_rotation<N>(m, c, s);
    double na = 0, nb = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double t0 = c * r.v[k] + s * m.aj[k];
        const double t1 = -s * r.v[k] + c * m.aj[k];
        m.nv[k] = t0;
        m.nj[k] = t1;
        na += t0 * t0;
        nb += t1 * t1;
    }
    m.na = na;
    m.nb = nb;
}

template <int ST, int N>
__device__ __forceinline__ void jacobi_pair_commit(LP<ST> At, LP<ST> W, JacobiRow<N> &r, int j,
                                                   const PairMath<N> &m, bool &changed)
{
    if (!m.rotate)
        return;
    const LP<ST> Aj = At + j * N;
#pragma unroll
    for (int k = 0; k < N; k++) {
        r.v[k] = m.nv[k];
        Aj[k] = m.nj[k];
    }
    r.w = m.na;
    W[j] = m.nb;
    changed = true;
    r.dirty = true;
}

// Pairs (ra, ja) and (rb, jb) touch four different rows, so they commute; issuing their
// dependent fp64 chains together nearly halves the time a lone wave spends waiting on
// its own results.  Each pair's operations are those of jacobi_pair, bit for bit.
template <int ST, int N>
__device__ __forceinline__ void jacobi_pair2(LP<ST> At, LP<ST> W, JacobiRow<N> &ra, int ja,
                                             JacobiRow<N> &rb, int jb, double eps, bool &changed)
{
    PairMath<N> ma, mb;
    jacobi_pair_test<ST, N>(At, W, ra, ja, eps, ma);
    jacobi_pair_test<ST, N>(At, W, rb, jb, eps, mb);
    if (__builtin_amdgcn_ballot_w64(ma.rotate | mb.rotate) == 0)
        return; // both pairs converged in every lane
    jacobi_pair_math<N>(ra, ma);
    jacobi_pair_math<N>(rb, mb);
    jacobi_pair_commit<ST, N>(At, W, ra, ja, ma, changed);
    jacobi_pair_commit<ST, N>(At, W, rb, jb, mb, changed);
}

// One pair with both rows in LDS, in the straight-line form of jacobi_pair2; the rotated
// rows go straight back to LDS.  True if this lane's pair was rotated.
template <int ST, int N>
__device__ __forceinline__ bool jacobi_pair_lds(LP<ST> At, LP<ST> W, int i, int j, double eps)
{
    const LP<ST> Ai = At + i * N, Aj = At + j * N;
    JacobiRow<N> r;
    r.load(Ai, W[i]);
    PairMath<N> m;
    jacobi_pair_test<ST, N>(At, W, r, j, eps, m);
    if (__builtin_amdgcn_ballot_w64(m.rotate) == 0)
        return false;
    jacobi_pair_math<N>(r, m);
    if (m.rotate) {
#pragma unroll
        for (int k = 0; k < N; k++) {
            Ai[k] = m.nv[k];
            Aj[k] = m.nj[k];
        }
        W[i] = m.na;
        W[j] = m.nb;
    }
    return m.rotate;
}

// jacobi_pair_lds by TWO lanes: both load the two rows, test and compute the rotation, then
// the first lane rotates and stores row i with its norm and the second (second == true)
// row j with its norm, a third fewer fp64 instructions per lane.  Each row's elements and
// norm are computed by the operations of jacobi_pair in their order (row j's element is
// -s * v + c * aj either way), so the result is bit-identical.  True if the pair was rotated.
template <int ST, int N>
__device__ __forceinline__ bool jacobi_pair_lds_half(LP<ST> At, LP<ST> W, int i, int j, double eps,
                                                     bool second)
{
    const LP<ST> Ai = At + i * N, Aj = At + j * N;
    JacobiRow<N> r;
    r.load(Ai, W[i]);
    PairMath<N> m;
    jacobi_pair_test<ST, N>(At, W, r, j, eps, m);
    if (__builtin_amdgcn_ballot_w64(m.rotate) == 0)
        return false;
    double c, s;
    jacobi_pair_rotation<N>(m, c, s);
    const double cv = second ? -s : c, cj = second ? c : s;
    double out[N], norm2 = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double t = cv * r.v[k] + cj * m.aj[k];
        out[k] = t;
        norm2 += t * t;
    }
    if (m.rotate) {
        const LP<ST> row = second ? Aj : Ai;
#pragma unroll
        for (int k = 0; k < N; k++)
            row[k] = out[k];
        W[second ? j : i] = norm2;
    }
    return m.rotate;
}

// The 12 x 12 instance of jacobi_svd<ST, false>.  Row i of a pass is held in registers
// across the inner j loop (half the LDS traffic per rotation), and passes i and i + 1 of
// a sweep run together: at a step pass i works on (i, s) and pass i + 1 on (i + 1, s - 1),
// which is after (i, i + 1) and (i, s - 1), the only earlier pairs that touch its two
// rows.  Pairs on disjoint rows commute exactly (each rotation reads and writes its own
// two rows and norms only), so every pair performs the operations it performs in the
// cyclic order and the results are bit-identical to the generic routine.
// W[i] = squared norm of row i (the state the sweeps start from).
template <int ST> __device__ __forceinline__ void jacobi_init_norms_12(LP<ST> At, LP<ST> W)
{
    constexpr int N = 12;
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const double t = At[i * N + k];
            sd += t * t;
        }
        W[i] = sd;
    }
}

template <int ST> __device__ void jacobi_finish_12(LP<ST> At, LP<ST> W);

template <int ST> __device__ void jacobi_svd_12(LP<ST> At, LP<ST> W)
{
    constexpr int N = 12;
    const double eps = DBL_EPSILON * 10;
    jacobi_init_norms_12<ST>(At, W);
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
        // Passes i and i + 1 run together, pass i + 1 one partner row behind pass i.
        for (int i = 0; i < N - 1; i += 2) {
            const LP<ST> Ai = At + i * N;
            JacobiRow<N> ra, rb;
            ra.load(Ai, W[i]);
            jacobi_pair<ST, N>(At, W, ra, i + 1, eps, changed);
            if (i + 1 >= N - 1) { // the last pass has no companion
                ra.store(Ai, W[i]);
                break;
            }
            jacobi_pair<ST, N>(At, W, ra, i + 2, eps, changed);
            rb.load(Ai + N, W[i + 1]);
            for (int sj = i + 3; sj < N; sj++)
                jacobi_pair2<ST, N>(At, W, ra, sj, rb, sj - 1, eps, changed);
            jacobi_pair<ST, N>(At, W, rb, N - 1, eps, changed);
            ra.store(Ai, W[i]);
            rb.store(Ai + N, W[i + 1]);
        }
        if (!changed)
            break;
    }
    jacobi_finish_12<ST>(At, W);
}

// What follows the sweeps: singular values, descending order, completion of zero rows.
template <int ST> __device__ void jacobi_finish_12(LP<ST> At, LP<ST> W)
{
    constexpr int N = 12;
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const double t = At[i * N + k];
            sd += t * t;
        }
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < N - 1; i++) {
        int j = i;
        for (int k = i + 1; k < N; k++)
            if (W[j] < W[k])
                j = k;
        if (i != j) {
            double t = W[i];
            W[i] = W[j];
            W[j] = t;
            for (int k = 0; k < N; k++) {
                t = At[i * N + k];
                At[i * N + k] = At[j * N + k];
                At[j * N + k] = t;
            }
        }
    }
    Rng rng{0x12345678};
    for (int i = 0; i < N; i++) {
        double sd = W[i];
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / N;
            for (int k = 0; k < N; k++)
                At[i * N + k] = (rng.next() & 256) != 0 ? val0 : -val0;
            for (int it = 0; it < 2; it++)
                for (int j = 0; j < i; j++) {
                    sd = 0;
                    for (int k = 0; k < N; k++)
                        sd += At[i * N + k] * At[j * N + k];
                    double asum = 0;
                    for (int k = 0; k < N; k++) {
                        const double t = At[i * N + k] - sd * At[j * N + k];
                        At[i * N + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (int k = 0; k < N; k++)
                        At[i * N + k] *= asum;
                }
            sd = 0;
            for (int k = 0; k < N; k++) {
                const double t = At[i * N + k];
                sd += t * t;
            }
            sd = sqrt(sd);
        }
        const double s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < N; k++)
            At[i * N + k] *= s;
    }
}

// Rounds of one 12 x 12 sweep for 2 and 4 lanes per matrix: byte g of a word = the pair
// (i << 4 | j) of member g, 0xFF = none.  A list schedule of the pair dependencies (pair
// (i, j) needs the previous pairs on row i and on row j of the cyclic order: (i, j-1), or
// (i-1, i) when j == i+1, and (i-1, j)), longest remaining chain first: 35 and 23 rounds,
// against 39 and 26 when each step i + j - 1 = T is finished before the next begins.
__device__ const uint32_t JACOBI_ROUNDS_2[35] = {
    0xFFFFFF01u, 0xFFFFFF02u, 0xFFFF1203u, 0xFFFF1304u, 0xFFFF1405u, 0xFFFF0623u, 0xFFFF2415u,
    0xFFFF1607u, 0xFFFF3425u, 0xFFFF1708u, 0xFFFF3526u, 0xFFFF1809u, 0xFFFF3627u, 0xFFFF0A45u,
    0xFFFF2819u, 0xFFFF4637u, 0xFFFF1A0Bu, 0xFFFF3829u, 0xFFFF5647u, 0xFFFF2A1Bu, 0xFFFF4839u,
    0xFFFF2B57u, 0xFFFF493Au, 0xFFFF6758u, 0xFFFF4A3Bu, 0xFFFF6859u, 0xFFFF5A4Bu, 0xFFFF7869u,
    0xFFFF6A5Bu, 0xFFFF6B79u, 0xFFFF897Au, 0xFFFF8A7Bu, 0xFFFF9A8Bu, 0xFFFFFF9Bu, 0xFFFFFFABu};
__device__ const uint32_t JACOBI_ROUNDS_4[23] = {
    0xFFFFFF01u, 0xFFFFFF02u, 0xFFFF1203u, 0xFFFF1304u, 0xFF231405u, 0xFF241506u, 0x34251607u,
    0x35261708u, 0x36271809u, 0x28190A45u, 0x1A0B4637u, 0x56473829u, 0x48392A1Bu, 0x493A2B57u,
    0x4A3B6758u, 0xFF4B6859u, 0xFF78695Au, 0xFF796A5Bu, 0xFF897A6Bu, 0xFFFF8A7Bu, 0xFFFF9A8Bu,
    0xFFFFFF9Bu, 0xFFFFFFABu};

// The sweeps of jacobi_svd_12 with G lanes per matrix.  In the cyclic order pair (i, j)
// depends only on the earlier pairs that touch row i or row j, which are (i, j - 1) and
// (i - 1, j) and what precedes those; so all pairs with i + j - 1 == T can run once step
// T - 1 is done, they touch 2 * count different rows, and each performs exactly the
// operations it performs in the sequential order: a sweep is 21 steps of 1..6 independent
// pairs instead of 66 pairs in a row.  The next sweep starts 12 steps after this one, not
// 21: its pair (i, j) then runs at step i + j + 11 of this sweep, after (j, 11) at step
// j + 10 and (i, 11) at step i + 10, the last pairs here that touch its rows, and the two
// sweeps' pairs of one step are on different rows (the new sweep's are <= T + 1, the old
// one's >= T + 2).  Steady state: 5 or 6 independent pairs per step, 12 steps per sweep.
// A sweep that rotates nothing ends the decomposition; the first 9 steps of its
// successor have run by then, on rows that sweep has left as they were, so they repeat
// its tests and rotate nothing either: the result is that of the sequential order, bit
// for bit.  At most 30 sweeps, as there.
// The lanes g = 0..G-1 that share a matrix take the pairs of a step G at a time.  The
// matrix and the row norms are in LDS (At, W: the same addresses in every lane of a
// group).  Lane `hl + HPW * g` is member g of group hl; all members of a group must
// enter together.  OVERLAP = false starts a sweep only when the one before is complete
// (period 21): with 2 or 4 lanes per matrix the overlap saves 7 % of the rounds but fewer
// rounds are then free of rotations in every lane of the wave, and it measured slower.
template <int ST, int G, int HPW, bool OVERLAP>
__device__ void jacobi_sweeps_12_shared(LP<ST> At, LP<ST> W, int g, int hl)
{
    constexpr int N = 12, MAX_SWEEPS = 30, LAST = 2 * N - 4, PERIOD = OVERLAP ? N : LAST + 1;
    const double eps = DBL_EPSILON * 10;
    unsigned long long group = 0;
    for (int m = 0; m < G; m++)
        group |= 1ull << (m * HPW);
    if constexpr (G == 2 || G == 4) {
        const uint32_t *rounds = G == 2 ? JACOBI_ROUNDS_2 : JACOBI_ROUNDS_4;
        constexpr int N_ROUNDS = G == 2 ? 35 : 23;
        for (int s = 0; s < MAX_SWEEPS; s++) {
            bool rotated = false;
            for (int r = 0; r < N_ROUNDS; r++) {
                const uint32_t e = rounds[r] >> (8 * g) & 0xFF;
                if (e != 0xFF)
                    rotated |= jacobi_pair_lds<ST, N>(At, W, e >> 4, e & 15, eps);
                wave_sync();
            }
            if ((__builtin_amdgcn_ballot_w64(rotated) >> hl & group) == 0)
                return;
        }
        return;
    }
    // twelve or more lanes per matrix: two lanes per pair (a step has at most six pairs)
    constexpr bool HALVES = G >= 12;
    constexpr int PAIRS = HALVES ? G / 2 : G;
    bool rot_old = false, rot_new = false; // a rotation in sweep s - 1 / in sweep s
    for (int s = 0; s <= MAX_SWEEPS; s++) {
        for (int T0 = 0; T0 < PERIOD; T0++) {
            // sweep s is at its step T0, sweep s - 1 at its step T0 + PERIOD
            const int T1 = T0 + PERIOD;
            const int lo0 = T0 > N - 2 ? T0 - (N - 2) : 0, lo1 = T1 - (N - 2);
            const int c0 = s < MAX_SWEEPS ? (T0 >> 1) - lo0 + 1 : 0;
            const int c1 = OVERLAP && s > 0 && T1 <= LAST ? (T1 >> 1) - lo1 + 1 : 0;
            for (int q0 = 0; q0 < c0 + c1; q0 += PAIRS) {
                const int q = q0 + (HALVES ? g >> 1 : g);
                if (q < c0 + c1) {
                    const bool old = OVERLAP && q >= c0;
                    const int i = old ? lo1 + q - c0 : lo0 + q;
                    const int j = (old ? T1 : T0) + 1 - i;
                    const bool rotated = HALVES
                                             ? jacobi_pair_lds_half<ST, N>(At, W, i, j, eps, g & 1)
                                             : jacobi_pair_lds<ST, N>(At, W, i, j, eps);
                    rot_old |= old & rotated;
                    rot_new |= !old & rotated;
                }
                wave_sync();
            }
            if (OVERLAP && s > 0 && T1 == LAST) { // sweep s - 1 is complete
                if (s == MAX_SWEEPS || (__builtin_amdgcn_ballot_w64(rot_old) >> hl & group) == 0)
                    return;
            }
        }
        if (!OVERLAP) { // sweep s is complete
            if (s == MAX_SWEEPS - 1 || (__builtin_amdgcn_ballot_w64(rot_new) >> hl & group) == 0)
                return;
        }
        rot_old = rot_new;
        rot_new = false;
    }
}

// A lane's LDS arrays (offsets in doubles; LDS_DOUBLES per hypothesis).  Lifetimes
// overlap so that the peak is the 12 x 12 matrix plus its 12 row norms:
//   ut  12 x 12: MtM, then its left singular vectors (rows), with w = the row norms.
//       Only rows 8..11 are read after the decomposition: row 11 moves to registers
//       and rows 8..10 to v (the last 36 doubles), which frees ut[0..120) for
//   l   6 x 10, and the scratch of the small SVDs sa / sv / sw (At up to 5 x 6, Vt up
//       to 5 x 5, W), which also serve the control-point and basis steps that run
//       before MtM is formed;
//   ga  the 6 x 4 Gauss-Newton matrix shares sa (no small SVD runs meanwhile).
constexpr int OFF_UT = 0;
constexpr int OFF_L = 0;     // 60
constexpr int OFF_SA = 60;   // 30
constexpr int OFF_GA = 60;   // 24
constexpr int OFF_SV = 90;   // 25
constexpr int OFF_SW = 115;  // 5
constexpr int OFF_V = 120;   // 36: rows 8, 9, 10 of ut
constexpr int OFF_W = 144;   // 12
constexpr int LDS_DOUBLES = 156; // 32 hypotheses = 39 KiB: four workgroups per CU
// Element stride of a solve workgroup with fewer hypotheses than this (part of its LDS is
// then unused): 4096 hypotheses at 8 per wave take 0.26 ms with stride 32 and 0.28 ms
// with stride 8 (the Gauss-Newton step, the most LDS-bound one, is 1.5 times slower).
constexpr int LDS_MIN_STRIDE = 32;

template <int ST> struct EpnpLds {
    LP<ST> ut, w, sa, sv, sw, l, ga, v;
    __device__ explicit EpnpLds(lds_double *lane_base)
        : ut{lane_base + OFF_UT * ST}, w{lane_base + OFF_W * ST}, sa{lane_base + OFF_SA * ST},
          sv{lane_base + OFF_SV * ST}, sw{lane_base + OFF_SW * ST}, l{lane_base + OFF_L * ST},
          ga{lane_base + OFF_GA * ST}, v{lane_base + OFF_V * ST}
    {
    }
};

// SVD of a square matrix given TRANSPOSED in S.sa (n x n, n <= 3 here): after
// the call S.sa holds Ut (rows = left singular vectors), S.sv holds Vt.
template <int ST> __device__ void svd_small_square(const EpnpLds<ST> &S, int n)
{
    if (n == 3)
        jacobi_svd<ST, true, 3, 3>(S.sa, 3, S.sw, S.sv, 3, 3, 3);
    else
        jacobi_svd<ST, true>(S.sa, n, S.sw, S.sv, n, n, n);
}

// cv::solve(A, b, x, DECOMP_SVD) with A (m x n) already TRANSPOSED in S.sa (n x m).
template <int ST>
__device__ void solve_svd(const EpnpLds<ST> &S, int m, int n, const double *b, double *x,
                          LaneGroup grp = LaneGroup())
{
    if (m == 6 && n == 4)
        jacobi_svd<ST, true, 6, 4>(S.sa, 6, S.sw, S.sv, 4, 6, 4, grp);
    else if (m == 6 && n == 3)
        jacobi_svd<ST, true, 6, 3>(S.sa, 6, S.sw, S.sv, 3, 6, 3);
    else if (m == 6 && n == 5)
        jacobi_svd<ST, true, 6, 5>(S.sa, 6, S.sw, S.sv, 5, 6, 5, grp);
    else
        jacobi_svd<ST, true>(S.sa, m, S.sw, S.sv, n, m, n);
    double threshold = 0;
    for (int i = 0; i < n; i++) {
        x[i] = 0;
        threshold += S.sw[i];
    }
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < n; i++) {
        double wi = S.sw[i];
        if (fabs(wi) <= threshold)
            continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < m; j++)
            s += S.sa[i * m + j] * b[j];
        s *= wi;
        for (int j = 0; j < n; j++)
            x[j] = x[j] + s * S.sv[i * n + j];
    }
}

__device__ inline double dot3(const double *a, const double *b)
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

__device__ inline double dist2(const double *p1, const double *p2)
{
    return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) +
           (p1[2] - p2[2]) * (p1[2] - p2[2]);
}

// Correspondences of one solve: rows idx[0..n) of obj/img (idx == nullptr: rows 0..n).
struct Points {
    const float *obj;
    const float *img;
    const int32_t *idx;
    int n;
    // Sums over the points run i = first, first + step, ...: (0, 1) when a lane
    // owns its problem, (lane, 64) when the 64 lanes of a wave share one problem
    // and combine their partial sums with wave_sum().
    int first, step;
    double fu, fv, uc, vc, ifx, ify;
    // The points reach solvePnP as CV_32F (the caller's arrays, RANSAC's samples) or as CV_64F
    // (solvePnPRansac converts the consensus set to Point3d / Point2d before its final solve).
    bool points_are_f32;
    __device__ void world(int i, double *pw) const
    {
        const int64_t r = idx ? idx[i] : i;
        pw[0] = obj[3 * r];
        pw[1] = obj[3 * r + 1];
        pw[2] = obj[3 * r + 2];
    }
    // undistortPoints with zero distortion stores (u - cx)/fx in the type of the points it was
    // given (rounded to float for CV_32F points, kept for CV_64F ones); EPnP maps it back.
    __device__ void pixel(int i, double &u, double &v) const
    {
        const int64_t r = idx ? idx[i] : i;
        double xn = ((double)img[2 * r] - uc) * ifx;
        double yn = ((double)img[2 * r + 1] - vc) * ify;
        if (points_are_f32) {
            xn = (float)xn;
            yn = (float)yn;
        }
        u = xn * fu + uc;
        v = yn * fv + vc;
    }
};

// Sum of x over the 64 lanes (fixed butterfly order; every lane gets the same value).
__device__ inline double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        x += __shfl_xor(x, off, 64);
    return x;
}

struct EpnpState {
    double cws[4][3];
    double ci[9]; // inverse of the control-point basis
    double ccs[4][3];
    bool flip;    // solve_for_sign decision of the current candidate
};

__device__ inline void alphas_of(const EpnpState &e, const double *pi, double *a)
{
    for (int j = 0; j < 3; j++)
        a[1 + j] = e.ci[3 * j] * (pi[0] - e.cws[0][0]) + e.ci[3 * j + 1] * (pi[1] - e.cws[0][1]) +
                   e.ci[3 * j + 2] * (pi[2] - e.cws[0][2]);
    a[0] = 1.0f - a[1] - a[2] - a[3];
}

__device__ inline void pc_of(const EpnpState &e, const double *a, double *pc)
{
    for (int j = 0; j < 3; j++) {
        const double v = a[0] * e.ccs[0][j] + a[1] * e.ccs[1][j] + a[2] * e.ccs[2][j] +
                         a[3] * e.ccs[3][j];
        pc[j] = e.flip ? -v : v;
    }
}

template <int ST>
__device__ void choose_control_points(const Points &P, EpnpState &e, const EpnpLds<ST> &S)
{
    const int n = P.n;
    e.cws[0][0] = e.cws[0][1] = e.cws[0][2] = 0;
    for (int i = P.first; i < n; i += P.step) {
        double pw[3];
        P.world(i, pw);
        for (int j = 0; j < 3; j++)
            e.cws[0][j] += pw[j];
    }
    if (P.step > 1)
        for (int j = 0; j < 3; j++)
            e.cws[0][j] = wave_sum(e.cws[0][j]);
    for (int j = 0; j < 3; j++)
        e.cws[0][j] /= n;
    double m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0;
    for (int i = P.first; i < n; i += P.step) {
        double pw[3];
        P.world(i, pw);
        const double d0 = pw[0] - e.cws[0][0], d1 = pw[1] - e.cws[0][1], d2 = pw[2] - e.cws[0][2];
        m00 += d0 * d0; m01 += d0 * d1; m02 += d0 * d2;
        m11 += d1 * d1; m12 += d1 * d2; m22 += d2 * d2;
    }
    if (P.step > 1) {
        m00 = wave_sum(m00); m01 = wave_sum(m01); m02 = wave_sum(m02);
        m11 = wave_sum(m11); m12 = wave_sum(m12); m22 = wave_sum(m22);
    }
    // symmetric, so the transposed copy the SVD works on equals the matrix
    S.sa[0] = m00; S.sa[1] = m01; S.sa[2] = m02;
    S.sa[3] = m01; S.sa[4] = m11; S.sa[5] = m12;
    S.sa[6] = m02; S.sa[7] = m12; S.sa[8] = m22;
    svd_small_square(S, 3);
    for (int i = 1; i < 4; i++) {
        const double k = sqrt(S.sw[i - 1] / n);
        for (int j = 0; j < 3; j++)
            e.cws[i][j] = e.cws[0][j] + k * S.sa[3 * (i - 1) + j];
    }
}

// cv::invert(CC, DECOMP_SVD) of the 3 x 3 control-point basis.
template <int ST> __device__ void compute_basis_inverse(EpnpState &e, const EpnpLds<ST> &S)
{
    double cc[9];
    for (int i = 0; i < 3; i++)
        for (int j = 1; j < 4; j++)
            cc[3 * i + j - 1] = e.cws[j][i] - e.cws[0][i];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            S.sa[i * 3 + j] = cc[j * 3 + i];
    svd_small_square(S, 3);
    const double threshold = (S.sw[0] + S.sw[1] + S.sw[2]) * (DBL_EPSILON * 2);
    for (int i = 0; i < 9; i++)
        e.ci[i] = 0;
    for (int i = 0; i < 3; i++) {
        double wi = S.sw[i];
        if (fabs(wi) <= threshold)
            continue;
        wi = 1 / wi;
        double buffer[3];
        for (int j = 0; j < 3; j++)
            buffer[j] = S.sa[i * 3 + j] * wi;
        for (int r = 0; r < 3; r++) {
            const double s = S.sv[i * 3 + r];
            for (int j = 0; j < 3; j++)
                e.ci[r * 3 + j] = e.ci[r * 3 + j] + s * buffer[j];
        }
    }
}

// MtM = M^T M accumulated point by point (rows 2i and 2i+1 of M in that order);
// the structurally zero products are skipped, which leaves every sum unchanged.
template <int ST>
__device__ void accumulate_mtm(const Points &P, const EpnpState &e, const EpnpLds<ST> &S)
{
    double acc[78]; // upper triangle, row-major
#pragma unroll
    for (int i = 0; i < 78; i++)
        acc[i] = 0;
    for (int i = P.first; i < P.n; i += P.step) {
        double pw[3], as[4], u, v, m1[12], m2[12];
        P.world(i, pw);
        P.pixel(i, u, v);
        alphas_of(e, pw, as);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            m1[3 * k] = as[k] * P.fu;
            m1[3 * k + 1] = 0.0;
            m1[3 * k + 2] = as[k] * (P.uc - u);
            m2[3 * k] = 0.0;
            m2[3 * k + 1] = as[k] * P.fv;
            m2[3 * k + 2] = as[k] * (P.vc - v);
        }
        int o = 0;
#pragma unroll
        for (int a = 0; a < 12; a++)
#pragma unroll
            for (int b = a; b < 12; b++, o++) {
                if (a % 3 != 1 && b % 3 != 1)
                    acc[o] += m1[a] * m1[b];
                if (a % 3 != 0 && b % 3 != 0)
                    acc[o] += m2[a] * m2[b];
            }
    }
    int o = 0;
#pragma unroll
    for (int a = 0; a < 12; a++)
#pragma unroll
        for (int b = a; b < 12; b++, o++) {
            const double v = P.step > 1 ? wave_sum(acc[o]) : acc[o];
            S.ut[a * 12 + b] = v;
            S.ut[b * 12 + a] = v;
        }
}

// The null-space rows of ut, detached from it: row 11 in registers, rows 8..10 in
// S.v.  element(i, k) = ut[(11 - i) * 12 + k], i.e. i counts from the smallest
// singular value as EPnP does.
template <int ST> struct NullRows {
    double v11[12];
    LP<ST> v;
    __device__ double at(int i, int k) const { return i == 0 ? v11[k] : v[(3 - i) * 12 + k]; }
};

template <int ST> __device__ void detach_null_rows(const EpnpLds<ST> &S, NullRows<ST> &N)
{
    N.v = S.v;
#pragma unroll
    for (int k = 0; k < 12; k++)
        N.v11[k] = S.ut[11 * 12 + k];
    // rows 8..10 (ut[96..132)) -> v (ut[120..156)): overlapping, so copy from the top
    for (int e = 35; e >= 0; e--)
        S.v[e] = S.ut[96 + e];
}

template <int ST> __device__ void compute_L_6x10(const NullRows<ST> &N, LP<ST> l)
{
    double dv[4][6][3];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int a = 0, b = 1;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            dv[i][j][0] = N.at(i, 3 * a) - N.at(i, 3 * b);
            dv[i][j][1] = N.at(i, 3 * a + 1) - N.at(i, 3 * b + 1);
            dv[i][j][2] = N.at(i, 3 * a + 2) - N.at(i, 3 * b + 2);
            b++;
            if (b > 3) {
                a++;
                b = a + 1;
            }
        }
    }
    for (int i = 0; i < 6; i++) {
        const LP<ST> row = l + 10 * i;
        row[0] = dot3(dv[0][i], dv[0][i]);
        row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
        row[2] = dot3(dv[1][i], dv[1][i]);
        row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
        row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
        row[5] = dot3(dv[2][i], dv[2][i]);
        row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
        row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
        row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
        row[9] = dot3(dv[3][i], dv[3][i]);
    }
}

// betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44]
template <int ST>
__device__ void find_betas_approx(int which, const EpnpLds<ST> &S, const double *rho, double *betas,
                                  LaneGroup grp = LaneGroup())
{
    const LP<ST> l = S.l;
    double b5[5];
    if (which == 1) { // [B11 B12 B13 B14]; S.sa = transpose of the 6 x 4 system
        for (int i = 0; i < 6; i++) {
            S.sa[0 * 6 + i] = l[10 * i];
            S.sa[1 * 6 + i] = l[10 * i + 1];
            S.sa[2 * 6 + i] = l[10 * i + 3];
            S.sa[3 * 6 + i] = l[10 * i + 6];
        }
        solve_svd(S, 6, 4, rho, b5, grp);
        if (b5[0] < 0) {
            betas[0] = sqrt(-b5[0]);
            betas[1] = -b5[1] / betas[0];
            betas[2] = -b5[2] / betas[0];
            betas[3] = -b5[3] / betas[0];
        } else {
            betas[0] = sqrt(b5[0]);
            betas[1] = b5[1] / betas[0];
            betas[2] = b5[2] / betas[0];
            betas[3] = b5[3] / betas[0];
        }
        return;
    }
    const int nc = which == 2 ? 3 : 5; // [B11 B12 B22] or [B11 B12 B22 B13 B23]
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < nc; j++)
            S.sa[j * 6 + i] = l[10 * i + j];
    solve_svd(S, 6, nc, rho, b5, grp);
    if (b5[0] < 0) {
        betas[0] = sqrt(-b5[0]);
        betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0;
    } else {
        betas[0] = sqrt(b5[0]);
        betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0;
    }
    if (b5[1] < 0)
        betas[0] = -betas[0];
    betas[2] = which == 2 ? 0.0 : b5[3] / betas[0];
    betas[3] = 0.0;
}

// epnp::qr_solve on the 6 x 4 system in S.ga (Householder, in place).  The
// column-maximum scan covers rows k..nr-2 only, as in the source it follows.
template <int ST> __device__ void qr_solve(LP<ST> pA, double *pb, double *pX)
{
    const int nr = 6, nc = 4;
    double A1[4], A2[4];
    LP<ST> ppAkk = pA;
    for (int k = 0; k < nc; k++) {
        LP<ST> ppAik1 = ppAkk;
        double eta = fabs(*ppAik1);
        for (int i = k + 1; i < nr; i++) {
            const double elt = fabs(*ppAik1);
            if (eta < elt)
                eta = elt;
            ppAik1 += nc;
        }
        if (eta == 0)
            return;
        LP<ST> ppAik2 = ppAkk;
        double sum2 = 0.0;
        const double inv_eta = 1. / eta;
        for (int i = k; i < nr; i++) {
            *ppAik2 *= inv_eta;
            sum2 += *ppAik2 * *ppAik2;
            ppAik2 += nc;
        }
        double sigma = sqrt(sum2);
        if (*ppAkk < 0)
            sigma = -sigma;
        *ppAkk += sigma;
        A1[k] = sigma * *ppAkk;
        A2[k] = -eta * sigma;
        for (int j = k + 1; j < nc; j++) {
            LP<ST> ppAik = ppAkk;
            double sum = 0;
            for (int i = k; i < nr; i++) {
                sum += *ppAik * ppAik[j - k];
                ppAik += nc;
            }
            const double tau = sum / A1[k];
            ppAik = ppAkk;
            for (int i = k; i < nr; i++) {
                ppAik[j - k] -= tau * *ppAik;
                ppAik += nc;
            }
        }
        ppAkk += nc + 1;
    }
    LP<ST> ppAjj = pA;
    for (int j = 0; j < nc; j++) {
        LP<ST> ppAij = ppAjj;
        double tau = 0;
        for (int i = j; i < nr; i++) {
            tau += *ppAij * pb[i];
            ppAij += nc;
        }
        tau /= A1[j];
        ppAij = ppAjj;
        for (int i = j; i < nr; i++) {
            pb[i] -= tau * *ppAij;
            ppAij += nc;
        }
        ppAjj += nc + 1;
    }
    pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; i--) {
        LP<ST> ppAij = pA + (i * nc + (i + 1));
        double sum = 0;
        for (int j = i + 1; j < nc; j++) {
            sum += *ppAij * pX[j];
            ppAij++;
        }
        pX[i] = (pb[i] - sum) / A2[i];
    }
}

template <int ST>
__device__ void gauss_newton(const EpnpLds<ST> &S, const double *rho, double *bt)
{
    double b[6], x[4] = {0, 0, 0, 0};
    for (int it = 0; it < 5; it++) {
        for (int i = 0; i < 6; i++) {
            const LP<ST> rowL = S.l + i * 10;
            const LP<ST> rowA = S.ga + i * 4;
            rowA[0] = 2 * rowL[0] * bt[0] + rowL[1] * bt[1] + rowL[3] * bt[2] + rowL[6] * bt[3];
            rowA[1] = rowL[1] * bt[0] + 2 * rowL[2] * bt[1] + rowL[4] * bt[2] + rowL[7] * bt[3];
            rowA[2] = rowL[3] * bt[0] + rowL[4] * bt[1] + 2 * rowL[5] * bt[2] + rowL[8] * bt[3];
            rowA[3] = rowL[6] * bt[0] + rowL[7] * bt[1] + rowL[8] * bt[2] + 2 * rowL[9] * bt[3];
            b[i] = rho[i] -
                   (rowL[0] * bt[0] * bt[0] + rowL[1] * bt[0] * bt[1] + rowL[2] * bt[1] * bt[1] +
                    rowL[3] * bt[0] * bt[2] + rowL[4] * bt[1] * bt[2] + rowL[5] * bt[2] * bt[2] +
                    rowL[6] * bt[0] * bt[3] + rowL[7] * bt[1] * bt[3] + rowL[8] * bt[2] * bt[3] +
                    rowL[9] * bt[3] * bt[3]);
        }
        qr_solve(S.ga, b, x);
        for (int i = 0; i < 4; i++)
            bt[i] += x[i];
    }
}

// compute_ccs, compute_pcs, solve_for_sign, estimate_R_and_t, reprojection_error.
template <int ST>
__device__ double compute_R_and_t(const Points &P, EpnpState &e, const EpnpLds<ST> &S,
                                  const NullRows<ST> &N, const double *betas, double R[3][3],
                                  double t[3])
{
    const int n = P.n;
    for (int i = 0; i < 4; i++)
        e.ccs[i][0] = e.ccs[i][1] = e.ccs[i][2] = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                e.ccs[j][k] += betas[i] * N.at(i, 3 * j + k);
    }
    // solve_for_sign: the first point's depth decides the sign of everything
    {
        double pw[3], a[4], pc[3];
        P.world(0, pw);
        alphas_of(e, pw, a);
        e.flip = false;
        pc_of(e, a, pc);
        e.flip = pc[2] < 0.0;
    }
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = P.first; i < n; i += P.step) {
        double pw[3], a[4], pc[3];
        P.world(i, pw);
        alphas_of(e, pw, a);
        pc_of(e, a, pc);
        for (int j = 0; j < 3; j++) {
            pc0[j] += pc[j];
            pw0[j] += pw[j];
        }
    }
    if (P.step > 1)
        for (int j = 0; j < 3; j++) {
            pc0[j] = wave_sum(pc0[j]);
            pw0[j] = wave_sum(pw0[j]);
        }
    for (int j = 0; j < 3; j++) {
        pc0[j] /= n;
        pw0[j] /= n;
    }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = P.first; i < n; i += P.step) {
        double pw[3], a[4], pc[3];
        P.world(i, pw);
        alphas_of(e, pw, a);
        pc_of(e, a, pc);
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    if (P.step > 1)
        for (int j = 0; j < 9; j++)
            abt[j] = wave_sum(abt[j]);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            S.sa[i * 3 + j] = abt[j * 3 + i];
    svd_small_square(S, 3);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i][j] = S.sa[0 * 3 + i] * S.sv[0 * 3 + j] + S.sa[1 * 3 + i] * S.sv[1 * 3 + j] +
                      S.sa[2 * 3 + i] * S.sv[2 * 3 + j];
    const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] +
                       R[0][2] * R[1][0] * R[2][1] - R[0][2] * R[1][1] * R[2][0] -
                       R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) {
        R[2][0] = -R[2][0];
        R[2][1] = -R[2][1];
        R[2][2] = -R[2][2];
    }
    t[0] = pc0[0] - dot3(R[0], pw0);
    t[1] = pc0[1] - dot3(R[1], pw0);
    t[2] = pc0[2] - dot3(R[2], pw0);
    double sum2 = 0.0;
    for (int i = P.first; i < n; i += P.step) {
        double pw[3], u, v;
        P.world(i, pw);
        P.pixel(i, u, v);
        const double Xc = dot3(R[0], pw) + t[0];
        const double Yc = dot3(R[1], pw) + t[1];
        const double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
        const double ue = P.uc + P.fu * Xc * inv_Zc;
        const double ve = P.vc + P.fv * Yc * inv_Zc;
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    if (P.step > 1)
        sum2 = wave_sum(sum2);
    return sum2 / n;
}

// How the lanes of a wave are used by epnp_pose.
enum PoseMode {
    POSE_LANE,    // every lane solves its own problem by itself
    POSE_HELPED,  // lanes ST*g + hl, g = 0..64/ST-1, serve problem hl: all of them run the
                  // solve (same instructions, same LDS copy, identical values) and split
                  // the independent Jacobi pairs and the independent candidates
    POSE_WAVE     // ST == 1: the 64 lanes serve one problem as in POSE_HELPED, but the sums
                  // over the points are split over the lanes (Points::step == 64), so the
                  // candidates, whose pose step has such sums, run one after the other
};

// First half of epnp::compute_pose: control points, barycentric basis, MtM and its 12 x 12
// decomposition.  On return rows 8..11 of S.ut (the null-space vectors) and e.cws / e.ci
// are everything the second half needs.
template <int ST, PoseMode MODE = POSE_LANE, int HPW = ST>
__device__ __forceinline__ void epnp_front(const Points &P, const EpnpLds<ST> &S, EpnpState &e,
                                           int lane = 0)
{
    constexpr int G = MODE == POSE_LANE ? 1 : 64 / HPW;
    const int member = MODE == POSE_LANE ? 0 : lane / HPW;
    PNP_STAMP(0);
    choose_control_points(P, e, S);
    PNP_STAMP(1);
    compute_basis_inverse(e, S);
    PNP_STAMP(2);
    accumulate_mtm(P, e, S);
    PNP_STAMP(3);
    // MtM is symmetric, so it is its own transposed copy; EPnP reads only U
    if constexpr (MODE != POSE_LANE) {
        jacobi_init_norms_12<ST>(S.ut, S.w);
        wave_sync();
        // six or more lanes per matrix take every pair of a step at once: then
        // overlapping the sweeps pays (12 rounds per sweep instead of 21)
        jacobi_sweeps_12_shared<ST, G, HPW, G >= 6>(S.ut, S.w, member, lane % HPW);
        jacobi_finish_12<ST>(S.ut, S.w);
    } else {
        jacobi_svd_12<ST>(S.ut, S.w);
    }
    PNP_STAMP(4);
}

// Second half of epnp::compute_pose, from the null-space vectors in rows 8..11 of S.ut and
// e.cws / e.ci.  Returns true in the lane that reports the result (one per problem).
// HPW = problems per wave in the grouped modes (the lane stride between the members of
// a group); the LDS element stride ST may be larger.
// only = 1..3: run that candidate alone and report its R, t and reprojection error
// *err_out; the caller applies the selection rule over the three.
template <int ST, PoseMode MODE = POSE_LANE, int HPW = ST>
__device__ __forceinline__ bool epnp_back(const Points &P, const EpnpLds<ST> &S, EpnpState &e,
                                          double R[3][3], double t[3], int lane = 0, int only = 0,
                                          double *err_out = nullptr)
{
    constexpr int G = MODE == POSE_LANE ? 1 : 64 / HPW;
    const int member = MODE == POSE_LANE ? 0 : lane / HPW;
    const LaneGroup grp = MODE == POSE_LANE ? LaneGroup() : LaneGroup{member, lane % HPW, HPW, G};
    NullRows<ST> N;
    detach_null_rows(S, N);
    compute_L_6x10(N, S.l);
    PNP_STAMP(5);
    double rho[6];
    rho[0] = dist2(e.cws[0], e.cws[1]);
    rho[1] = dist2(e.cws[0], e.cws[2]);
    rho[2] = dist2(e.cws[0], e.cws[3]);
    rho[3] = dist2(e.cws[1], e.cws[2]);
    rho[4] = dist2(e.cws[1], e.cws[3]);
    rho[5] = dist2(e.cws[2], e.cws[3]);
    if (MODE == POSE_HELPED && only == 0) {
        // The three candidates (beta initialisations) are independent of one another
        // after find_betas_approx, and Gauss-Newton and the pose step are the same code
        // for each: the members of a group take different candidates at once.  L stays
        // where it is while Gauss-Newton runs, so two 6 x 4 matrices fit beside it;
        // the pose step needs 21 doubles per candidate and L is dead by then.
        double betas[3][4];
        find_betas_approx(1, S, rho, betas[0], grp);
        PNP_STAMP(6);
        find_betas_approx(2, S, rho, betas[1], grp);
        PNP_STAMP(7);
        find_betas_approx(3, S, rho, betas[2], grp);
        PNP_STAMP(8);
        wave_sync();
        {
            const bool second = member & 1;
            EpnpLds<ST> Sm = S;
            Sm.ga = S.ga + (second ? 24 : 0);
            double b[4];
            for (int k = 0; k < 4; k++)
                b[k] = second ? betas[1][k] : betas[0][k];
            gauss_newton(Sm, rho, b);
            // every member needs every candidate's betas for the pose step
            for (int k = 0; k < 4; k++) {
                const double other = __shfl_xor(b[k], HPW, 64);
                betas[0][k] = second ? other : b[k];
                betas[1][k] = second ? b[k] : other;
            }
            PNP_STAMP(9);
            wave_sync();
            gauss_newton(S, rho, betas[2]);
            PNP_STAMP(10);
            wave_sync();
        }
        // pose step: candidates per round
        constexpr int PER_ROUND = G >= 3 ? 3 : 2;
        const LP<ST> res = S.ut + 64; // 3 x 13 results, clear of the 3 x 21 scratch doubles
        for (int c0 = 0; c0 < 3; c0 += PER_ROUND) {
            const int slot = member % PER_ROUND, c = c0 + slot < 3 ? c0 + slot : c0;
            EpnpLds<ST> Sm = S;
            Sm.sa = S.ut + 21 * (c - c0);
            Sm.sv = Sm.sa + 9;
            Sm.sw = Sm.sv + 9;
            double b[4], Rc[3][3], tc[3];
            for (int k = 0; k < 4; k++)
                b[k] = c == 0 ? betas[0][k] : c == 1 ? betas[1][k] : betas[2][k];
            const double err = compute_R_and_t(P, e, Sm, N, b, Rc, tc);
            const LP<ST> out = res + 13 * c;
            out[0] = err;
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++)
                    out[1 + 3 * i + j] = Rc[i][j];
                out[10 + i] = tc[i];
            }
            PNP_STAMP(11 + c0 / PER_ROUND);
        }
        wave_sync();
        // N = 1; if (err2 < err1) N = 2; if (err3 < err[N]) N = 3
        int best = 0;
        if (res[13] < res[0])
            best = 1;
        if (res[26] < res[13 * best])
            best = 2;
        const LP<ST> win = res + 13 * best;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++)
                R[i][j] = win[1 + 3 * i + j];
            t[i] = win[10 + i];
        }
        wave_sync();
        return member == 0;
    }
    // one candidate after the other, or the one asked for (every member of a group runs it)
    double best_err = 0;
    for (int c = 1; c <= 3; c++) {
        if (only != 0 && c != only)
            continue;
        double betas[4], Rc[3][3], tc[3];
        find_betas_approx(c, S, rho, betas, grp);
        PNP_STAMP(6 + 3 * (c - 1));
        gauss_newton(S, rho, betas);
        PNP_STAMP(7 + 3 * (c - 1));
        const double err = compute_R_and_t(P, e, S, N, betas, Rc, tc);
        PNP_STAMP(8 + 3 * (c - 1));
        // N = 1; if (err2 < err1) N = 2; if (err3 < err[N]) N = 3
        if (c == 1 || only != 0 || err < best_err) {
            best_err = err;
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++)
                    R[i][j] = Rc[i][j];
                t[i] = tc[i];
            }
        }
    }
    if (err_out)
        *err_out = best_err;
    return MODE == POSE_HELPED ? member == 0 : true;
}

// epnp::compute_pose on the correspondences P: both halves in one go.
template <int ST, PoseMode MODE = POSE_LANE, int HPW = ST>
__device__ __forceinline__ bool epnp_pose(const Points &P, const EpnpLds<ST> &S, double R[3][3], double t[3],
                          int lane = 0, int only = 0, double *err_out = nullptr)
{
    EpnpState e;
    epnp_front<ST, MODE, HPW>(P, S, e, lane);
    return epnp_back<ST, MODE, HPW>(P, S, e, R, t, lane, only, err_out);
}

// cv::Rodrigues matrix -> vector (R re-orthonormalised by SVD first).
template <int ST>
__device__ void rodrigues_mat_to_vec(const EpnpLds<ST> &S, const double Rin[3][3], double r[3])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            if (!(fabs(Rin[i][j]) < 100.)) {
                r[0] = r[1] = r[2] = 0;
                return;
            }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            S.sa[i * 3 + j] = Rin[j][i];
    svd_small_square(S, 3);
    double R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = S.sa[0 * 3 + i] * S.sv[0 * 3 + j] + S.sa[1 * 3 + i] * S.sv[1 * 3 + j] +
                           S.sa[2 * 3 + i] * S.sv[2 * 3 + j];
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0)
            rx = ry = rz = 0;
        else {
            double q;
            q = (R[0] + 1) * 0.5;
            rx = sqrt(q > 0. ? q : 0.);
            q = (R[4] + 1) * 0.5;
            ry = sqrt(q > 0. ? q : 0.) * (R[1] < 0 ? -1. : 1.);
            q = (R[8] + 1) * 0.5;
            rz = sqrt(q > 0. ? q : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0))
                rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta;
            ry *= theta;
            rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth;
        ry *= vth;
        rz *= vth;
    }
    r[0] = rx;
    r[1] = ry;
    r[2] = rz;
}

// cv::Rodrigues vector -> matrix.
__device__ void rodrigues_vec_to_mat(const double r[3], double R[9])
{
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++)
            R[i] = (i % 4 == 0) ? 1. : 0.;
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    const double x = r[0] * itheta, y = r[1] * itheta, z = r[2] * itheta;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double r_x[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int i = 0; i < 9; i++)
        R[i] = c * ((i % 4 == 0) ? 1. : 0.) + c1 * rrt[i] + s * r_x[i];
}

// projectPoints (zero distortion) + float squared error + threshold test of
// PnPRansacCallback::computeError / findInliers for point i.
__device__ __forceinline__ bool inlier_of(double X, double Y, double Z, float u, float v,
                                          const double *R, const double *t, double fx, double fy,
                                          double cx, double cy, float thr2)
{
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    const float px = (float)(x * fx + cx), py = (float)(y * fy + cy);
    const float dx = u - px, dy = v - py;
    const float err = dx * dx + dy * dy;
    return err <= thr2;
}

__device__ inline bool is_inlier(const float *obj, const float *img, int64_t i, const double *R,
                                 const double *t, double fx, double fy, double cx, double cy,
                                 float thr2)
{
    return inlier_of(obj[3 * i], obj[3 * i + 1], obj[3 * i + 2], img[2 * i], img[2 * i + 1], R, t,
                     fx, fy, cx, cy, thr2);
}

__device__ inline Points make_points(const float *obj, const float *img, const int32_t *idx, int n,
                                     CameraK K, bool points_are_f32 = true)
{
    Points P;
    P.points_are_f32 = points_are_f32;
    P.obj = obj;
    P.img = img;
    P.idx = idx;
    P.n = n;
    P.first = 0;
    P.step = 1;
    P.fu = K.fx;
    P.fv = K.fy;
    P.uc = K.cx;
    P.vc = K.cy;
    P.ifx = 1. / K.fx;
    P.ify = 1. / K.fy;
    return P;
}

} // namespace

// Solve kernel: lane l < HPW of workgroup b solves hypothesis h = b*HPW + l (EPnP
// on rows subsets[5h .. 5h+5)) and writes its model (rvec, tvec) to models[6h..]; the
// lanes l + HPW, l + 2*HPW, ... of the wave join it for the Jacobi sweeps (POSE_HELPED).
// Dynamic LDS: max(HPW, LDS_MIN_STRIDE) * LDS_DOUBLES doubles.
// A batch holds n_problems independent problems of hyp_per_problem hypotheses
// each; problem p owns the correspondence rows [begins[p], ends[p]) (begins ==
// nullptr: one problem over all rows) and its subsets index rows of that range.
template <int HPW>
__global__ __launch_bounds__(64) void ransac_solve_kernel(const float *__restrict__ obj,
                                                          const float *__restrict__ img, CameraK K,
                                                          const int32_t *__restrict__ subsets,
                                                          const int32_t *__restrict__ begins,
                                                          int hyp_per_problem, int h0, int n_hyp,
                                                          double *__restrict__ models)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x, hl = lane % HPW;
    const int h = h0 + blockIdx.x * HPW + hl;
    if (h >= n_hyp)
        return;
    constexpr int ST = HPW < LDS_MIN_STRIDE ? LDS_MIN_STRIDE : HPW;
    const EpnpLds<ST> S((lds_double *)lds + hl);
    const int64_t row0 = begins ? begins[h / hyp_per_problem] : 0;
    const Points P = make_points(obj + 3 * row0, img + 2 * row0, subsets + (int64_t)h * 5, 5, K);
    double R[3][3], t[3], r[3];
    if (!epnp_pose<ST, HPW < 64 ? POSE_HELPED : POSE_LANE, HPW>(P, S, R, t, lane))
        return;
    rodrigues_mat_to_vec(S, R, r);
    PNP_STAMP(15);
    double *m = models + (int64_t)h * 6;
    m[0] = r[0]; m[1] = r[1]; m[2] = r[2];
    m[3] = t[0]; m[4] = t[1]; m[5] = t[2];
}

// The solve of a large batch as TWO kernels -- an experiment that is kept selectable
// (ransac_set_solve_variant) but is NOT the default, because it measured slower.  The
// 12 x 12 decomposition (60 % of a solve) needs few registers beside its two rows, so the
// first half can run with TWO waves per SIMD: 16 hypotheses per wave, four lanes per
// hypothesis, LDS stride 16 -- 8 workgroups of 20 KB per CU, the same 128 hypotheses per CU
// as the single kernel.  It hands the null-space vectors (rows 8..11 of ut), the control
// points and the basis inverse to the second half through HBM, 69 doubles per hypothesis,
// element-major (front[e * stride + slot], so that a wave's accesses coalesce).  Every
// operation of a hypothesis is the one ransac_solve_kernel performs, in its order: the
// models are bit-identical.  Measured per 32 768 hypotheses: first half 266-283 us + second
// half 144-154 us against 392-408 us for the single kernel.  The reason is the fp64 pipe: a
// lone wave issues an fp64 instruction every 7.2 clocks, two waves on a SIMD one every 5.7
// between them (the pipe's limit is 4.5), so the second wave adds a quarter, not a half,
// which the 23 rounds per sweep of four lanes per matrix against 35 of two lanes at twice
// the hypotheses per wave give back exactly.
constexpr int FRONT_DOUBLES = 48 + 12 + 9;

template <int HPW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void ransac_front_kernel(
    const float *__restrict__ obj, const float *__restrict__ img, CameraK K,
    const int32_t *__restrict__ subsets, const int32_t *__restrict__ begins, int hyp_per_problem,
    int h0, int n_hyp, double *__restrict__ front, int64_t stride)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x, hl = lane % HPW;
    const int h = h0 + blockIdx.x * HPW + hl;
    if (h >= n_hyp)
        return;
    const EpnpLds<HPW> S((lds_double *)lds + hl);
    const int64_t row0 = begins ? begins[h / hyp_per_problem] : 0;
    const Points P = make_points(obj + 3 * row0, img + 2 * row0, subsets + (int64_t)h * 5, 5, K);
    EpnpState e;
    epnp_front<HPW, POSE_HELPED, HPW>(P, S, e, lane);
    if (lane >= HPW) // every member of a group holds the same values: the first one reports
        return;
    double *out = front + (h - h0);
#pragma unroll
    for (int k = 0; k < 48; k++)
        out[k * stride] = S.ut[96 + k];
#pragma unroll
    for (int k = 0; k < 12; k++)
        out[(48 + k) * stride] = e.cws[k / 3][k % 3];
#pragma unroll
    for (int k = 0; k < 9; k++)
        out[(60 + k) * stride] = e.ci[k];
}

template <int HPW, int WAVES_PER_SIMD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES_PER_SIMD, WAVES_PER_SIMD)))
void ransac_back_kernel(const float *__restrict__ obj, const float *__restrict__ img, CameraK K,
                        const int32_t *__restrict__ subsets, const int32_t *__restrict__ begins,
                        int hyp_per_problem, int h0, int n_hyp, const double *__restrict__ front,
                        int64_t stride, double *__restrict__ models)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x, hl = lane % HPW;
    const int h = h0 + blockIdx.x * HPW + hl;
    if (h >= n_hyp)
        return;
    constexpr int ST = WAVES_PER_SIMD > 1 ? HPW : (HPW < LDS_MIN_STRIDE ? LDS_MIN_STRIDE : HPW);
    const EpnpLds<ST> S((lds_double *)lds + hl);
    const int64_t row0 = begins ? begins[h / hyp_per_problem] : 0;
    const Points P = make_points(obj + 3 * row0, img + 2 * row0, subsets + (int64_t)h * 5, 5, K);
    EpnpState e;
    const double *in = front + (h - h0);
    // (the members of a group all write the same values to the same words)
#pragma unroll
    for (int k = 0; k < 48; k++)
        S.ut[96 + k] = in[k * stride];
#pragma unroll
    for (int k = 0; k < 12; k++)
        e.cws[k / 3][k % 3] = in[(48 + k) * stride];
#pragma unroll
    for (int k = 0; k < 9; k++)
        e.ci[k] = in[(60 + k) * stride];
    wave_sync();
    double R[3][3], t[3], r[3];
    if (!epnp_back<ST, POSE_HELPED, HPW>(P, S, e, R, t, lane))
        return;
    rodrigues_mat_to_vec(S, R, r);
    double *m = models + (int64_t)h * 6;
    m[0] = r[0]; m[1] = r[1]; m[2] = r[2];
    m[3] = t[0]; m[4] = t[1]; m[5] = t[2];
}

// ransac_solve_kernel for batches so small that most of the GPU would idle: a workgroup of
// THREE waves serves HPW hypotheses, wave w solving candidate w + 1 (beta initialisation)
// of each on its own LDS copy after repeating the common part for itself, as in
// epnp_solve_three_waves; the candidates' errors and poses meet in LDS behind the three
// copies (HPW x 3 x 13 doubles) and wave 0 applies the selection rule.  Every operation
// of a candidate is the one ransac_solve_kernel performs: the models are bit-identical.
// Dynamic LDS: (3 * max(HPW, LDS_MIN_STRIDE) * LDS_DOUBLES + HPW * 39) doubles.
template <int HPW>
__global__ __launch_bounds__(64 * 3) void ransac_solve3_kernel(
    const float *__restrict__ obj, const float *__restrict__ img, CameraK K,
    const int32_t *__restrict__ subsets, const int32_t *__restrict__ begins, int hyp_per_problem,
    int h0, int n_hyp, double *__restrict__ models)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane % HPW;
    const int h = h0 + blockIdx.x * HPW + hl;
    constexpr int ST = HPW < LDS_MIN_STRIDE ? LDS_MIN_STRIDE : HPW;
    const EpnpLds<ST> S((lds_double *)lds + wave * (ST * LDS_DOUBLES) + hl);
    lds_double *res = (lds_double *)lds + 3 * (ST * LDS_DOUBLES) + hl * 39;
    const bool live = h < n_hyp; // a dead lane skips the solve but its wave reaches the barrier
    double R[3][3], t[3], r[3];
    bool reports = false;
    if (live) {
        const int64_t row0 = begins ? begins[h / hyp_per_problem] : 0;
        const Points P =
            make_points(obj + 3 * row0, img + 2 * row0, subsets + (int64_t)h * 5, 5, K);
        double err;
        reports = epnp_pose<ST, POSE_HELPED, HPW>(P, S, R, t, lane, wave + 1, &err);
        if (reports) {
            lds_double *out = res + 13 * wave;
            out[0] = err;
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++)
                    out[1 + 3 * i + j] = R[i][j];
                out[10 + i] = t[i];
            }
        }
    }
    __syncthreads();
    if (wave != 0 || !reports)
        return;
    // N = 1; if (err2 < err1) N = 2; if (err3 < err[N]) N = 3
    int best = 0;
    if (res[13] < res[0])
        best = 1;
    if (res[26] < res[13 * best])
        best = 2;
    const lds_double *win = res + 13 * best;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            R[i][j] = win[1 + 3 * i + j];
        t[i] = win[10 + i];
    }
    rodrigues_mat_to_vec(S, R, r);
    PNP_STAMP(15);
    double *m = models + (int64_t)h * 6;
    m[0] = r[0]; m[1] = r[1]; m[2] = r[2];
    m[3] = t[0]; m[4] = t[1]; m[5] = t[2];
}

// A wave-uniform double moved to scalar registers (every lane holds the same value).
__device__ __forceinline__ double to_sgpr(double v)
{
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// Scoring kernel.  Every wave scores SCORE_HPW consecutive hypotheses of one problem, two
// at a time: a correspondence is loaded once (from L2: a problem's rows are read by thousands
// of waves) for two models; lanes stride over the rows; inlier count by ballot + popcount.
//
// The verdict of a (row, model) is that of inlier_of() -- projectPoints in fp64 in the
// order of the sequential statement, squared error in fp32, err <= thr2 -- but it is
// seldom COMPUTED that way.  A first pass evaluates the squared error in packed fp32 (both
// models of the pair in one v_pk_* instruction) together with a bound B on how far that
// value can lie from the one inlier_of() computes; err32 + B <= thr2 is a certain inlier,
// err32 - B > thr2 a certain outlier, and only when some lane of the wave can say neither
// (a point within about 0.5 % of the threshold circle, a point at almost no depth, a NaN
// model) the wave runs inlier_of() for that batch of 64 rows.  Counts are therefore
// identical to the exact evaluation, at about half its instructions.
//
// The bound (fast_verdicts): with u = 2^-24, S = |X| + |Y| + |Z| + max|t_i| >= the sum of
// the magnitudes in a camera-frame coordinate (|R_ij| <= 1), every fp32 coordinate is within
// 5 u S of the true one; with k = 2^-20 S / |z| (>= 3 x that over the depth) the quotient
// x/z is off by at most k (1 + |q|) + 2^-21.4 |q|, the pixel by eps (G + |p - c| + |d|) with
// eps = k + 2^-19 + 2^-21 + 2^-22, G = f + 2 |c|, and the same bounds hold, with room, for
// the fp64 evaluation rounded to fp32 as inlier_of() does.  In squares (no absolute
// values, which packed fp32 lacks): Delta^2 <= 4 (k^2 + ceps^2) (G^2 + 2 M^2),
// M^2 = f^2 (qx^2 + qy^2) + err32, and |err32 - err| <= 2 sqrt(2 err32) Delta + 2 Delta^2 +
// 8u err32 <= err32 / 512 + 1026 Delta^2 + 8u err32 for each of the two evaluations
// (2ab <= a^2 / 512 + 512 b^2): B = err32 / 251 + 2075 Delta^2.  k <= 2^-4 is required (depth
// not below S / 65536, so that the quotient's own error stays a small fraction of it), and
// anything non-finite fails every comparison, i.e. is "uncertain".
// tools/score_filter_check measures the bound against the exact error on benign and wild
// models: no certain verdict differs, the observed |err32 - err| / B stays below 0.01.
// (Staging the rows in LDS -- 40 KB per workgroup for 2000 rows -- was measured: alone the
// kernel runs no faster, and the LDS it takes keeps solve workgroups of the next round off
// the CU: the step got 8 % slower.)
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat2(float v) { return f2{v, v}; }

// Two models in fp32, component x = the first, y = the second.
struct FastModels {
    f2 R[9], t[3];
    f2 tmax;          // max |t_i| of each model, rounded up
    float fx, fy, cx, cy;
    float f2max;      // max(fx, fy)^2, rounded up
    float G2;         // (max(fx, fy) + 2 max(|cx|, |cy|))^2, rounded up
};

__device__ __forceinline__ FastModels fast_models(const double *Ra, const double *ta,
                                                  const double *Rb, const double *tb, double fx,
                                                  double fy, double cx, double cy)
{
    FastModels F;
#pragma unroll
    for (int e = 0; e < 9; e++)
        F.R[e] = f2{(float)Ra[e], (float)Rb[e]};
#pragma unroll
    for (int e = 0; e < 3; e++)
        F.t[e] = f2{(float)ta[e], (float)tb[e]};
    const double ma = fmax(fabs(ta[0]), fmax(fabs(ta[1]), fabs(ta[2])));
    const double mb = fmax(fabs(tb[0]), fmax(fabs(tb[1]), fabs(tb[2])));
    F.tmax = f2{(float)(ma * 1.000001), (float)(mb * 1.000001)};
    F.fx = (float)fx;
    F.fy = (float)fy;
    F.cx = (float)cx;
    F.cy = (float)cy;
    const double f = fmax(fabs(fx), fabs(fy)), c = fmax(fabs(cx), fabs(cy));
    F.f2max = (float)(f * f * 1.000001);
    F.G2 = (float)((f + 2 * c) * (f + 2 * c) * 1.000001);
    return F;
}

// Whether the fp32 first pass may be used for this pair of models at all (wave-uniform).
__device__ __forceinline__ bool fast_models_usable(const FastModels &F)
{
    // false for NaN and infinity as well; the lower limit keeps 2^-20 S a normal number
    return (F.tmax.x <= 0x1p59f) & (F.tmax.y <= 0x1p59f) & (F.tmax.x >= 0x1p-60f) &
           (F.tmax.y >= 0x1p-60f);
}

// First pass of the scoring kernel for one row and two models: err32 and the bound B
// (see the kernel's comment).  certain-inlier / certain-outlier / neither, per model.
struct FastVerdict {
    bool in_a, out_a, in_b, out_b; // "tame" already folded in: neither set = uncertain
    f2 err, bound;
};

__device__ __forceinline__ FastVerdict fast_verdicts(const FastModels &F, float X, float Y, float Z,
                                                     float u, float v, float thr2)
{
    const float C1 = 0x1p-20f, CE2 = 9.0e-12f /* > (2^-19 + 2^-21 + 2^-22)^2 */, K2MAX = 0x1p-8f;
    const f2 x = fma2(F.R[2], splat2(Z), fma2(F.R[1], splat2(Y), fma2(F.R[0], splat2(X), F.t[0])));
    const f2 y = fma2(F.R[5], splat2(Z), fma2(F.R[4], splat2(Y), fma2(F.R[3], splat2(X), F.t[1])));
    const f2 z = fma2(F.R[8], splat2(Z), fma2(F.R[7], splat2(Y), fma2(F.R[6], splat2(X), F.t[2])));
    const f2 iz = f2{__builtin_amdgcn_rcpf(z.x), __builtin_amdgcn_rcpf(z.y)};
    const f2 qx = x * iz, qy = y * iz;
    const f2 px = fma2(qx, splat2(F.fx), splat2(F.cx)), py = fma2(qy, splat2(F.fy), splat2(F.cy));
    const f2 dx = splat2(u) - px, dy = splat2(v) - py;
    const f2 e2 = fma2(dy, dy, dx * dx);
    const float S0 = fabsf(X) + fabsf(Y) + fabsf(Z);
    const float Sp = S0 * C1;
    const f2 w = fma2(F.tmax, splat2(C1), splat2(Sp)) * iz; // k = 2^-20 S / z (its sign squares away)
    const f2 k2 = w * w;
    const f2 M2 = fma2(fma2(qy, qy, qx * qx), splat2(F.f2max), e2);
    const f2 T = fma2(M2, splat2(2.f), splat2(F.G2));
    const f2 D2 = (k2 + splat2(CE2)) * T;                       // Delta^2 / 4
    const f2 B = fma2(D2, splat2(8300.f), e2 * splat2(0x1.05p-8f)); // err32 / 251 + 2075 Delta^2
    const f2 up = e2 + B, dn = e2 - B;
    FastVerdict r;
    // (magnitudes: the caller admits only models with max |t_i| <= 2^59; with the row's
    // coordinates below that too, no coordinate exceeds 2^60 and v_rcp_f32, which flushes
    // denormal results to zero, stays in its accurate range; a depth too small for it
    // comes back infinite and fails the k test)
    const bool row_ok = S0 <= 0x1p59f;
    const bool tame_a = (k2.x <= K2MAX) & row_ok, tame_b = (k2.y <= K2MAX) & row_ok;
    r.in_a = (up.x <= thr2) & tame_a;
    r.out_a = (dn.x > thr2) & tame_a;
    r.in_b = (up.y <= thr2) & tame_b;
    r.out_b = (dn.y > thr2) & tame_b;
    r.err = e2;
    r.bound = B;
    return r;
}

#ifdef SCORE_STATS
// tools/score_filter_check: [0] (row, model) verdicts, [1] uncertain ones, [2] certain
// verdicts that differ from inlier_of(), [3] batches of 64 rows sent to the exact path,
// [4] batches, [5] max over certain verdicts of |err32 - err| / B as float bits.
__device__ unsigned long long g_score_stats[8];
__device__ __forceinline__ float exact_err(double X, double Y, double Z, float u, float v,
                                           const double *R, const double *t, double fx, double fy,
                                           double cx, double cy)
{
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    const float px = (float)(x * fx + cx), py = (float)(y * fy + cy);
    const float dx = u - px, dy = v - py;
    return dx * dx + dy * dy;
}
#endif

constexpr int SCORE_WAVES = 4;
constexpr int SCORE_HPW = 2;
constexpr int SCORE_HPB = SCORE_WAVES * SCORE_HPW;

__global__ __launch_bounds__(64 * SCORE_WAVES) void ransac_score_kernel(
    const float *__restrict__ obj, const float *__restrict__ img, int n, CameraK K,
    const double *__restrict__ models, const int32_t *__restrict__ begins,
    const int32_t *__restrict__ ends, int hyp_per_problem, int h0, int n_hyp, float thr2,
    int32_t *__restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ha = h0 + (blockIdx.x * SCORE_WAVES + wave) * SCORE_HPW, hb = ha + 1;
    if (ha >= n_hyp)
        return;
    const bool vb = hb < n_hyp; // (an odd tail: the second chain repeats the first)
    const double fx = K.fx, fy = K.fy, cx = K.cx, cy = K.cy;
    double Ra[9], ta[3], Rb[9], tb[3];
    {
        const double *ma = models + (int64_t)ha * 6;
        const double *mb = models + (int64_t)(vb ? hb : ha) * 6;
        const double ra[3] = {ma[0], ma[1], ma[2]}, rb[3] = {mb[0], mb[1], mb[2]};
        rodrigues_vec_to_mat(ra, Ra);
        rodrigues_vec_to_mat(rb, Rb);
#pragma unroll
        for (int e = 0; e < 9; e++) {
            Ra[e] = to_sgpr(Ra[e]);
            Rb[e] = to_sgpr(Rb[e]);
        }
#pragma unroll
        for (int e = 0; e < 3; e++) { // (scalar registers are spent on R: t stays in vector ones)
            ta[e] = ma[3 + e];
            tb[e] = mb[3 + e];
        }
    }
    // the two hypotheses may belong to different problems (hyp_per_problem odd)
    const int pa = begins ? ha / hyp_per_problem : 0, pb = begins && vb ? hb / hyp_per_problem : pa;
    int ga = 0, gb = 0;
    if (pa == pb) {
        const int64_t row0 = begins ? begins[pa] : 0;
        const int rows = begins ? ends[pa] - begins[pa] : n;
        const float *o = obj + 3 * row0, *im = img + 2 * row0;
        const FastModels F = fast_models(Ra, ta, Rb, tb, fx, fy, cx, cy);
        const bool usable = __builtin_amdgcn_ballot_w64(!fast_models_usable(F)) == 0;
        for (int base = 0; base < rows; base += 64) {
            const int i = base + lane;
            const bool in = i < rows;
            const int ii = in ? i : 0;
            const float Xf = o[3 * ii], Yf = o[3 * ii + 1], Zf = o[3 * ii + 2];
            const float u = im[2 * ii], v = im[2 * ii + 1];
            const FastVerdict q = fast_verdicts(F, Xf, Yf, Zf, u, v, thr2);
            const bool unsure = in & !((q.in_a | q.out_a) & (q.in_b | q.out_b) & usable);
            bool fa = q.in_a & in, fb = q.in_b & in;
#ifdef SCORE_STATS
            {
                const float ea = exact_err(Xf, Yf, Zf, u, v, Ra, ta, fx, fy, cx, cy);
                const float eb = exact_err(Xf, Yf, Zf, u, v, Rb, tb, fx, fy, cx, cy);
                const bool ca = in & usable & (q.in_a | q.out_a), cb = in & usable & (q.in_b | q.out_b);
                const int wrong = (ca & (q.in_a != (ea <= thr2))) + (cb & (q.in_b != (eb <= thr2)));
                const float ra = ca ? fabsf(q.err.x - ea) / q.bound.x : 0.f;
                const float rb = cb ? fabsf(q.err.y - eb) / q.bound.y : 0.f;
                const float rr = ra > rb ? ra : rb;
                if (in) {
                    atomicAdd(&g_score_stats[0], 2ull);
                    atomicAdd(&g_score_stats[1], (unsigned long long)(!ca + !cb));
                    atomicAdd(&g_score_stats[2], (unsigned long long)wrong);
                    atomicMax(&g_score_stats[5], (unsigned long long)__float_as_uint(rr));
                }
                const bool slow = __ballot(unsure) != 0;
                if (lane == 0) {
                    atomicAdd(&g_score_stats[4], 1ull);
                    if (slow)
                        atomicAdd(&g_score_stats[3], 1ull);
                }
            }
#endif
            if (__ballot(unsure) != 0) { // rare: the exact evaluation for this batch of rows
                double X = Xf, Y = Yf, Z = Zf;
                KEEP(X); KEEP(Y); KEEP(Z); // (keeps the fp64 chains from being hoisted out of the branch)
                fa = inlier_of(X, Y, Z, u, v, Ra, ta, fx, fy, cx, cy, thr2) & in;
                fb = inlier_of(X, Y, Z, u, v, Rb, tb, fx, fy, cx, cy, thr2) & in;
            }
            ga += __builtin_popcountll(__ballot(fa));
            gb += __builtin_popcountll(__ballot(fb));
        }
    } else {
        for (int side = 0; side < 2; side++) {
            const int p = side ? pb : pa;
            const int64_t row0 = begins[p];
            const int rows = ends[p] - begins[p];
            int g = 0;
            for (int base = 0; base < rows; base += 64) {
                const int i = base + lane;
                const bool f = i < rows && is_inlier(obj + 3 * row0, img + 2 * row0, i,
                                                     side ? Rb : Ra, side ? tb : ta, fx, fy, cx,
                                                     cy, thr2);
                g += __builtin_popcountll(__ballot(f));
            }
            if (side)
                gb = g;
            else
                ga = g;
        }
    }
    if (lane == 0) {
        counts[ha] = ga;
        if (vb)
            counts[hb] = gb;
    }
}

// EPnP on the correspondences P by a workgroup of THREE waves, one per candidate (beta
// initialisation): the candidates are independent chains after the null-space step, and a
// lone wave is bound by the latency of its own instruction stream, so the three chains run
// on three SIMDs at once instead of one after the other.  Each wave works on its own LDS
// copy (lds + wave * LDS_DOUBLES) and repeats the common part (control points, MtM, the
// 12 x 12 decomposition, L) for itself, which costs no time; every lane of a wave runs the
// whole solve on the wave's copy (all lanes write the same values), the independent Jacobi
// pairs of a step go to different lanes, and the sums over the points are split over the
// lanes and combined with wave_sum() when P.n >= 64 (the split changes the order of those
// sums: the result differs from a sequential statement by rounding, ~1e-13 relative;
// fewer points run the sequential order on every lane).  The candidates' errors and poses
// meet in LDS behind the three copies (3 x 13 doubles), and wave 0 applies OpenCV's
// selection rule.  Returns true in wave 0, whose lanes all hold rvec|tvec in r, t.
constexpr int REFIT_WAVES = 3;
constexpr int REFIT_LDS_DOUBLES = REFIT_WAVES * LDS_DOUBLES + 3 * 13;

__device__ __forceinline__ bool epnp_solve_three_waves(Points P, lds_double *lds, double r[3],
                                                       double t[3])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (P.n >= 64) {
        P.first = lane;
        P.step = 64;
    }
    const EpnpLds<1> S(lds + wave * LDS_DOUBLES);
    lds_double *res = lds + REFIT_WAVES * LDS_DOUBLES;
    double R[3][3], err;
    epnp_pose<1, POSE_WAVE>(P, S, R, t, lane, wave + 1, &err);
    if (lane == 0) {
        lds_double *out = res + 13 * wave;
        out[0] = err;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++)
                out[1 + 3 * i + j] = R[i][j];
            out[10 + i] = t[i];
        }
    }
    __syncthreads();
    if (wave != 0)
        return false;
    // N = 1; if (err2 < err1) N = 2; if (err3 < err[N]) N = 3
    int best = 0;
    if (res[13] < res[0])
        best = 1;
    if (res[26] < res[13 * best])
        best = 2;
    const lds_double *win = res + 13 * best;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            R[i][j] = win[1 + 3 * i + j];
        t[i] = win[10 + i];
    }
    rodrigues_mat_to_vec(S, R, r);
    return true;
}

// solvePnP(SOLVEPNP_EPNP) on rows idx[0..n) (idx may be null): model[6] = rvec|tvec.
// One workgroup of three waves, one problem (epnp_solve_three_waves): a refit on thousands
// of inliers costs less than a 5-point solve of one lane.
__global__ __launch_bounds__(64 * REFIT_WAVES) void epnp_solve_kernel(
    const float *__restrict__ obj, const float *__restrict__ img, const int32_t *__restrict__ idx,
    int n, CameraK K, double *__restrict__ model)
{
    extern __shared__ double lds[];
    double r[3], t[3];
    if (epnp_solve_three_waves(make_points(obj, img, idx, n, K), (lds_double *)lds, r, t) &&
        threadIdx.x == 0) {
        model[0] = r[0]; model[1] = r[1]; model[2] = r[2];
        model[3] = t[0]; model[4] = t[1]; model[5] = t[2];
    }
}

// The end of solvePnPRansac for a batch, one workgroup of three waves per problem: the
// inliers of the problem's best model (all rows where use_all_rows[b] != 0: a problem of
// exactly five rows has no model) are listed in ascending row order, then EPnP is solved on
// them as in epnp_solve_kernel.  Problem b owns rows [begins[b], ends[b]); its list (rows
// relative to begins[b]) goes to idx_dev + begins[b], which the solve reads back, and to
// idx_out + begins[b]; n_inliers[b] = its length; models_out[6b..] = rvec|tvec (left
// alone, like idx_out, when fewer than 4 rows are listed, which a model with 5 or more
// inliers cannot produce).  Every wave builds the list for itself (the three write equal
// values to the same words), so none waits for another before its solve.
// begins == nullptr: one problem, rows 0..one.rows, model one.best (passed by value: a
// single call then reads no table).
// RANSACUpdateNumIters with the two libm results it needs taken from the host (num_log =
// log(max(1 - confidence, DBL_MIN)), denom_log = log(1 - (good / n)^5), NaN where the host's
// rule returns 0), so that the product, the comparison, the correctly rounded quotient and the
// round-to-nearest-even are the host's own operations on the host's own operands.
__device__ __forceinline__ int replay_update_iters(double denom_log, double num_log, int max_iters)
{
    if (denom_log != denom_log)
        return 0;
    if (denom_log >= 0 || -num_log >= max_iters * (-denom_log))
        return max_iters;
    return (int)__builtin_rint(num_log / denom_log);
}

// RANSACPointSetRegistrator::run's selection rule over the inlier counts of `total` hypotheses,
// in their order: a hypothesis wins iff it lies inside the current budget and its count is
// strictly larger than every earlier winner's and than 4; each win shrinks the budget.  The
// whole wave runs it together, 64 counts at a time; every lane returns the same values.
// Returns the index of the last winner (-1: none); run = iterations the sequential loop
// would have gone through.
__device__ __forceinline__ int replay_selection(const RefitReplay &rp, int lane, int &run)
{
    int niters = rp.total, max_good = 4, best = -1;
    for (int base = 0; base < niters; base += 64) {
        const int i = base + lane;
        const int c = i < rp.total ? rp.counts[i] : 0;
        unsigned long long cand = __ballot(c > max_good && i < niters);
        while (cand) {
            const int p = __builtin_ctzll(cand);
            cand &= cand - 1;
            if (base + p >= niters) // the budget has shrunk below this hypothesis: the loop has ended
                break;
            const int cp = __shfl(c, p);
            if (cp > max_good) {
                max_good = cp;
                best = base + p;
                niters = replay_update_iters(rp.iter_table[cp], rp.num_log, niters);
            }
        }
    }
    run = niters > best + 1 ? niters : best + 1;
    return best;
}

__global__ __launch_bounds__(64 * REFIT_WAVES) void epnp_refit_batched_kernel(
    const float *__restrict__ obj, const float *__restrict__ img,
    const int32_t *__restrict__ begins, const int32_t *__restrict__ ends,
    const double *__restrict__ best_models, const uint8_t *__restrict__ use_all_rows,
    RefitOne one, RefitReplay replay, CameraK K, float thr2, int32_t *idx_dev,
    int32_t *__restrict__ idx_out, int32_t *__restrict__ n_inliers,
    double *__restrict__ models_out)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x;
    // a lone call whose hypotheses were scored on the device: choose the model here (every wave
    // for itself, with the same result), instead of on the host between two synchronisations
    const double *chosen = nullptr;
    if (!begins && replay.counts) {
        int run;
        const int best = replay_selection(replay, lane, run);
        if (threadIdx.x == 0) {
            replay.out[0] = run;
            replay.out[1] = best;
        }
        if (best < 0) { // no hypothesis with 5 or more inliers: no pose
            if (threadIdx.x == 0)
                n_inliers[0] = 0;
            return;
        }
        chosen = replay.models + (int64_t)best * 6;
    }
    const int64_t row0 = begins ? begins[b] : 0;
    const int rows = begins ? ends[b] - begins[b] : one.rows;
    obj += 3 * row0;
    img += 2 * row0;
    int32_t *list = idx_dev + row0;
    int k = 0;
    // a problem of exactly five rows is solved on the caller's float points; a consensus set is
    // converted to CV_64F first
    const bool all = begins ? use_all_rows[b] != 0 : false;
    {
        // constant indices: a loop over one.best would put the by-value struct in scratch
        double rv[3] = {one.best[0], one.best[1], one.best[2]};
        double tv[3] = {one.best[3], one.best[4], one.best[5]};
        if (begins || chosen) {
            const double *m = begins ? best_models + (int64_t)b * 6 : chosen;
            rv[0] = m[0]; rv[1] = m[1]; rv[2] = m[2];
            tv[0] = m[3]; tv[1] = m[4]; tv[2] = m[5];
        }
        double Rm[9];
        rodrigues_vec_to_mat(rv, Rm);
        for (int base = 0; base < rows; base += 64) {
            const int i = base + lane;
            const bool f = i < rows &&
                           (all || is_inlier(obj, img, i, Rm, tv, K.fx, K.fy, K.cx, K.cy, thr2));
            const unsigned long long votes = __ballot(f);
            if (f)
                list[k + __builtin_popcountll(votes & ((1ull << lane) - 1))] = i;
            k += __builtin_popcountll(votes);
        }
    }
    if (threadIdx.x == 0)
        n_inliers[b] = k;
    if (k < 4) // the same k in every wave: no wave is left waiting at the barrier below
        return;
    // the list was written by other lanes of this wave: make it visible to every lane's loads
    __threadfence();
    if (threadIdx.x < 64)
        for (int i = lane; i < k; i += 64)
            idx_out[row0 + i] = list[i];
    double r[3], t[3];
    if (epnp_solve_three_waves(make_points(obj, img, list, k, K, all), (lds_double *)lds, r, t) &&
        threadIdx.x == 0) {
        double *model = models_out + (int64_t)b * 6;
        model[0] = r[0]; model[1] = r[1]; model[2] = r[2];
        model[3] = t[0]; model[4] = t[1]; model[5] = t[2];
    }
}

template <int HPW>
static hipError_t launch_solve(const float *d_obj, const float *d_img, CameraK K,
                               const int32_t *d_subsets, const int32_t *d_begins,
                               int hyp_per_problem, int h0, int n_hyp, double *d_models,
                               hipStream_t stream)
{
    const size_t lds_bytes =
        (size_t)(HPW < LDS_MIN_STRIDE ? LDS_MIN_STRIDE : HPW) * LDS_DOUBLES * sizeof(double);
    // opt in to > 64 KiB of dynamic LDS (per device, so not cached across calls)
    hipError_t e = hipFuncSetAttribute((const void *)ransac_solve_kernel<HPW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess)
        return e;
    ransac_solve_kernel<HPW><<<(n_hyp - h0 + HPW - 1) / HPW, 64, lds_bytes, stream>>>(
        d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0, n_hyp, d_models);
    return hipGetLastError();
}

template <int HPW>
static hipError_t launch_solve3(const float *d_obj, const float *d_img, CameraK K,
                                const int32_t *d_subsets, const int32_t *d_begins,
                                int hyp_per_problem, int h0, int n_hyp, double *d_models,
                                hipStream_t stream)
{
    const size_t lds_bytes =
        ((size_t)3 * (HPW < LDS_MIN_STRIDE ? LDS_MIN_STRIDE : HPW) * LDS_DOUBLES + HPW * 39) *
        sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void *)ransac_solve3_kernel<HPW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess)
        return e;
    ransac_solve3_kernel<HPW><<<(n_hyp - h0 + HPW - 1) / HPW, 64 * 3, lds_bytes, stream>>>(
        d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0, n_hyp, d_models);
    return hipGetLastError();
}

// The two-kernel solve (ransac_front_kernel + ransac_back_kernel) of hypotheses [h0, n_hyp);
// d_front: FRONT_DOUBLES * (n_hyp - h0) doubles of workspace.
template <int BACK_HPW, int BACK_WAVES>
static hipError_t launch_solve_split(const float *d_obj, const float *d_img, CameraK K,
                                     const int32_t *d_subsets, const int32_t *d_begins,
                                     int hyp_per_problem, int h0, int n_hyp, double *d_models,
                                     double *d_front, hipStream_t stream)
{
    constexpr int FRONT_HPW = 16;
    const int count = n_hyp - h0;
    const size_t lds_front = (size_t)FRONT_HPW * LDS_DOUBLES * sizeof(double);
    constexpr int BACK_ST =
        BACK_WAVES > 1 ? BACK_HPW : (BACK_HPW < LDS_MIN_STRIDE ? LDS_MIN_STRIDE : BACK_HPW);
    const size_t lds_back = (size_t)BACK_ST * LDS_DOUBLES * sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void *)ransac_front_kernel<FRONT_HPW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_front);
    if (e != hipSuccess)
        return e;
    e = hipFuncSetAttribute((const void *)ransac_back_kernel<BACK_HPW, BACK_WAVES>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_back);
    if (e != hipSuccess)
        return e;
    ransac_front_kernel<FRONT_HPW><<<(count + FRONT_HPW - 1) / FRONT_HPW, 64, lds_front, stream>>>(
        d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0, n_hyp, d_front, count);
    if ((e = hipGetLastError()) != hipSuccess)
        return e;
    ransac_back_kernel<BACK_HPW, BACK_WAVES>
        <<<(count + BACK_HPW - 1) / BACK_HPW, 64, lds_back, stream>>>(
            d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0, n_hyp, d_front, count,
            d_models);
    return hipGetLastError();
}

static std::atomic<int> g_solve_variant{0}; // a process-wide measurement setting, read by every context

void ransac_set_solve_variant(int variant) { g_solve_variant = variant; }
int ransac_solve_variant() { return g_solve_variant; }

size_t ransac_front_workspace_bytes(int n_cus)
{
    return (size_t)FRONT_DOUBLES * (size_t)n_cus * 128 * sizeof(double);
}

hipError_t ransac_hypotheses_launch(const float *d_obj, const float *d_img, int n, CameraK K,
                                    const int32_t *d_subsets, const int32_t *d_begins,
                                    const int32_t *d_ends, int n_problems, int hyp_per_problem,
                                    float thr2, double *d_models, int32_t *d_counts, int n_cus,
                                    hipStream_t stream, const RansacOverlap *overlap,
                                    hipEvent_t *events, double *d_front)
{
    const int64_t total = (int64_t)n_problems * hyp_per_problem;
    if (total <= 0)
        return hipSuccess;
    if (total > 0x7FFFFFFF)
        return hipErrorInvalidValue;
    const int n_hyp = (int)total;
    // Hypotheses per wave: one wave's fp64 instruction stream keeps its SIMD busy, and
    // LDS capacity (1248 B per hypothesis -> 128 per CU) sets how many hypotheses are
    // in flight, so many hypotheses go 32 to a wave (4 workgroups per CU, one per SIMD,
    // two lanes per hypothesis).  Fewer go 16 or 8 to a wave: more lanes per hypothesis
    // shorten its solve (8 lanes take all the independent Jacobi pairs of a step and
    // all three candidates at once) while the waves still fit the SIMDs in one round.
    // One workgroup per CU or fewer (a single solvePnPRansac call with OpenCV's default 100
    // iterations): three waves per 8 hypotheses, one per candidate; per 4 hypotheses where
    // that still fits (16 lanes per hypothesis: two lanes per Jacobi pair).  Measured, solve
    // only: 100 hypotheses 0.25 ms (one wave per 8) -> 0.21 -> 0.19 ms, 2048 0.25 -> 0.22 ms.
    // Up to four hypotheses per SIMD (4096 on this chip) a wave takes 4, so that every SIMD has
    // a wave and a hypothesis 16 lanes: 4096 hypotheses x 2000 rows, solve + score, 0.237 ->
    // 0.224 ms per call against 8 per wave on half the SIMDs.
    const bool lone4 = n_hyp <= n_cus * 4;
    const bool lone = n_hyp <= n_cus * 8;
    const bool quad = n_hyp <= n_cus * 4 * 4;
    const bool tiny = n_hyp <= n_cus * 4 * 8;
    const bool small = n_hyp <= n_cus * 4 * 16;
    // A large batch goes down in rounds of one GPU-full of solves; the scoring of a
    // round runs on a second stream underneath the next round's solves (the solve
    // kernel leaves most of every SIMD's issue slots and registers for one more wave).
    const int round = n_cus * 128;
    const bool split = overlap && !small && n_hyp > round;
    // Large batches take the two-kernel solve when the caller brought its workspace (which
    // holds one round: such a batch then always goes down in rounds).
    const int variant = g_solve_variant;
    const bool two_kernel = !small && d_front != nullptr && variant >= 2;
    const bool rounds = split || (two_kernel && n_hyp > round);
    hipError_t e;
    for (int h0 = 0; h0 < n_hyp; h0 += rounds ? round : n_hyp) {
        const int h1 = rounds ? (n_hyp - h0 > round ? h0 + round : n_hyp) : n_hyp;
        const bool first = h0 == 0;
        if (events && first && (e = hipEventRecord(events[0], stream)) != hipSuccess)
            return e;
        if (events && h1 == n_hyp && (e = hipEventRecord(events[2], stream)) != hipSuccess)
            return e;
        if (lone4)
            e = launch_solve3<4>(d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0, h1,
                                 d_models, stream);
        else if (lone)
            e = launch_solve3<8>(d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0, h1,
                                 d_models, stream);
        else if (quad)
            e = launch_solve<4>(d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0, h1,
                                d_models, stream);
        else if (tiny)
            e = launch_solve<8>(d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0, h1,
                                d_models, stream);
        else if (small)
            e = launch_solve<16>(d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0, h1,
                                 d_models, stream);
        else if (two_kernel && variant == 3)
            e = launch_solve_split<16, 2>(d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0,
                                          h1, d_models, d_front, stream);
        else if (two_kernel)
            e = launch_solve_split<32, 1>(d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0,
                                          h1, d_models, d_front, stream);
        else
            e = launch_solve<32>(d_obj, d_img, K, d_subsets, d_begins, hyp_per_problem, h0, h1,
                                 d_models, stream);
        if (e != hipSuccess)
            return e;
        const bool last = h1 == n_hyp;
        if (events && first && (e = hipEventRecord(events[1], stream)) != hipSuccess)
            return e;
        if (events && last && (e = hipEventRecord(events[3], stream)) != hipSuccess)
            return e;
        hipStream_t ss = stream;
        if (split && !last) {
            if ((e = hipEventRecord(overlap->solved, stream)) != hipSuccess)
                return e;
            if ((e = hipStreamWaitEvent(overlap->side, overlap->solved, 0)) != hipSuccess)
                return e;
            ss = overlap->side;
        }
        ransac_score_kernel<<<(h1 - h0 + SCORE_HPB - 1) / SCORE_HPB, 64 * SCORE_WAVES, 0, ss>>>(
            d_obj, d_img, n, K, d_models, d_begins, d_ends, hyp_per_problem, h0, h1, thr2,
            d_counts);
        if ((e = hipGetLastError()) != hipSuccess)
            return e;
        if (events && last && (e = hipEventRecord(events[4], stream)) != hipSuccess)
            return e;
    }
    if (split) { // the caller's stream continues only after the side stream's scoring
        if ((e = hipEventRecord(overlap->scored, overlap->side)) != hipSuccess)
            return e;
        if ((e = hipStreamWaitEvent(stream, overlap->scored, 0)) != hipSuccess)
            return e;
    }
    return hipSuccess;
}

// ---- fp64 VALU peak -------------------------------------------------------------
// Eight independent chains per lane of v_fma_f64 (FMA = true) or of v_mul_f64 + v_add_f64
// pairs (the solve is built without contraction, so this is its instruction mix), 8 waves
// per SIMD on every CU.
template <bool FMA>
__global__ __launch_bounds__(256) void fp64_peak_kernel(double *out, double a, double b, int iters)
{
    double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5,
           x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (FMA)
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n"
                             "v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n"
                             "v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6),
                               "+v"(x7)
                             : "v"(a), "v"(b));
            else
                asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n"
                             "v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                             "v_add_f64 %4, %4, %9\n v_add_f64 %5, %5, %9\n"
                             "v_add_f64 %6, %6, %9\n v_add_f64 %7, %7, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6),
                               "+v"(x7)
                             : "v"(a), "v"(b));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

hipError_t valu_fp64_peak(int n_cus, int iters, double *d_scratch, hipStream_t stream,
                          double *fma_per_s, double *mul_add_per_s)
{
    const int blocks = n_cus * 8; // 8 workgroups x 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    hipError_t e;
    if ((e = hipEventCreate(&e0)) != hipSuccess)
        return e;
    if ((e = hipEventCreate(&e1)) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return e;
    }
    float best[2] = {0.f, 0.f};
    for (int form = 0; form < 2 && e == hipSuccess; form++)
        for (int r = 0; r < 4; r++) { // first pass warms up
            (void)hipEventRecord(e0, stream);
            if (form == 0)
                fp64_peak_kernel<true><<<blocks, 256, 0, stream>>>(d_scratch, 1.0000001, 1e-9, iters);
            else
                fp64_peak_kernel<false><<<blocks, 256, 0, stream>>>(d_scratch, 1.0000001, 1e-9, iters);
            (void)hipEventRecord(e1, stream);
            if ((e = hipEventSynchronize(e1)) != hipSuccess)
                break;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && (best[form] == 0.f || ms < best[form]))
                best[form] = ms;
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess)
        return e;
    const double ops = (double)blocks * 256.0 * (double)iters * 8 * 8;
    *fma_per_s = ops / (best[0] * 1e-3);
    *mul_add_per_s = ops / (best[1] * 1e-3);
    return hipGetLastError();
}

hipError_t epnp_solve_launch(const float *d_obj, const float *d_img, const int32_t *d_idx, int n,
                             CameraK K, double *d_model, hipStream_t stream)
{
    const size_t lds_bytes = (size_t)REFIT_LDS_DOUBLES * sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void *)epnp_solve_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess)
        return e;
    epnp_solve_kernel<<<1, 64 * REFIT_WAVES, lds_bytes, stream>>>(d_obj, d_img, d_idx, n, K, d_model);
    return hipGetLastError();
}

hipError_t epnp_refit_batched_launch(const float *d_obj, const float *d_img,
                                     const int32_t *d_begins, const int32_t *d_ends,
                                     const double *d_best_models, const uint8_t *d_use_all_rows,
                                     const RefitOne &one, int n_problems, CameraK K, float thr2,
                                     int32_t *d_idx,
                                     int32_t *d_idx_out, int32_t *d_n_inliers,
                                     double *d_models_out, hipStream_t stream,
                                     const RefitReplay *replay)
{
    if (n_problems <= 0)
        return hipSuccess;
    if (replay && (d_begins || n_problems != 1 || !replay->counts || !replay->models ||
                   !replay->iter_table || !replay->out || replay->total < 1))
        return hipErrorInvalidValue;
    const size_t lds_bytes = (size_t)REFIT_LDS_DOUBLES * sizeof(double);
    epnp_refit_batched_kernel<<<n_problems, 64 * REFIT_WAVES, lds_bytes, stream>>>(
        d_obj, d_img, d_begins, d_ends, d_best_models, d_use_all_rows, one,
        replay ? *replay : RefitReplay{}, K, thr2, d_idx, d_idx_out, d_n_inliers, d_models_out);
    return hipGetLastError();
}

} // namespace posehip
