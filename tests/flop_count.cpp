// Test infrastructure: counts the fp64 operations of one EPnP hypothesis solve, exactly,
// by compiling the CPU oracle (oracle/pnp_ransac.c) with `double` replaced by a counting
// type.  The GPU solve kernels perform the oracle's operations in the oracle's order (their
// models are bit-identical), so this is the algorithmic operation count of a hypothesis on
// the GPU as well; the lanes that help a hypothesis repeat some operations, which is
// overhead, not algorithmic work.
//
// Convention: add / sub / mul / div / sqrt count one FLOP each, acos / sin / cos /
// pow / log one each as well (listed separately); comparisons, fabs, negation and
// conversions count zero.  The oracle is built without contraction, so there is no fma.
//
// usage: flop_count n_points outlier_frac seed n_hyp   (the scene of tests/datagen.py is
// passed in by the test through a binary file: obj[n*3] f32, img[n*2] f32, K[9] f64,
// subsets[n_hyp*5] i32)
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

struct Counts {
    unsigned long long add = 0, mul = 0, div = 0, sqrt_ = 0, trans = 0, cmp = 0;
};
static Counts g_n;

struct CD {
    double v;
    CD() = default;
    CD(double x) : v(x) {}
    explicit operator double() const { return v; }
    explicit operator float() const { return (float)v; }
    explicit operator int() const { return (int)v; }
    explicit operator bool() const { return v != 0.0; }
    CD &operator+=(CD o) { g_n.add++; v += o.v; return *this; }
    CD &operator-=(CD o) { g_n.add++; v -= o.v; return *this; }
    CD &operator*=(CD o) { g_n.mul++; v *= o.v; return *this; }
    CD &operator/=(CD o) { g_n.div++; v /= o.v; return *this; }
    CD operator-() const { return CD(-v); }
};
static inline CD operator+(CD a, CD b) { g_n.add++; return CD(a.v + b.v); }
static inline CD operator-(CD a, CD b) { g_n.add++; return CD(a.v - b.v); }
static inline CD operator*(CD a, CD b) { g_n.mul++; return CD(a.v * b.v); }
static inline CD operator/(CD a, CD b) { g_n.div++; return CD(a.v / b.v); }
#define CD_CMP(op) static inline bool operator op(CD a, CD b) { g_n.cmp++; return a.v op b.v; }
CD_CMP(<) CD_CMP(<=) CD_CMP(>) CD_CMP(>=) CD_CMP(==) CD_CMP(!=)
static inline CD fabs(CD a) { return CD(fabs(a.v)); }
static inline CD sqrt(CD a) { g_n.sqrt_++; return CD(sqrt(a.v)); }
static inline CD acos(CD a) { g_n.trans++; return CD(acos(a.v)); }
static inline CD sin(CD a) { g_n.trans++; return CD(sin(a.v)); }
static inline CD cos(CD a) { g_n.trans++; return CD(cos(a.v)); }
static inline CD log(CD a) { g_n.trans++; return CD(log(a.v)); }
static inline CD pow(CD a, CD b) { g_n.trans++; return CD(pow(a.v, b.v)); }
static inline long lrint(CD a) { return lrint(a.v); }

#define double CD
#include "../oracle/pnp_ransac.c"
#undef double

int main(int argc, char **argv)
{
    if (argc != 4) {
        fprintf(stderr, "usage: flop_count scene.bin n_points n_hyp\n");
        return 2;
    }
    const int n = atoi(argv[2]), n_hyp = atoi(argv[3]);
    FILE *f = fopen(argv[1], "rb");
    if (!f)
        return 2;
    std::vector<float> obj((size_t)n * 3), img((size_t)n * 2);
    double K[9];
    std::vector<int32_t> sub((size_t)n_hyp * 5);
    if (fread(obj.data(), 4, obj.size(), f) != obj.size() || fread(img.data(), 4, img.size(), f) != img.size() ||
        fread(K, 8, 9, f) != 9 || fread(sub.data(), 4, sub.size(), f) != sub.size())
        return 2;
    fclose(f);
    CD Kc[9];
    for (int i = 0; i < 9; i++)
        Kc[i].v = K[i];
    Counts solve{}, score{};
    unsigned long long min_solve = ~0ull, max_solve = 0;
    for (int h = 0; h < n_hyp; h++) {
        float so[15], si[10];
        for (int i = 0; i < 5; i++) {
            memcpy(so + 3 * i, &obj[3 * (size_t)sub[5 * h + i]], 12);
            memcpy(si + 2 * i, &img[2 * (size_t)sub[5 * h + i]], 8);
        }
        CD r[3], t[3];
        g_n = Counts{};
        const int ok = orc_solvepnp_epnp(so, si, 5, Kc, r, t);
        const unsigned long long tot = g_n.add + g_n.mul + g_n.div + g_n.sqrt_ + g_n.trans;
        solve.add += g_n.add; solve.mul += g_n.mul; solve.div += g_n.div;
        solve.sqrt_ += g_n.sqrt_; solve.trans += g_n.trans; solve.cmp += g_n.cmp;
        min_solve = tot < min_solve ? tot : min_solve;
        max_solve = tot > max_solve ? tot : max_solve;
        if (ok != 0)
            continue;
        g_n = Counts{};
        orc_pnp_count_inliers(obj.data(), img.data(), n, Kc, r, t, CD(8.0), NULL, NULL);
        score.add += g_n.add; score.mul += g_n.mul; score.div += g_n.div;
        score.sqrt_ += g_n.sqrt_; score.trans += g_n.trans; score.cmp += g_n.cmp;
    }
    const double H = n_hyp;
    printf("{\"n_hyp\": %d, \"n_points\": %d, "
           "\"solve_fp64_per_hyp\": {\"add\": %.3f, \"mul\": %.3f, \"div\": %.3f, \"sqrt\": %.3f, "
           "\"transcendental\": %.3f, \"compare\": %.3f, \"flop\": %.3f, \"min\": %llu, \"max\": %llu}, "
           "\"score_fp64_per_hyp\": {\"add\": %.3f, \"mul\": %.3f, \"div\": %.3f, \"sqrt\": %.3f, "
           "\"transcendental\": %.3f, \"flop\": %.3f}}\n",
           n_hyp, n, solve.add / H, solve.mul / H, solve.div / H, solve.sqrt_ / H, solve.trans / H,
           solve.cmp / H, (solve.add + solve.mul + solve.div + solve.sqrt_ + solve.trans) / H,
           min_solve, max_solve, score.add / H, score.mul / H, score.div / H, score.sqrt_ / H,
           score.trans / H, (score.add + score.mul + score.div + score.sqrt_ + score.trans) / H);
    return 0;
}
