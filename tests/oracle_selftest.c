/* Exercises the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (host code
 * only): exactly-sized heap buffers around every call, odd sizes, strided rows, fewer train
 * rows than k, degenerate PnP inputs.  Built and run by tests/test_oracle_sanitizers.py. */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } orc_dmatch;
int orc_knn_hamming256(const uint8_t *, int, size_t, const uint8_t *, int, size_t, int, int32_t *, int32_t *);
int orc_knn_hamming256_mt(const uint8_t *, int, size_t, const uint8_t *, int, size_t, int, int32_t *, int32_t *, int);
int orc_match(const uint8_t *, int, size_t, const uint8_t *, int, size_t, orc_dmatch *);
int orc_match_crosscheck(const uint8_t *, int, size_t, const uint8_t *, int, size_t, orc_dmatch *);
int orc_ratio_filter(const int32_t *, const int32_t *, int, float, orc_dmatch *);
int orc_min_dist_filter(const orc_dmatch *, int, orc_dmatch *);
void orc_sort_by_distance(orc_dmatch *, int);
void orc_ransac_subsets(uint64_t, int, int, int, int32_t *);
int orc_solvepnp_epnp(const float *, const float *, int, const double *, double *, double *);
int orc_pnp_hypotheses(const float *, const float *, int, const double *, const int32_t *, int, double, double *, int32_t *, int);
int orc_solvepnp_ransac(const float *, const float *, int, const double *, int, float, double, uint64_t, double *, double *, int32_t *, int *, int *);

static uint64_t s = 88172645463325252ull;
static uint32_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); }
static double unit(void) { return rnd() / 4294967296.0 * 2 - 1; }

static void matcher_case(int nq, int nt, size_t step, int k)
{
    uint8_t *q = malloc((size_t)(nq ? nq : 1) * step), *t = malloc((size_t)(nt ? nt : 1) * step);
    for (size_t i = 0; i < (size_t)nq * step; i++) q[i] = (uint8_t)(rnd() & 3); /* many ties */
    for (size_t i = 0; i < (size_t)nt * step; i++) t[i] = (uint8_t)(rnd() & 3);
    int32_t *idx = malloc(sizeof(int32_t) * (size_t)(nq ? nq : 1) * k), *dist = malloc(sizeof(int32_t) * (size_t)(nq ? nq : 1) * k);
    int32_t *idx2 = malloc(sizeof(int32_t) * (size_t)(nq ? nq : 1) * k), *dist2 = malloc(sizeof(int32_t) * (size_t)(nq ? nq : 1) * k);
    orc_dmatch *m = malloc(sizeof(orc_dmatch) * (size_t)(nq ? nq : 1)), *m2 = malloc(sizeof(orc_dmatch) * (size_t)(nq ? nq : 1));
    if (orc_knn_hamming256(q, nq, step, t, nt, step, k, idx, dist) != 0) exit(2);
    if (orc_knn_hamming256_mt(q, nq, step, t, nt, step, k, idx2, dist2, 3) != 0) exit(3);
    if (nq && nt && (memcmp(idx, idx2, sizeof(int32_t) * (size_t)nq * k) || memcmp(dist, dist2, sizeof(int32_t) * (size_t)nq * k))) exit(4);
    int n = orc_match(q, nq, step, t, nt, step, m);
    if (n != (nq > 0 && nt > 0 ? nq : 0)) exit(5);
    n = orc_min_dist_filter(m, n, m2);
    orc_sort_by_distance(m2, n);
    n = orc_match_crosscheck(q, nq, step, t, nt, step, m);
    if (n < 0 || n > nq) exit(6);
    if (k == 2 && nq && nt) {
        n = orc_ratio_filter(idx, dist, nq, 0.8f, m);
        if (n < 0 || n > nq) exit(7);
    }
    free(q); free(t); free(idx); free(dist); free(idx2); free(dist2); free(m); free(m2);
}

static void pnp_case(int n, int mode, int iters)
{
    const double f = 300.0 / tan(22.5 * M_PI / 180), K[9] = {f, 0, 400, 0, f, 300, 0, 0, 1};
    float *obj = malloc(sizeof(float) * 3 * (size_t)n), *img = malloc(sizeof(float) * 2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        double X = unit(), Y = 1.3 * unit(), Z = 0.4 * unit();
        if (mode == 1) { Y = 0.5 * X; Z = 0.2 * X; }      /* collinear */
        if (mode == 2) { X = 0.3; Y = -0.2; Z = 0.1; }    /* coincident */
        if (mode == 3) { X = Y = Z = 0; }                 /* all zero */
        const double x = cos(0.6) * X + sin(0.6) * Z, y = Y, z = -sin(0.6) * X + cos(0.6) * Z + 6;
        obj[3 * i] = (float)X; obj[3 * i + 1] = (float)Y; obj[3 * i + 2] = (float)Z;
        const int out = (rnd() % 10) < 3;
        img[2 * i] = (float)(out ? 400 + 400 * unit() : x / z * f + 400 + unit());
        img[2 * i + 1] = (float)(out ? 300 + 300 * unit() : y / z * f + 300 + unit());
    }
    double rvec[3], tvec[3];
    int32_t *inl = malloc(sizeof(int32_t) * (size_t)n);
    int n_inl = 0, it = 0;
    (void)orc_solvepnp_ransac(obj, img, n, K, iters, 8.0f, 0.99, 0xFFFFFFFFFFFFFFFFull, rvec, tvec, inl, &n_inl, &it);
    if (n_inl < 0 || n_inl > n) exit(8);
    if (n >= 4) (void)orc_solvepnp_epnp(obj, img, n, K, rvec, tvec);
    if (n > 5) {
        const int H = 24;
        int32_t *sub = malloc(sizeof(int32_t) * 5 * H), *cnt = malloc(sizeof(int32_t) * H);
        double *models = malloc(sizeof(double) * 6 * H);
        orc_ransac_subsets(0xFFFFFFFFFFFFFFFFull, n, 5, H, sub);
        if (orc_pnp_hypotheses(obj, img, n, K, sub, H, 8.0, models, cnt, 3) != 0) exit(9);
        free(sub); free(cnt); free(models);
    }
    free(obj); free(img); free(inl);
}

int main(void)
{
    const int sizes[] = {0, 1, 2, 3, 63, 64, 65, 257};
    for (unsigned a = 0; a < sizeof sizes / sizeof *sizes; a++)
        for (unsigned b = 0; b < sizeof sizes / sizeof *sizes; b++)
            for (int k = 1; k <= 2; k++) {
                matcher_case(sizes[a], sizes[b], 32, k);
                matcher_case(sizes[a], sizes[b], 41, k);
            }
    const int ns[] = {5, 6, 7, 64, 300};
    for (unsigned a = 0; a < sizeof ns / sizeof *ns; a++)
        for (int mode = 0; mode < 4; mode++)
            pnp_case(ns[a], mode, 60);
    puts("oracle selftest ok");
    return 0;
}
