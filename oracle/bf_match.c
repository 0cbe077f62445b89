/*
 * oracle/bf_match.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the brute-force Hamming matcher that the reference calls
 * through OpenCV:
 *   - /root/reference/src/cereal_box.cpp:53,344   (DescriptorMatcher "BruteForce-Hamming", match())
 *   - /root/reference/src/cereal_box.cpp:345-356  (min-distance filter)
 *   - /root/reference/src/orb_feature_matching/matcher.py:18,21,24
 *                                                 (BFMatcher(NORM_HAMMING, crossCheck=True).match, sort)
 *
 * The arithmetic itself lives in a third-party dependency that is absent from
 * the reference tree: OpenCV 4.6.0 (README.md:57-61 of the reference),
 * modules/core/src/batch_distance.cpp and modules/features2d/src/matchers.cpp.
 * This file restates that published algorithm.  PARITY UNPINNED: the reference
 * holds no test, golden vector or fixture for this path and OpenCV is not
 * installable here, so the restatement is pinned only by hand-computed
 * known-answer cases in tests/ and by the one vector of OpenCV's own test suite for
 * this path (the cross-check regression test, see orc_match_crosscheck).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object; the product library never does.
 */
#include <limits.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;
    float distance;
} orc_dmatch; /* layout of cv::DMatch (16 bytes) */

/* NORM_HAMMING over one 32-byte ORB descriptor: popcount(a XOR b) in [0,256]. */
static inline int hamming256(const uint8_t *a, const uint8_t *b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

/*
 * One query row against all train rows, cv::batchDistance semantics with K>0:
 * a K-long list kept sorted ascending, initialised to (INT_MAX, -1); train rows
 * visited in ascending index; both comparisons strict, so among equal
 * distances the lowest train index ranks first and a later equal distance
 * never displaces an earlier one.
 */
static void knn_one_query(const uint8_t *q, const uint8_t *t, int nt, size_t t_step, int K,
                          int32_t *idx, int32_t *dist)
{
    for (int k = 0; k < K; k++) {
        dist[k] = INT_MAX;
        idx[k] = -1;
    }
    for (int j = 0; j < nt; j++) {
        int d = hamming256(q, t + (size_t)j * t_step);
        if (d < dist[K - 1]) {
            int k = K - 2;
            while (k >= 0 && dist[k] > d) {
                dist[k + 1] = dist[k];
                idx[k + 1] = idx[k];
                k--;
            }
            dist[k + 1] = d;
            idx[k + 1] = j;
        }
    }
}

/*
 * knnMatch(query, train, k): out_idx/out_dist are nq*k, row-major per query.
 * Slots past min(k, nt) hold idx=-1, dist=INT_MAX (OpenCV ends the row there).
 * Empty query or train is not an error: nothing is written (OpenCV clears the
 * match list).  Returns 0, or -1 on invalid arguments.
 */
int orc_knn_hamming256(const uint8_t *q, int nq, size_t q_step, const uint8_t *t, int nt,
                       size_t t_step, int k, int32_t *out_idx, int32_t *out_dist)
{
    if (nq < 0 || nt < 0 || k < 1 || q_step < 32 || t_step < 32)
        return -1;
    if (nq == 0)
        return 0;
    int K = k < nt ? k : nt;
    for (int i = 0; i < nq; i++) {
        int32_t *idx = out_idx + (size_t)i * k, *dist = out_dist + (size_t)i * k;
        for (int s = 0; s < k; s++) {
            idx[s] = -1;
            dist[s] = INT_MAX;
        }
        if (K > 0)
            knn_one_query(q + (size_t)i * q_step, t, nt, t_step, K, idx, dist);
    }
    return 0;
}

/* ---- multi-threaded variant: the timing baseline (query rows are independent) ---- */
typedef struct {
    const uint8_t *q, *t;
    int q0, q1, nt, k, K;
    size_t q_step, t_step;
    int32_t *out_idx, *out_dist;
} knn_job;

static void *knn_worker(void *p)
{
    knn_job *j = (knn_job *)p;
    for (int i = j->q0; i < j->q1; i++) {
        int32_t *idx = j->out_idx + (size_t)i * j->k, *dist = j->out_dist + (size_t)i * j->k;
        for (int s = 0; s < j->k; s++) {
            idx[s] = -1;
            dist[s] = INT_MAX;
        }
        if (j->K > 0)
            knn_one_query(j->q + (size_t)i * j->q_step, j->t, j->nt, j->t_step, j->K, idx, dist);
    }
    return NULL;
}

int orc_knn_hamming256_mt(const uint8_t *q, int nq, size_t q_step, const uint8_t *t, int nt,
                          size_t t_step, int k, int32_t *out_idx, int32_t *out_dist, int nthreads)
{
    if (nq < 0 || nt < 0 || k < 1 || q_step < 32 || t_step < 32 || nthreads < 1)
        return -1;
    if (nthreads > 256)
        nthreads = 256;
    if (nthreads > nq)
        nthreads = nq > 0 ? nq : 1;
    pthread_t th[256];
    knn_job jobs[256];
    int K = k < nt ? k : nt;
    for (int w = 0; w < nthreads; w++) {
        knn_job *j = &jobs[w];
        j->q = q; j->t = t; j->nt = nt; j->k = k; j->K = K;
        j->q_step = q_step; j->t_step = t_step;
        j->out_idx = out_idx; j->out_dist = out_dist;
        j->q0 = (int)((long long)nq * w / nthreads);
        j->q1 = (int)((long long)nq * (w + 1) / nthreads);
        if (pthread_create(&th[w], NULL, knn_worker, j) != 0)
            return -2;
    }
    for (int w = 0; w < nthreads; w++)
        pthread_join(th[w], NULL);
    return 0;
}

/*
 * match(query, train) without crossCheck == knnMatch(k=1, compact) flattened:
 * one DMatch per query in query order when nt>0, none otherwise.
 * Returns the number of matches written (out must hold nq).
 */
int orc_match(const uint8_t *q, int nq, size_t q_step, const uint8_t *t, int nt, size_t t_step,
              orc_dmatch *out)
{
    if (nq <= 0 || nt <= 0)
        return 0;
    int n = 0;
    for (int i = 0; i < nq; i++) {
        int32_t idx, dist;
        knn_one_query(q + (size_t)i * q_step, t, nt, t_step, 1, &idx, &dist);
        if (idx >= 0) {
            out[n].queryIdx = i;
            out[n].trainIdx = idx;
            out[n].imgIdx = 0;
            out[n].distance = (float)dist;
            n++;
        }
    }
    return n;
}

/*
 * match(query, train) with crossCheck=true (BFMatcher::knnMatchImpl with K == 1 hands the
 * flag to cv::batchDistance, whose crosscheck branch runs BOTH 1-NN searches):
 *   1. dist[q]=INT_MAX, nidx[q]=-1 for every query;
 *   2. tidx[i] / tdist[i]: every TRAIN row's nearest QUERY row (roles swapped; lowest
 *      query index on ties);
 *   3. sidx[q]: every QUERY row's nearest TRAIN row (lowest train index on ties);
 *   4. for train i ascending: q=tidx[i]; if (tdist[i] < dist[q] && sidx[q] == i) take it;
 *   5. emit (q, nidx[q], dist[q]) for queries with nidx[q]>=0, ascending q.
 * A pair survives iff each row is the other's nearest neighbour ("the BFMatcher will only
 * return consistent pairs").  OpenCV before the fix of its issue 11855 (2018; every 4.x release has it) lacked step 3 and its condition
 * (a query was paired with the closest train row among those whose nearest query it was,
 * which is not mutual); the library's regression test Features2d_DMatch.issue_11855
 * (modules/features2d/test/test_matchers_algorithmic.cpp) pins the mutual form, and its
 * vector is part of tests/golden/matcher_kat.json.  The reference names OpenCV 4.6.0.
 * Returns the number of matches written (out must hold nq).
 */
int orc_match_crosscheck(const uint8_t *q, int nq, size_t q_step, const uint8_t *t, int nt,
                         size_t t_step, orc_dmatch *out)
{
    if (nq <= 0 || nt <= 0)
        return 0;
    int32_t *dist = (int32_t *)malloc(sizeof(int32_t) * (size_t)nq);
    int32_t *nidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)nq);
    int32_t *sidx = (int32_t *)malloc(sizeof(int32_t) * (size_t)nq);
    if (!dist || !nidx || !sidx) {
        free(dist);
        free(nidx);
        free(sidx);
        return -1;
    }
    for (int i = 0; i < nq; i++) {
        int32_t sd;
        dist[i] = INT_MAX;
        nidx[i] = -1;
        knn_one_query(q + (size_t)i * q_step, t, nt, t_step, 1, &sidx[i], &sd);
    }
    for (int i = 0; i < nt; i++) {
        int32_t tq, td;
        knn_one_query(t + (size_t)i * t_step, q, nq, q_step, 1, &tq, &td);
        if (tq >= 0 && td < dist[tq] && sidx[tq] == i) {
            dist[tq] = td;
            nidx[tq] = i;
        }
    }
    int n = 0;
    for (int i = 0; i < nq; i++) {
        if (nidx[i] >= 0) {
            out[n].queryIdx = i;
            out[n].trainIdx = nidx[i];
            out[n].imgIdx = 0;
            out[n].distance = (float)dist[i];
            n++;
        }
    }
    free(dist);
    free(nidx);
    free(sidx);
    return n;
}

/*
 * Lowe ratio test on a knnMatch(k=2) result: keep m[0] iff the row has two
 * entries and m[0].distance < ratio * m[1].distance (float arithmetic, as a
 * caller of OpenCV writes it on DMatch::distance).  Not in the reference; the
 * benchmark configuration "kNN k=2 ratio test" defines it.
 */
int orc_ratio_filter(const int32_t *knn_idx, const int32_t *knn_dist, int nq, float ratio,
                     orc_dmatch *out)
{
    int n = 0;
    for (int i = 0; i < nq; i++) {
        if (knn_idx[2 * i] < 0 || knn_idx[2 * i + 1] < 0)
            continue;
        float d0 = (float)knn_dist[2 * i], d1 = (float)knn_dist[2 * i + 1];
        if (d0 < ratio * d1) {
            out[n].queryIdx = i;
            out[n].trainIdx = knn_idx[2 * i];
            out[n].imgIdx = 0;
            out[n].distance = d0;
            n++;
        }
    }
    return n;
}

/*
 * Min-distance filter of cereal_box.cpp:345-356 with the loop bounds made
 * explicit (the reference loops to the TRAIN row count over a vector that has
 * QUERY-count entries and never clears good_matches; here n is the number of
 * matches and out starts empty): min_dist starts at 10000, keep
 * distance <= max(2*min_dist, 30.0) compared in double.
 */
int orc_min_dist_filter(const orc_dmatch *m, int n, orc_dmatch *out)
{
    double min_dist = 10000;
    for (int i = 0; i < n; i++) {
        double d = m[i].distance;
        if (d < min_dist)
            min_dist = d;
    }
    double thr = 2 * min_dist > 30.0 ? 2 * min_dist : 30.0;
    int k = 0;
    for (int i = 0; i < n; i++)
        if ((double)m[i].distance <= thr)
            out[k++] = m[i];
    return k;
}

/* Stable ascending sort by distance: matcher.py:24 (Python's sorted is stable). */
void orc_sort_by_distance(orc_dmatch *m, int n)
{
    /* insertion sort keeps equal distances in their incoming order */
    for (int i = 1; i < n; i++) {
        orc_dmatch v = m[i];
        int j = i - 1;
        while (j >= 0 && m[j].distance > v.distance) {
            m[j + 1] = m[j];
            j--;
        }
        m[j + 1] = v;
    }
}
