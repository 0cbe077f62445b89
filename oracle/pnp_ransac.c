/*
 * oracle/pnp_ransac.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of cv::solvePnPRansac(..., SOLVEPNP_EPNP), the pose stage the
 * reference describes for its EPnP branch (docs/index.md:75,120 and the TODO at
 * src/cereal_box.cpp:292-295 of the reference; the call itself is not in the
 * mounted tree).  The arithmetic lives in OpenCV 4.6.0 (README.md:57-61 of the
 * reference), absent here: modules/calib3d/src/solvepnp.cpp, ptsetreg.cpp,
 * epnp.cpp, calibration.cpp (Rodrigues, projectPoints), modules/core/src/
 * lapack.cpp (JacobiSVD, SVBkSb) and rand.cpp (cv::RNG).  This file restates
 * those published algorithms in plain C.
 *
 * PARITY UNPINNED: the reference holds no fixture for this path and OpenCV
 * cannot be installed here, so nothing checks this restatement against OpenCV
 * itself.  tests/ pin it by synthetic pose recovery and by properties.
 *
 * Every fp64 operation is rounded separately (build with -ffp-contract=off).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object; the product library never does.
 */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ cv::RNG */
/* Multiply-with-carry generator of modules/core: state = lo32*4164903690 + hi32. */
typedef struct {
    uint64_t state;
} orc_rng;

static uint32_t rng_next(orc_rng *r)
{
    r->state = (uint64_t)(uint32_t)r->state * 4164903690u + (uint32_t)(r->state >> 32);
    return (uint32_t)r->state;
}

static int rng_uniform(orc_rng *r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (uint32_t)(b - a) + a); }

void orc_rng_init(orc_rng *r, uint64_t state) { r->state = state ? state : 0xffffffffu; }
uint32_t orc_rng_next(orc_rng *r) { return rng_next(r); }

/*
 * RANSACPointSetRegistrator::getSubset: for each of the model_points slots draw
 * uniform(0,count) and redraw while it repeats an earlier slot.  checkSubset of
 * the PnP callback always accepts, so the first attempt is returned.
 */
static void get_subset(orc_rng *rng, int count, int model_points, int *idx)
{
    for (int i = 0; i < model_points; i++) {
        int v, dup;
        do {
            v = rng_uniform(rng, 0, count);
            dup = 0;
            for (int j = 0; j < i; j++)
                if (idx[j] == v)
                    dup = 1;
        } while (dup);
        idx[i] = v;
    }
}

/* The first n_subsets draws of the RANSAC loop (seed (uint64)-1 is OpenCV's fixed default). */
void orc_ransac_subsets(uint64_t rng_state, int count, int model_points, int n_subsets,
                        int32_t *out)
{
    orc_rng rng;
    orc_rng_init(&rng, rng_state);
    for (int s = 0; s < n_subsets; s++)
        get_subset(&rng, count, model_points, out + (size_t)s * model_points);
}

/* RANSACUpdateNumIters of ptsetreg.cpp. */
int orc_ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0. ? p : 0.;
    p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.;
    ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN)
        return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* ------------------------------------------------------------- Jacobi SVD */
/*
 * lapack.cpp's own hypot, a function template in namespace cv that the unqualified call in
 * JacobiSVDImpl_ resolves to (it hides the C library's hypot there): the larger magnitude
 * times sqrt(1 + ratio^2).  Not the C library's correctly rounded hypot: the two differ in
 * the last bits, which the 5-point EPnP amplifies, so the form matters.
 */
static double cv_hypot(double a, double b)
{
    a = fabs(a);
    b = fabs(b);
    if (a > b) {
        b /= a;
        return a * sqrt(1 + b * b);
    }
    if (b > 0) {
        a /= b;
        return b * sqrt(1 + a * a);
    }
    return 0;
}

/*
 * JacobiSVDImpl_<double> of lapack.cpp: one-sided (Hestenes) Jacobi on the n
 * rows of At (each of length m), rotations accumulated in Vt (n x n).  On exit
 * row i of At is the i-th LEFT singular vector (unit length), W descending.
 * n1 rows of At are completed (n1 = n here).
 */
static void jacobi_svd(double *At, int astep, double *W_out, double *Vt, int vstep, int m, int n,
                       int n1)
{
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
    double W[16];
    int i, j, k, iter, max_iter = m > 30 ? m : 30;
    double c, s;
    for (i = 0; i < n; i++) {
        double sd = 0;
        for (k = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sd;
        if (Vt) {
            for (k = 0; k < n; k++)
                Vt[i * vstep + k] = 0;
            Vt[i * vstep + i] = 1;
        }
    }
    for (iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++)
                    p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b))
                    continue;
                p *= 2;
                double beta = a - b, gamma = cv_hypot(p, beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = 1;
                if (Vt) {
                    double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        double t0 = c * Vi[k] + s * Vj[k];
                        double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0;
                        Vj[k] = t1;
                    }
                }
            }
        if (!changed)
            break;
    }
    for (i = 0; i < n; i++) {
        double sd = 0;
        for (k = 0; k < m; k++) {
            double t = At[i * astep + k];
            sd += t * t;
        }
        W[i] = sqrt(sd);
    }
    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k])
                j = k;
        if (i != j) {
            double t = W[i];
            W[i] = W[j];
            W[j] = t;
            if (Vt) {
                for (k = 0; k < m; k++) {
                    t = At[i * astep + k];
                    At[i * astep + k] = At[j * astep + k];
                    At[j * astep + k] = t;
                }
                for (k = 0; k < n; k++) {
                    t = Vt[i * vstep + k];
                    Vt[i * vstep + k] = Vt[j * vstep + k];
                    Vt[j * vstep + k] = t;
                }
            }
        }
    }
    for (i = 0; i < n; i++)
        W_out[i] = W[i];
    if (!Vt)
        return;
    /* Left vectors of (numerically) zero singular values: a fixed pseudo-random
     * vector orthogonalised against the earlier rows (cv::RNG 0x12345678). */
    orc_rng rng;
    orc_rng_init(&rng, 0x12345678);
    for (i = 0; i < n1; i++) {
        double sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            const double val0 = 1. / m;
            for (k = 0; k < m; k++) {
                double val = (rng_next(&rng) & 256) != 0 ? val0 : -val0;
                At[i * astep + k] = val;
            }
            for (iter = 0; iter < 2; iter++) {
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++)
                        sd += At[i * astep + k] * At[j * astep + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) {
                        double t = At[i * astep + k] - sd * At[j * astep + k];
                        At[i * astep + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (k = 0; k < m; k++)
                        At[i * astep + k] *= asum;
                }
            }
            sd = 0;
            for (k = 0; k < m; k++) {
                double t = At[i * astep + k];
                sd += t * t;
            }
            sd = sqrt(sd);
        }
        s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++)
            At[i * astep + k] *= s;
    }
}

/*
 * cv::SVD::compute for a square n x n matrix A (row-major): A = U diag(w) Vt.
 * Returns Ut (rows = left singular vectors) and Vt.  n <= 12.
 */
static void svd_square(const double *A, int n, double *w, double *Ut, double *Vt)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            Ut[i * n + j] = A[j * n + i]; /* temp_a = transpose(src) */
    jacobi_svd(Ut, n, w, Vt, n, n, n, n);
}

/*
 * cv::solve(A, b, x, DECOMP_SVD) for an m x n system (m >= n, one right-hand
 * side): Jacobi SVD of A, then SVBkSb: x = sum_i v_i * (u_i . b) / w_i over the
 * singular values above DBL_EPSILON*2*sum(w).
 */
static void solve_svd(const double *A, int m, int n, const double *b, double *x)
{
    double At[12 * 12], Vt[12 * 12], w[12];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++)
            At[i * m + j] = A[j * n + i];
    jacobi_svd(At, m, w, Vt, n, m, n, n);
    double threshold = 0;
    for (int i = 0; i < n; i++) {
        x[i] = 0;
        threshold += w[i];
    }
    threshold *= DBL_EPSILON * 2;
    for (int i = 0; i < n; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold)
            continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < m; j++)
            s += At[i * m + j] * b[j];
        s *= wi;
        for (int j = 0; j < n; j++)
            x[j] = x[j] + s * Vt[i * n + j];
    }
}

/* cv::invert(A, DECOMP_SVD) for 3 x 3: pseudo-inverse by SVD + SVBkSb without rhs. */
static void invert3_svd(const double *A, double *inv)
{
    double Ut[9], Vt[9], w[3], buffer[3];
    svd_square(A, 3, w, Ut, Vt);
    double threshold = (w[0] + w[1] + w[2]) * (DBL_EPSILON * 2);
    for (int i = 0; i < 9; i++)
        inv[i] = 0;
    for (int i = 0; i < 3; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold)
            continue;
        wi = 1 / wi;
        for (int j = 0; j < 3; j++)
            buffer[j] = Ut[i * 3 + j] * wi; /* u(:, i) scaled */
        for (int r = 0; r < 3; r++) {
            double s = Vt[i * 3 + r];
            for (int j = 0; j < 3; j++)
                inv[r * 3 + j] = inv[r * 3 + j] + s * buffer[j];
        }
    }
}

/* ------------------------------------------------------------------- EPnP */
typedef struct {
    double uc, vc, fu, fv;
    int n;
    double *pws, *us, *alphas, *pcs; /* 3n, 2n, 4n, 3n */
    double cws[4][3], ccs[4][3];
} epnp_t;

static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static double dist2(const double *p1, const double *p2)
{
    return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) +
           (p1[2] - p2[2]) * (p1[2] - p2[2]);
}

static void choose_control_points(epnp_t *e)
{
    const int n = e->n;
    e->cws[0][0] = e->cws[0][1] = e->cws[0][2] = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++)
            e->cws[0][j] += e->pws[3 * i + j];
    for (int j = 0; j < 3; j++)
        e->cws[0][j] /= n;
    /* PW0^T PW0 (cvMulTransposed order 1), then its SVD with U transposed */
    double m33[9] = {0}, dc[3], uct[9], vt[9];
    for (int a = 0; a < 3; a++)
        for (int b = a; b < 3; b++) {
            double s = 0;
            for (int i = 0; i < n; i++)
                s += (e->pws[3 * i + a] - e->cws[0][a]) * (e->pws[3 * i + b] - e->cws[0][b]);
            m33[a * 3 + b] = s;
            m33[b * 3 + a] = s;
        }
    svd_square(m33, 3, dc, uct, vt);
    for (int i = 1; i < 4; i++) {
        double k = sqrt(dc[i - 1] / n);
        for (int j = 0; j < 3; j++)
            e->cws[i][j] = e->cws[0][j] + k * uct[3 * (i - 1) + j];
    }
}

static void compute_barycentric_coordinates(epnp_t *e)
{
    double cc[9], ci[9];
    for (int i = 0; i < 3; i++)
        for (int j = 1; j < 4; j++)
            cc[3 * i + j - 1] = e->cws[j][i] - e->cws[0][i];
    invert3_svd(cc, ci);
    for (int i = 0; i < e->n; i++) {
        const double *pi = e->pws + 3 * i;
        double *a = e->alphas + 4 * i;
        for (int j = 0; j < 3; j++)
            a[1 + j] = ci[3 * j] * (pi[0] - e->cws[0][0]) + ci[3 * j + 1] * (pi[1] - e->cws[0][1]) +
                       ci[3 * j + 2] * (pi[2] - e->cws[0][2]);
        a[0] = 1.0f - a[1] - a[2] - a[3];
    }
}

static void compute_L_6x10(const double *ut, double *l)
{
    const double *v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
        int a = 0, b = 1;
        for (int j = 0; j < 6; j++) {
            dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
            dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
            dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
            b++;
            if (b > 3) {
                a++;
                b = a + 1;
            }
        }
    }
    for (int i = 0; i < 6; i++) {
        double *row = l + 10 * i;
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

static void compute_rho(const epnp_t *e, double *rho)
{
    rho[0] = dist2(e->cws[0], e->cws[1]);
    rho[1] = dist2(e->cws[0], e->cws[2]);
    rho[2] = dist2(e->cws[0], e->cws[3]);
    rho[3] = dist2(e->cws[1], e->cws[2]);
    rho[4] = dist2(e->cws[1], e->cws[3]);
    rho[5] = dist2(e->cws[2], e->cws[3]);
}

/* betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44]; approx_1 uses [B11 B12 B13 B14] */
static void find_betas_approx_1(const double *l, const double *rho, double *betas)
{
    double l4[6 * 4], b4[4];
    for (int i = 0; i < 6; i++) {
        l4[4 * i] = l[10 * i];
        l4[4 * i + 1] = l[10 * i + 1];
        l4[4 * i + 2] = l[10 * i + 3];
        l4[4 * i + 3] = l[10 * i + 6];
    }
    solve_svd(l4, 6, 4, rho, b4);
    if (b4[0] < 0) {
        betas[0] = sqrt(-b4[0]);
        betas[1] = -b4[1] / betas[0];
        betas[2] = -b4[2] / betas[0];
        betas[3] = -b4[3] / betas[0];
    } else {
        betas[0] = sqrt(b4[0]);
        betas[1] = b4[1] / betas[0];
        betas[2] = b4[2] / betas[0];
        betas[3] = b4[3] / betas[0];
    }
}

/* approx_2 uses [B11 B12 B22] */
static void find_betas_approx_2(const double *l, const double *rho, double *betas)
{
    double l3[6 * 3], b3[3];
    for (int i = 0; i < 6; i++) {
        l3[3 * i] = l[10 * i];
        l3[3 * i + 1] = l[10 * i + 1];
        l3[3 * i + 2] = l[10 * i + 2];
    }
    solve_svd(l3, 6, 3, rho, b3);
    if (b3[0] < 0) {
        betas[0] = sqrt(-b3[0]);
        betas[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0;
    } else {
        betas[0] = sqrt(b3[0]);
        betas[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0;
    }
    if (b3[1] < 0)
        betas[0] = -betas[0];
    betas[2] = 0.0;
    betas[3] = 0.0;
}

/* approx_3 uses [B11 B12 B22 B13 B23] */
static void find_betas_approx_3(const double *l, const double *rho, double *betas)
{
    double l5[6 * 5], b5[5];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 5; j++)
            l5[5 * i + j] = l[10 * i + j];
    solve_svd(l5, 6, 5, rho, b5);
    if (b5[0] < 0) {
        betas[0] = sqrt(-b5[0]);
        betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0;
    } else {
        betas[0] = sqrt(b5[0]);
        betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0;
    }
    if (b5[1] < 0)
        betas[0] = -betas[0];
    betas[2] = b5[3] / betas[0];
    betas[3] = 0.0;
}

static void compute_A_and_b_gauss_newton(const double *l, const double *rho, const double *bt,
                                         double *A, double *b)
{
    for (int i = 0; i < 6; i++) {
        const double *rowL = l + i * 10;
        double *rowA = A + i * 4;
        rowA[0] = 2 * rowL[0] * bt[0] + rowL[1] * bt[1] + rowL[3] * bt[2] + rowL[6] * bt[3];
        rowA[1] = rowL[1] * bt[0] + 2 * rowL[2] * bt[1] + rowL[4] * bt[2] + rowL[7] * bt[3];
        rowA[2] = rowL[3] * bt[0] + rowL[4] * bt[1] + 2 * rowL[5] * bt[2] + rowL[8] * bt[3];
        rowA[3] = rowL[6] * bt[0] + rowL[7] * bt[1] + rowL[8] * bt[2] + 2 * rowL[9] * bt[3];
        b[i] = rho[i] - (rowL[0] * bt[0] * bt[0] + rowL[1] * bt[0] * bt[1] + rowL[2] * bt[1] * bt[1] +
                         rowL[3] * bt[0] * bt[2] + rowL[4] * bt[1] * bt[2] + rowL[5] * bt[2] * bt[2] +
                         rowL[6] * bt[0] * bt[3] + rowL[7] * bt[1] * bt[3] + rowL[8] * bt[2] * bt[3] +
                         rowL[9] * bt[3] * bt[3]);
    }
}

/*
 * epnp::qr_solve: Householder QR of the 6 x 4 system, in place.  The scan for
 * the column maximum starts at row k and advances AFTER reading, so it covers
 * rows k..nr-2 only -- kept as in the source.  Returns without touching X when a
 * column is all zero.
 */
static void qr_solve(double *pA, int nr, int nc, double *pb, double *pX)
{
    double A1[8], A2[8];
    double *ppAkk = pA;
    for (int k = 0; k < nc; k++) {
        double *ppAik1 = ppAkk, eta = fabs(*ppAik1);
        for (int i = k + 1; i < nr; i++) {
            double elt = fabs(*ppAik1);
            if (eta < elt)
                eta = elt;
            ppAik1 += nc;
        }
        if (eta == 0) {
            A1[k] = A2[k] = 0.0;
            return;
        } else {
            double *ppAik2 = ppAkk, sum2 = 0.0, inv_eta = 1. / eta;
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
                double *ppAik = ppAkk, sum = 0;
                for (int i = k; i < nr; i++) {
                    sum += *ppAik * ppAik[j - k];
                    ppAik += nc;
                }
                double tau = sum / A1[k];
                ppAik = ppAkk;
                for (int i = k; i < nr; i++) {
                    ppAik[j - k] -= tau * *ppAik;
                    ppAik += nc;
                }
            }
        }
        ppAkk += nc + 1;
    }
    /* b <- Qt b */
    double *ppAjj = pA;
    for (int j = 0; j < nc; j++) {
        double *ppAij = ppAjj, tau = 0;
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
    /* X = R^-1 b */
    pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; i--) {
        double *ppAij = pA + i * nc + (i + 1), sum = 0;
        for (int j = i + 1; j < nc; j++) {
            sum += *ppAij * pX[j];
            ppAij++;
        }
        pX[i] = (pb[i] - sum) / A2[i];
    }
}

static void gauss_newton(const double *l, const double *rho, double *betas)
{
    double a[6 * 4], b[6], x[4] = {0, 0, 0, 0};
    for (int k = 0; k < 5; k++) {
        compute_A_and_b_gauss_newton(l, rho, betas, a, b);
        qr_solve(a, 6, 4, b, x);
        for (int i = 0; i < 4; i++)
            betas[i] += x[i];
    }
}

static void compute_ccs(epnp_t *e, const double *betas, const double *ut)
{
    for (int i = 0; i < 4; i++)
        e->ccs[i][0] = e->ccs[i][1] = e->ccs[i][2] = 0.0f;
    for (int i = 0; i < 4; i++) {
        const double *v = ut + 12 * (11 - i);
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++)
                e->ccs[j][k] += betas[i] * v[3 * j + k];
    }
}

static void compute_pcs(epnp_t *e)
{
    for (int i = 0; i < e->n; i++) {
        const double *a = e->alphas + 4 * i;
        double *pc = e->pcs + 3 * i;
        for (int j = 0; j < 3; j++)
            pc[j] = a[0] * e->ccs[0][j] + a[1] * e->ccs[1][j] + a[2] * e->ccs[2][j] +
                    a[3] * e->ccs[3][j];
    }
}

static void solve_for_sign(epnp_t *e)
{
    if (e->pcs[2] < 0.0) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 3; j++)
                e->ccs[i][j] = -e->ccs[i][j];
        for (int i = 0; i < e->n; i++) {
            e->pcs[3 * i] = -e->pcs[3 * i];
            e->pcs[3 * i + 1] = -e->pcs[3 * i + 1];
            e->pcs[3 * i + 2] = -e->pcs[3 * i + 2];
        }
    }
}

static void estimate_R_and_t(const epnp_t *e, double R[3][3], double t[3])
{
    const int n = e->n;
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        const double *pc = e->pcs + 3 * i, *pw = e->pws + 3 * i;
        for (int j = 0; j < 3; j++) {
            pc0[j] += pc[j];
            pw0[j] += pw[j];
        }
    }
    for (int j = 0; j < 3; j++) {
        pc0[j] /= n;
        pw0[j] /= n;
    }
    double abt[9] = {0}, abt_d[3], abt_ut[9], abt_vt[9];
    for (int i = 0; i < n; i++) {
        const double *pc = e->pcs + 3 * i, *pw = e->pws + 3 * i;
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    svd_square(abt, 3, abt_d, abt_ut, abt_vt);
    /* R = U * V^T: R[i][j] = sum_k U[i][k] * V[j][k] with U[i][k] = Ut[k][i], V[j][k] = Vt[k][j] */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i][j] = abt_ut[0 * 3 + i] * abt_vt[0 * 3 + j] + abt_ut[1 * 3 + i] * abt_vt[1 * 3 + j] +
                      abt_ut[2 * 3 + i] * abt_vt[2 * 3 + j];
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
}

static double reprojection_error(const epnp_t *e, double R[3][3], const double t[3])
{
    double sum2 = 0.0;
    for (int i = 0; i < e->n; i++) {
        const double *pw = e->pws + 3 * i;
        double Xc = dot3(R[0], pw) + t[0];
        double Yc = dot3(R[1], pw) + t[1];
        double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
        double ue = e->uc + e->fu * Xc * inv_Zc;
        double ve = e->vc + e->fv * Yc * inv_Zc;
        double u = e->us[2 * i], v = e->us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / e->n;
}

static double compute_R_and_t(epnp_t *e, const double *ut, const double *betas, double R[3][3],
                              double t[3])
{
    compute_ccs(e, betas, ut);
    compute_pcs(e);
    solve_for_sign(e);
    estimate_R_and_t(e, R, t);
    return reprojection_error(e, R, t);
}

/*
 * solvePnP(SOLVEPNP_EPNP) up to (R, t): image points are first normalised by
 * undistortPoints (zero distortion: x = (u - cx)/fx computed in double, stored
 * in the type of the points it was given), then epnp::init_points maps them
 * back with us = x*fu + uc.  points_are_f32 != 0: the points reach solvePnP as
 * CV_32F (the caller's arrays, RANSAC's five-point samples), so x is rounded to
 * float on the way; 0: they reach it as CV_64F (solvePnPRansac converts the
 * consensus set to Point3d / Point2d before its final solve), so x stays double.
 * obj: n x 3 float, img: n x 2 float (the values are floats either way), K
 * row-major 3 x 3.
 */
static int epnp_pose(const float *obj, const float *img, int n, const double *K, int points_are_f32,
                     double R[3][3], double t[3])
{
    epnp_t e;
    e.fu = K[0]; e.fv = K[4]; e.uc = K[2]; e.vc = K[5];
    e.n = n;
    double *buf = (double *)malloc(sizeof(double) * (size_t)n * 12 + sizeof(double) * (size_t)n * 24);
    if (!buf)
        return -1;
    e.pws = buf;
    e.us = buf + 3 * (size_t)n;
    e.alphas = e.us + 2 * (size_t)n;
    e.pcs = e.alphas + 4 * (size_t)n;
    double *M = e.pcs + 3 * (size_t)n; /* 2n x 12 */
    const double ifx = 1. / K[0], ify = 1. / K[4];
    for (int i = 0; i < n; i++) {
        e.pws[3 * i] = obj[3 * i];
        e.pws[3 * i + 1] = obj[3 * i + 1];
        e.pws[3 * i + 2] = obj[3 * i + 2];
        double xn = ((double)img[2 * i] - K[2]) * ifx;
        double yn = ((double)img[2 * i + 1] - K[5]) * ify;
        if (points_are_f32) {
            xn = (float)xn;
            yn = (float)yn;
        }
        e.us[2 * i] = xn * e.fu + e.uc;
        e.us[2 * i + 1] = yn * e.fv + e.vc;
    }
    choose_control_points(&e);
    compute_barycentric_coordinates(&e);
    for (int i = 0; i < n; i++) { /* fill_M */
        const double *as = e.alphas + 4 * i;
        double *M1 = M + (size_t)(2 * i) * 12, *M2 = M1 + 12;
        const double u = e.us[2 * i], v = e.us[2 * i + 1];
        for (int k = 0; k < 4; k++) {
            M1[3 * k] = as[k] * e.fu;
            M1[3 * k + 1] = 0.0;
            M1[3 * k + 2] = as[k] * (e.uc - u);
            M2[3 * k] = 0.0;
            M2[3 * k + 1] = as[k] * e.fv;
            M2[3 * k + 2] = as[k] * (e.vc - v);
        }
    }
    double mtm[144], d[12], ut[144], vt[144];
    for (int a = 0; a < 12; a++) /* cvMulTransposed(M, MtM, 1) */
        for (int b = a; b < 12; b++) {
            double s = 0;
            for (int k = 0; k < 2 * n; k++)
                s += M[(size_t)k * 12 + a] * M[(size_t)k * 12 + b];
            mtm[a * 12 + b] = s;
            mtm[b * 12 + a] = s;
        }
    svd_square(mtm, 12, d, ut, vt);
    double l_6x10[60], rho[6];
    compute_L_6x10(ut, l_6x10);
    compute_rho(&e, rho);
    double Betas[4][4], rep_errors[4], Rs[4][3][3], ts[4][3];
    find_betas_approx_1(l_6x10, rho, Betas[1]);
    gauss_newton(l_6x10, rho, Betas[1]);
    rep_errors[1] = compute_R_and_t(&e, ut, Betas[1], Rs[1], ts[1]);
    find_betas_approx_2(l_6x10, rho, Betas[2]);
    gauss_newton(l_6x10, rho, Betas[2]);
    rep_errors[2] = compute_R_and_t(&e, ut, Betas[2], Rs[2], ts[2]);
    find_betas_approx_3(l_6x10, rho, Betas[3]);
    gauss_newton(l_6x10, rho, Betas[3]);
    rep_errors[3] = compute_R_and_t(&e, ut, Betas[3], Rs[3], ts[3]);
    int N = 1;
    if (rep_errors[2] < rep_errors[1])
        N = 2;
    if (rep_errors[3] < rep_errors[N])
        N = 3;
    memcpy(R, Rs[N], sizeof(double) * 9);
    memcpy(t, ts[N], sizeof(double) * 3);
    free(buf);
    return 0;
}

/* -------------------------------------------------------------- Rodrigues */
/* cv::Rodrigues, 3 x 3 matrix -> rotation vector (R is re-orthonormalised by SVD first). */
static void rodrigues_mat_to_vec(const double Rin[3][3], double r[3])
{
    double w[3], Ut[9], Vt[9], R[9];
    for (int i = 0; i < 9; i++)
        if (!(fabs(((const double *)Rin)[i]) < 100.)) { /* checkRange(-100, 100), NaN fails */
            r[0] = r[1] = r[2] = 0;
            return;
        }
    svd_square((const double *)Rin, 3, w, Ut, Vt);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = Ut[0 * 3 + i] * Vt[0 * 3 + j] + Ut[1 * 3 + i] * Vt[1 * 3 + j] +
                           Ut[2 * 3 + i] * Vt[2 * 3 + j];
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0)
            rx = ry = rz = 0;
        else {
            double t;
            t = (R[0] + 1) * 0.5;
            rx = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5;
            ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
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

/* cv::Rodrigues, rotation vector -> 3 x 3 matrix. */
static void rodrigues_vec_to_mat(const double r[3], double R[9])
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++)
            R[i] = (i % 4 == 0) ? 1. : 0.;
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    double x = r[0] * itheta, y = r[1] * itheta, z = r[2] * itheta;
    double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    double r_x[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int i = 0; i < 9; i++)
        R[i] = c * ((i % 4 == 0) ? 1. : 0.) + c1 * rrt[i] + s * r_x[i];
}

void orc_rodrigues_mat_to_vec(const double *R9, double *r3)
{
    double R[3][3];
    memcpy(R, R9, sizeof(R));
    rodrigues_mat_to_vec(R, r3);
}

void orc_rodrigues_vec_to_mat(const double *r3, double *R9) { rodrigues_vec_to_mat(r3, R9); }

static int solvepnp_epnp(const float *obj, const float *img, int n, const double *K,
                         int points_are_f32, double *rvec, double *tvec)
{
    double R[3][3], t[3];
    if (n < 4)
        return -1;
    if (epnp_pose(obj, img, n, K, points_are_f32, R, t) != 0)
        return -1;
    rodrigues_mat_to_vec(R, rvec);
    tvec[0] = t[0];
    tvec[1] = t[1];
    tvec[2] = t[2];
    return 0;
}

/* solvePnP(obj, img, K, no distortion, SOLVEPNP_EPNP) on CV_32F points -> rvec, tvec. */
int orc_solvepnp_epnp(const float *obj, const float *img, int n, const double *K, double *rvec,
                      double *tvec)
{
    return solvepnp_epnp(obj, img, n, K, 1, rvec, tvec);
}

/* The same on the values of float points handed over as CV_64F, which is how solvePnPRansac
 * calls solvePnP on its consensus set. */
int orc_solvepnp_epnp_f64points(const float *obj, const float *img, int n, const double *K,
                                double *rvec, double *tvec)
{
    return solvepnp_epnp(obj, img, n, K, 0, rvec, tvec);
}

/*
 * PnPRansacCallback::computeError + findInliers: projectPoints in double
 * (zero distortion), projected pixel stored as float, squared L2 distance in
 * float, inlier iff err <= (float)(threshold*threshold).  Returns the count;
 * mask (may be NULL) gets one byte per point.
 */
int orc_pnp_count_inliers(const float *obj, const float *img, int n, const double *K,
                          const double *rvec, const double *tvec, double threshold, uint8_t *mask,
                          float *err_out)
{
    double R[9];
    rodrigues_vec_to_mat(rvec, R);
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const float t = (float)(threshold * threshold);
    int good = 0;
    for (int i = 0; i < n; i++) {
        const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + tvec[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + tvec[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        const float px = (float)(x * fx + cx), py = (float)(y * fy + cy);
        const float dx = img[2 * i] - px, dy = img[2 * i + 1] - py;
        const float err = dx * dx + dy * dy;
        const int f = err <= t;
        if (mask)
            mask[i] = (uint8_t)f;
        if (err_out)
            err_out[i] = err;
        good += f;
    }
    return good;
}

/*
 * One RANSAC hypothesis: EPnP on the 5 sampled correspondences -> (rvec, tvec),
 * then its inlier count over all n points.  Returns the count (>= 0).
 */
int orc_pnp_hypothesis(const float *obj, const float *img, int n, const double *K,
                       const int32_t *subset, int model_points, double threshold, double *rvec,
                       double *tvec)
{
    float so[3 * 8], si[2 * 8];
    for (int i = 0; i < model_points; i++) {
        memcpy(so + 3 * i, obj + 3 * (size_t)subset[i], 3 * sizeof(float));
        memcpy(si + 2 * i, img + 2 * (size_t)subset[i], 2 * sizeof(float));
    }
    if (orc_solvepnp_epnp(so, si, model_points, K, rvec, tvec) != 0)
        return 0;
    return orc_pnp_count_inliers(obj, img, n, K, rvec, tvec, threshold, NULL, NULL);
}

/* All n_hyp hypotheses of given subsets, optionally on several threads (timing baseline). */
typedef struct {
    const float *obj, *img;
    int n, mp, h0, h1;
    const double *K;
    const int32_t *subsets;
    double threshold;
    double *models;
    int32_t *counts;
} hyp_job;

static void *hyp_worker(void *p)
{
    hyp_job *j = (hyp_job *)p;
    for (int h = j->h0; h < j->h1; h++)
        j->counts[h] = orc_pnp_hypothesis(j->obj, j->img, j->n, j->K, j->subsets + (size_t)h * j->mp,
                                          j->mp, j->threshold, j->models + 6 * (size_t)h,
                                          j->models + 6 * (size_t)h + 3);
    return NULL;
}

int orc_pnp_hypotheses(const float *obj, const float *img, int n, const double *K,
                       const int32_t *subsets, int n_hyp, double threshold, double *models,
                       int32_t *counts, int nthreads)
{
    if (nthreads < 1)
        nthreads = 1;
    if (nthreads > 64)
        nthreads = 64;
    pthread_t th[64];
    hyp_job jobs[64];
    for (int w = 0; w < nthreads; w++) {
        hyp_job *j = &jobs[w];
        j->obj = obj; j->img = img; j->n = n; j->mp = 5; j->K = K; j->subsets = subsets;
        j->threshold = threshold; j->models = models; j->counts = counts;
        j->h0 = (int)((long long)n_hyp * w / nthreads);
        j->h1 = (int)((long long)n_hyp * (w + 1) / nthreads);
        if (nthreads == 1)
            hyp_worker(j);
        else if (pthread_create(&th[w], NULL, hyp_worker, j) != 0)
            return -1;
    }
    if (nthreads > 1)
        for (int w = 0; w < nthreads; w++)
            pthread_join(th[w], NULL);
    return 0;
}

/*
 * cv::solvePnPRansac(obj, img, K, no distortion, rvec, tvec, useExtrinsicGuess
 * = false, iterationsCount, reprojectionError, confidence, inliers,
 * SOLVEPNP_EPNP).  RANSAC (RANSACPointSetRegistrator::run): fixed RNG state
 * rng_state ((uint64)-1 in OpenCV), 5-point EPnP per iteration, a model
 * replaces the best iff its count is strictly larger than max(best, 4), and the
 * iteration budget shrinks by RANSACUpdateNumIters on every replacement.  Then
 * one EPnP solve over all inliers of the best model.
 * Returns 1 (pose found), 0 (no pose).  inliers must hold n ints.
 * iters_run (optional) receives the number of loop iterations consumed.
 */
int orc_solvepnp_ransac(const float *obj, const float *img, int n, const double *K,
                        int iterations_count, float reprojection_error, double confidence,
                        uint64_t rng_state, double *rvec, double *tvec, int32_t *inliers,
                        int *n_inliers, int *iters_run)
{
    const int model_points = 5;
    if (n_inliers)
        *n_inliers = 0;
    if (iters_run)
        *iters_run = 0;
    if (n < 4)
        return -1;
    if (n == 4)
        return -1; /* OpenCV switches to P3P for exactly four points: outside this path */
    if (n == model_points) {
        if (orc_solvepnp_epnp(obj, img, n, K, rvec, tvec) != 0)
            return 0;
        for (int i = 0; i < n; i++)
            inliers[i] = i;
        if (n_inliers)
            *n_inliers = n;
        return 1;
    }
    const double threshold = reprojection_error;
    orc_rng rng;
    orc_rng_init(&rng, rng_state);
    uint8_t *mask = (uint8_t *)malloc((size_t)n), *best_mask = (uint8_t *)malloc((size_t)n);
    if (!mask || !best_mask) {
        free(mask);
        free(best_mask);
        return -1;
    }
    double best_r[3] = {0, 0, 0}, best_t[3] = {0, 0, 0};
    int max_good = 0, niters = iterations_count > 1 ? iterations_count : 1, iter;
    for (iter = 0; iter < niters; iter++) {
        int32_t idx[5];
        float so[15], si[10];
        double r[3], t[3];
        get_subset(&rng, n, model_points, idx);
        for (int i = 0; i < model_points; i++) {
            memcpy(so + 3 * i, obj + 3 * (size_t)idx[i], 3 * sizeof(float));
            memcpy(si + 2 * i, img + 2 * (size_t)idx[i], 2 * sizeof(float));
        }
        if (orc_solvepnp_epnp(so, si, model_points, K, r, t) != 0)
            continue;
        int good = orc_pnp_count_inliers(obj, img, n, K, r, t, threshold, mask, NULL);
        if (good > (max_good > model_points - 1 ? max_good : model_points - 1)) {
            uint8_t *tmp = mask;
            mask = best_mask;
            best_mask = tmp;
            memcpy(best_r, r, sizeof(r));
            memcpy(best_t, t, sizeof(t));
            max_good = good;
            niters = orc_ransac_update_num_iters(confidence, (double)(n - good) / n, model_points,
                                                 niters);
        }
    }
    if (iters_run)
        *iters_run = iter;
    int result = 0;
    if (max_good > 0) {
        /* refit on the inliers, converted to CV_64F (exact for float inputs; what changes is
         * that undistortPoints then keeps the normalised image points in double) */
        float *io = (float *)malloc(sizeof(float) * 5 * (size_t)max_good);
        if (io) {
            float *ii = io + 3 * (size_t)max_good;
            int k = 0;
            for (int i = 0; i < n; i++)
                if (best_mask[i]) {
                    memcpy(io + 3 * k, obj + 3 * (size_t)i, 3 * sizeof(float));
                    memcpy(ii + 2 * k, img + 2 * (size_t)i, 2 * sizeof(float));
                    inliers[k] = i;
                    k++;
                }
            if (orc_solvepnp_epnp_f64points(io, ii, k, K, rvec, tvec) == 0) {
                result = 1;
                if (n_inliers)
                    *n_inliers = k;
            } else {
                memcpy(rvec, best_r, sizeof(best_r));
                memcpy(tvec, best_t, sizeof(best_t));
            }
            free(io);
        }
    }
    free(mask);
    free(best_mask);
    return result;
}
