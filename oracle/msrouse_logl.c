/*
 * TEST INFRASTRUCTURE -- CPU oracle for the Rouse Kalman-filter log-likelihood.
 *
 * Plain-C restatement of the reference algorithm.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call this file; the product library
 * (bild_amd/csrc) never does.
 *
 * Parity status: PINNED.  Checked against the reference's own Cython kernel (compiled
 * unmodified, oracle/build_ref.py) and NumPy kernel on the committed golden vectors
 * (tests/golden/, tests/test_oracle.py).
 *
 * Follows, statement by statement:
 *   flavor CYTHON (1): /root/reference/bild/src/MSRouse_logL.pyx
 *       Kalman_update          :19-90
 *       first update           :186-190
 *       mean predict           :206-216   (dsymv "u" on C-ordered B == row-major lower triangle)
 *       covariance predict     :220-241   (row n0 at a time: BCn0 = C.B[n0,:], C_post[n0,:] = Sig[n0,:] + B.BCn0)
 *       masked update          :244-248
 *       final sum              :251-256   (sequential over frames, then dims)
 *   flavor NUMPY (0):  /root/reference/bild/src/MSRouse_logL_py.py
 *       Kalman_update          :5-52      (full matrices, -0.5*(v^2/S + log S + log 2pi))
 *       loop                   :96-121    (empty valid_times -> 0.0)
 *
 * BLAS routines are restated as plain loops, so agreement with the reference is to
 * rounding (observed < 1e-9 at T = 1000), not bit-exact.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BILD_ORACLE_FLAVOR_NUMPY  0
#define BILD_ORACLE_FLAVOR_CYTHON 1

static const double LOG_2PI = 1.8378770664093453; /* log(2*pi) */

/* y = A x for symmetric A of which only the row-major lower triangle is read
 * (what dsymv("u") sees on C-ordered data, pyx:55,210,227,235) */
static void symv_lower(int N, const double *A, const double *x, int incx, double *y, int incy, double beta)
{
    for (int i = 0; i < N; ++i) {
        double acc = 0.0;
        for (int c = 0; c <= i; ++c) acc += A[(size_t)i * N + c] * x[(size_t)c * incx];
        for (int r = i + 1; r < N; ++r) acc += A[(size_t)r * N + i] * x[(size_t)r * incx];
        y[(size_t)i * incy] = (beta == 0.0 ? 0.0 : beta * y[(size_t)i * incy]) + acc;
    }
}

/* y = A x, full matrix (NumPy flavor) */
static void gemv_full(int N, const double *A, const double *x, int incx, double *y, int incy, double beta)
{
    for (int i = 0; i < N; ++i) {
        double acc = 0.0;
        for (int c = 0; c < N; ++c) acc += A[(size_t)i * N + c] * x[(size_t)c * incx];
        y[(size_t)i * incy] = (beta == 0.0 ? 0.0 : beta * y[(size_t)i * incy]) + acc;
    }
}

typedef struct {
    int N, d, dstar, flavor;
    double *M;      /* N x d   */
    double *C;      /* d* x N x N */
    double *Cw, *K; /* d* x N  */
    double *Sinv;   /* d*      */
    double *M_post, *C_post, *BCn0, *tmp;
} work_t;

static int work_alloc(work_t *wk, int N, int d, int dstar, int flavor)
{
    wk->N = N; wk->d = d; wk->dstar = dstar; wk->flavor = flavor;
    size_t n = (size_t)N;
    wk->M = (double *)malloc(sizeof(double) * n * d);
    wk->C = (double *)malloc(sizeof(double) * dstar * n * n);
    wk->Cw = (double *)malloc(sizeof(double) * dstar * n);
    wk->K = (double *)malloc(sizeof(double) * dstar * n);
    wk->Sinv = (double *)malloc(sizeof(double) * dstar);
    wk->M_post = (double *)malloc(sizeof(double) * n * d);
    wk->C_post = (double *)malloc(sizeof(double) * n * n);
    wk->BCn0 = (double *)malloc(sizeof(double) * n);
    wk->tmp = (double *)malloc(sizeof(double) * n * n);
    return (wk->M && wk->C && wk->Cw && wk->K && wk->Sinv && wk->M_post && wk->C_post && wk->BCn0 && wk->tmp) ? 0 : -1;
}

static void work_free(work_t *wk)
{
    free(wk->M); free(wk->C); free(wk->Cw); free(wk->K); free(wk->Sinv);
    free(wk->M_post); free(wk->C_post); free(wk->BCn0); free(wk->tmp);
}

/* pyx:19-90 / _py.py:5-52 ; writes d per-dimension terms to logL */
static void kalman_update(work_t *wk, const double *w, const double *x, const double *s2,
                          const int32_t *Cind, double *logL)
{
    const int N = wk->N, d = wk->d;
    for (int e = 0; e < wk->dstar; ++e) {
        double *C = wk->C + (size_t)e * N * N;
        double *Cw = wk->Cw + (size_t)e * N;
        double *K = wk->K + (size_t)e * N;
        if (wk->flavor == BILD_ORACLE_FLAVOR_CYTHON) symv_lower(N, C, w, 1, Cw, 1, 0.0);
        else gemv_full(N, C, w, 1, Cw, 1, 0.0);
        double S = 0.0;
        for (int i = 0; i < N; ++i) S += Cw[i] * w[i];
        S += s2[e];
        if (wk->flavor == BILD_ORACLE_FLAVOR_CYTHON) {
            wk->Sinv[e] = 1.0 / S;
            for (int i = 0; i < N; ++i) K[i] = wk->Sinv[e] * Cw[i];
        } else {
            wk->Sinv[e] = S; /* NumPy flavor keeps S itself */
            for (int i = 0; i < N; ++i) K[i] = Cw[i] / S;
        }
        /* dger: C -= K (x) Cw, all N*N entries */
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j)
                C[(size_t)i * N + j] -= K[i] * Cw[j];
    }
    for (int k = 0; k < d; ++k) {
        double m = 0.0;
        for (int i = 0; i < N; ++i) m += w[i] * wk->M[(size_t)i * d + k];
        const double xmm = x[k] - m;
        const double *K = wk->K + (size_t)Cind[k] * N;
        for (int i = 0; i < N; ++i) wk->M[(size_t)i * d + k] += xmm * K[i];
        if (wk->flavor == BILD_ORACLE_FLAVOR_CYTHON) {
            const double Sinv = wk->Sinv[Cind[k]];
            logL[k] = -0.5 * (xmm * xmm * Sinv - log(Sinv) + LOG_2PI);
        } else {
            const double S = wk->Sinv[Cind[k]];
            logL[k] = -0.5 * (xmm * xmm / S + log(S) + LOG_2PI);
        }
    }
}

/* pyx:206-241 / _py.py:109-110 */
static void predict(work_t *wk, const double *B, const double *G, const double *Sig)
{
    const int N = wk->N, d = wk->d;
    if (wk->flavor == BILD_ORACLE_FLAVOR_CYTHON) {
        for (int k = 0; k < d; ++k) {
            for (int i = 0; i < N; ++i) wk->M_post[(size_t)i * d + k] = G[(size_t)i * d + k];
            symv_lower(N, B, wk->M + k, d, wk->M_post + k, d, 1.0);
        }
        memcpy(wk->M, wk->M_post, sizeof(double) * N * d);
        for (int e = 0; e < wk->dstar; ++e) {
            double *C = wk->C + (size_t)e * N * N;
            for (int n0 = 0; n0 < N; ++n0) {
                memcpy(wk->C_post + (size_t)n0 * N, Sig + (size_t)n0 * N, sizeof(double) * N);
                symv_lower(N, C, B + (size_t)n0 * N, 1, wk->BCn0, 1, 0.0);
                symv_lower(N, B, wk->BCn0, 1, wk->C_post + (size_t)n0 * N, 1, 1.0);
            }
            memcpy(C, wk->C_post, sizeof(double) * N * N);
        }
    } else {
        /* M = B @ M + G */
        for (int k = 0; k < d; ++k) {
            gemv_full(N, B, wk->M + k, d, wk->M_post + k, d, 0.0);
            for (int i = 0; i < N; ++i) wk->M_post[(size_t)i * d + k] += G[(size_t)i * d + k];
        }
        memcpy(wk->M, wk->M_post, sizeof(double) * N * d);
        /* C = (B @ C) @ B + Sig */
        for (int e = 0; e < wk->dstar; ++e) {
            double *C = wk->C + (size_t)e * N * N;
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) {
                    double acc = 0.0;
                    for (int k = 0; k < N; ++k) acc += B[(size_t)i * N + k] * C[(size_t)k * N + j];
                    wk->tmp[(size_t)i * N + j] = acc;
                }
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) {
                    double acc = 0.0;
                    for (int k = 0; k < N; ++k) acc += wk->tmp[(size_t)i * N + k] * B[(size_t)k * N + j];
                    wk->C_post[(size_t)i * N + j] = acc + Sig[(size_t)i * N + j];
                }
            memcpy(C, wk->C_post, sizeof(double) * N * N);
        }
    }
}

/*
 * One (profile, trajectory) evaluation.
 *
 *   B, Sig, C0 : S x N x N      G, M0 : S x N x d     w : N         (row-major f64)
 *   s2 : d* unique squared localization errors (ascending), Cind : d  (dim -> d* index)
 *   x  : T x d, NaN = missing (a frame is missing iff any coordinate is NaN, pyx:178)
 *   states : T, states[0] selects the steady state, states[t] the propagator into frame t
 */
double bild_oracle_logl(int N, int d, int S, const double *B, const double *G, const double *Sig,
                        const double *M0, const double *C0, const double *w,
                        int dstar, const double *s2, const int32_t *Cind,
                        int T, const double *x, const int32_t *states, int flavor)
{
    (void)S;
    work_t wk;
    if (T <= 0 || work_alloc(&wk, N, d, dstar, flavor) != 0) return NAN;
    const size_t nn = (size_t)N * N, nd = (size_t)N * d;

    const int s0 = states[0];
    memcpy(wk.M, M0 + s0 * nd, sizeof(double) * nd);
    for (int e = 0; e < dstar; ++e) memcpy(wk.C + e * nn, C0 + s0 * nn, sizeof(double) * nn);

    double total = 0.0;
    double *terms = (double *)malloc(sizeof(double) * (size_t)(d > 0 ? d : 1));
    for (int t = 0; t < T; ++t) {
        if (t > 0) {
            const int s = states[t];
            predict(&wk, B + s * nn, G + s * nd, Sig + s * nn);
        }
        int valid = 1;
        for (int k = 0; k < d; ++k) if (isnan(x[(size_t)t * d + k])) valid = 0;
        if (valid) {
            kalman_update(&wk, w, x + (size_t)t * d, s2, Cind, terms);
            for (int k = 0; k < d; ++k) total += terms[k];
        }
    }
    free(terms);
    work_free(&wk);
    return total;
}

/* batch over expanded profiles: states is n x T (row stride ld) */
void bild_oracle_logl_batch(int N, int d, int S, const double *B, const double *G, const double *Sig,
                            const double *M0, const double *C0, const double *w,
                            int dstar, const double *s2, const int32_t *Cind,
                            int T, const double *x, int64_t n, const int32_t *states, int64_t ld,
                            int flavor, double *out)
{
    for (int64_t r = 0; r < n; ++r)
        out[r] = bild_oracle_logl(N, d, S, B, G, Sig, M0, C0, w, dstar, s2, Cind, T, x,
                                  states + r * ld, flavor);
}
