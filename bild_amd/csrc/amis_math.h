// Per-sample arithmetic of the AMIS bookkeeping (reference bild/amis.py:805-906), shared by the host passes
// (amis_host.cpp) and the device passes (amis_device.hip): the same expressions in the same order on both sides, so
// that the host implementation -- which the CPU tests compare with the NumPy formulation step by step -- is also the
// specification of the device one.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define BILD_HD __host__ __device__
#else
#define BILD_HD
#endif

namespace bild {

// what the passes over the pooled samples read: plain pointers, host or device
struct AmisView {
    int k1, k, n;
    // proposals: a (Q x k1), Dirichlet normalisation (Q), log probability of the first state (Q x n), of each
    // (slot, previous state, state) transition (Q x k*n*n)
    const double *a, *dir_norm, *head, *pair;
    // pooled samples
    const double *ss, *log_ss;  // P x k1
    const uint8_t *has_zero;    // P
    const int32_t *first;       // P
    const int32_t *pcode;       // P x k: (slot * n + previous state) * n + state
    const int32_t *theta;       // P x k1
    const double *logL;         // P
};

BILD_HD inline double amis_neg_inf() { return -HUGE_VAL; }

BILD_HD inline double amis_logaddexp(double x, double y)
{
    if (x == y) return x + 0.6931471805599453; // also covers equal infinities
    const double d = x - y;
    // the smaller term is below 2e-22 of the larger: the sum rounds to the larger (no exp / log1p needed);
    // this is the common case for old samples under a proposal that has since concentrated elsewhere
    if (d > 50) return x;
    if (d < -50) return y;
    if (d > 0) return x + log1p(exp(-d));
    if (d <= 0) return y + log1p(exp(d));
    return x + y; // NaN
}

// log density of proposal q at pooled sample p
BILD_HD inline double amis_log_q(const AmisView &v, int64_t q, int64_t p)
{
    const int k1 = v.k1, k = v.k, n = v.n;
    const double *A = v.a + (size_t)q * k1;
    double out = v.dir_norm[q];
    if (!v.has_zero[p]) {
        const double *ls = v.log_ss + (size_t)p * k1;
        for (int j = 0; j < k1; ++j) out += (A[j] - 1.0) * ls[j];
    } else { // x log(0): 0 for x = 0; a pole of the density (s = 0, a < 1) is +inf (tests/test_amis.py:51-54 of the reference)
        const double *s = v.ss + (size_t)p * k1;
        bool pole = false;
        for (int j = 0; j < k1; ++j) {
            const double x = A[j] - 1.0;
            if (s[j] == 0) {
                if (A[j] < 1) pole = true;
                if (x != 0) out += x * amis_neg_inf();
            } else {
                out += x * log(s[j]);
            }
        }
        if (pole) out = -amis_neg_inf();
    }
    double disc = v.head[(size_t)q * n + v.first[p]];
    const int32_t *pc = v.pcode + (size_t)p * k;
    const double *pr = v.pair + (size_t)q * k * n * n;
    for (int i = 0; i < k; ++i) disc += pr[pc[i]];
    // a trace of probability zero has density zero, also at a pole of the Dirichlet factor (+inf + -inf is not NaN here)
    return disc == amis_neg_inf() ? amis_neg_inf() : out + disc;
}

// relative weight of a sample (pass B): exp underflows to exactly 0 below -745.2
BILD_HD inline double amis_rel_weight(double logw, double top)
{
    const double dlt = logw - top;
    return dlt < -746.0 ? 0.0 : exp(dlt);
}

// ---- device passes (amis_device.hip; stubs in asan_stubs.cpp).  All buffers device memory; `partial` holds one row of
// `width` doubles per block.
struct AmisDeviceOut; // host-side staging owned by amis_host.cpp
constexpr int kAmisBlock = 64;        // lanes per block (one accumulator column in LDS per lane)
constexpr int kAmisPerLane = 8;       // samples per lane (passes B and C: one row of partial sums per block comes down)
constexpr int kAmisPerLaneA = 2;      // ... of pass A over the samples drawn so far: two numbers per block come down, and the pass is
                                      // bound by what ONE lane does in sequence (a log-density with its exp / log1p per sample), not by
                                      // the chip: 440 waves for 450 000 samples left most SIMDs idle
constexpr int kAmisMaxNm = 64;        // n * k1 the device passes support
int amis_dev_pass_a_rows(int64_t P0, int64_t P); // rows of `partial` pass A writes
// N samples from the proposal (a: k1 concentrations; prob: n x k1 slot weights, normalised per slot; trans: n x n) drawn on
// the device from the counter-based stream (seed, step): ss (N x k1) and theta8 (N x k1) are device memory
int amis_dev_draw(int k1, int n, int64_t N, uint64_t seed, uint64_t first /* pool index of the first new sample */, const double *a, const double *prob, const uint8_t *trans,
                  double *ss, uint8_t *theta8, void *stream);
// (the launches go to `stream`, a hipStream_t; nothing is waited for: the caller's copy of `partial` does that.  theta8 ...
// theta: see AmisDerive in amis_device.hip -- null when the host has prepared the new samples)
int amis_dev_pass_a(const AmisView &v, int64_t Q, int64_t P0, int64_t P, double logQ, double *log_ss /* = v.log_ss: written for the new samples */,
                    double *cur, double *logd, double *logw, double *partial /* rows x 2: max, any NaN */, int *rows, void *stream,
                    const uint8_t *theta8, uint8_t *has_zero, int32_t *first, int32_t *pcode, int32_t *theta,
                    double *lq_keep /* device, Q x (P - P0) doubles, or null: scratch for the new samples' log-densities */);
int amis_dev_pass_b(const AmisView &v, int64_t P, double top, int top_finite, const double *logw, double *rel,
                    double *partial /* blocks x (2 + k1 + n k1): W, S, acc, marg */, int blocks, void *stream);
int amis_dev_pass_c(const AmisView &v, int64_t P, const double *mean /* k1, device */, double ev, const double *rel, const double *cur,
                    double *partial /* blocks x (k1 + 2): var, sq, kl */, int blocks, void *stream);

} // namespace bild
