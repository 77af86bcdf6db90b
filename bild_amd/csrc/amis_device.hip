// Device passes of the AMIS bookkeeping (reference bild/amis.py:819-906): the three sweeps
// over ALL samples drawn so far that an AMIS step consists of once the likelihood of the new batch is known --
//   A  mixture denominators and deterministic-mixture weights (one log-density of the newest proposal per old sample,
//      all proposals for the new ones; exp + log1p each),
//   B  relative weights and the weighted first moments / slot marginals / weight sum of the refit,
//   C  second moments, spread of the weights, KL sum.
// On the host these are 3.9 of the 5.2 ms of a step at N = 10 000 (pool of 2e5 samples; threads do not help on this
// pool), 28 times the likelihood of the batch.  The pooled samples therefore stay in HBM next to
// the likelihood's results, and a step moves the new samples up and a few hundred partial sums down.
//
// HBM-bound by construction (about 130 B per sample and pass) but small: 2e5 samples are 26 MB per pass -- launch and
// synchronisation latency is what a step pays, so the passes are three plain kernels and two tiny host reductions.
// Per-sample arithmetic: amis_math.h, shared with the host implementation.  Sums: every lane adds its samples in index
// order into its own LDS column, the columns are added by a fixed tree, blocks are added in block order on the host --
// a result does not depend on scheduling.
#include <hip/hip_runtime.h>

#include "amis_math.h"

namespace bild {
namespace {

__device__ inline void tree_sum(double *col, int width, int tid)
{
    // col: width entries x kAmisBlock lanes, entry e of lane t at col[e * kAmisBlock + t]
    for (int half = kAmisBlock / 2; half > 0; half >>= 1) {
        __syncthreads();
        if (tid < half)
            for (int e = 0; e < width; ++e) col[e * kAmisBlock + tid] += col[e * kAmisBlock + tid + half];
    }
    __syncthreads();
}

// samples [lo, hi), `per_lane` consecutive ones per lane; block b writes its partial to row row0 + b
// what pass A derives for NEW samples that were not prepared on the host (fused step: the samples went up as (s, theta)
// rows for the likelihood and stay where they are): pointers into the pool, null when the host has prepared them
struct AmisDerive {
    const uint8_t *theta8; // P x k1, the states as they went up
    uint8_t *has_zero;
    int32_t *first, *pcode, *theta;
};

__global__ void __launch_bounds__(kAmisBlock) pass_a_kernel(AmisView v, int64_t Q, int64_t P0, int64_t lo, int64_t hi, int per_lane,
                                                            int row0, double logQ, double *log_ss, double *cur, double *logd,
                                                            double *logw, double *partial, AmisDerive dv)
{
    __shared__ double top_s[kAmisBlock];
    __shared__ int nan_s;
    const int tid = threadIdx.x;
    if (tid == 0) nan_s = 0;
    __syncthreads();
    const int64_t base = lo + ((int64_t)blockIdx.x * kAmisBlock + tid) * per_lane;
    double top = amis_neg_inf();
    bool any_nan = false;
    for (int64_t p = base; p < base + per_lane && p < hi; ++p) {
        double cq, ld;
        if (p < P0) {
            cq = amis_log_q(v, Q - 1, p);
            ld = amis_logaddexp(logd[p], cq);
        } else {
            // a new sample: its logs first (the host skipped them), as the host takes them: log(0) stays out (0 stands in)
            bool z = false;
            for (int j = 0; j < v.k1; ++j) {
                const double sv = v.ss[(size_t)p * v.k1 + j];
                z |= sv == 0;
                log_ss[(size_t)p * v.k1 + j] = sv == 0 ? 0.0 : log(sv);
            }
            if (dv.theta8 != nullptr) { // ... and what bild_amis_step's host loop derives from the states (amis_host.cpp)
                const uint8_t *t8 = dv.theta8 + (size_t)p * v.k1;
                dv.has_zero[p] = z ? 1 : 0;
                dv.first[p] = t8[0];
                for (int i = 0; i < v.k1; ++i) dv.theta[(size_t)p * v.k1 + i] = t8[i];
                for (int i = 0; i < v.k; ++i) dv.pcode[(size_t)p * v.k + i] = (i * v.n + t8[i]) * v.n + t8[i + 1];
            }
            // log-sum-exp over all proposals used so far, as the host's lse(): largest first, NaN if any term is
            double mx = amis_neg_inf();
            bool nan_q = false;
            cq = 0.0;
            for (int64_t q = 0; q < Q; ++q) {
                const double lq = amis_log_q(v, q, p);
                if (q == Q - 1) cq = lq;
                nan_q |= lq != lq;
                mx = lq > mx ? lq : mx;
            }
            if (nan_q) {
                ld = nan("");
            } else {
                if (!(mx > amis_neg_inf() && mx < -amis_neg_inf())) mx = 0.0;
                double s = 0.0;
                for (int64_t q = 0; q < Q; ++q) s += exp(amis_log_q(v, q, p) - mx);
                ld = log(s) + mx;
            }
        }
        cur[p] = cq;
        logd[p] = ld;
        const double lw = v.logL[p] - ld + logQ;
        logw[p] = lw;
        any_nan |= lw != lw;
        top = (top < lw) ? lw : top; // NaN never replaces the maximum (std::max semantics of the host pass)
    }
    top_s[tid] = top;
    if (any_nan) atomicOr(&nan_s, 1);
    for (int half = kAmisBlock / 2; half > 0; half >>= 1) {
        __syncthreads();
        if (tid < half) top_s[tid] = top_s[tid] < top_s[tid + half] ? top_s[tid + half] : top_s[tid];
    }
    __syncthreads();
    if (tid == 0) {
        partial[2 * (row0 + blockIdx.x)] = top_s[0];
        partial[2 * (row0 + blockIdx.x) + 1] = nan_s ? 1.0 : 0.0;
    }
}

__global__ void __launch_bounds__(kAmisBlock) pass_b_kernel(AmisView v, int64_t P, double top, int top_finite, const double *logw,
                                                            double *rel, double *partial)
{
    extern __shared__ double col[]; // (2 + k1 + n k1) x kAmisBlock
    const int tid = threadIdx.x, k1 = v.k1, nm = v.n * v.k1, width = 2 + k1 + nm;
    for (int e = 0; e < width; ++e) col[e * kAmisBlock + tid] = 0.0;
    const double tiny = 2.2250738585072014e-308;
    const int64_t base = ((int64_t)blockIdx.x * kAmisBlock + tid) * kAmisPerLane;
    for (int64_t p = base; p < base + kAmisPerLane && p < P; ++p) {
        const double w = amis_rel_weight(logw[p], top);
        rel[p] = w;
        if (w >= 1e-100) {
            col[0 * kAmisBlock + tid] += w;
            const double *sp = v.ss + (size_t)p * k1;
            for (int j = 0; j < k1; ++j) col[(2 + j) * kAmisBlock + tid] += w * sp[j];
        }
        if (top_finite && w != 0) {
            const int32_t *th = v.theta + (size_t)p * k1;
            for (int i = 0; i < k1; ++i) col[(2 + k1 + th[i] * k1 + i) * kAmisBlock + tid] += w;
        }
        if (w >= tiny) col[1 * kAmisBlock + tid] += w; // subnormal weights: no effect on the sums
    }
    tree_sum(col, width, tid);
    for (int e = tid; e < width; e += kAmisBlock) partial[(size_t)blockIdx.x * width + e] = col[e * kAmisBlock];
}

__global__ void __launch_bounds__(kAmisBlock) pass_c_kernel(AmisView v, int64_t P, const double *mean, double ev, const double *rel,
                                                            const double *cur, double *partial)
{
    extern __shared__ double col[]; // (k1 + 2) x kAmisBlock
    const int tid = threadIdx.x, k1 = v.k1, width = k1 + 2;
    for (int e = 0; e < width; ++e) col[e * kAmisBlock + tid] = 0.0;
    const double tiny = 2.2250738585072014e-308;
    const int64_t base = ((int64_t)blockIdx.x * kAmisBlock + tid) * kAmisPerLane;
    for (int64_t p = base; p < base + kAmisPerLane && p < P; ++p) {
        const double w = rel[p];
        if (w >= 1e-100) {
            const double *sp = v.ss + (size_t)p * k1;
            for (int j = 0; j < k1; ++j) {
                const double dv = sp[j] - mean[j];
                col[j * kAmisBlock + tid] += w * dv * dv;
            }
        }
        const double we = w < tiny ? 0.0 : w;
        const double dv = we - ev;
        col[k1 * kAmisBlock + tid] += dv * dv;
        const double term = we * (v.logL[p] - cur[p]);
        if (term == term) col[(k1 + 1) * kAmisBlock + tid] += term; // zero-weight samples the current proposal cannot produce: dropped
    }
    tree_sum(col, width, tid);
    for (int e = tid; e < width; e += kAmisBlock) partial[(size_t)blockIdx.x * width + e] = col[e * kAmisBlock];
}

int finish(hipStream_t st, bool wait)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && wait) e = hipStreamSynchronize(st);
    return e == hipSuccess ? 0 : 1;
}

} // namespace

int amis_dev_pass_a(const AmisView &v, int64_t Q, int64_t P0, int64_t P, double logQ, double *log_ss, double *cur, double *logd,
                    double *logw, double *partial, int *rows, void *stream, const uint8_t *theta8, uint8_t *has_zero, int32_t *first,
                    int32_t *pcode, int32_t *theta)
{
    hipStream_t st = (hipStream_t)stream;
    const AmisDerive dv{theta8, has_zero, first, pcode, theta};
    // the samples drawn so far: one log-density each, kAmisPerLane per lane; the new ones: all Q proposals each (twice:
    // maximum, then sum) -- one per lane, or ten blocks would work while the rest of the chip looks on
    const int64_t per_block = (int64_t)kAmisBlock * kAmisPerLane;
    const int old_blocks = (int)((P0 + per_block - 1) / per_block), new_blocks = (int)((P - P0 + kAmisBlock - 1) / kAmisBlock);
    if (old_blocks)
        hipLaunchKernelGGL(pass_a_kernel, dim3(old_blocks), dim3(kAmisBlock), 0, st, v, Q, P0, (int64_t)0, P0, kAmisPerLane, 0, logQ,
                           log_ss, cur, logd, logw, partial, dv);
    if (new_blocks)
        hipLaunchKernelGGL(pass_a_kernel, dim3(new_blocks), dim3(kAmisBlock), 0, st, v, Q, P0, P0, P, 1, old_blocks, logQ, log_ss,
                           cur, logd, logw, partial, dv);
    *rows = old_blocks + new_blocks;
    return finish(st, false);
}

int amis_dev_pass_a_rows(int64_t P0, int64_t P)
{
    const int64_t per_block = (int64_t)kAmisBlock * kAmisPerLane;
    return (int)((P0 + per_block - 1) / per_block) + (int)((P - P0 + kAmisBlock - 1) / kAmisBlock);
}

int amis_dev_pass_b(const AmisView &v, int64_t P, double top, int top_finite, const double *logw, double *rel, double *partial,
                    int blocks, void *stream)
{
    const size_t lds = (size_t)(2 + v.k1 + v.n * v.k1) * kAmisBlock * sizeof(double);
    hipLaunchKernelGGL(pass_b_kernel, dim3(blocks), dim3(kAmisBlock), lds, (hipStream_t)stream, v, P, top, top_finite, logw, rel, partial);
    return finish((hipStream_t)stream, false);
}

int amis_dev_pass_c(const AmisView &v, int64_t P, const double *mean, double ev, const double *rel, const double *cur, double *partial,
                    int blocks, void *stream)
{
    const size_t lds = (size_t)(v.k1 + 2) * kAmisBlock * sizeof(double);
    hipLaunchKernelGGL(pass_c_kernel, dim3(blocks), dim3(kAmisBlock), lds, (hipStream_t)stream, v, P, mean, ev, rel, cur, partial);
    return finish((hipStream_t)stream, false);
}

} // namespace bild
