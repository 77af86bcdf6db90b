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
                                                            double *logw, double *partial, AmisDerive dv, double *lq_keep)
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
            // (the log-densities are kept for the second loop -- lq_keep: Q rows of hi - lo values, neighbouring lanes next to
            // each other -- instead of being computed twice: they are most of this kernel)
            const int64_t n_new = hi - lo, col = p - lo;
            for (int64_t q = 0; q < Q; ++q) {
                const double lq = amis_log_q(v, q, p);
                if (lq_keep) lq_keep[q * n_new + col] = lq;
                if (q == Q - 1) cq = lq;
                nan_q |= lq != lq;
                mx = lq > mx ? lq : mx;
            }
            if (nan_q) {
                ld = nan("");
            } else {
                if (!(mx > amis_neg_inf() && mx < -amis_neg_inf())) mx = 0.0;
                double s = 0.0;
                for (int64_t q = 0; q < Q; ++q) s += exp((lq_keep ? lq_keep[q * n_new + col] : amis_log_q(v, q, p)) - mx);
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

// ---- drawing the samples of a step on the device (opt-in: FixedkSampler(rng='device')) -----------------------------------
// The reference draws from NumPy's global stream (scipy.stats.dirichlet.rvs, np.random.choice, np.random.rand:
// bild/amis.py:66-81, 223-256), and so does this package by default -- 1.0 of the 1.8 ms of a step at N = 10 000.  This is
// the same distribution from a counter-based generator: Philox-4x32-10 keyed by the sampler's seed, one stream per
// (number of intervals k + 1, index of the sample in the POOL) -- the pool index grows with every step, also a failed one, and
// survives pickling / bild_amis_restore (a step counter did not: a restored sampler drew its first batches again); k + 1 keeps
// the samplers of one adaptive-k run, which share the user's seed, on distinct streams; gamma variates by Marsaglia-Tsang (with the boost gamma(a) = gamma(a + 1) U^(1/a) below one), the
// Dirichlet point as their normalised vector, the trace slot by slot from the CFC weights as bild_amis_sample_traces does.
// Not the reference's random numbers -- the same sampler in distribution (tests: evidences agree within their errors).
struct Philox {
    uint32_t c[4], k[2];
    uint32_t out[4];
    int have;
    __device__ Philox(uint64_t seed, uint64_t stream, uint64_t sample)
    {
        c[0] = (uint32_t)sample;
        c[1] = (uint32_t)(sample >> 32);
        c[2] = (uint32_t)stream;
        c[3] = 0; // block counter of this stream
        k[0] = (uint32_t)seed;
        k[1] = (uint32_t)(seed >> 32);
        have = 0;
    }
    __device__ void block()
    {
        uint32_t x0 = c[0], x1 = c[1], x2 = c[2], x3 = c[3], k0 = k[0], k1 = k[1];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
            const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1, y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
            x0 = y0;
            x1 = y1;
            x2 = y2;
            x3 = y3;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        out[0] = x0;
        out[1] = x1;
        out[2] = x2;
        out[3] = x3;
        ++c[3];
        have = 2;
    }
    // uniform in [0, 1) with 53 random bits
    __device__ double uniform()
    {
        if (!have) block();
        --have;
        const uint64_t bits = ((uint64_t)out[2 * have] << 32) | out[2 * have + 1];
        return (double)(bits >> 11) * 0x1p-53;
    }
    __device__ double normal() // Box-Muller, one of the pair
    {
        const double u1 = 1.0 - uniform(), u2 = uniform(); // u1 in (0, 1]
        return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
    }
    __device__ double gamma(double a)
    {
        if (!(a > 0)) return 0.0;
        double boost = 1.0;
        if (a < 1.0) { // gamma(a) = gamma(a + 1) U^(1/a)
            boost = pow(1.0 - uniform(), 1.0 / a);
            a += 1.0;
        }
        const double d = a - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * d);
        for (int it = 0; it < 64; ++it) { // (acceptance > 95 % per trial)
            const double x = normal(), t = 1.0 + cc * x;
            if (t <= 0) continue;
            const double v = t * t * t, u = 1.0 - uniform();
            if (u < 1.0 - 0.0331 * (x * x) * (x * x) || log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return boost * d * v;
        }
        return boost * d;
    }
};

// one sample per lane: ss (N x k1) and the states (N x k1 bytes) written where the likelihood and the passes read them
__global__ void __launch_bounds__(256) draw_kernel(int k1, int n, int64_t N, uint64_t seed, uint64_t first, const double *__restrict__ a,
                                                   const double *__restrict__ prob /* n x k1, [state][slot], normalised per slot */,
                                                   const uint8_t *__restrict__ trans, double *__restrict__ ss, uint8_t *__restrict__ theta8)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    Philox rng(seed, (uint64_t)k1, first + (uint64_t)r);
    double *row = ss + (size_t)r * k1;
    double tot = 0, asum = 0;
    for (int j = 0; j < k1; ++j) {
        const double g = rng.gamma(a[j]);
        row[j] = g;
        tot += g;
        asum += a[j];
    }
    if (tot > 0 && tot < HUGE_VAL) {
        for (int j = 0; j < k1; ++j) row[j] /= tot;
    } else {
        // all concentrations tiny: every variate underflows.  The distribution is then, to all digits, a mixture of point
        // masses at the corners of the simplex with probabilities a_j / sum(a) (as Dirichlet.sample on the host, amis.py)
        const double u = rng.uniform() * asum;
        double c = 0;
        int corner = k1 - 1;
        for (int j = 0; j < k1; ++j) {
            c += a[j];
            if (u < c) {
                corner = j;
                break;
            }
        }
        for (int j = 0; j < k1; ++j) row[j] = j == corner ? 1.0 : 0.0;
    }
    // the trace (bild_amis_sample_traces: first state from the slot-0 weights, later ones from the allowed successors)
    uint8_t *th = theta8 + (size_t)r * k1;
    int prev;
    {
        double last = 0;
        for (int s = 0; s < n; ++s) last += prob[(size_t)s * k1];
        const double u = rng.uniform();
        double c = 0;
        int idx = n - 1;
        for (int s = 0; s < n; ++s) {
            c += prob[(size_t)s * k1];
            if (c / last > u) {
                idx = s;
                break;
            }
        }
        th[0] = (uint8_t)idx;
        prev = idx;
    }
    for (int i = 1; i < k1; ++i) {
        double last = 0;
        for (int s = 0; s < n; ++s) last += prob[(size_t)s * k1 + i] * (trans[(size_t)prev * n + s] ? 1.0 : 0.0);
        const double u = rng.uniform();
        double c = 0;
        int idx = 0; // first crossing, 0 if there is none
        for (int s = 0; s < n; ++s) {
            c += prob[(size_t)s * k1 + i] * (trans[(size_t)prev * n + s] ? 1.0 : 0.0);
            if (c / last > u) {
                idx = s;
                break;
            }
        }
        th[i] = (uint8_t)idx;
        prev = idx;
    }
}

int finish(hipStream_t st, bool wait)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && wait) e = hipStreamSynchronize(st);
    return e == hipSuccess ? 0 : 1;
}


// Pass A for the NEW samples of a step, four lanes per sample: a new sample needs its log-density under every proposal used so
// far (Q of them, twice: for the maximum, then for the sum) -- one lane per sample left 157 waves walking 2 Q densities each
// while the chip looked on.  Lane `sub` of a sample's quad takes the proposals q = sub, sub + 4, ...; maximum, NaN flag and sum
// are combined across the quad.  (What the host's loop derives from the states is written by all four lanes -- the same values --
// so that each lane reads back its own stores.)  The sum over q is formed in four interleaved parts, (s0 + s1) + (s2 + s3): it
// agrees with the host's sequential sum to rounding.
constexpr int kAmisQuad = 4;
__global__ void __launch_bounds__(kAmisBlock) pass_a_new_kernel(AmisView v, int64_t Q, int64_t lo, int64_t hi, int row0, double logQ,
                                                                double *log_ss, double *cur, double *logd, double *logw,
                                                                double *partial, AmisDerive dv, double *lq_keep)
{
    __shared__ double top_s[kAmisBlock];
    __shared__ int nan_s;
    const int tid = threadIdx.x, sub = tid & (kAmisQuad - 1);
    if (tid == 0) nan_s = 0;
    __syncthreads();
    const int64_t p = lo + (int64_t)blockIdx.x * (kAmisBlock / kAmisQuad) + (tid >> 2);
    const bool live = p < hi;
    const int64_t n_new = hi - lo, colx = p - lo;
    double cq = 0.0, mx = amis_neg_inf(), s = 0.0;
    bool nan_q = false;
    if (live) {
        bool z = false;
        for (int j = 0; j < v.k1; ++j) {
            const double sv = v.ss[(size_t)p * v.k1 + j];
            z |= sv == 0;
            log_ss[(size_t)p * v.k1 + j] = sv == 0 ? 0.0 : log(sv);
        }
        if (dv.theta8 != nullptr) {
            const uint8_t *t8 = dv.theta8 + (size_t)p * v.k1;
            dv.has_zero[p] = z ? 1 : 0;
            dv.first[p] = t8[0];
            for (int i = 0; i < v.k1; ++i) dv.theta[(size_t)p * v.k1 + i] = t8[i];
            for (int i = 0; i < v.k; ++i) dv.pcode[(size_t)p * v.k + i] = (i * v.n + t8[i]) * v.n + t8[i + 1];
        }
        for (int64_t q = sub; q < Q; q += kAmisQuad) {
            const double lq = amis_log_q(v, q, p);
            lq_keep[q * n_new + colx] = lq;
            if (q == Q - 1) cq = lq;
            nan_q |= lq != lq;
            mx = lq > mx ? lq : mx;
        }
    }
    // across the quad (all lanes take part in the exchanges; a lane without a sample carries neutral values)
    for (int off = 1; off < kAmisQuad; off <<= 1) {
        const double m2 = __shfl_xor(mx, off);
        mx = m2 > mx ? m2 : mx;
        nan_q |= __shfl_xor(nan_q ? 1 : 0, off) != 0;
        cq += __shfl_xor(cq, off); // (exactly one lane of the quad holds it, the others 0)
    }
    double ld;
    if (nan_q) {
        ld = nan("");
    } else {
        if (!(mx > amis_neg_inf() && mx < -amis_neg_inf())) mx = 0.0;
        if (live)
            for (int64_t q = sub; q < Q; q += kAmisQuad) s += exp(lq_keep[q * n_new + colx] - mx);
        for (int off = 1; off < kAmisQuad; off <<= 1) s += __shfl_xor(s, off);
        ld = log(s) + mx;
    }
    double top = amis_neg_inf();
    bool any_nan = false;
    if (live && sub == 0) {
        cur[p] = cq;
        logd[p] = ld;
        const double lw = v.logL[p] - ld + logQ;
        logw[p] = lw;
        any_nan = lw != lw;
        top = lw == lw ? lw : top; // NaN never replaces the maximum
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
} // namespace

int amis_dev_pass_a(const AmisView &v, int64_t Q, int64_t P0, int64_t P, double logQ, double *log_ss, double *cur, double *logd,
                    double *logw, double *partial, int *rows, void *stream, const uint8_t *theta8, uint8_t *has_zero, int32_t *first,
                    int32_t *pcode, int32_t *theta, double *lq_keep)
{
    hipStream_t st = (hipStream_t)stream;
    const AmisDerive dv{theta8, has_zero, first, pcode, theta};
    // the samples drawn so far: one log-density each, kAmisPerLaneA per lane; the new ones: all Q proposals each (twice:
    // maximum, then sum) -- one per lane, or ten blocks would work while the rest of the chip looks on
    const int64_t per_block = (int64_t)kAmisBlock * kAmisPerLaneA;
    const int per_new = kAmisBlock / kAmisQuad; // (the rows of `partial` are laid out for the quad kernel: amis_dev_pass_a_rows)
    const int old_blocks = (int)((P0 + per_block - 1) / per_block), new_blocks = (int)((P - P0 + per_new - 1) / per_new);
    if (old_blocks)
        hipLaunchKernelGGL(pass_a_kernel, dim3(old_blocks), dim3(kAmisBlock), 0, st, v, Q, P0, (int64_t)0, P0, kAmisPerLaneA, 0, logQ,
                           log_ss, cur, logd, logw, partial, dv, (double *)nullptr);
    if (new_blocks && lq_keep)
        hipLaunchKernelGGL(pass_a_new_kernel, dim3(new_blocks), dim3(kAmisBlock), 0, st, v, Q, P0, P, old_blocks, logQ, log_ss, cur, logd,
                           logw, partial, dv, lq_keep);
    int new_rows = new_blocks;
    if (new_blocks && !lq_keep) { // (no scratch memory: one lane per sample, every density twice)
        new_rows = (int)((P - P0 + kAmisBlock - 1) / kAmisBlock);
        hipLaunchKernelGGL(pass_a_kernel, dim3(new_rows), dim3(kAmisBlock), 0, st, v, Q, P0, P0, P, 1, old_blocks, logQ, log_ss, cur, logd,
                           logw, partial, dv, (double *)nullptr);
    }
    *rows = old_blocks + new_rows;
    return finish(st, false);
}

int amis_dev_draw(int k1, int n, int64_t N, uint64_t seed, uint64_t first, const double *a, const double *prob, const uint8_t *trans,
                  double *ss, uint8_t *theta8, void *stream)
{
    hipLaunchKernelGGL(draw_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k1, n, N, seed, first, a, prob, trans,
                       ss, theta8);
    return finish((hipStream_t)stream, false);
}

int amis_dev_pass_a_rows(int64_t P0, int64_t P)
{
    const int64_t per_block = (int64_t)kAmisBlock * kAmisPerLaneA, per_new = kAmisBlock / kAmisQuad;
    return (int)((P0 + per_block - 1) / per_block) + (int)((P - P0 + per_new - 1) / per_new);
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
