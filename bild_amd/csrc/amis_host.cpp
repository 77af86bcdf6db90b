// Host-side bookkeeping of one fixed-k AMIS sampler: the part of
// reference bild/amis.py FixedkSampler.step (amis.py:805-906) that is left once the likelihood of a
// batch is known -- mixture denominators of all samples drawn so far, deterministic-mixture weights,
// the weighted refit of both proposal families (Dirichlet method of moments amis.py:110-151, CFC
// marginals and their inversion amis.py:284-399), the brakes, and evidence / standard error / KL.
//
// Why native: with the likelihood on the GPU this bookkeeping IS an AMIS step.  In NumPy it is ~50 array
// calls per step (0.7 ms at the reference's default N = 100, 10 ms at N = 10 000); here it is one pass
// over the pooled samples.  bild_amd/amis.py keeps the NumPy formulation as the specification and the
// tests compare the two step by step.  Random numbers are NOT drawn here: the caller draws them from
// the NumPy stream exactly as the reference does, so runs stay reproducible against it.
//
// Plain host C++ (no GPU involved); exported through the same C ABI (include/bild_amd.h).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>

#include <hip/hip_runtime_api.h>

#include "../../include/bild_amd.h"
#include "amis_math.h"
#include "config.h"
#include "internal.h"

namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();

// log(sum(exp(a[i]) for selected i)), shifted by the largest selected entry; nothing selected or all -inf: -inf
template <typename Sel>
double lse(int n, const double *a, int stride, Sel selected)
{
    double top = -kInf;
    for (int i = 0; i < n; ++i)
        if (selected(i)) top = std::max(top, a[(size_t)i * stride]); // NaN entries are ignored by max, summed below
    bool any_nan = false;
    for (int i = 0; i < n; ++i)
        if (selected(i) && std::isnan(a[(size_t)i * stride])) any_nan = true;
    if (any_nan) return std::numeric_limits<double>::quiet_NaN();
    if (!std::isfinite(top)) top = 0.0;
    double s = 0.0;
    for (int i = 0; i < n; ++i)
        if (selected(i)) s += std::exp(a[(size_t)i * stride] - top);
    return std::log(s) + top;
}

double logaddexp(double x, double y) { return bild::amis_logaddexp(x, y); }

} // namespace

struct bild_amis {
    int k1 = 0, k = 0, n = 0;
    double brake_c = 0, brake_p = 0, logprior = 0;
    std::vector<uint8_t> trans; // n x n, [from][to]
    // proposals, in order of use: a (k1), logp (n x k1, [state][slot]) and what is derived from them
    std::vector<std::vector<double>> a, logp, head, pair;
    std::vector<double> dir_norm;
    std::vector<double> a_flat, head_flat, pair_flat; // the same, proposal after proposal (what AmisView points at)
    // pooled samples
    std::vector<double> ss, log_ss;     // P x k1
    std::vector<uint8_t> has_zero;      // P
    std::vector<int32_t> first, pcode;  // P, P x k
    std::vector<int32_t> theta;         // P x k1
    std::vector<double> logL, logd, cur, logw;
    int64_t steps = 0;
    std::string err;
    int mom_maxiter = 1000;
    double mom_precision = 1e-2;

    int64_t P() const { return (int64_t)logL.size(); } // samples the HOST holds
    // all samples: fused steps (bild_amis_step_fused) leave the new ones in HBM only, the host catches up when somebody looks
    int64_t total() const { return dev.on && dev.P > P() ? dev.P : P(); }

    void derive(size_t q)
    {
        const std::vector<double> &A = a[q], &L = logp[q];
        double sum = 0, lg = 0;
        for (int j = 0; j < k1; ++j) {
            sum += A[j];
            lg += std::lgamma(A[j]);
        }
        dir_norm.push_back(std::lgamma(sum) - lg);
        std::vector<double> h(n), pr((size_t)k * n * n);
        const double n0 = lse(n, L.data(), k1, [](int) { return true; });
        for (int s = 0; s < n; ++s) h[s] = L[(size_t)s * k1] - n0;
        for (int i = 1; i < k1; ++i)
            for (int prev = 0; prev < n; ++prev) {
                const double norm = lse(n, L.data() + i, k1, [&](int c) { return trans[(size_t)prev * n + c] != 0; });
                for (int c = 0; c < n; ++c) {
                    const double v = L[(size_t)c * k1 + i];
                    // a state of weight exactly zero has probability zero, also where -inf - (-inf) would be NaN
                    pr[((size_t)(i - 1) * n + prev) * n + c] = (v == -kInf) ? -kInf : v - norm;
                }
            }
        a_flat.insert(a_flat.end(), A.begin(), A.end());
        head_flat.insert(head_flat.end(), h.begin(), h.end());
        pair_flat.insert(pair_flat.end(), pr.begin(), pr.end());
        head.push_back(std::move(h));
        pair.push_back(std::move(pr));
    }

    bild::AmisView view() const
    {
        bild::AmisView v{};
        v.k1 = k1;
        v.k = k;
        v.n = n;
        v.a = a_flat.data();
        v.dir_norm = dir_norm.data();
        v.head = head_flat.data();
        v.pair = pair_flat.data();
        v.ss = ss.data();
        v.log_ss = log_ss.data();
        v.has_zero = has_zero.data();
        v.first = first.data();
        v.pcode = pcode.data();
        v.theta = theta.data();
        v.logL = logL.data();
        return v;
    }

    // log density of proposal q at pooled sample p (amis_math.h: the expression the device passes evaluate too)
    double log_q(size_t q, int64_t p) const { return bild::amis_log_q(view(), (int64_t)q, p); }

    // ---- device mirror (bild_amis_use_device): the pooled samples and the proposals in HBM -----------------------------
    struct Dev {
        bool on = false;
        bool host_stale = false; // logd / cur / logw of the host are older than the device's (pulled on demand)
        int64_t cap = 0, P = 0;  // samples: room, mirrored
        int64_t qcap = 0, Q = 0; // proposals
        double *a = nullptr, *dir_norm = nullptr, *head = nullptr, *pair = nullptr;
        double *ss = nullptr, *log_ss = nullptr, *logL = nullptr, *logd = nullptr, *cur = nullptr, *logw = nullptr, *rel = nullptr;
        uint8_t *has_zero = nullptr;
        int32_t *first = nullptr, *pcode = nullptr, *theta = nullptr;
        uint8_t *theta8 = nullptr; // fused step: the states as they went up for the likelihood (P x k1)
        double *lq_keep = nullptr; // pass A: the new samples' log-densities under all proposals, between its two loops
        int64_t lq_cap = 0;
        double *partial = nullptr, *mean = nullptr; // partial: PINNED HOST memory the passes write their block sums into
        int64_t partial_cap = 0;
        double *draw_par = nullptr;     // device-side draws: [a (k1) | slot weights (n x k1) | transitions (n x n bytes)]
        void *stage = nullptr; // pinned host memory: the new samples of a step on their way up
        size_t stage_bytes = 0;
        void *qstage = nullptr; // pinned host memory: the new proposal(s) of a step on their way up (fused step: asynchronous)
        size_t qstage_bytes = 0;
    };
    int64_t log_ss_valid = 0; // samples whose log(s) the HOST holds (with the device mirror on, the device takes the logs)
    mutable Dev dev;
    ~bild_amis();
};

namespace {

// The passes over the pooled samples are cut into chunks of kChunk samples; every chunk produces its own partial
// sums, which are then added in chunk order -- so the result does not depend on how many threads ran the chunks.
// One thread by default: measured on the MI355X host at N = 10 000 (pool of 1e5 samples) 2-16 threads started per
// pass were SLOWER (5.0-5.5 ms per AMIS step against 4.65 ms) -- the step is then dominated by drawing the samples
// and by the likelihood, not by these passes.  BILD_AMIS_THREADS=<n> enables n threads for pools of >= 8 chunks.
constexpr int64_t kChunk = 4096;

int worker_count(int64_t nchunks)
{
    if (nchunks < 8) return 1;
    int want = 1;
    want = bild::config().amis_threads;
    return (int)std::max<int64_t>(1, std::min<int64_t>(want, nchunks / 2));
}

template <typename F>
void for_chunks(int64_t P, F fn)
{
    const int64_t nchunks = (P + kChunk - 1) / kChunk;
    const int workers = worker_count(nchunks);
    auto run = [&](int w) {
        for (int64_t c = w; c < nchunks; c += workers) fn(c, c * kChunk, std::min(P, (c + 1) * kChunk));
    };
    if (workers == 1) {
        run(0);
        return;
    }
    std::vector<std::thread> pool;
    for (int w = 1; w < workers; ++w) pool.emplace_back(run, w);
    run(0);
    for (std::thread &t : pool) t.join();
}

// fixed-point inversion of one slot's marginals (amis.py:339-399): 0 ok, 1 did not converge
int solve_marginals_single(const bild_amis &m, const double *logf, const double *logg, double *out)
{
    const int n = m.n;
    bool f_has0 = false, g_has0 = false;
    for (int s = 0; s < n; ++s) {
        f_has0 |= logf[s] == 0;
        g_has0 |= logg[s] == 0;
    }
    if (f_has0 || g_has0) { // delta-like marginals need no iteration
        std::copy(logf, logf + n, out);
        return 0;
    }
    std::vector<double> cur(logf, logf + n), nw(n), out_norm(n), flow(n), inflow(n);
    for (int it = 0; it < m.mom_maxiter; ++it) {
        for (int a = 0; a < n; ++a) {
            out_norm[a] = lse(n, cur.data(), 1, [&](int c) { return m.trans[(size_t)a * n + c] != 0; });
            if (logg[a] == -kInf) out_norm[a] = 0;
            flow[a] = logg[a] - out_norm[a];
        }
        for (int c = 0; c < n; ++c) {
            inflow[c] = lse(n, flow.data(), 1, [&](int a) { return m.trans[(size_t)a * n + c] != 0; });
            if (logf[c] == -kInf) inflow[c] = 0;
            nw[c] = logf[c] - inflow[c];
        }
        const double norm = lse(n, nw.data(), 1, [](int) { return true; });
        double worst = 0;
        bool bad = false;
        for (int c = 0; c < n; ++c) {
            nw[c] -= norm;
            if (logf[c] != -kInf) {
                const double dlt = std::fabs(nw[c] - cur[c]);
                if (std::isnan(dlt)) bad = true;
                worst = std::max(worst, dlt);
            }
        }
        if (!bad && worst < m.mom_precision) {
            std::copy(nw.begin(), nw.end(), out);
            return 0;
        }
        cur = nw;
    }
    return 1;
}


// ---- device mirror ---------------------------------------------------------------------------------------------------
#define AMIS_HIP(call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            m.err = std::string("device bookkeeping: ") + #call + ": " + hipGetErrorString(e_);     \
            return BILD_ERR_HIP;                                                                    \
        }                                                                                           \
    } while (0)

template <typename T>
int dev_regrow(bild_amis &m, T *&ptr, size_t old_count, size_t new_count)
{
    T *fresh = nullptr;
    AMIS_HIP(hipMalloc((void **)&fresh, std::max<size_t>(new_count, 1) * sizeof(T)));
    if (ptr && old_count) AMIS_HIP(hipMemcpy(fresh, ptr, old_count * sizeof(T), hipMemcpyDeviceToDevice));
    if (ptr) (void)hipFree(ptr);
    ptr = fresh;
    return BILD_OK;
}

void dev_release(bild_amis::Dev &d)
{
    void *all[] = {d.a, d.dir_norm, d.head, d.pair, d.ss, d.log_ss, d.logL, d.logd, d.cur, d.logw, d.rel, d.has_zero, d.first,
                   d.pcode, d.theta, d.theta8, d.mean, d.draw_par, d.lq_keep};
    for (void *q : all)
        if (q) (void)hipFree(q);
    if (d.stage) (void)hipHostFree(d.stage);
    if (d.qstage) (void)hipHostFree(d.qstage);
    if (d.partial) (void)hipHostFree(d.partial);
    d = bild_amis::Dev();
}

// room for P samples and Q proposals; what is mirrored already is kept
int dev_reserve(bild_amis &m, int64_t P, int64_t Q)
{
    bild_amis::Dev &d = m.dev;
    const size_t k1 = m.k1, k = m.k, n = m.n;
    int rc;
    bool regrown = false;
    if (P > d.cap) {
        regrown = true;
        const int64_t cap = std::max<int64_t>(P, d.cap * 2);
        const size_t have = (size_t)d.P;
        if ((rc = dev_regrow(m, d.ss, have * k1, cap * k1)) || (rc = dev_regrow(m, d.log_ss, have * k1, cap * k1)) ||
            (rc = dev_regrow(m, d.logL, have, (size_t)cap)) || (rc = dev_regrow(m, d.logd, have, (size_t)cap)) ||
            (rc = dev_regrow(m, d.cur, have, (size_t)cap)) || (rc = dev_regrow(m, d.logw, have, (size_t)cap)) ||
            (rc = dev_regrow(m, d.rel, 0, (size_t)cap)) || (rc = dev_regrow(m, d.has_zero, have, (size_t)cap)) ||
            (rc = dev_regrow(m, d.first, have, (size_t)cap)) || (rc = dev_regrow(m, d.pcode, have * k, cap * std::max<size_t>(k, 1))) ||
            (rc = dev_regrow(m, d.theta, have * k1, cap * k1)) || (rc = dev_regrow(m, d.theta8, have * k1, cap * k1)))
            return rc;
        d.cap = cap;
    }
    if (Q > d.qcap) {
        regrown = true;
        const int64_t qcap = std::max<int64_t>(Q, d.qcap * 2 + 8);
        const size_t have = (size_t)d.Q;
        if ((rc = dev_regrow(m, d.a, have * k1, qcap * k1)) || (rc = dev_regrow(m, d.dir_norm, have, (size_t)qcap)) ||
            (rc = dev_regrow(m, d.head, have * n, qcap * n)) || (rc = dev_regrow(m, d.pair, have * k * n * n, qcap * std::max<size_t>(k * n * n, 1))))
            return rc;
        d.qcap = qcap;
    }
    if (!d.mean) AMIS_HIP(hipMalloc((void **)&d.mean, k1 * sizeof(double)));
    // (the device-to-device copies of a regrowth ran on the null stream; the passes may run on another one)
    if (regrown) AMIS_HIP(hipDeviceSynchronize());
    return BILD_OK;
}

// upload what the host holds beyond the mirror: samples [d.P, P) (static data; with_state: also logd / cur / logw, for a
// mirror that is switched on late) and proposals [d.Q, Q)
// `stream` (fused step): the proposals go up asynchronously on the stream the passes run on, out of a pinned block of their own
// -- four synchronous copies out of pageable memory were 50 us of every step; the step's last synchronisation covers them
int dev_push(bild_amis &m, bool with_state, hipStream_t stream = nullptr, bool async_proposals = false)
{
    bild_amis::Dev &d = m.dev;
    const size_t k1 = m.k1, k = m.k, n = m.n;
    const int64_t P = m.P(), Q = (int64_t)m.a.size();
    int rc;
    if ((rc = dev_reserve(m, P, Q))) return rc;
    const size_t lo = (size_t)d.P, cnt = P > d.P ? (size_t)(P - d.P) : 0;
    if (cnt) {
        // one pinned staging block, asynchronous copies out of it, one synchronisation (seven synchronous copies out of
        // pageable memory cost 0.2 ms per step)
        const bool logs = (int64_t)(lo + cnt) <= m.log_ss_valid; // the host has the logs of these samples: send them
        struct Piece { const void *src; void *dst; size_t bytes; };
        std::vector<Piece> pieces = {
            {m.ss.data() + lo * k1, d.ss + lo * k1, cnt * k1 * sizeof(double)},
            {m.logL.data() + lo, d.logL + lo, cnt * sizeof(double)},
            {m.first.data() + lo, d.first + lo, cnt * sizeof(int32_t)},
            {m.theta.data() + lo * k1, d.theta + lo * k1, cnt * k1 * sizeof(int32_t)},
            {m.has_zero.data() + lo, d.has_zero + lo, cnt},
        };
        if (k) pieces.push_back({m.pcode.data() + lo * k, d.pcode + lo * k, cnt * k * sizeof(int32_t)});
        if (logs) pieces.push_back({m.log_ss.data() + lo * k1, d.log_ss + lo * k1, cnt * k1 * sizeof(double)});
        if (with_state) {
            pieces.push_back({m.logd.data() + lo, d.logd + lo, cnt * sizeof(double)});
            pieces.push_back({m.cur.data() + lo, d.cur + lo, cnt * sizeof(double)});
            pieces.push_back({m.logw.data() + lo, d.logw + lo, cnt * sizeof(double)});
        }
        size_t total = 0;
        for (const Piece &pc : pieces) total += (pc.bytes + 15) & ~(size_t)15;
        if (total > d.stage_bytes) {
            if (d.stage) (void)hipHostFree(d.stage);
            d.stage = nullptr;
            d.stage_bytes = 0;
            AMIS_HIP(hipHostMalloc(&d.stage, total * 2, hipHostMallocDefault));
            d.stage_bytes = total * 2;
        }
        size_t off = 0;
        for (const Piece &pc : pieces) {
            std::memcpy((char *)d.stage + off, pc.src, pc.bytes);
            AMIS_HIP(hipMemcpyAsync(pc.dst, (char *)d.stage + off, pc.bytes, hipMemcpyHostToDevice, nullptr));
            off += (pc.bytes + 15) & ~(size_t)15;
        }
        AMIS_HIP(hipStreamSynchronize(nullptr));
        d.P = P;
    }
    const size_t qlo = (size_t)d.Q, qcnt = (size_t)(Q - d.Q);
    if (qcnt && async_proposals) {
        struct Piece { const double *src; double *dst; size_t count; };
        const Piece pieces[4] = {{m.a_flat.data() + qlo * k1, d.a + qlo * k1, qcnt * k1},
                                 {m.dir_norm.data() + qlo, d.dir_norm + qlo, qcnt},
                                 {m.head_flat.data() + qlo * n, d.head + qlo * n, qcnt * n},
                                 {m.pair_flat.data() + qlo * k * n * n, d.pair + qlo * k * n * n, qcnt * k * n * n}};
        size_t total = 0;
        for (const Piece &pc : pieces) total += pc.count;
        if (total * sizeof(double) > d.qstage_bytes) {
            if (d.qstage) (void)hipHostFree(d.qstage);
            d.qstage = nullptr;
            d.qstage_bytes = 0;
            AMIS_HIP(hipHostMalloc(&d.qstage, total * sizeof(double) * 2, hipHostMallocDefault));
            d.qstage_bytes = total * sizeof(double) * 2;
        }
        double *q = (double *)d.qstage;
        for (const Piece &pc : pieces) {
            if (!pc.count) continue;
            std::memcpy(q, pc.src, pc.count * sizeof(double));
            AMIS_HIP(hipMemcpyAsync(pc.dst, q, pc.count * sizeof(double), hipMemcpyHostToDevice, stream));
            q += pc.count;
        }
        d.Q = Q;
    } else if (qcnt) {
        AMIS_HIP(hipMemcpy(d.a + qlo * k1, m.a_flat.data() + qlo * k1, qcnt * k1 * sizeof(double), hipMemcpyHostToDevice));
        AMIS_HIP(hipMemcpy(d.dir_norm + qlo, m.dir_norm.data() + qlo, qcnt * sizeof(double), hipMemcpyHostToDevice));
        AMIS_HIP(hipMemcpy(d.head + qlo * n, m.head_flat.data() + qlo * n, qcnt * n * sizeof(double), hipMemcpyHostToDevice));
        if (k) AMIS_HIP(hipMemcpy(d.pair + qlo * k * n * n, m.pair_flat.data() + qlo * k * n * n, qcnt * k * n * n * sizeof(double), hipMemcpyHostToDevice));
        d.Q = Q;
    }
    return BILD_OK;
}

// the per-sample results of the device passes, back into the host arrays (exports, a later host-side step)
// samples that fused steps left in HBM only: the host's copy of the static per-sample data catches up
int host_catch_up(bild_amis &m)
{
    bild_amis::Dev &d = m.dev;
    if (!d.on || d.P <= m.P()) return BILD_OK;
    const size_t k1 = m.k1, k = m.k, lo = (size_t)m.P(), P = (size_t)d.P, cnt = P - lo;
    m.ss.resize(P * k1);
    m.log_ss.resize(P * k1);
    m.has_zero.resize(P);
    m.first.resize(P);
    m.pcode.resize(P * k);
    m.theta.resize(P * k1);
    m.logL.resize(P);
    m.logd.resize(P);
    m.cur.resize(P);
    m.logw.resize(P);
    AMIS_HIP(hipDeviceSynchronize());
    AMIS_HIP(hipMemcpy(m.ss.data() + lo * k1, d.ss + lo * k1, cnt * k1 * sizeof(double), hipMemcpyDeviceToHost));
    AMIS_HIP(hipMemcpy(m.log_ss.data() + lo * k1, d.log_ss + lo * k1, cnt * k1 * sizeof(double), hipMemcpyDeviceToHost));
    AMIS_HIP(hipMemcpy(m.has_zero.data() + lo, d.has_zero + lo, cnt, hipMemcpyDeviceToHost));
    AMIS_HIP(hipMemcpy(m.first.data() + lo, d.first + lo, cnt * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (k) AMIS_HIP(hipMemcpy(m.pcode.data() + lo * k, d.pcode + lo * k, cnt * k * sizeof(int32_t), hipMemcpyDeviceToHost));
    AMIS_HIP(hipMemcpy(m.theta.data() + lo * k1, d.theta + lo * k1, cnt * k1 * sizeof(int32_t), hipMemcpyDeviceToHost));
    AMIS_HIP(hipMemcpy(m.logL.data() + lo, d.logL + lo, cnt * sizeof(double), hipMemcpyDeviceToHost));
    m.log_ss_valid = (int64_t)P;
    d.host_stale = true; // logd / cur / logw of these samples are on the device only
    return BILD_OK;
}

int dev_pull(const bild_amis &mc)
{
    bild_amis &m = const_cast<bild_amis &>(mc);
    bild_amis::Dev &d = m.dev;
    if (int rc = host_catch_up(m)) return rc;
    if (!d.on || !d.host_stale) return BILD_OK;
    const size_t P = (size_t)std::min<int64_t>(d.P, m.P());
    if (P) {
        AMIS_HIP(hipMemcpy(m.logd.data(), d.logd, P * sizeof(double), hipMemcpyDeviceToHost));
        AMIS_HIP(hipMemcpy(m.cur.data(), d.cur, P * sizeof(double), hipMemcpyDeviceToHost));
        AMIS_HIP(hipMemcpy(m.logw.data(), d.logw, P * sizeof(double), hipMemcpyDeviceToHost));
    }
    d.host_stale = false;
    return BILD_OK;
}

bild::AmisView dev_view(const bild_amis &m)
{
    bild::AmisView v = m.view();
    const bild_amis::Dev &d = m.dev;
    v.a = d.a;
    v.dir_norm = d.dir_norm;
    v.head = d.head;
    v.pair = d.pair;
    v.ss = d.ss;
    v.log_ss = d.log_ss;
    v.has_zero = d.has_zero;
    v.first = d.first;
    v.pcode = d.pcode;
    v.theta = d.theta;
    v.logL = d.logL;
    return v;
}

// the partial sums of the pass just launched on `st`, through pinned memory; waits for the stream
int dev_partials(bild_amis &m, int64_t doubles, std::vector<double> &host, hipStream_t st)
{
    // (the passes write their partial sums straight into pinned host memory -- a few tens of KB of posted writes per pass --:
    // no copy command to enqueue and wait for between a pass and the host's sums)
    bild_amis::Dev &d = m.dev;
    AMIS_HIP(hipStreamSynchronize(st));
    host.assign(d.partial, d.partial + doubles);
    return BILD_OK;
}

} // namespace

bild_amis::~bild_amis() { dev_release(dev); }

extern "C" {

int bild_amis_create(int k1, int n, const uint8_t *transitions, double brake_c, double brake_p, double logprior,
                     const double *a0, const double *logp0, bild_amis **out)
{
    if (!out || k1 < 1 || n < 1 || !transitions || !a0 || !logp0) return BILD_ERR_INVALID;
    bild_amis *m = new bild_amis;
    m->k1 = k1;
    m->k = k1 - 1;
    m->n = n;
    m->brake_c = brake_c;
    m->brake_p = brake_p;
    m->logprior = logprior;
    m->trans.assign(transitions, transitions + (size_t)n * n);
    m->a.emplace_back(a0, a0 + k1);
    m->logp.emplace_back(logp0, logp0 + (size_t)n * k1);
    m->derive(0);
    *out = m;
    return BILD_OK;
}

int bild_amis_destroy(bild_amis *m)
{
    delete m;
    return BILD_OK;
}

const char *bild_amis_error(const bild_amis *m) { return m ? m->err.c_str() : ""; }

int64_t bild_amis_pool_size(const bild_amis *m) { return m ? m->total() : 0; }
int64_t bild_amis_num_proposals(const bild_amis *m) { return m ? (int64_t)m->a.size() : 0; }

int bild_amis_params(const bild_amis *m, int64_t which, double *a, double *logp)
{
    if (!m) return BILD_ERR_INVALID;
    const int64_t Q = (int64_t)m->a.size();
    if (which < 0) which += Q;
    if (which < 0 || which >= Q) return BILD_ERR_INVALID;
    if (a) std::copy(m->a[which].begin(), m->a[which].end(), a);
    if (logp) std::copy(m->logp[which].begin(), m->logp[which].end(), logp);
    return BILD_OK;
}

int bild_amis_pool(const bild_amis *m, int what, double *out)
{
    if (!m || !out) return BILD_ERR_INVALID;
    if (int rc = dev_pull(*m)) return rc;
    const std::vector<double> *src = what == 0 ? &m->logL : what == 1 ? &m->logd : what == 2 ? &m->cur : what == 3 ? &m->logw : nullptr;
    if (!src) return BILD_ERR_INVALID;
    std::copy(src->begin(), src->end(), out);
    return BILD_OK;
}

// Rebuild a sampler from saved state (pickling / copying on the Python side): `Q_extra` further proposals after
// the initial one, and the pooled samples with the per-sample arrays as bild_amis_pool returned them.
int bild_amis_restore(bild_amis *m, int64_t Q_extra, const double *a, const double *logp, int64_t P, const double *ss,
                      const int64_t *thetas, const double *logLs, const double *logd, const double *cur, const double *logw)
{
    if (!m || m->total() != 0 || m->a.size() != 1 || Q_extra < 0 || P < 0) return BILD_ERR_INVALID;
    if ((Q_extra && (!a || !logp)) || (P && (!ss || !thetas || !logLs || !logd || !cur || !logw))) return BILD_ERR_INVALID;
    const int k1 = m->k1, k = m->k, n = m->n;
    for (int64_t q = 0; q < Q_extra; ++q) {
        m->a.emplace_back(a + (size_t)q * k1, a + (size_t)(q + 1) * k1);
        m->logp.emplace_back(logp + (size_t)q * n * k1, logp + (size_t)(q + 1) * n * k1);
        m->derive(m->a.size() - 1);
    }
    m->ss.assign(ss, ss + (size_t)P * k1);
    m->log_ss.resize((size_t)P * k1);
    m->has_zero.resize(P);
    m->first.resize(P);
    m->pcode.resize((size_t)P * k);
    m->theta.resize((size_t)P * k1);
    for (int64_t p = 0; p < P; ++p) {
        bool z = false;
        for (int j = 0; j < k1; ++j) {
            const double v = ss[(size_t)p * k1 + j];
            z |= v == 0;
            m->log_ss[(size_t)p * k1 + j] = v == 0 ? 0.0 : std::log(v);
            const int64_t th = thetas[(size_t)p * k1 + j];
            if (th < 0 || th >= n) return BILD_ERR_INVALID;
            m->theta[(size_t)p * k1 + j] = (int32_t)th;
        }
        m->has_zero[p] = z;
        m->first[p] = m->theta[(size_t)p * k1];
        for (int i = 0; i < k; ++i)
            m->pcode[(size_t)p * k + i] = (int32_t)((i * n + m->theta[(size_t)p * k1 + i]) * n + m->theta[(size_t)p * k1 + i + 1]);
    }
    m->log_ss_valid = P;
    m->logL.assign(logLs, logLs + P);
    m->logd.assign(logd, logd + P);
    m->cur.assign(cur, cur + P);
    m->logw.assign(logw, logw + P);
    return BILD_OK;
}

// State traces from the current proposal (amis.py:223-256): the caller supplies the uniform random numbers it
// drew from the NumPy stream in the reference's order -- u[0..N) for the first slot (what np.random.choice
// consumes), then one block of N per later slot (np.random.rand(N, 1)).
int bild_amis_sample_traces(const bild_amis *m, int64_t N, const double *u, int64_t *thetas)
{
    if (!m || !u || !thetas || N < 0) return BILD_ERR_INVALID;
    const int n = m->n, k1 = m->k1;
    const std::vector<double> &L = m->logp.back();
    std::vector<double> p((size_t)n * k1), cdf(n);
    for (int i = 0; i < k1; ++i) {
        const double norm = lse(n, L.data() + i, k1, [](int) { return true; });
        for (int s = 0; s < n; ++s) p[(size_t)s * k1 + i] = std::exp(L[(size_t)s * k1 + i] - norm);
    }
    {   // np.random.choice(n, p=p0): cdf = cumsum(p0) / last; index = searchsorted(cdf, u, side='right')
        double c = 0;
        for (int s = 0; s < n; ++s) cdf[s] = (c += p[(size_t)s * k1]);
        for (int s = 0; s < n; ++s) cdf[s] /= cdf[n - 1];
        for (int64_t r = 0; r < N; ++r) {
            int idx = 0;
            while (idx < n && cdf[idx] <= u[r]) ++idx;
            thetas[(size_t)r * k1] = std::min(idx, n - 1);
        }
    }
    for (int i = 1; i < k1; ++i) {
        const double *ui = u + (size_t)i * N;
        for (int64_t r = 0; r < N; ++r) {
            const int prev = (int)thetas[(size_t)r * k1 + i - 1];
            double c = 0;
            for (int s = 0; s < n; ++s) cdf[s] = (c += p[(size_t)s * k1 + i] * (m->trans[(size_t)prev * n + s] ? 1.0 : 0.0));
            const double last = cdf[n - 1];
            int idx = 0; // np.argmax(cdf / last > r): first crossing, 0 if there is none
            for (int s = 0; s < n; ++s)
                if (cdf[s] / last > ui[r]) {
                    idx = s;
                    break;
                }
            thetas[(size_t)r * k1 + i] = idx;
        }
    }
    return BILD_OK;
}

// Keep the pooled samples in HBM and run the three passes of a step there (amis_device.hip).  Worth it for large
// batches (the passes are 3.9 of the 5.2 ms of a step at N = 10 000 on the host, ~0.3 ms on the device); for the
// reference's default N = 100 the host passes take microseconds and stay the default.  enable = 0: back to the host
// (state is pulled first).  BILD_ERR_UNSUPPORTED for n * k1 > 64; BILD_ERR_NO_DEVICE without a GPU.
int bild_amis_use_device(bild_amis *m, int enable)
{
    if (!m) return BILD_ERR_INVALID;
    if (!enable) {
        if (!m->dev.on) return BILD_OK;
        if (int rc = dev_pull(*m)) return rc;
        const int64_t P = m->P();
        if (m->log_ss_valid < P) { // the logs the device took
            const size_t lo = (size_t)m->log_ss_valid * m->k1, cnt = (size_t)(P - m->log_ss_valid) * m->k1;
            if (hipMemcpy(m->log_ss.data() + lo, m->dev.log_ss + lo, cnt * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
                m->err = "device bookkeeping: copying the pool back failed";
                return BILD_ERR_HIP;
            }
            m->log_ss_valid = P;
        }
        dev_release(m->dev);
        return BILD_OK;
    }
    if (m->dev.on) return BILD_OK;
    if (m->n * m->k1 > bild::kAmisMaxNm) {
        m->err = "device bookkeeping: n_states * (k + 1) > 64";
        return BILD_ERR_UNSUPPORTED;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
        (void)hipGetLastError();
        m->err = "device bookkeeping: no GPU";
        return BILD_ERR_NO_DEVICE;
    }
    m->dev.on = true;
    if (int rc = dev_push(*m, true)) {
        dev_release(m->dev);
        return rc;
    }
    return BILD_OK;
}

// One AMIS iteration after the likelihood of the new batch is known (amis.py:819-906).
// evidence[3] = (logev, dlogev, KL).  Returns BILD_ERR_INVALID with "Iteration did not converge" in
// bild_amis_error when the CFC fit does not converge (the reference raises RuntimeError there).
// `model` given: the FUSED step -- the new samples go up once, as the (s, theta) rows the likelihood kernels read
// (walk.hip), into the pool where the passes below find them; their likelihood is written into the pool by the kernels;
// what the host loop of the plain step derives per sample (has_zero, first, pcode) is derived by pass A on the device; and
// everything runs on the model's stream, so the only synchronisations are the three copies of partial sums.  The host's
// copy of the pool catches up when somebody asks for it (host_catch_up).
static int amis_step_impl(bild_amis *m, int64_t N, const double *ss, const int64_t *thetas, const double *logLs, double *evidence,
                          const bild_model *model, const bild_trajset *ts, unsigned flags, const uint64_t *rng_seed = nullptr)
{
    const bool fused = model != nullptr;
    const bool draw = fused && rng_seed != nullptr; // the samples are drawn on the device (amis_device.hip: draw_kernel)
    if (!m || (!draw && (!ss || !thetas)) || (!fused && !logLs) || !evidence || N < 1) return BILD_ERR_INVALID;
    if (fused && (!m->dev.on || !ts)) {
        m->err = "fused step: the pooled samples must be on the device (bild_amis_use_device)";
        return BILD_ERR_INVALID;
    }
    if (!fused)
        if (int rc = host_catch_up(*m)) return rc; // (earlier fused steps)
    const int k1 = m->k1, k = m->k, n = m->n;
    const size_t Q = m->a.size(); // proposals used so far, the current one last
    const int64_t P0 = m->total();
    hipStream_t st = nullptr;
    if (!draw) {
        uint64_t bad = 0; // (one branch-free pass: the states are checked again where they are narrowed / used)
        for (size_t i = 0, e = (size_t)N * k1; i < e; ++i) bad |= (uint64_t)((uint64_t)thetas[i] >= (uint64_t)n);
        if (bad) {
            m->err = "state index out of range";
            return BILD_ERR_INVALID;
        }
    }

    // BILD_AMIS_TRACE=1: where a step spends its time (microseconds), on stderr
    const bool trace = bild::config().amis_trace;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "%s %.0f  ", what, std::chrono::duration<double, std::micro>(now - t_last).count());
        t_last = now;
    };
    // 2. the new samples and their own denominators: all proposals used so far
    const int64_t P = P0 + N;
    if (fused) {
        bild_amis::Dev &d = m->dev;
        int rc;
        // everything of this step goes to the model's own stream: the copies, the likelihood, the passes
        void *stream = bild::internal_model_stream(model);
        if (!stream) {
            m->err = bild_last_error();
            return BILD_ERR_HIP;
        }
        st = (hipStream_t)stream;
        if ((rc = dev_push(*m, false, st, true))) return rc; // proposals not yet mirrored (and room for them)
        if ((rc = dev_reserve(*m, P, (int64_t)Q))) return rc;
        const size_t nseg = (size_t)N * k1;
        const size_t par_bytes = ((size_t)(k1 + n * k1) * sizeof(double) + (size_t)n * n + 15) & ~(size_t)15;
        const size_t need = (draw ? par_bytes : nseg * sizeof(double) + ((nseg + 15) & ~(size_t)15)) + 64;
        if (need > d.stage_bytes) {
            if (d.stage) (void)hipHostFree(d.stage);
            d.stage = nullptr;
            d.stage_bytes = 0;
            if (hipHostMalloc(&d.stage, need * 2, hipHostMallocDefault) != hipSuccess) {
                m->err = "fused step: out of pinned memory";
                return BILD_ERR_NOMEM;
            }
            d.stage_bytes = need * 2;
        }
        uint8_t *t8 = (uint8_t *)d.stage + nseg * sizeof(double);
        int32_t *status = draw ? (int32_t *)((char *)d.stage + par_bytes) : (int32_t *)(t8 + ((nseg + 15) & ~(size_t)15));
        status[0] = status[1] = 0;
        if (draw) {
            // parameters of the current proposal for the draw kernel: concentrations, slot weights as probabilities
            // (bild_amis_sample_traces: exp(logp - logsumexp over the states of a slot)), allowed transitions
            double *par = (double *)d.stage;
            const std::vector<double> &A = m->a.back(), &L = m->logp.back();
            for (int j = 0; j < k1; ++j) par[j] = A[j];
            for (int i = 0; i < k1; ++i) {
                const double norm = lse(n, L.data() + i, k1, [](int) { return true; });
                for (int s2 = 0; s2 < n; ++s2) par[k1 + (size_t)s2 * k1 + i] = std::exp(L[(size_t)s2 * k1 + i] - norm);
            }
            std::memcpy(par + k1 + (size_t)n * k1, m->trans.data(), (size_t)n * n);
            if (!d.draw_par && hipMalloc((void **)&d.draw_par, par_bytes) != hipSuccess) {
                m->err = "fused step: out of device memory";
                return BILD_ERR_NOMEM;
            }
        } else {
            std::memcpy(d.stage, ss, nseg * sizeof(double));
            for (size_t i = 0; i < nseg; ++i) t8[i] = (uint8_t)thetas[i]; // (range checked above)
        }
        lap("[amis step fused] stage");
        if (draw) {
            const double *dp = d.draw_par;
            if (hipMemcpyAsync(d.draw_par, d.stage, par_bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
                bild::amis_dev_draw(k1, n, N, *rng_seed, (uint64_t)P0, dp, dp + k1, (const uint8_t *)(dp + k1 + (size_t)n * k1),
                                    d.ss + (size_t)P0 * k1, d.theta8 + (size_t)P0 * k1, (void *)st)) {
                m->err = "fused step: drawing the samples failed";
                return BILD_ERR_HIP;
            }
        } else if (hipMemcpyAsync(d.ss + (size_t)P0 * k1, d.stage, nseg * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
                   hipMemcpyAsync(d.theta8 + (size_t)P0 * k1, t8, nseg, hipMemcpyHostToDevice, st) != hipSuccess) {
            m->err = "fused step: upload failed";
            return BILD_ERR_HIP;
        }
        rc = bild::internal_logl_st_resident(model, ts, N, k1, d.ss + (size_t)P0 * k1, d.theta8 + (size_t)P0 * k1, flags, d.logL + P0,
                                             status, &stream);
        if (rc) {
            (void)hipStreamSynchronize(st);
            m->err = bild_last_error();
            return rc;
        }
        lap("likelihood enqueued");
    } else {
    m->ss.insert(m->ss.end(), ss, ss + (size_t)N * k1);
    m->log_ss.resize((size_t)P * k1);
    m->has_zero.resize(P);
    m->first.resize(P);
    m->pcode.resize((size_t)P * k);
    m->theta.resize((size_t)P * k1);
    m->logL.insert(m->logL.end(), logLs, logLs + N);
    m->logd.resize(P);
    m->cur.resize(P);
    m->logw.resize(P);
    for (int64_t r = 0; r < N; ++r) {
        const int64_t p = P0 + r;
        bool z = false;
        for (int j = 0; j < k1; ++j) {
            const double v = ss[(size_t)r * k1 + j];
            z |= v == 0;
            if (!m->dev.on) m->log_ss[(size_t)p * k1 + j] = v == 0 ? 0.0 : std::log(v); // (device mirror: pass A takes the logs)
        }
        m->has_zero[p] = z;
        m->first[p] = (int32_t)thetas[(size_t)r * k1];
        for (int i = 0; i < k1; ++i) m->theta[(size_t)p * k1 + i] = (int32_t)thetas[(size_t)r * k1 + i];
        for (int i = 0; i < k; ++i)
            m->pcode[(size_t)p * k + i] = (int32_t)((i * n + thetas[(size_t)r * k1 + i]) * n + thetas[(size_t)r * k1 + i + 1]);
    }
    if (!m->dev.on) m->log_ss_valid = P;
    lap("[amis step] append");
    }
    const int64_t nchunks = (P + kChunk - 1) / kChunk;
    const double logQ = std::log((double)Q);
    const int nm = n * k1;
    const double tiny = std::numeric_limits<double>::min();
    double top = -kInf, W = 0, sum = 0, ev = 0, sq = 0, kl = 0;
    std::vector<double> mean(k1, 0.0), marg(nm, 0.0), var(k1, 0.0);
    if (m->dev.on) {
        // ---- the three passes on the device (amis_device.hip), partial sums added here in block order ----------------
        bild_amis &mm = *m;
        int rc;
        if (!fused) {
            if ((rc = dev_push(mm, false))) return rc;
            lap("upload");
        }
        const int blocks = (int)((P + (int64_t)bild::kAmisBlock * bild::kAmisPerLane - 1) / ((int64_t)bild::kAmisBlock * bild::kAmisPerLane));
        const int rows_a = bild::amis_dev_pass_a_rows(P0, P);
        const int64_t need = std::max<int64_t>((int64_t)blocks * (2 + k1 + nm), (int64_t)rows_a * 2);
        if (need > m->dev.partial_cap) {
            if (m->dev.partial) (void)hipHostFree(m->dev.partial);
            m->dev.partial = nullptr;
            if (hipHostMalloc((void **)&m->dev.partial, (size_t)need * 2 * sizeof(double), hipHostMallocDefault) != hipSuccess) {
                m->err = "device bookkeeping: out of memory";
                return BILD_ERR_HIP;
            }
            m->dev.partial_cap = need * 2;
        }
        const bild::AmisView dv = dev_view(*m);
        std::vector<double> part;
        int rows = 0;
        bild_amis::Dev &d = m->dev;
        if ((int64_t)Q * N > d.lq_cap) { // (scratch only: nothing to keep across a regrowth; without it the pass computes twice)
            if (d.lq_keep) (void)hipFree(d.lq_keep);
            d.lq_keep = nullptr;
            d.lq_cap = 0;
            if (hipMalloc((void **)&d.lq_keep, (size_t)Q * N * 2 * sizeof(double)) == hipSuccess) d.lq_cap = (int64_t)Q * N * 2;
            else (void)hipGetLastError();
        }
        if (bild::amis_dev_pass_a(dv, (int64_t)Q, P0, P, logQ, d.log_ss, d.cur, d.logd, d.logw, d.partial, &rows, (void *)st,
                                  fused ? d.theta8 : nullptr, d.has_zero, d.first, d.pcode, d.theta, d.lq_keep)) {
            m->err = "device bookkeeping: pass A failed";
            return BILD_ERR_HIP;
        }
        m->dev.host_stale = true;
        if ((rc = dev_partials(mm, (int64_t)rows * 2, part, st))) return rc;
        if (fused) {
            // the stream has been waited for: the likelihood's verdict on the rows is in, and the samples are pooled
            const size_t par_bytes2 = ((size_t)(k1 + n * k1) * sizeof(double) + (size_t)n * n + 15) & ~(size_t)15;
            const int32_t *status = draw ? (const int32_t *)((char *)d.stage + par_bytes2)
                                         : (const int32_t *)((uint8_t *)d.stage + (size_t)N * k1 * sizeof(double) + (((size_t)N * k1 + 15) & ~(size_t)15));
            if (status[0] != 0) {
                m->err = "interval lengths of a sample are not non-negative finite numbers (of a point on the simplex)";
                return BILD_ERR_INVALID;
            }
            d.P = P;
        }
        bool nan_w = false;
        for (int b = 0; b < rows; ++b) {
            top = std::max(top, part[(size_t)2 * b]);
            nan_w |= part[(size_t)2 * b + 1] != 0;
        }
        if (nan_w) top = std::numeric_limits<double>::quiet_NaN(); // as np.max
        const bool top_finite = std::isfinite(top);
        lap("A");
        if (bild::amis_dev_pass_b(dv, P, top, top_finite ? 1 : 0, m->dev.logw, m->dev.rel, m->dev.partial, blocks, (void *)st)) {
            m->err = "device bookkeeping: pass B failed";
            return BILD_ERR_HIP;
        }
        const int wb = 2 + k1 + nm;
        if ((rc = dev_partials(mm, (int64_t)blocks * wb, part, st))) return rc;
        for (int b = 0; b < blocks; ++b) {
            const double *row = part.data() + (size_t)b * wb;
            W += row[0];
            sum += row[1];
            for (int j = 0; j < k1; ++j) mean[j] += row[2 + j];
            for (int i = 0; i < nm; ++i) marg[i] += row[2 + k1 + i];
        }
        for (int j = 0; j < k1; ++j) mean[j] /= W;
        ev = sum / (double)P;
        lap("B");
        if (hipMemcpyAsync(m->dev.mean, mean.data(), (size_t)k1 * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
            bild::amis_dev_pass_c(dv, P, m->dev.mean, ev, m->dev.rel, m->dev.cur, m->dev.partial, blocks, (void *)st)) {
            m->err = "device bookkeeping: pass C failed";
            return BILD_ERR_HIP;
        }
        const int wc = k1 + 2;
        if ((rc = dev_partials(mm, (int64_t)blocks * wc, part, st))) return rc;
        for (int b = 0; b < blocks; ++b) {
            const double *row = part.data() + (size_t)b * wc;
            for (int j = 0; j < k1; ++j) var[j] += row[j];
            sq += row[k1];
            kl += row[k1 + 1];
        }
    } else {
        const bild::AmisView hv = m->view();
        // pass A: 1. the mixture denominator of every earlier sample gains the current proposal; 2. the new samples'
        // own denominators (all proposals used so far); 3. deterministic-mixture weights L / mean over proposals
        std::vector<double> ctop(nchunks, -kInf);
        std::vector<uint8_t> cnan(nchunks, 0);
        for_chunks(P, [&](int64_t c, int64_t lo, int64_t hi) {
            std::vector<double> lq(Q);
            double top_c = -kInf;
            bool nan_c = false;
            for (int64_t p = lo; p < hi; ++p) {
                if (p < P0) {
                    const double cq = bild::amis_log_q(hv, (int64_t)Q - 1, p);
                    m->cur[p] = cq;
                    m->logd[p] = logaddexp(m->logd[p], cq);
                } else {
                    for (size_t q = 0; q < Q; ++q) lq[q] = bild::amis_log_q(hv, (int64_t)q, p);
                    m->cur[p] = lq[Q - 1];
                    m->logd[p] = lse((int)Q, lq.data(), 1, [](int) { return true; });
                }
                const double lw = m->logL[p] - m->logd[p] + logQ;
                m->logw[p] = lw;
                nan_c |= std::isnan(lw);
                top_c = std::max(top_c, lw);
            }
            ctop[c] = top_c;
            cnan[c] = nan_c;
        });
        bool nan_w = false;
        for (int64_t c = 0; c < nchunks; ++c) {
            top = std::max(top, ctop[c]);
            nan_w |= cnan[c] != 0;
        }
        if (nan_w) top = std::numeric_limits<double>::quiet_NaN(); // as np.max

        // ---- refit -------------------------------------------------------------------------------
        // pass B: relative weights; first moments of the Dirichlet fit (weights below 1e-100 of the largest are
        // dropped there), slot marginals of the CFC fit, sum of the weights for the evidence
        std::vector<double> rel(P), cW(nchunks, 0.0), cS(nchunks, 0.0), cacc((size_t)nchunks * k1, 0.0), cmarg((size_t)nchunks * nm, 0.0);
        const bool top_finite = std::isfinite(top);
        for_chunks(P, [&](int64_t c, int64_t lo, int64_t hi) {
            double *acc = cacc.data() + (size_t)c * k1, *mg = cmarg.data() + (size_t)c * nm;
            double W = 0, S = 0;
            for (int64_t p = lo; p < hi; ++p) {
                const double dlt = m->logw[p] - top;
                const double w = dlt < -746.0 ? 0.0 : std::exp(dlt); // exp underflows to exactly 0 below -745.2
                rel[p] = w;
                if (w >= 1e-100) {
                    W += w;
                    const double *sp = m->ss.data() + (size_t)p * k1;
                    for (int j = 0; j < k1; ++j) acc[j] += w * sp[j];
                }
                if (top_finite && w != 0) {
                    const int32_t *th = m->theta.data() + (size_t)p * k1;
                    for (int i = 0; i < k1; ++i) mg[(size_t)th[i] * k1 + i] += w;
                }
                if (w >= tiny) S += w; // subnormal weights: no effect on the sums
            }
            cW[c] = W;
            cS[c] = S;
        });
        for (int64_t c = 0; c < nchunks; ++c) {
            W += cW[c];
            sum += cS[c];
            for (int j = 0; j < k1; ++j) mean[j] += cacc[(size_t)c * k1 + j];
            for (int i = 0; i < nm; ++i) marg[i] += cmarg[(size_t)c * nm + i];
        }
        for (int j = 0; j < k1; ++j) mean[j] /= W;
        ev = sum / (double)P;
        // pass C: second moments of the Dirichlet fit; spread of the weights and the KL sum for the evidence block
        std::vector<double> cvar((size_t)nchunks * k1, 0.0), csq(nchunks, 0.0), ckl(nchunks, 0.0);
        for_chunks(P, [&](int64_t c, int64_t lo, int64_t hi) {
            double *acc = cvar.data() + (size_t)c * k1;
            double sq = 0, kl = 0;
            for (int64_t p = lo; p < hi; ++p) {
                const double w = rel[p];
                if (w >= 1e-100) {
                    const double *sp = m->ss.data() + (size_t)p * k1;
                    for (int j = 0; j < k1; ++j) {
                        const double dv = sp[j] - mean[j];
                        acc[j] += w * dv * dv;
                    }
                }
                const double we = w < tiny ? 0.0 : w;
                const double dv = we - ev;
                sq += dv * dv;
                const double term = we * (m->logL[p] - m->cur[p]);
                if (!std::isnan(term)) kl += term; // zero-weight samples the current proposal cannot produce: dropped
            }
            csq[c] = sq;
            ckl[c] = kl;
        });
        for (int64_t c = 0; c < nchunks; ++c) {
            for (int j = 0; j < k1; ++j) var[j] += cvar[(size_t)c * k1 + j];
            sq += csq[c];
            kl += ckl[c];
        }
    }
    lap("passes (C)");
    std::vector<double> new_a(k1);
    {
        bool degenerate = false;
        double tot = 0;
        for (int j = 0; j < k1; ++j) {
            var[j] /= W;
            degenerate |= var[j] == 0;
        }
        if (degenerate) {
            tot = 1e10; // very concentrated but finite: the brake takes over
        } else {
            for (int j = 0; j < k1; ++j) tot += mean[j] * (1 - mean[j]) / var[j];
            tot = tot / k1 - 1;
        }
        // weight on the boundary of the simplex (m_j = 1 up to rounding, or 0): negative / zero / non-finite
        // concentration, with which the reference ends in "alpha <= 0"; treated like the degenerate case
        if (!(std::isfinite(tot) && tot > 0)) tot = 1e10;
        for (int j = 0; j < k1; ++j) new_a[j] = std::max(tot * mean[j], std::numeric_limits<double>::min());
    }
    // CFC: weighted slot marginals -> weights
    std::vector<double> new_logp((size_t)n * k1);
    {
        std::vector<double> lm((size_t)n * k1);
        for (size_t i = 0; i < lm.size(); ++i) lm[i] = std::log((double)marg[i]) + top;
        for (int i = 0; i < k1; ++i) {
            const double norm = lse(n, lm.data() + i, k1, [](int) { return true; });
            for (int s = 0; s < n; ++s) lm[(size_t)s * k1 + i] -= norm;
        }
        std::vector<double> f(n), g(n), o(n);
        for (int s = 0; s < n; ++s) new_logp[(size_t)s * k1] = lm[(size_t)s * k1];
        for (int i = 1; i < k1; ++i) {
            for (int s = 0; s < n; ++s) {
                f[s] = lm[(size_t)s * k1 + i];
                g[s] = lm[(size_t)s * k1 + i - 1];
            }
            if (solve_marginals_single(*m, f.data(), g.data(), o.data())) {
                m->err = "Iteration did not converge";
                // leave the sampler as the reference's would be after the exception: samples appended, no new proposal
                return BILD_ERR_INVALID;
            }
            for (int s = 0; s < n; ++s) new_logp[(size_t)s * k1 + i] = o[s];
        }
    }
    // ---- brakes (amis.py:856-874) ----------------------------------------------------------------
    {
        const std::vector<double> &a_cur = m->a.back(), &logp_cur = m->logp.back();
        const double limit_c = (double)N * m->brake_c;
        double sn = 0, sc = 0;
        for (int j = 0; j < k1; ++j) {
            sn += new_a[j];
            sc += a_cur[j];
        }
        const double log_ratio = std::log(sn / sc);
        if (std::fabs(log_ratio) > limit_c) {
            const double sgn = log_ratio > 0 ? 1.0 : (log_ratio < 0 ? -1.0 : 0.0);
            const double f = std::exp(sgn * limit_c - log_ratio);
            for (int j = 0; j < k1; ++j) new_a[j] *= f;
        }
        const double limit_p = (double)N * m->brake_p;
        for (int i = 0; i < k1; ++i) {
            double biggest = 0;
            bool nan_d = false;
            std::vector<double> delta(n), pold(n);
            for (int s = 0; s < n; ++s) {
                pold[s] = std::exp(logp_cur[(size_t)s * k1 + i]);
                delta[s] = std::exp(new_logp[(size_t)s * k1 + i]) - pold[s];
                nan_d |= std::isnan(delta[s]);
                biggest = std::max(biggest, std::fabs(delta[s]));
            }
            if (!nan_d && biggest > limit_p)
                for (int s = 0; s < n; ++s) new_logp[(size_t)s * k1 + i] = std::log(pold[s] + limit_p * delta[s] / biggest);
        }
    }
    m->a.push_back(new_a);
    m->logp.push_back(new_logp);
    m->derive(m->a.size() - 1);

    // ---- evidence, its standard error, KL(posterior || current proposal) (amis.py:876-903) ---------
    {
        const double logev = std::log(ev) + top + m->logprior;
        const double sd = P > 1 ? std::sqrt(sq / (double)(P - 1)) : std::numeric_limits<double>::quiet_NaN();
        evidence[0] = logev;
        evidence[1] = sd / std::sqrt((double)P) / ev;
        evidence[2] = kl / (double)P / ev - logev + m->logprior;
    }
    lap("refit");
    if (trace) fprintf(stderr, "us  (pool %lld)\n", (long long)P);
    m->steps += 1;
    m->err.clear();
    return BILD_OK;
}

int bild_amis_step(bild_amis *m, int64_t N, const double *ss, const int64_t *thetas, const double *logLs, double *evidence)
{
    return amis_step_impl(m, N, ss, thetas, logLs, evidence, nullptr, nullptr, 0);
}

int bild_amis_step_fused(bild_amis *m, const bild_model *model, const bild_trajset *ts, int64_t N, const double *ss,
                         const int64_t *thetas, unsigned flags, double *evidence)
{
    if (!model || !ts) return BILD_ERR_INVALID;
    return amis_step_impl(m, N, ss, thetas, nullptr, evidence, model, ts, flags);
}

int bild_amis_step_device_rng(bild_amis *m, const bild_model *model, const bild_trajset *ts, int64_t N, uint64_t seed, unsigned flags,
                              double *evidence)
{
    if (!model || !ts) return BILD_ERR_INVALID;
    return amis_step_impl(m, N, nullptr, nullptr, nullptr, evidence, model, ts, flags, &seed);
}

// the pooled samples themselves (fused steps and device-side draws leave them in HBM until somebody asks):
// ss (P x k1), thetas (P x k1 int64); either may be NULL
int bild_amis_pool_samples(const bild_amis *m, double *ss, int64_t *thetas)
{
    if (!m) return BILD_ERR_INVALID;
    if (int rc = host_catch_up(const_cast<bild_amis &>(*m))) return rc;
    if (ss) std::copy(m->ss.begin(), m->ss.end(), ss);
    if (thetas)
        for (size_t i = 0; i < m->theta.size(); ++i) thetas[i] = m->theta[i];
    return BILD_OK;
}

// Choice of the next k in the adaptive-k loop (reference bild/choicesampler.py:115-210): for every row of the
// common random sample the "best k" is the smallest k whose perturbed evidence x_k = rvs_k + mu_k lies within dE
// of the row maximum (NaN entries -- omitted k -- are ignored, as np.nanmax / np.nanargmax do).  One pass gives
//   n0[k]        histogram of the best k,
//   dn[kc][k]    histogram with mu[kc] + dmu[kc]/2 minus histogram with mu[kc] - dmu[kc]/2   (ChoiceSampler.Dn),
//   n_omit[k]    histogram with the entries flagged in `omit` ignored                            (KLD_omitK),
// instead of 2 kmax + 2 passes of array operations.  Any output may be NULL.
int bild_choice_counts(int64_t samplesize, int kmax, const double *rvs, const double *mu, const double *dmu, double dE,
                       const uint8_t *omit, int64_t *n0, int64_t *dn, int64_t *n_omit)
{
    if (samplesize < 0 || kmax < 1 || !rvs || !mu) return BILD_ERR_INVALID;
    if (dn && !dmu) return BILD_ERR_INVALID;
    if (n0) std::fill(n0, n0 + kmax, 0);
    if (dn) std::fill(dn, dn + (size_t)kmax * kmax, 0);
    if (n_omit) std::fill(n_omit, n_omit + kmax, 0);
    std::vector<double> x(kmax);
    // first k with top - dE - x_k <= 0 over the entries that are not NaN; 0 if there is none (argmax of all-False)
    auto best = [&](const double *v) {
        double top = -kInf;
        bool any = false;
        for (int k = 0; k < kmax; ++k)
            if (!std::isnan(v[k])) {
                top = any ? std::max(top, v[k]) : v[k];
                any = true;
            }
        for (int k = 0; k < kmax; ++k)
            if (top - dE - v[k] <= 0) return k; // false for NaN
        return 0;
    };
    for (int64_t r = 0; r < samplesize; ++r) {
        const double *row = rvs + (size_t)r * kmax;
        for (int k = 0; k < kmax; ++k) x[k] = row[k] + mu[k];
        if (n0) n0[best(x.data())] += 1;
        if (dn)
            for (int kc = 0; kc < kmax; ++kc) {
                const double keep = x[kc];
                x[kc] = row[kc] + (mu[kc] + 0.5 * dmu[kc]);
                dn[(size_t)kc * kmax + best(x.data())] += 1;
                x[kc] = row[kc] + (mu[kc] + -0.5 * dmu[kc]);
                dn[(size_t)kc * kmax + best(x.data())] -= 1;
                x[kc] = keep;
            }
        if (n_omit && omit) {
            for (int k = 0; k < kmax; ++k)
                if (omit[k]) x[k] = std::numeric_limits<double>::quiet_NaN();
            n_omit[best(x.data())] += 1;
        }
    }
    return BILD_OK;
}

// Weighted state occupancy per frame (reference bild/amis.py:945-972 sums, per frame, the weights of the samples
// whose profile is in each state): post[s][t] = sum of w[p] over samples p whose interval covering frame t has
// state s.  Profiles are given as in bild_logl_segments (seg_start / seg_state, P x k1, seg_start[p][0] = 0,
// non-decreasing).  Every interval is added to O(log T) nodes of a range-add tree and a frame's value is the sum
// of the O(log T) nodes above it: only non-negative terms are ever added, so tiny marginals keep their relative
// accuracy (a difference array would not), at O(P k1 log T + n T log T) instead of O(P T) operations.
int bild_interval_marginals(int64_t P, int k1, int n, int64_t T, const int32_t *seg_start, const int32_t *seg_state,
                            const double *w, double *post)
{
    if (P < 0 || k1 < 1 || n < 1 || T < 1 || !post || (P && (!seg_start || !seg_state || !w))) return BILD_ERR_INVALID;
    int64_t T2 = 1;
    while (T2 < T) T2 <<= 1;
    std::vector<double> tree((size_t)n * 2 * T2, 0.0);
    for (int64_t p = 0; p < P; ++p) {
        const double wp = w[p];
        if (!(wp > 0)) continue;
        for (int i = 0; i < k1; ++i) {
            const int st = seg_state[(size_t)p * k1 + i];
            if (st < 0 || st >= n) return BILD_ERR_INVALID;
            int64_t lo = std::min<int64_t>(seg_start[(size_t)p * k1 + i], T);
            int64_t hi = i + 1 < k1 ? std::min<int64_t>(seg_start[(size_t)p * k1 + i + 1], T) : T;
            if (lo < 0 || hi <= lo) continue;
            double *tr = tree.data() + (size_t)st * 2 * T2;
            for (lo += T2, hi += T2; lo < hi; lo >>= 1, hi >>= 1) {
                if (lo & 1) tr[lo++] += wp;
                if (hi & 1) tr[--hi] += wp;
            }
        }
    }
    for (int st = 0; st < n; ++st) {
        const double *tr = tree.data() + (size_t)st * 2 * T2;
        for (int64_t t = 0; t < T; ++t) {
            double acc = 0;
            for (int64_t node = t + T2; node >= 1; node >>= 1) acc += tr[node];
            post[(size_t)st * T + t] = acc;
        }
    }
    return BILD_OK;
}

} // extern "C"
