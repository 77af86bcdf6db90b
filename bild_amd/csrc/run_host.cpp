// The inference driver: the adaptive-k loops of MANY trajectories as one native state machine, one ROUND at a time
// (reference bild/core.py:138-227, bild/amis.py:741-906, bild/choicesampler.py:83-210).
//
// Why: with the likelihood at tens of microseconds per call, BASELINE configs[4] (64 trajectories, default settings: 4 600
// AMIS steps of 100 candidates) spent 99 % of its wall time in the Python that drove it -- 250 us per sampler step.  Here a
// round does, for every live trajectory at once, what one iteration of `core.sample`'s loop does:
//
//   plan      which sampler of each trajectory takes its next AMIS step, which new samplers are opened (their exhaustive
//             enumerations are staged as ordinary candidate rows), which trajectories will rate their evidence curve afterwards;
//             says how many random numbers the round needs: gamma variates (with their shape parameters), uniforms, normals;
//   stage     turns the caller's random numbers into candidate rows: Dirichlet points (normalised gamma variates, exactly as
//             np.random.dirichlet forms them), state traces (bild_amis_sample_traces), all rows of the round in ONE block
//             (padded to the longest list of the round with empty intervals);
//   likelihood  ONE call of bild_logl_st over all rows (traj_id per row) -- or the caller's own evaluation of the rows
//             (bild_run_rows / bild_run_finish: CPU tests, models that are not this library's);
//   finish    per trajectory, on a pool of host threads: the bookkeeping of the step (bild_amis_step), exact evidences of
//             enumerated samplers, the choice sampler (bild_choice_counts) and the stop rules.
//
// Random numbers are NOT drawn here: the caller draws them in bulk, once per round, from the NumPy stream (three vectorised
// calls for all samplers together).  Per trajectory they are consumed in the order the reference consumes them (gamma
// variates of the Dirichlet draw, the uniforms of the traces, the normals of the choice sampler), so a run of ONE trajectory
// walks through exactly the random numbers `core.sample` would, and reproduces it bit for bit (tests/test_run.py).
//
// Composes the C ABI's own building blocks (bild_amis_*, bild_choice_counts, bild_logl_st); plain host C++.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bild_amd.h"
#include "config.h"

namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr double kNaN = std::numeric_limits<double>::quiet_NaN();

// ---- a small persistent pool of host threads --------------------------------------------------------------------------
// Items are handed out one at a time through an atomic counter (a judgement costs ten AMIS steps: static shares would
// leave threads idle); what an item computes never depends on which thread ran it.
class Workers {
  public:
    explicit Workers(int n)
    {
        for (int i = 1; i < n; ++i) threads_.emplace_back([this] { loop(); });
    }
    ~Workers()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
            ++generation_;
        }
        cv_.notify_all();
        for (std::thread &t : threads_) t.join();
    }
    int size() const { return (int)threads_.size() + 1; }
    void run(int64_t n, const std::function<void(int64_t)> &fn)
    {
        if (n <= 0) return;
        if (threads_.empty() || n == 1) {
            for (int64_t i = 0; i < n; ++i) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_ = n;
            next_.store(0);
            busy_ = (int)threads_.size();
            ++generation_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return busy_ == 0; });
        fn_ = nullptr;
    }

  private:
    void work()
    {
        for (;;) {
            const int64_t i = next_.fetch_add(1);
            if (i >= n_) return;
            (*fn_)(i);
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (quit_) return;
            }
            work();
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--busy_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int64_t)> *fn_ = nullptr;
    int64_t n_ = 0;
    std::atomic<int64_t> next_{0};
    int busy_ = 0;
    uint64_t generation_ = 0;
    bool quit_ = false;
};

// np.sum of a contiguous float64 vector: NumPy's pairwise summation (blocks of 128, eight accumulators), so that the
// evidence of an enumerated sampler carries the same bits as FixedkSampler.fix_exhaustive's np.mean
double np_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double r = 0.0;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
}

enum Kind { kDegenerate = 0, kExhaustive = 1, kAmis = 2 };
enum State { kRunning = 0, kDone = 1, kFailed = 2 };

struct Sampler {
    int kind = kDegenerate, k = 0;
    bild_amis *core = nullptr; // owned until bild_run_take_core hands it to the caller
    bool exhausted = false;
    int64_t steps = 0;       // AMIS steps taken (= len(FixedkSampler.samples))
    std::vector<double> ev;  // evidences: (logev, dlogev, KL) per entry
    // enumerated samplers: their rows and log-likelihoods (FixedkSampler.fix_exhaustive, amis.py:741-803)
    std::vector<double> ss, logL;
    std::vector<int32_t> thetas;
    int64_t n_rows = 0;
    bool evaluated = false;
};

struct LogRow {
    int k = 0;
    bool has_kld = false, has_ila = false;
    std::vector<double> pk, kld; // pk empty: the row was never annotated
    double i_la = 0;
};

struct Traj {
    int T = 0;
    std::vector<Sampler> samplers;
    std::vector<LogRow> rows;
    int target = 0;
    int state = kRunning;
    int err_kind = 0; // 1 RuntimeError ("Iteration did not converge"), 2 ValueError, 3 other
    std::string err;
    // the initial runs of a freshly opened sampler (core.py:194-199)
    int init_left = 0, init_k = -1;
    bool stepped_any = false;
    // ---- this round ----
    std::vector<int> exh; // samplers whose enumeration is evaluated in this round
    int step_k = -1;      // the sampler that takes an AMIS step in this round
    bool judge = false;   // ... and whether that step ends the round's work (the evidence curve is rated afterwards)
    int64_t g_off = 0, u_off = 0, z_off = 0, row_off = 0; // offsets into the round's random numbers / rows
    int64_t n_rows = 0;
    std::vector<double> ss;   // the step's samples, N x k1
    std::vector<int64_t> th;
    std::vector<double> a_cur;
};

} // namespace

struct bild_run {
    bild_run_settings s{};
    int n_states = 0;
    std::vector<uint8_t> trans;
    int n_k = 0;
    std::vector<std::vector<double>> logp0; // per k: n x (k + 1)
    std::vector<double> logprior, n_total;
    std::vector<std::vector<int32_t>> traces; // per k: all valid traces (n_traces x (k + 1)), or empty
    std::vector<Traj> trajs;
    Workers *workers = nullptr;
    // ---- the round ----
    int phase = 0; // 0 idle, 1 planned, 2 staged
    int64_t n_gamma = 0, n_uniform = 0, n_normal = 0, n_rows = 0;
    int K1 = 1;
    std::vector<double> shapes;
    std::vector<double> ss;      // n_rows x K1
    std::vector<int64_t> thetas; // n_rows x K1
    std::vector<int32_t> traj_id;
    std::vector<double> logL;
    int64_t rounds = 0, evaluations = 0;
    std::string err;
    ~bild_run()
    {
        delete workers;
        for (Traj &t : trajs)
            for (Sampler &sm : t.samplers)
                if (sm.core) bild_amis_destroy(sm.core);
    }
};

namespace {

int fail_run(bild_run &r, int code, const std::string &msg)
{
    r.err = msg;
    bild_set_last_error(msg.c_str());
    return code;
}

void fail_traj(Traj &t, int kind, const std::string &msg)
{
    t.state = kFailed;
    t.err_kind = kind;
    t.err = msg;
}

bool exhausted_after_next_step(const bild_run &r, const Sampler &sm)
{
    // amis.py:903-904 after the step: (len(samples) + 1) * N >= max_fev
    return (sm.steps + 2) * r.s.N >= r.s.max_fev;
}

// core.py:216-227 (and the two guards of bild_amd/core.py: goes_on)
bool goes_on(const bild_run &r, const Traj &t, bool stepped)
{
    const int k = t.target, opened = (int)t.samplers.size();
    if (k == opened) return k <= r.s.k_max;
    if (!stepped && t.samplers[k].exhausted) return false;
    const LogRow &newest = t.rows.back();
    double top = -kInf;
    for (double p : newest.pk) top = std::max(top, p);
    if (top >= r.s.certainty_in_k) return false;
    return !newest.has_kld || newest.kld[k] > 0;
}

// a round of the loop in which no step was taken (core.py:141-146)
void after_idle_work(const bild_run &r, Traj &t)
{
    t.target = t.rows.empty() ? (int)t.samplers.size() : t.rows.back().k;
    if (!goes_on(r, t, false)) t.state = kDone;
}

// FixedkSampler.__init__ (amis.py:623-668) for sampler k of trajectory t; enumerations are staged for this round
void open_sampler(bild_run &r, Traj &t, int k)
{
    Sampler sm;
    sm.k = k;
    if (k >= t.T) { // more switches than frames (amis.py:641-648)
        sm.kind = kDegenerate;
        sm.ev = {-kInf, 1e-10, kInf};
        sm.exhausted = true;
        t.samplers.push_back(std::move(sm));
        return;
    }
    if (k >= r.n_k) {
        fail_traj(t, 3, "internal: sampler k = " + std::to_string(k) + " beyond the prepared range");
        return;
    }
    // fix_exhaustive (amis.py:741-803)
    const double Nmax = (double)std::min(r.s.max_fcomplete, r.s.max_fev);
    double nprof = r.n_total[k];
    bool impractical = false;
    for (int i = 0; i < k; ++i) {
        nprof *= (double)(t.T - i - 1);
        if (nprof > Nmax) {
            impractical = true;
            break;
        }
    }
    if (!impractical) {
        const int k1 = k + 1;
        if (r.n_total[k] > Nmax || r.traces[k].empty()) { // CFC.full_sample refuses (amis.py:499-536): a plain ValueError
            char buf[128];
            std::snprintf(buf, sizeof buf, "Full sample would be %.0f > Nmax = %.0f traces", r.n_total[k], Nmax);
            fail_traj(t, 2, buf);
            return;
        }
        // all k-subsets of the T - 1 half-integer switch positions, lexicographically (itertools.combinations), as interval
        // lengths: np.diff([0, (i + 0.5) / (T - 1) ..., 1])
        std::vector<double> ssu;
        std::vector<int> idx(k);
        for (int i = 0; i < k; ++i) idx[i] = i;
        const int npos = t.T - 1;
        const double Tm1 = (double)(t.T - 1);
        int64_t n_ss = 0;
        if (k <= npos)
            for (;;) {
                double prev = 0.0;
                for (int i = 0; i < k; ++i) {
                    const double c = ((double)idx[i] + 0.5) / Tm1;
                    ssu.push_back(c - prev);
                    prev = c;
                }
                ssu.push_back(1.0 - prev);
                ++n_ss;
                int i = k - 1;
                while (i >= 0 && idx[i] == npos - k + i) --i;
                if (i < 0) break;
                ++idx[i];
                for (int j = i + 1; j < k; ++j) idx[j] = idx[j - 1] + 1;
            }
        const int64_t n_th = (int64_t)r.traces[k].size() / k1;
        sm.kind = kExhaustive;
        sm.exhausted = true;
        sm.n_rows = n_ss * n_th; // ss tiled, thetas repeated (amis.py:776-780)
        sm.ss.resize((size_t)sm.n_rows * k1);
        sm.thetas.resize((size_t)sm.n_rows * k1);
        for (int64_t a = 0; a < n_th; ++a)
            for (int64_t b = 0; b < n_ss; ++b) {
                const int64_t row = a * n_ss + b;
                std::memcpy(sm.ss.data() + (size_t)row * k1, ssu.data() + (size_t)b * k1, (size_t)k1 * sizeof(double));
                std::memcpy(sm.thetas.data() + (size_t)row * k1, r.traces[k].data() + (size_t)a * k1, (size_t)k1 * sizeof(int32_t));
            }
        if (sm.n_rows == 0) { // (cannot happen for k < T; np.max of an empty array raises in the reference)
            fail_traj(t, 2, "zero-size enumeration");
            return;
        }
        t.samplers.push_back(std::move(sm));
        t.exh.push_back((int)t.samplers.size() - 1);
        return;
    }
    sm.kind = kAmis;
    const int k1 = k + 1;
    std::vector<double> ones(k1, 1.0);
    if (bild_amis_create(k1, r.n_states, r.trans.data(), r.s.concentration_brake, r.s.polarization_brake, r.logprior[k], ones.data(),
                         r.logp0[k].data(), &sm.core) != BILD_OK) {
        fail_traj(t, 3, "bild_amis_create failed");
        return;
    }
    t.samplers.push_back(std::move(sm));
}

void plan_step(bild_run &r, Traj &t, int k)
{
    Sampler &sm = t.samplers[k];
    t.step_k = k;
    t.a_cur.resize(k + 1);
    bild_amis_params(sm.core, -1, t.a_cur.data(), nullptr);
}

// one trajectory up to its next AMIS step (or its end): everything that needs neither random numbers nor likelihoods
void plan_traj(bild_run &r, Traj &t)
{
    t.exh.clear();
    t.step_k = -1;
    t.judge = false;
    while (t.state == kRunning) {
        if (t.init_left > 0) { // the initial runs of sampler init_k: init_runs calls of step(), whatever they return
            Sampler &sm = t.samplers[t.init_k];
            if (sm.exhausted) { // the remaining calls do nothing (amis.py:812-813)
                t.init_left = 0;
                if (t.stepped_any) { // cannot happen: the step that exhausted the sampler ended the work in its own round
                    fail_traj(t, 3, "internal: initial runs out of step");
                    return;
                }
                after_idle_work(r, t);
                continue;
            }
            plan_step(r, t, t.init_k);
            t.init_left -= 1;
            if (t.init_left == 0 || exhausted_after_next_step(r, sm)) t.judge = true;
            return;
        }
        const int k = t.target, opened = (int)t.samplers.size();
        if (k > opened) {
            fail_traj(t, 1, "Trying to sample outside of existing range; this is a bug");
            return;
        }
        if (k < opened) {
            if (t.samplers[k].exhausted) {
                after_idle_work(r, t);
                continue;
            }
            plan_step(r, t, k);
            t.judge = true;
            return;
        }
        open_sampler(r, t, k);
        if (t.state != kRunning) return;
        const Sampler &sm = t.samplers.back();
        if (sm.kind == kAmis && r.s.init_runs > 0) {
            t.init_left = r.s.init_runs;
            t.init_k = k;
            t.stepped_any = false;
            continue;
        }
        after_idle_work(r, t);
    }
}

// exact evidence of an enumerated sampler (amis.py:787-803)
void exhaustive_evidence(Sampler &sm)
{
    const int64_t n = sm.n_rows;
    double top = -kInf;
    bool any_nan = false;
    for (int64_t i = 0; i < n; ++i) {
        any_nan |= std::isnan(sm.logL[i]);
        top = std::max(top, sm.logL[i]);
    }
    if (any_nan) top = kNaN;
    std::vector<double> rel(n), lr(n);
    for (int64_t i = 0; i < n; ++i) {
        rel[i] = std::exp(sm.logL[i] - top);
        lr[i] = sm.logL[i] * rel[i];
    }
    const double ev = np_sum(rel.data(), n) / (double)n;
    const double logev = std::log(ev) + top;
    const double KL = np_sum(lr.data(), n) / (double)n / ev - logev;
    sm.ev = {logev, 1e-10, KL};
    sm.evaluated = true;
}

// core.py:138-192 (bild_amd/core.py: judge) with the choice sampler of bild/choicesampler.py:83-210; z: samplesize x kmax normals
void judge(const bild_run &r, Traj &t, const double *z)
{
    const int opened = (int)t.samplers.size(), kmax = opened;
    const bool may_open = opened <= r.s.k_max;
    const int64_t S = r.s.choice_samplesize;
    std::vector<double> mu(kmax), root(kmax), dmu(kmax);
    for (int k = 0; k < kmax; ++k) {
        const Sampler &sm = t.samplers[k];
        const double *e = sm.ev.data() + sm.ev.size() - 3;
        const double shat = e[1] * e[1];
        const double budget = sm.exhausted ? kInf : (double)sm.steps;
        mu[k] = e[0];
        root[k] = std::sqrt(shat);
        dmu[k] = std::sqrt(shat / (budget + 1.0));
    }
    std::vector<double> rvs((size_t)S * kmax);
    for (int64_t i = 0; i < S; ++i)
        for (int k = 0; k < kmax; ++k) rvs[(size_t)i * kmax + k] = root[k] * z[(size_t)i * kmax + k];
    const bool straight_on = opened <= r.s.k_lookahead && may_open; // every sampler lies inside the look-ahead window
    const bool with_window = !straight_on && opened > r.s.k_lookahead;
    std::vector<uint8_t> omit(kmax, 0);
    if (with_window)
        for (int k = opened - r.s.k_lookahead; k < opened; ++k) omit[k] = 1;
    std::vector<int64_t> n0(kmax), dn(straight_on ? 0 : (size_t)kmax * kmax), n_omit(kmax);
    bild_choice_counts(S, kmax, rvs.data(), mu.data(), dmu.data(), r.s.dE, with_window ? omit.data() : nullptr, n0.data(),
                       straight_on ? nullptr : dn.data(), with_window ? n_omit.data() : nullptr);
    LogRow &row = t.rows.back();
    row.pk.resize(kmax);
    for (int k = 0; k < kmax; ++k) row.pk[k] = (double)n0[k] / (double)S;
    if (straight_on) {
        row.has_ila = true;
        row.i_la = kInf;
        t.target = opened;
        return;
    }
    const double c = 0.5 / (double)S;
    std::vector<double> term(kmax);
    row.kld.resize(kmax);
    row.has_kld = true;
    for (int kc = 0; kc < kmax; ++kc) {
        for (int k = 0; k < kmax; ++k) {
            const int64_t d = dn[(size_t)kc * kmax + k];
            term[k] = (double)(d * d) / (double)(n0[k] + 1);
        }
        row.kld[kc] = c * np_sum(term.data(), kmax);
    }
    row.has_ila = true;
    if (with_window) {
        int64_t tot = 0;
        for (int k = 0; k < kmax; ++k) tot += n_omit[k];
        for (int k = 0; k < kmax; ++k) {
            const double nw = (double)n_omit[k] / (double)tot * (double)S;
            const double d = omit[k] ? 0.0 : (double)n0[k] - nw;
            term[k] = d * d / (nw + 1.0);
        }
        row.i_la = c * np_sum(term.data(), kmax);
    } else {
        row.i_la = kInf;
    }
    int refine = 0;
    for (int k = 1; k < kmax; ++k)
        if (row.kld[k] > row.kld[refine]) refine = k;
    t.target = (may_open && row.i_la > row.kld[refine]) ? opened : refine;
}

} // namespace

extern "C" {

int bild_run_create(int n_traj, const int32_t *T, int n_states, const uint8_t *transitions, const bild_run_settings *settings, int n_k,
                    const double *logp0, const double *logprior, const double *n_total, const int64_t *n_traces, const int32_t *traces,
                    bild_run **out)
{
    if (!out || n_traj < 0 || (n_traj && !T) || n_states < 1 || !transitions || !settings || n_k < 1 || !logp0 || !logprior || !n_total ||
        !n_traces) {
        bild_set_last_error("bild_run_create: bad arguments");
        return BILD_ERR_INVALID;
    }
    const bild_run_settings &s = *settings;
    if (s.N < 1 || s.init_runs < 0 || s.k_lookahead < 0 || s.k_max < 0 || s.k_max + 1 > n_k || s.choice_samplesize < 1 || s.max_fev < 0 ||
        s.max_fcomplete < 0) {
        bild_set_last_error("bild_run_create: bad settings");
        return BILD_ERR_INVALID;
    }
    bild_run *r = new bild_run;
    r->s = s;
    r->n_states = n_states;
    r->trans.assign(transitions, transitions + (size_t)n_states * n_states);
    r->n_k = n_k;
    const double *lp = logp0;
    const int32_t *tr = traces;
    for (int k = 0; k < n_k; ++k) {
        r->logp0.emplace_back(lp, lp + (size_t)n_states * (k + 1));
        lp += (size_t)n_states * (k + 1);
        r->logprior.push_back(logprior[k]);
        r->n_total.push_back(n_total[k]);
        if (n_traces[k] > 0 && traces) {
            r->traces.emplace_back(tr, tr + (size_t)n_traces[k] * (k + 1));
            tr += (size_t)n_traces[k] * (k + 1);
        } else {
            r->traces.emplace_back();
        }
    }
    r->trajs.resize(n_traj);
    for (int j = 0; j < n_traj; ++j) {
        if (T[j] < 1) {
            delete r;
            bild_set_last_error("bild_run_create: a trajectory without frames");
            return BILD_ERR_INVALID;
        }
        r->trajs[j].T = T[j];
    }
    int threads = bild::config().host_threads;
    if (threads <= 0) threads = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    threads = std::max(1, std::min(threads, std::max(n_traj, 1)));
    r->workers = new Workers(threads);
    *out = r;
    return BILD_OK;
}

int bild_run_destroy(bild_run *r)
{
    delete r;
    return BILD_OK;
}

const char *bild_run_error(const bild_run *r) { return r ? r->err.c_str() : ""; }

// counts: [0] gamma variates, [1] uniforms, [2] normals, [3] candidate rows, [4] trajectories still running (after this
// round's planning; a finished trajectory may still have enumerations in this round), [5] AMIS steps in this round,
// [6] trajectories that have failed so far (counts must hold 8 entries).
// A round with counts[3] == 0 and counts[4] == 0 needs no stage / finish: the run is over.
int bild_run_plan(bild_run *r, int64_t *counts, const double **gamma_shapes)
{
    if (!r || !counts) return BILD_ERR_INVALID;
    if (r->phase != 0) return fail_run(*r, BILD_ERR_INVALID, "bild_run_plan: the previous round has not been finished");
    int64_t g = 0, u = 0, z = 0, rows = 0, live = 0, steps = 0, failed = 0;
    int K1 = 1;
    for (Traj &t : r->trajs) {
        t.n_rows = 0;
        t.exh.clear();
        t.step_k = -1;
        t.judge = false;
        if (t.state == kRunning) plan_traj(*r, t);
        t.g_off = g;
        t.u_off = u;
        t.z_off = z;
        t.row_off = rows;
        for (int i : t.exh) {
            t.n_rows += t.samplers[i].n_rows;
            K1 = std::max(K1, t.samplers[i].k + 1);
        }
        if (t.step_k >= 0) {
            const int k1 = t.step_k + 1;
            g += r->s.N * k1;
            u += r->s.N * k1;
            t.n_rows += r->s.N;
            K1 = std::max(K1, k1);
            ++steps;
            if (t.judge) z += r->s.choice_samplesize * (int64_t)t.samplers.size();
        }
        rows += t.n_rows;
        live += t.state == kRunning;
        failed += t.state == kFailed;
    }
    r->n_gamma = g;
    r->n_uniform = u;
    r->n_normal = z;
    r->n_rows = rows;
    r->K1 = K1;
    r->shapes.resize((size_t)g);
    for (Traj &t : r->trajs)
        if (t.step_k >= 0) {
            const int k1 = t.step_k + 1;
            double *dst = r->shapes.data() + t.g_off;
            for (int64_t i = 0; i < r->s.N; ++i)
                for (int j = 0; j < k1; ++j) dst[(size_t)i * k1 + j] = t.a_cur[j];
        }
    counts[0] = g;
    counts[1] = u;
    counts[2] = z;
    counts[3] = rows;
    counts[4] = live;
    counts[5] = steps;
    counts[6] = failed;
    if (gamma_shapes) *gamma_shapes = r->shapes.data();
    r->phase = (rows > 0 || live > 0) ? 1 : 0;
    return BILD_OK;
}

// gammas: counts[0] standard gamma variates with the shapes bild_run_plan returned; uniforms: counts[1] numbers in [0, 1)
int bild_run_stage(bild_run *r, const double *gammas, const double *uniforms)
{
    if (!r) return BILD_ERR_INVALID;
    if (r->phase != 1) return fail_run(*r, BILD_ERR_INVALID, "bild_run_stage: no round planned");
    if ((r->n_gamma && !gammas) || (r->n_uniform && !uniforms)) return fail_run(*r, BILD_ERR_INVALID, "bild_run_stage: NULL random numbers");
    const int K1 = r->K1;
    const int64_t N = r->s.N;
    r->ss.assign((size_t)r->n_rows * K1, 0.0);
    r->thetas.resize((size_t)r->n_rows * K1);
    r->traj_id.resize((size_t)r->n_rows);
    r->logL.resize((size_t)r->n_rows);
    std::atomic<int> bad{0};
    r->workers->run((int64_t)r->trajs.size(), [&](int64_t j) {
        Traj &t = r->trajs[j];
        if (t.n_rows == 0) return;
        int64_t row = t.row_off;
        // rows shorter than the round's lists are padded with empty intervals in the last state: the expanded profile, and
        // with it the result, is the same (include/bild_amd.h: a result depends on the expanded profile only)
        auto put = [&](const double *s, auto th_of, int k1) {
            double *ds = r->ss.data() + (size_t)row * K1;
            int64_t *dt = r->thetas.data() + (size_t)row * K1;
            for (int i = 0; i < k1; ++i) {
                ds[i] = s[i];
                dt[i] = th_of(i);
            }
            for (int i = k1; i < K1; ++i) dt[i] = dt[k1 - 1];
            r->traj_id[row] = (int32_t)j;
            ++row;
        };
        for (int idx : t.exh) {
            const Sampler &sm = t.samplers[idx];
            const int k1 = sm.k + 1;
            for (int64_t i = 0; i < sm.n_rows; ++i) {
                const int32_t *th = sm.thetas.data() + (size_t)i * k1;
                put(sm.ss.data() + (size_t)i * k1, [&](int c) { return (int64_t)th[c]; }, k1);
            }
        }
        if (t.step_k < 0) return;
        Sampler &sm = t.samplers[t.step_k];
        const int k1 = sm.k + 1;
        t.ss.resize((size_t)N * k1);
        t.th.resize((size_t)N * k1);
        const double *g = gammas + t.g_off, *u = uniforms + t.u_off;
        // Dirichlet points as np.random.dirichlet forms them: sequential sum of the gamma variates, ONE reciprocal, products
        for (int64_t i = 0; i < N; ++i) {
            double acc = 0.0;
            for (int c = 0; c < k1; ++c) acc = acc + g[(size_t)i * k1 + c];
            const double inv = 1.0 / acc;
            double tot = 0.0;
            for (int c = 0; c < k1; ++c) tot += (t.ss[(size_t)i * k1 + c] = g[(size_t)i * k1 + c] * inv);
            if (!std::isfinite(tot)) {
                // all concentrations tiny: every variate underflowed (0 / 0).  The distribution is then, to all digits, a
                // mixture of point masses at the corners of the simplex with probabilities a_c / sum(a) (bild_amd/amis.py:
                // Dirichlet.sample).  No extra random number is consumed: the corner comes from the low-order bits of the
                // row's first trace uniform, which the choice of the first state (its leading bits) does not use.
                double asum = 0.0;
                for (int c = 0; c < k1; ++c) asum += t.a_cur[c];
                const double f = u[i] * 1048576.0, v = (f - std::floor(f)) * asum;
                int corner = k1 - 1;
                double cum = 0.0;
                for (int c = 0; c < k1; ++c) {
                    cum += t.a_cur[c];
                    if (v < cum) {
                        corner = c;
                        break;
                    }
                }
                for (int c = 0; c < k1; ++c) t.ss[(size_t)i * k1 + c] = c == corner ? 1.0 : 0.0;
            }
        }
        if (bild_amis_sample_traces(sm.core, N, u, t.th.data()) != BILD_OK) {
            bad.store(1);
            return;
        }
        for (int64_t i = 0; i < N; ++i) {
            const int64_t *th = t.th.data() + (size_t)i * k1;
            put(t.ss.data() + (size_t)i * k1, [&](int c) { return th[c]; }, k1);
        }
    });
    if (bad.load()) return fail_run(*r, BILD_ERR_INVALID, "bild_run_stage: sampling the traces failed");
    r->phase = 2;
    return BILD_OK;
}

// the staged rows of the round, for a caller that evaluates them itself: ss (n x K1 float64), thetas (n x K1 int64), traj_id (n)
int bild_run_rows(const bild_run *r, int64_t *n, int *K1, const double **ss, const int64_t **thetas, const int32_t **traj_id)
{
    if (!r || r->phase != 2) return BILD_ERR_INVALID;
    if (n) *n = r->n_rows;
    if (K1) *K1 = r->K1;
    if (ss) *ss = r->ss.data();
    if (thetas) *thetas = r->thetas.data();
    if (traj_id) *traj_id = r->traj_id.data();
    return BILD_OK;
}

// logLs: the log-likelihoods of the staged rows (host, counts[3] doubles); normals: counts[2] standard normal numbers
int bild_run_finish(bild_run *r, const double *logLs, const double *normals)
{
    if (!r) return BILD_ERR_INVALID;
    if (r->phase != 2) return fail_run(*r, BILD_ERR_INVALID, "bild_run_finish: no round staged");
    if ((r->n_rows && !logLs) || (r->n_normal && !normals)) return fail_run(*r, BILD_ERR_INVALID, "bild_run_finish: NULL input");
    const int64_t N = r->s.N;
    r->workers->run((int64_t)r->trajs.size(), [&](int64_t j) {
        Traj &t = r->trajs[j];
        if (t.n_rows == 0) return;
        const double *L = logLs + t.row_off;
        for (int idx : t.exh) {
            Sampler &sm = t.samplers[idx];
            sm.logL.assign(L, L + sm.n_rows);
            L += sm.n_rows;
            exhaustive_evidence(sm);
        }
        t.exh.clear();
        if (t.step_k < 0) return;
        Sampler &sm = t.samplers[t.step_k];
        const int k = t.step_k;
        t.step_k = -1;
        double ev[3];
        const int rc = bild_amis_step(sm.core, N, t.ss.data(), t.th.data(), L, ev);
        if (rc != BILD_OK) { // the reference raises out of sample(): this trajectory's run ends here
            const std::string msg = bild_amis_error(sm.core);
            sm.steps += 1; // (the samples were pooled, as the reference appends them before the fit fails)
            fail_traj(t, msg.find("converge") != std::string::npos ? 1 : 3, msg);
            return;
        }
        sm.ev.insert(sm.ev.end(), ev, ev + 3);
        sm.steps += 1;
        if ((sm.steps + 1) * N >= r->s.max_fev) sm.exhausted = true;
        LogRow row;
        row.k = k;
        t.rows.push_back(std::move(row));
        t.stepped_any = true;
        if (!t.judge) return;
        t.judge = false;
        t.init_left = 0;
        judge(*r, t, normals + t.z_off);
        if (!goes_on(*r, t, true)) t.state = kDone;
    });
    r->evaluations += r->n_rows;
    r->rounds += 1;
    r->phase = 0;
    return BILD_OK;
}

// stage + ONE likelihood call on the GPU over all rows of the round + finish
int bild_run_round(bild_run *r, const bild_model *m, const bild_trajset *ts, unsigned flags, const double *gammas, const double *uniforms,
                   const double *normals)
{
    if (!r || !m || !ts) return BILD_ERR_INVALID;
    if (int rc = bild_run_stage(r, gammas, uniforms)) return rc;
    if (r->n_rows) {
        const int rc = bild_logl_st(m, ts, r->n_rows, r->K1, r->ss.data(), r->thetas.data(), r->traj_id.data(), flags, r->logL.data());
        if (rc != BILD_OK) {
            r->err = bild_last_error();
            r->phase = 0;
            return rc;
        }
    }
    return bild_run_finish(r, r->logL.data(), normals);
}

// ---- results --------------------------------------------------------------------------------------------------------
// info: [0] state (0 running, 1 done, 2 failed), [1] kind of failure (1 RuntimeError, 2 ValueError, 3 other), [2] samplers,
// [3] log rows, [4] widest pk / KLD row
int bild_run_traj_info(const bild_run *r, int j, int64_t *info, const char **message)
{
    if (!r || j < 0 || j >= (int)r->trajs.size() || !info) return BILD_ERR_INVALID;
    const Traj &t = r->trajs[j];
    size_t width = 1;
    for (const LogRow &row : t.rows) width = std::max({width, row.pk.size(), row.kld.size()});
    info[0] = t.state;
    info[1] = t.err_kind;
    info[2] = (int64_t)t.samplers.size();
    info[3] = (int64_t)t.rows.size();
    info[4] = (int64_t)width;
    if (message) *message = t.err.c_str();
    return BILD_OK;
}

// the log of trajectory j (core.py:117-122): k (rows), flags (rows: bit 0 pk given, bit 1 KLD given, bit 2 I_la given,
// bits 8-15 length of pk, bits 16-23 length of KLD),
// I_la (rows), pk and KLD (rows x width, NaN-padded); any output may be NULL
int bild_run_traj_log(const bild_run *r, int j, int32_t *k, int32_t *flags, double *i_la, double *pk, double *kld)
{
    if (!r || j < 0 || j >= (int)r->trajs.size()) return BILD_ERR_INVALID;
    const Traj &t = r->trajs[j];
    size_t width = 1;
    for (const LogRow &row : t.rows) width = std::max({width, row.pk.size(), row.kld.size()});
    for (size_t i = 0; i < t.rows.size(); ++i) {
        const LogRow &row = t.rows[i];
        if (k) k[i] = row.k;
        if (flags) flags[i] = (row.pk.empty() ? 0 : 1) | (row.has_kld ? 2 : 0) | (row.has_ila ? 4 : 0) | ((int32_t)row.pk.size() << 8) | ((int32_t)row.kld.size() << 16);
        if (i_la) i_la[i] = row.has_ila ? row.i_la : kNaN;
        for (size_t c = 0; c < width; ++c) {
            if (pk) pk[i * width + c] = c < row.pk.size() ? row.pk[c] : kNaN;
            if (kld) kld[i * width + c] = c < row.kld.size() ? row.kld[c] : kNaN;
        }
    }
    return BILD_OK;
}

// sampler k of trajectory j: info = [0] kind (0: k >= T, 1 enumerated, 2 AMIS), [1] exhausted, [2] AMIS steps taken,
// [3] evidences, [4] enumerated rows
int bild_run_sampler_info(const bild_run *r, int j, int k, int64_t *info)
{
    if (!r || j < 0 || j >= (int)r->trajs.size() || !info) return BILD_ERR_INVALID;
    const Traj &t = r->trajs[j];
    if (k < 0 || k >= (int)t.samplers.size()) return BILD_ERR_INVALID;
    const Sampler &sm = t.samplers[k];
    info[0] = sm.kind;
    info[1] = sm.exhausted ? 1 : 0;
    info[2] = sm.steps;
    info[3] = (int64_t)sm.ev.size() / 3;
    info[4] = sm.kind == kExhaustive && sm.evaluated ? sm.n_rows : 0;
    return BILD_OK;
}

// evidences (info[3] x 3), and for an enumerated sampler its rows: ss (rows x (k+1)), thetas (rows x (k+1) int64), logLs
int bild_run_sampler_data(const bild_run *r, int j, int k, double *evidences, double *ss, int64_t *thetas, double *logLs)
{
    if (!r || j < 0 || j >= (int)r->trajs.size()) return BILD_ERR_INVALID;
    const Traj &t = r->trajs[j];
    if (k < 0 || k >= (int)t.samplers.size()) return BILD_ERR_INVALID;
    const Sampler &sm = t.samplers[k];
    if (evidences) std::copy(sm.ev.begin(), sm.ev.end(), evidences);
    if (sm.kind == kExhaustive && sm.evaluated) {
        if (ss) std::copy(sm.ss.begin(), sm.ss.end(), ss);
        if (thetas)
            for (size_t i = 0; i < sm.thetas.size(); ++i) thetas[i] = sm.thetas[i];
        if (logLs) std::copy(sm.logL.begin(), sm.logL.end(), logLs);
    }
    return BILD_OK;
}

// hands the native bookkeeping of an AMIS sampler (proposals, pooled samples) over to the caller, who destroys it with
// bild_amis_destroy; NULL for samplers that have none (or whose core was taken already)
int bild_run_take_core(bild_run *r, int j, int k, bild_amis **out)
{
    if (!r || !out || j < 0 || j >= (int)r->trajs.size()) return BILD_ERR_INVALID;
    Traj &t = r->trajs[j];
    if (k < 0 || k >= (int)t.samplers.size()) return BILD_ERR_INVALID;
    *out = t.samplers[k].core;
    t.samplers[k].core = nullptr;
    return BILD_OK;
}

// totals: [0] rounds, [1] likelihood evaluations, [2] host threads
int bild_run_totals(const bild_run *r, int64_t *totals)
{
    if (!r || !totals) return BILD_ERR_INVALID;
    totals[0] = r->rounds;
    totals[1] = r->evaluations;
    totals[2] = r->workers ? r->workers->size() : 1;
    return BILD_OK;
}

} // extern "C"
