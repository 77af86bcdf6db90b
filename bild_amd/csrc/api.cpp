// C ABI of bild_amd (include/bild_amd.h): host-side model analysis, device residency,
// launches.  Compiled with hipcc together with kernels.hip into libbild_amd.so.
#include <hip/hip_runtime_api.h>

#include <climits>
#include <emmintrin.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <unordered_map>
#include <queue>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bild_amd.h"
#include "common.h"
#include "config.h"
#include "host_linalg.h"
#include "internal.h"

using namespace bild;
using la::Mat;

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
namespace {


thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? BILD_ERR_NO_DEVICE \
                                                                              : BILD_ERR_HIP,      \
                        "%s failed: %s", #call, hipGetErrorString(e_));                            \
    } while (0)

bool all_finite(const double *p, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(p[i])) return false;
    return true;
}

struct DeviceBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return BILD_OK;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&ptr, want);
        if (e != hipSuccess) {
            ptr = nullptr;
            return fail(BILD_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return BILD_OK;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

// Table memory is recycled.  hipFree of a block of two megabytes or more unmaps it -- 0.22 ms a piece, whatever its size: 0.9 ms per
// trajectory set between the scratch of its builders and its three large tables, a tenth of what `sample` spends on an ordinary
// trajectory when they come one after the other (rocprofv3 --hip-trace of tools/first_call.py).  Blocks of 128 KiB and more are therefore
// rounded up to a size class (powers of two in quarter steps) and kept for the next trajectory set when they are released -- up to
// BILD_TABLE_CACHE_BYTES (default 4 GB; 0: every block goes back to the driver at once).  A request that cannot be met flushes the cache
// and asks again.  Nothing is returned at process exit (the runtime may be gone by then).
struct TableCache {
    std::mutex mu;
    std::multimap<size_t, void *> idle;         // size class -> block
    std::unordered_map<void *, size_t> classes; // every block handed out or idle that came from here
    size_t held = 0;
    static size_t size_class(size_t bytes)
    {
        size_t c = (size_t)128 << 10;
        while (c < bytes) {
            const size_t q = c / 4;
            for (int k = 1; k <= 4 && c < bytes; ++k) c += q; // c, 1.25 c, 1.5 c, 1.75 c, 2 c
        }
        return c;
    }
    void flush_locked()
    {
        for (auto &kv : idle) {
            classes.erase(kv.second);
            (void)hipFree(kv.second);
        }
        idle.clear();
        held = 0;
    }
};
TableCache g_tables;

hipError_t tab_malloc(void **out, size_t bytes)
{
    *out = nullptr;
    const int64_t cap = config().table_cache_bytes;
    if (cap <= 0 || bytes < ((size_t)128 << 10)) return hipMalloc(out, bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t bytes_cls = TableCache::size_class(bytes);
    const size_t cls = bytes_cls | ((size_t)dev << 56); // (blocks stay on the device they were made on)
    std::lock_guard<std::mutex> lk(g_tables.mu);
    auto it = g_tables.idle.find(cls);
    if (it != g_tables.idle.end()) {
        *out = it->second;
        g_tables.held -= bytes_cls;
        g_tables.idle.erase(it);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes_cls);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        g_tables.flush_locked();
        e = hipMalloc(out, bytes_cls);
    }
    if (e == hipSuccess) g_tables.classes[*out] = cls;
    return e;
}

void tab_free(void *ptr)
{
    if (!ptr) return;
    std::unique_lock<std::mutex> lk(g_tables.mu);
    auto it = g_tables.classes.find(ptr);
    if (it == g_tables.classes.end()) { // (a small block, or the cache is off)
        lk.unlock();
        (void)hipFree(ptr);
        return;
    }
    const size_t cls = it->second, bytes_cls = cls & (((size_t)1 << 56) - 1);
    const int64_t cap = config().table_cache_bytes;
    if (cap > 0 && g_tables.held + bytes_cls <= (size_t)cap) {
        g_tables.idle.emplace(cls, ptr);
        g_tables.held += bytes_cls;
        return;
    }
    g_tables.classes.erase(it);
    lk.unlock();
    (void)hipFree(ptr);
}

// page-locked host staging: copies to and from it are true asynchronous DMA transfers
struct PinnedBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return BILD_OK;
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&ptr, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            ptr = nullptr;
            return fail(BILD_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return BILD_OK;
    }
    void release()
    {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};

// kernel timing (bench.py roofline leg)
std::atomic<int32_t *> g_frames_task{nullptr}; // diagnostics: bild_debug_frames_per_task
std::mutex g_time_mu;
int g_time_on = 0; // 0: off; p >= 1: every p-th launch is bracketed by events (sampling keeps the events out of most steps)
uint64_t g_time_count = 0;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_time_events;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_walk_events; // the table walk in front of a split launch (walk.hip)
std::string g_time_name;

} // namespace

// ------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------
struct bild_model {
    int N = 0, d = 0, S = 0;
    unsigned flags = 0;
    // inputs as given
    Mat B, G, Sig, M0, C0, w;
    // invariant-subspace reduction: V is N x n (columns orthonormal); r* are the projected arrays
    int n = 0;
    Mat V;
    Mat rB, rG, rSig, rM0, rC0, rw;
    bool has_G = false;
    // modal analysis (in reduced coordinates)
    bool modal_ok = false;
    std::string modal_why;
    Mat lam, sigd, Q, wq, R, C0q, M0q, Gq; // S*n, S*n, S*n*n, S*n, S*S*n*n, S*n*n, S*n*d, S*n*d
    // packed for the kernels
    int NP = 0; // padded row count of the modal packing; the launch geometry is chosen per batch (geometry_for)
    int NPm[2] = {0, 0}; // padded row count per path (kDense, kModal): the dense packing is rounded up to whole 4x4 tiles
                         // where the matrix-pipe kernel applies (dense_mfma.hip)
    bool wide = false; // NP > kMidMaxNP: LDS-resident kernel (wide.hip), modal path only
    bool mid = false;  // kMaxNP < NP <= kMidMaxNP: tile-register kernel (modal_mfma.hip), modal path only
    bool symmetric = true; // B, Sig, C0 symmetric (what the reference's dsymv calls assume)
    bool tab_factored = false; // modal table of the vector kernels holds Q[s], Q[s]^T instead of all pairs R[s2][s]
    Mat blob_states[2], blob_tab[2];
    // device residency
    mutable std::mutex mu;      // device residency and workspace growth
    mutable std::mutex call_mu; // host-buffer evaluations share one workspace: one at a time per model
    mutable int device = -1;
    mutable double *d_states[2] = {nullptr, nullptr};
    mutable double *d_tab[2] = {nullptr, nullptr};
    // host-buffer entry points (one call at a time per model, call_mu): ONE packed device buffer
    // [seg_start | seg_state | traj_id] filled by one copy out of pinned staging, results back through
    // pinned staging, all on the model's own stream
    mutable DeviceBuf ws_in, ws_out, ws_sched; // ws_sched: workspace of the device-side launch order (schedule.hip)
    mutable PinnedBuf h_in, h_out;
    mutable hipStream_t stream = nullptr;
    mutable unsigned long long *d_frames = nullptr; // frames the tasks ran themselves, summed while kernel timing is on
    mutable hipEvent_t h_in_event = nullptr; // behind the last kernel of a call that left its results on the device (nobody waited)
    mutable bool h_in_busy = false;
    // work lists of split launches (walk.hip): [two sets of kWorkBuckets counters, padded to 128 bytes] [kWorkBuckets lists].
    // Launches alternate between the two sets, and the walk kernel of a launch zeroes the OTHER set -- last touched by the
    // launch before, which has finished -- for the launch after: no launch of its own is needed to zero sixteen integers.
    // The block serves every launch on the stream that used it first (launches on one stream run one after another);
    // launches on other streams get a stream-ordered allocation of their own.
    // Two such blocks: the first for the model's own stream (host-buffer calls), the second for the first other stream that
    // launches on the model (a caller's stream: bench.py, dist.ShardedModel); launches on further streams allocate.
    struct WorkSlot {
        DeviceBuf ws_work;
        DeviceBuf ws_lists; // same rule: segment lists the walk kernel writes for the frame loop ((s, theta) input resident in HBM)
        hipStream_t stream = nullptr;
        bool taken = false;
        int work_set = 0;
        // held from the flip of `work_set` until BOTH kernels of the launch are enqueued: two threads launching on one
        // (model, stream) must enqueue in the order in which they flipped, or a walk would count into a set the frame loop of
        // the launch in front of it still reads
        std::mutex launch_mu;
    };
    mutable WorkSlot slots[2];
    // the block of stream `st`, or null (caller holds `mu`)
    WorkSlot *slot_for(hipStream_t st) const
    {
        if (st == stream && stream != nullptr) {
            slots[0].stream = st;
            slots[0].taken = true;
            return &slots[0];
        }
        if (!slots[1].taken) {
            slots[1].stream = st;
            slots[1].taken = true;
        }
        return slots[1].stream == st ? &slots[1] : nullptr;
    }
    mutable PinnedBuf h_status; // (s, theta) rows refused on the device by calls nobody waited for: sticky until bild_logl_st_status
};

struct bild_trajset {
    const bild_model *model = nullptr;
    int n_traj = 0;
    int d = 0;
    std::vector<TrajDesc> descs; // host copy; .x are device pointers
    int dstar_max = 1;
    int means_max = 1; // most dimensions any covariance chain carries
    int Tmax = 0;
    bool all_valid = true;
    int device = -1;
    double *d_x = nullptr; // all trajectories, each followed by kPadRows padding rows, then kZeroPad zeros
    double *d_zeros = nullptr;
    TrajDesc *d_descs = nullptr;
    // prefix table of the vector kernels' modal path (common.h: prefix_record_doubles), built at the first evaluation
    // that can use it: 0 not tried yet, 1 built, -1 not available for this set (too large, or the build failed)
    mutable std::mutex prefix_mu;
    mutable std::atomic<int> prefix_state{0};
    mutable double *d_prefix = nullptr;
    mutable double *d_tail_g = nullptr;   // first-order tails (tail.hip): kDMax x NP doubles per prefix record
    mutable double *d_prefix_L = nullptr; // running log-likelihood of every record, densely (walk.hip reads nothing else)
    mutable int64_t prefix_records = 0;
    mutable double prefix_build_ms = 0.0;
    // transient table (common.h: TransEntry), built right behind the prefix table: 0 not tried, 1 built, -1 none
    mutable std::atomic<int> trans_state{0};
    mutable TransEntry *d_trans = nullptr;
    mutable int64_t trans_entries = 0;
    mutable double trans_build_ms = 0.0;
    mutable std::atomic<int64_t> evals_seen{0}; // candidates evaluated on this set so far: the tables are built once they pay
    mutable std::atomic<int> trans2_state{0}; // pair table (two switches as one transient): 0 not tried, 1 built, -1 not available
    mutable TransEntry *d_trans2 = nullptr;
    mutable int64_t trans2_entries = 0;
    mutable double trans2_build_ms = 0.0;
    mutable int gap_max = 0;              // gaps 1 .. gap_max - 1 are in the pair table
    // the caller's declaration of how many evaluations the set will see (bild_trajset_expect; < 0: none given): which tables
    // are worth their build -- a property of the set and of that declaration, never of the call history
    mutable std::atomic<int64_t> expected_evals{-1};
    mutable double *d_strans = nullptr;   // transient state table (common.h), filled by the launch that builds the transient table
    mutable int64_t strans_records = 0;   // records of the state table as built: strans_entries * sgap
    mutable int64_t strans_entries = 0;   // (trajectory, chain, old state, new state, frame) combinations
    mutable int sgap = kStateGap;         // gaps 1 .. sgap - 1 behind a switch are covered (sized when the table is built) ...
    mutable int sstride = kStateStride, snq = 0; // ... by snq records per entry, one for every sstride-th gap
    mutable int trans_m_max = 0;          // longest converged transient of the single table
    mutable int two_switch_covered = 0;  // every candidate of <= 2 switches comes out of the tables (checked on the device when the pair table is built)
    mutable int trans_m_typ = 48;         // typical frames-to-convergence of the table's entries (90th percentile): the scheduler's yardstick
};
constexpr int kZeroPad = 8;
// bild_trajset_expect: below these many declared evaluations the prefix (+ transient) tables / the pair and state tables are not built
constexpr int64_t kExpectPrefix = 300, kExpectPairs = 3000;
// set by ensure_transients / ensure_pairs around their own launch: this launch FILLS a table (1: transients, 2: pairs) --
// a thread-local marker, not a bit of `flags`, so that no caller's flags can ever ask for it
thread_local int tl_building = 0;
struct BuildingScope {
    explicit BuildingScope(int what) { tl_building = what; }
    ~BuildingScope() { tl_building = 0; }
}; // internal flag of launch_batch: fill the transient table
// Candidates seen on a trajectory set before its tables are built.  Zero: at the first evaluation -- results then never depend
// on what was evaluated before (a call that runs frame by frame and a later one that uses the tables would differ by ~1e-12).
// Building costs about a millisecond per trajectory of 1000 frames (profiles/r02_transients.txt).
constexpr int64_t kPrefixAfter = 0, kTransientsAfter = 0;

namespace {

// Smallest subspace that contains w (and the mean sources M0, G) and is invariant under every
// B_s, Sig_s, C0_s.  All of these are symmetric, so the orthogonal complement is invariant as
// well and decouples exactly from the observable w.x: the filter restricted to the subspace
// gives the same likelihood.  (For the reference's default model -- free chain vs. chain with
// an end-to-end bond, end-to-end measurement -- this is the reflection-antisymmetric half of
// the modes, N/2 instead of N.)
//
// A Krylov construction is ill-conditioned here (the remainders decay smoothly, there is no
// gap to threshold on).  Instead: eigen-decompose ONE generic combination Z of all matrices;
// every common invariant subspace is spanned by eigenvectors of Z (generic Z has simple
// eigenvalues within each symmetry sector), so select the eigenvectors that overlap w and
// close the selection under the couplings  e_i^T X e_j  of every matrix X.  Overlaps and
// couplings are either O(1e-16) or macroscopic, which makes the threshold robust.
// Returns the basis as ROWS (n x N).
void invariant_subspace(const bild_model &m, Mat &rows, int &n)
{
    const int N = m.N, S = m.S;
    const double tol = 1e-11;
    std::vector<const double *> mats;
    for (int s = 0; s < S; ++s)
        for (const Mat *src : {&m.B, &m.Sig, &m.C0}) mats.push_back(src->data() + (size_t)s * N * N);
    Mat Z((size_t)N * N, 0.0);
    std::vector<double> scale(mats.size());
    for (size_t a = 0; a < mats.size(); ++a) {
        double mx = 0.0;
        for (int i = 0; i < N * N; ++i) mx = std::max(mx, std::fabs(mats[a][i]));
        scale[a] = std::max(mx, 1e-300);
        // fixed irrational-ish weights: reproducible, generic
        const double c = 0.5 + std::fmod(0.7548776662466927 * (double)(a + 1), 1.0);
        for (int i = 0; i < N * N; ++i) Z[i] += c * mats[a][i] / scale[a];
    }
    std::vector<double> ev;
    Mat E;
    la::jacobi_eigh(Z, N, ev, E); // columns
    Mat Et = la::transpose(E, N, N); // rows = eigenvectors

    std::vector<char> sel(N, 0);
    auto seed = [&](const double *v, int stride) {
        double nv = 0.0;
        for (int i = 0; i < N; ++i) nv += v[(size_t)i * stride] * v[(size_t)i * stride];
        nv = std::sqrt(nv);
        if (nv == 0.0) return;
        for (int e = 0; e < N; ++e) {
            double dot = 0.0;
            for (int i = 0; i < N; ++i) dot += Et[(size_t)e * N + i] * v[(size_t)i * stride];
            if (std::fabs(dot) > tol * nv) sel[e] = 1;
        }
    };
    seed(m.w.data(), 1);
    for (int s = 0; s < S; ++s)
        for (int k = 0; k < m.d; ++k) {
            seed(m.M0.data() + (size_t)s * N * m.d + k, m.d);
            seed(m.G.data() + (size_t)s * N * m.d + k, m.d);
        }
    // coupling matrices in the eigenbasis of Z
    std::vector<Mat> coup(mats.size());
    for (size_t a = 0; a < mats.size(); ++a) {
        Mat X(mats[a], mats[a] + (size_t)N * N);
        coup[a] = la::matmul(la::matmul(Et, X, N, N, N), E, N, N, N);
    }
    bool grew = true;
    while (grew) {
        grew = false;
        for (size_t a = 0; a < mats.size(); ++a)
            for (int i = 0; i < N; ++i) {
                if (sel[i]) continue;
                for (int j = 0; j < N; ++j)
                    if (sel[j] && std::fabs(coup[a][(size_t)i * N + j]) > tol * scale[a]) {
                        sel[i] = 1;
                        grew = true;
                        break;
                    }
            }
    }
    rows.clear();
    n = 0;
    for (int e = 0; e < N; ++e)
        if (sel[e]) {
            rows.insert(rows.end(), Et.begin() + (size_t)e * N, Et.begin() + (size_t)(e + 1) * N);
            ++n;
        }
}

int analyse(bild_model &m)
{
    const int N = m.N, d = m.d, S = m.S;
    m.has_G = la::max_abs(m.G) != 0.0;

    // ---- reduction ------------------------------------------------------------------
    bool symmetric = true;
    for (int s = 0; s < S && symmetric; ++s)
        for (const Mat *src : {&m.B, &m.Sig, &m.C0}) {
            const double *X = src->data() + (size_t)s * N * N;
            double scale = 0.0, asym = 0.0;
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) {
                    scale = std::max(scale, std::fabs(X[(size_t)i * N + j]));
                    asym = std::max(asym, std::fabs(X[(size_t)i * N + j] - X[(size_t)j * N + i]));
                }
            if (asym > 1e-12 * std::max(scale, 1e-300)) symmetric = false;
        }

    Mat rows;
    int n = N;
    bool reduced = false;
    if (symmetric && !(m.flags & BILD_MODEL_NO_REDUCE)) {
        invariant_subspace(m, rows, n);
        reduced = n < N && n >= 1;
    }
    if (!reduced) {
        n = N;
        rows.assign((size_t)N * N, 0.0);
        for (int i = 0; i < N; ++i) rows[(size_t)i * N + i] = 1.0;
    }
    m.n = n;
    m.V = la::transpose(rows, n, N); // N x n
    const Mat &Vt = rows;            // n x N

    auto project_sym = [&](const Mat &X3) {
        Mat out((size_t)S * n * n);
        for (int s = 0; s < S; ++s) {
            Mat X(X3.begin() + (size_t)s * N * N, X3.begin() + (size_t)(s + 1) * N * N);
            Mat t = la::matmul(Vt, X, n, N, N);
            Mat r = la::matmul(t, m.V, n, N, n);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) out[((size_t)s * n + i) * n + j] = reduced ? 0.5 * (r[(size_t)i * n + j] + r[(size_t)j * n + i]) : X[(size_t)i * N + j];
        }
        return out;
    };
    auto project_vecs = [&](const Mat &X3) {
        Mat out((size_t)S * n * d);
        for (int s = 0; s < S; ++s) {
            Mat X(X3.begin() + (size_t)s * N * d, X3.begin() + (size_t)(s + 1) * N * d);
            Mat r = la::matmul(Vt, X, n, N, d);
            std::copy(r.begin(), r.end(), out.begin() + (size_t)s * n * d);
        }
        return out;
    };
    m.rB = project_sym(m.B);
    m.rSig = project_sym(m.Sig);
    m.rC0 = project_sym(m.C0);
    m.rG = project_vecs(m.G);
    m.rM0 = project_vecs(m.M0);
    m.rw = la::matmul(Vt, m.w, n, N, 1);

    if (reduced) {
        // verify invariance: || X V - V (V^T X V) || small for every matrix; otherwise undo
        double worst = 0.0;
        for (int s = 0; s < S; ++s) {
            const Mat *full[3] = {&m.B, &m.Sig, &m.C0};
            const Mat *red[3] = {&m.rB, &m.rSig, &m.rC0};
            for (int a = 0; a < 3; ++a) {
                Mat X(full[a]->begin() + (size_t)s * N * N, full[a]->begin() + (size_t)(s + 1) * N * N);
                Mat Xr(red[a]->begin() + (size_t)s * n * n, red[a]->begin() + (size_t)(s + 1) * n * n);
                Mat XV = la::matmul(X, m.V, N, N, n);
                Mat VXr = la::matmul(m.V, Xr, N, n, n);
                double dev = 0.0;
                for (size_t i = 0; i < XV.size(); ++i) dev = std::max(dev, std::fabs(XV[i] - VXr[i]));
                worst = std::max(worst, dev / std::max(la::max_abs(X), 1e-300));
            }
        }
        if (worst > 1e-9) {
            // numerically not invariant enough: keep the full chain
            m.flags |= BILD_MODEL_NO_REDUCE;
            return analyse(m);
        }
    }

    // ---- modal analysis ---------------------------------------------------------------
    m.modal_ok = symmetric;
    m.symmetric = symmetric;
    m.modal_why = symmetric ? "" : "B, Sig or C0 is not symmetric";
    m.lam.assign((size_t)S * n, 0.0);
    m.sigd.assign((size_t)S * n, 0.0);
    m.Q.assign((size_t)S * n * n, 0.0);
    m.wq.assign((size_t)S * n, 0.0);
    m.R.assign((size_t)S * S * n * n, 0.0);
    m.C0q.assign((size_t)S * n * n, 0.0);
    m.M0q.assign((size_t)S * n * d, 0.0);
    m.Gq.assign((size_t)S * n * d, 0.0);
    if (m.modal_ok) {
        for (int s = 0; s < S; ++s) {
            Mat Bs(m.rB.begin() + (size_t)s * n * n, m.rB.begin() + (size_t)(s + 1) * n * n);
            Mat Ss(m.rSig.begin() + (size_t)s * n * n, m.rSig.begin() + (size_t)(s + 1) * n * n);
            // B and Sig of a Rouse model are functions of the same connectivity matrix and share
            // an eigenbasis.  Diagonalise a generic combination so that (near-)degenerate
            // eigenvalues of B alone (fast modes, exp(-ka) ~ 0) are still resolved.
            const double nb = std::max(la::fro(Bs), 1e-300), ns = std::max(la::fro(Ss), 1e-300);
            Mat mix((size_t)n * n);
            for (size_t i = 0; i < mix.size(); ++i) mix[i] = Bs[i] / nb + 0.61803398874989485 * Ss[i] / ns;
            std::vector<double> ev;
            Mat Q;
            la::jacobi_eigh(mix, n, ev, Q);
            Mat Qt = la::transpose(Q, n, n);
            Mat Bq = la::matmul(la::matmul(Qt, Bs, n, n, n), Q, n, n, n);
            Mat Sq = la::matmul(la::matmul(Qt, Ss, n, n, n), Q, n, n, n);
            double offB = 0.0, offS = 0.0;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    if (i != j) {
                        offB = std::max(offB, std::fabs(Bq[(size_t)i * n + j]));
                        offS = std::max(offS, std::fabs(Sq[(size_t)i * n + j]));
                    }
            if (offB > 1e-13 * std::max(la::max_abs(Bs), 1e-300) || offS > 1e-13 * std::max(la::max_abs(Ss), 1e-300)) {
                m.modal_ok = false;
                char buf[160];
                snprintf(buf, sizeof buf, "state %d: B and Sig do not share an eigenbasis (off-diagonal %.2e / %.2e)", s,
                         offB, offS);
                m.modal_why = buf;
                break;
            }
            for (int i = 0; i < n; ++i) {
                m.lam[(size_t)s * n + i] = Bq[(size_t)i * n + i];
                m.sigd[(size_t)s * n + i] = Sq[(size_t)i * n + i];
            }
            std::copy(Q.begin(), Q.end(), m.Q.begin() + (size_t)s * n * n);
            Mat ws(m.rw);
            Mat wqs = la::matmul(Qt, ws, n, n, 1);
            std::copy(wqs.begin(), wqs.end(), m.wq.begin() + (size_t)s * n);
            Mat C0s(m.rC0.begin() + (size_t)s * n * n, m.rC0.begin() + (size_t)(s + 1) * n * n);
            Mat C0qs = la::matmul(la::matmul(Qt, C0s, n, n, n), Q, n, n, n);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    m.C0q[((size_t)s * n + i) * n + j] = 0.5 * (C0qs[(size_t)i * n + j] + C0qs[(size_t)j * n + i]);
            Mat M0s(m.rM0.begin() + (size_t)s * n * d, m.rM0.begin() + (size_t)(s + 1) * n * d);
            Mat Gs(m.rG.begin() + (size_t)s * n * d, m.rG.begin() + (size_t)(s + 1) * n * d);
            Mat M0qs = la::matmul(Qt, M0s, n, n, d), Gqs = la::matmul(Qt, Gs, n, n, d);
            std::copy(M0qs.begin(), M0qs.end(), m.M0q.begin() + (size_t)s * n * d);
            std::copy(Gqs.begin(), Gqs.end(), m.Gq.begin() + (size_t)s * n * d);
        }
    }
    if (m.modal_ok) {
        for (int s2 = 0; s2 < S; ++s2)
            for (int s = 0; s < S; ++s) {
                Mat Q2(m.Q.begin() + (size_t)s2 * n * n, m.Q.begin() + (size_t)(s2 + 1) * n * n);
                Mat Q1(m.Q.begin() + (size_t)s * n * n, m.Q.begin() + (size_t)(s + 1) * n * n);
                Mat Rm = la::matmul(la::transpose(Q2, n, n), Q1, n, n, n);
                std::copy(Rm.begin(), Rm.end(), m.R.begin() + ((size_t)s2 * S + s) * n * n);
            }
    }

    // ---- pack --------------------------------------------------------------------------
    m.NP = padded_rows(n);
    if (!m.NP) {
        if (n > kWideMaxNP)
            return fail(BILD_ERR_UNSUPPORTED, "chain of %d effective modes exceeds the kernels (max %d)", n, kWideMaxNP);
        if (n <= kMidMaxNP) {
            m.NP = (n + 3) & ~3;
            m.mid = true;
        } else {
            m.NP = (n + 1) & ~1;
            m.wide = true;
        }
    }
    m.NPm[kModal] = m.NP;
    // register-resident kernels keep the basis-change matrices in LDS: all S*S pairs while that stays small
    // (two workgroups' worth of product images must still fit beside them), else the 2 S factors Q[s], Q[s]^T
    m.tab_factored = !m.wide && !m.mid && m.modal_ok && (size_t)S * S * table_stride(m.NP) * sizeof(double) > (size_t)32 * 1024;
    // the matrix-pipe kernel reads tiles transposed and relies on B, Sig, C0 being symmetric
    m.NPm[kDense] = (!m.wide && !m.mid && m.symmetric && dense_mfma_supported((n + 3) & ~3)) ? ((n + 3) & ~3) : m.NP;
    for (int mode = 0; mode < 2; ++mode) {
        const int NP = m.NPm[mode];
        const int SB = StateBlock::size(NP);
        const int MS = table_stride(NP);
        Mat &sb = m.blob_states[mode];
        Mat &tb = m.blob_tab[mode];
        sb.assign((size_t)S * SB, 0.0);
        const int ntab = mode == kDense ? 2 * S : S * S;
        tb.assign((size_t)ntab * MS, 0.0);
        if (mode == kModal && !m.modal_ok) continue;
        for (int s = 0; s < S; ++s) {
            double *b = sb.data() + (size_t)s * SB;
            const double *C0src = mode == kDense ? m.rC0.data() + (size_t)s * n * n : m.C0q.data() + (size_t)s * n * n;
            const double *M0src = mode == kDense ? m.rM0.data() + (size_t)s * n * d : m.M0q.data() + (size_t)s * n * d;
            const double *Gsrc = mode == kDense ? m.rG.data() + (size_t)s * n * d : m.Gq.data() + (size_t)s * n * d;
            for (int i = 0; i < n; ++i) {
                b[StateBlock::wq(NP) + i] = mode == kDense ? m.rw[i] : m.wq[(size_t)s * n + i];
                if (mode == kModal) {
                    b[StateBlock::lam(NP) + i] = m.lam[(size_t)s * n + i];
                    b[StateBlock::sig(NP) + i] = m.sigd[(size_t)s * n + i];
                }
                for (int k = 0; k < d; ++k) {
                    b[StateBlock::G(NP) + k * NP + i] = Gsrc[(size_t)i * d + k];
                    b[StateBlock::M0(NP) + k * NP + i] = M0src[(size_t)i * d + k];
                }
                for (int j = 0; j < n; ++j) b[StateBlock::C0(NP) + i * NP + j] = C0src[(size_t)i * n + j];
            }
        }
        auto put = [&](int slot, const double *X) {
            double *t = tb.data() + (size_t)slot * MS;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) t[i * NP + j] = X[(size_t)i * n + j];
        };
        if (mode == kDense) {
            for (int s = 0; s < S; ++s) {
                put(s, m.rB.data() + (size_t)s * n * n);
                put(S + s, m.rSig.data() + (size_t)s * n * n);
            }
        } else if (m.tab_factored) {
            // many states: S*S basis changes R[s2][s] = Q[s2]^T Q[s] would not fit LDS; keep the 2 S factors instead
            // (slot s: Q[s], modal -> common coordinates; slot S + s: Q[s]^T, back) and change basis in two steps
            tb.assign((size_t)2 * S * MS, 0.0);
            for (int s = 0; s < S; ++s) {
                Mat Q(m.Q.begin() + (size_t)s * n * n, m.Q.begin() + (size_t)(s + 1) * n * n);
                Mat Qt = la::transpose(Q, n, n);
                put(s, Q.data());
                put(S + s, Qt.data());
            }
        } else {
            for (int s2 = 0; s2 < S; ++s2)
                for (int s = 0; s < S; ++s) put(s2 * S + s, m.R.data() + ((size_t)s2 * S + s) * n * n);
        }
    }
    return BILD_OK;
}

size_t lds_bytes(const bild_model &m, const Geometry &geom, int mode)
{
    // matrix tables (dense: B_s, Sig_s; modal: basis changes R) + per-group product images
    const size_t groups = (size_t)geom.W * (64 / geom.G);
    const size_t image = (size_t)group_image_doubles(geom.NP) + group_seg_doubles();
    return (m.blob_tab[mode].size() + (size_t)m.S * state_header_doubles(geom.NP) + groups * image) * sizeof(double);
}

int ensure_device(const bild_model &m)
{
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(m.mu);
    if (m.device == dev) return BILD_OK;
    if (m.device != -1)
        return fail(BILD_ERR_INVALID, "model is resident on device %d but device %d is current (one process per GPU)", m.device, dev);
    for (int mode = 0; mode < 2; ++mode) {
        HIP_TRY(hipMalloc((void **)&m.d_states[mode], m.blob_states[mode].size() * sizeof(double)));
        HIP_TRY(hipMemcpy(m.d_states[mode], m.blob_states[mode].data(), m.blob_states[mode].size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&m.d_tab[mode], m.blob_tab[mode].size() * sizeof(double)));
        HIP_TRY(hipMemcpy(m.d_tab[mode], m.blob_tab[mode].data(), m.blob_tab[mode].size() * sizeof(double), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&m.h_in_event, hipEventDisableTiming));
    HIP_TRY(hipMalloc((void **)&m.d_frames, kFrameCounters * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(m.d_frames, 0, kFrameCounters * sizeof(unsigned long long)));
    HIP_TRY(hipDeviceSynchronize()); // (a memset is not ordered against the non-blocking stream just created)
    m.device = dev;
    return BILD_OK;
}

int pick_mode(const bild_model &m, unsigned flags, int *mode)
{
    switch (flags & 0xFu) {
    case BILD_PATH_AUTO: *mode = m.modal_ok ? kModal : kDense; return BILD_OK;
    case BILD_PATH_DENSE: *mode = kDense; return BILD_OK;
    case BILD_PATH_MODAL:
        if (!m.modal_ok) return fail(BILD_ERR_UNSUPPORTED, "modal path unavailable: %s", m.modal_why.c_str());
        *mode = kModal;
        return BILD_OK;
    default: return fail(BILD_ERR_INVALID, "unknown path selector %u", flags & 0xFu);
    }
}

int fill_params(const bild_model &m, const bild_trajset &ts, int mode, KParams &p)
{
    p.states = m.d_states[mode];
    p.tab = m.d_tab[mode];
    p.tab_doubles = (int32_t)m.blob_tab[mode].size();
    p.S = m.S;
    p.d = m.d;
    p.has_G = m.has_G ? 1 : 0;
    p.all_valid = ts.all_valid ? 1 : 0;
    p.trajs = ts.d_descs;
    p.dstar_max = ts.dstar_max;
    p.zeros = ts.d_zeros;
    p.tab_factored = (mode == kModal && m.tab_factored) ? 1 : 0;
    return BILD_OK;
}

// The prefix table of a trajectory set (common.h), built once: the likelihood kernel itself runs one task per
// (trajectory, covariance chain, initial state) with a profile that never switches and stores its state after every
// frame.  Synchronous (the one-time cost of a set, like its upload); afterwards any stream may read the table.
int ensure_prefix(const bild_model &m, const bild_trajset &ts, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(ts.prefix_mu);
    if (ts.prefix_state != 0) return BILD_OK;
    ts.prefix_state = -1;
    if (config().no_prefix) return BILD_OK;
    if (ts.expected_evals >= 0 && ts.expected_evals < kExpectPrefix) return BILD_OK; // a few hundred evaluations: cheaper frame by frame
    const int NP = m.NPm[kModal];
    Geometry geom{};
    if (!builder_geometry(NP, &geom)) return BILD_OK;
    const size_t lds = lds_bytes(m, geom, kModal);
    if (lds > 160 * 1024) return BILD_OK;
    const size_t bytes = (size_t)ts.prefix_records * prefix_record_doubles(NP) * sizeof(double);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return BILD_OK;
    if (bytes > free_b / 4 || bytes > ((size_t)16 << 30)) return BILD_OK; // the table is an optimisation, not a requirement
    const int S = m.S;
    const int64_t nb = (int64_t)ts.n_traj * S;
    std::vector<int32_t> host((size_t)3 * nb);
    for (int j = 0; j < ts.n_traj; ++j)
        for (int s = 0; s < S; ++s) {
            host[(size_t)j * S + s] = 0;                 // seg_start
            host[(size_t)nb + (size_t)j * S + s] = s;    // seg_state
            host[(size_t)2 * nb + (size_t)j * S + s] = j; // traj_id
        }
    int32_t *d_desc = nullptr;
    double *d_tab = nullptr, *d_sink = nullptr, *d_L = nullptr;
    auto cleanup = [&](bool keep) {
        if (d_desc) tab_free(d_desc);
        if (d_sink) tab_free(d_sink);
        if (!keep && d_tab) tab_free(d_tab);
        if (!keep && d_L) tab_free(d_L);
    };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = tab_malloc((void **)&d_desc, host.size() * sizeof(int32_t)) == hipSuccess &&
              tab_malloc((void **)&d_sink, (size_t)nb * ts.dstar_max * sizeof(double)) == hipSuccess &&
              tab_malloc((void **)&d_tab, bytes) == hipSuccess &&
              tab_malloc((void **)&d_L, (size_t)ts.prefix_records * sizeof(double)) == hipSuccess &&
              hipMemsetAsync(d_L, 0, (size_t)ts.prefix_records * sizeof(double), st) == hipSuccess &&
              hipMemcpy(d_desc, host.data(), host.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    if (ok) {
        KParams p{};
        fill_params(m, ts, kModal, p);
        p.ntasks = nb * ts.dstar_max;
        p.K1 = 1;
        p.seg_start = d_desc;
        p.seg_state = d_desc + nb;
        p.traj_id = d_desc + 2 * nb;
        p.out = d_sink;
        p.prefix_dump = d_tab;
        p.prefix_L_dump = d_L;
        const int64_t tpb = (int64_t)geom.W * geom.tasks_per_wave();
        const int grid = (int)std::min<int64_t>(std::max<int64_t>((p.ntasks + tpb - 1) / tpb, 1), 256 * 16);
        (void)hipEventRecord(e0, st);
        ok = launch_logl(geom, kModal, p, grid, lds, (void *)st) == 0 &&
             launch_prefix_L(ts.d_descs, ts.n_traj, S, NP, ts.dstar_max, ts.Tmax, d_tab, d_L, (void *)st) == 0; // (the records' running log-likelihoods)
        (void)hipEventRecord(e1, st);
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        float ms = 0.f;
        if (ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ts.prefix_build_ms = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    cleanup(ok);
    if (ok && !m.has_G && !config().no_tail) {
        // the first-order tails beside the table (tail.hip): one backward pass per (trajectory, chain, state)
        double *d_g = nullptr, *d_gain = nullptr;
        int64_t *d_first = nullptr;
        KParams q{};
        fill_params(m, ts, kModal, q);
        hipEvent_t t0 = nullptr, t1 = nullptr;
        const size_t gbytes = (size_t)ts.prefix_records * kDMax * NP * sizeof(double);
        std::vector<int64_t> first((size_t)ts.n_traj + 1, 0); // blocks of the parallel phase, trajectory by trajectory
        for (int j = 0; j < ts.n_traj; ++j) first[(size_t)j + 1] = first[j] + (int64_t)ts.dstar_max * S * ts.descs[j].T;
        bool good = first.back() == ts.prefix_records && first.back() < ((int64_t)1 << 31) &&
                    tab_malloc((void **)&d_g, gbytes) == hipSuccess &&
                    tab_malloc((void **)&d_gain, (size_t)ts.prefix_records * (NP + 4) * sizeof(double)) == hipSuccess &&
                    tab_malloc((void **)&d_first, first.size() * sizeof(int64_t)) == hipSuccess &&
                    hipMemcpy(d_first, first.data(), first.size() * sizeof(int64_t), hipMemcpyHostToDevice) == hipSuccess &&
                    hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess;
        if (good) {
            (void)hipEventRecord(t0, st);
            good = launch_tail(ts.d_descs, ts.n_traj, S, NP, m.d, ts.dstar_max, q.states, d_tab, d_first, first.back(), d_gain, d_g, (void *)st) == 0;
            (void)hipEventRecord(t1, st);
            good = good && hipStreamSynchronize(st) == hipSuccess;
            float ms = 0.f;
            if (good && hipEventElapsedTime(&ms, t0, t1) == hipSuccess) ts.prefix_build_ms += ms;
        }
        if (t0) (void)hipEventDestroy(t0);
        if (t1) (void)hipEventDestroy(t1);
        if (d_gain) tab_free(d_gain);
        if (d_first) tab_free(d_first);
        if (good) {
            ts.d_tail_g = d_g;
        } else {
            if (d_g) tab_free(d_g);
            (void)hipGetLastError();
        }
    }
    if (ok) {
        ts.d_prefix = d_tab;
        ts.d_prefix_L = d_L;
        ts.prefix_state = 1;
    } else {
        (void)hipGetLastError();
    }
    return BILD_OK;
}

// What a launch may be given beyond the segment lists (all device-visible, all optional)
struct SplitIn {
    // the sampler's own (s, theta) instead of segment lists (bild/amis.py:717-739): the walk kernel converts them on
    // the device and writes the lists the frame loop needs into d_seg_start / d_seg_state of the call (then WRITABLE)
    const double *d_ss = nullptr;
    const int8_t *d_thetas = nullptr;
    int32_t *status = nullptr;   // [0] != 0: a row was not a point on the simplex, [1]: such a row
};
int launch_batch(const bild_model &m, const bild_trajset &ts, int64_t n, int K1, const int32_t *d_seg_start,
                 const int32_t *d_seg_state, const int32_t *d_traj_id, const int32_t *d_order, unsigned flags,
                 hipStream_t st, double *d_out, const SplitIn *sp = nullptr);
// split launches (table walk + frame loop over the work lists) are possible for this many segments per candidate
constexpr int kSplitMaxK1 = kSegLds;
constexpr size_t kWorkHeader = 128; // two sets of work-list counters
static_assert(kWorkHeader >= 2 * kWorkBuckets * sizeof(int32_t), "header holds the counters");

// The transient table of a trajectory set (common.h: TransEntry), built once behind the prefix table: one ordinary
// two-segment candidate per (trajectory, old state, new state, switch frame), evaluated by the likelihood kernel in its
// table-building mode (it stops at the first successful convergence check and writes the entry instead of a result).
int ensure_transients(const bild_model &m, const bild_trajset &ts, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(ts.prefix_mu);
    if (ts.trans_state != 0) return BILD_OK;
    ts.trans_state = -1;
    if (ts.prefix_state != 1 || config().no_transients || config().no_jump || m.S < 2) return BILD_OK;
    const int S = m.S;
    int64_t nb = 0;
    for (const TrajDesc &td : ts.descs) nb += (int64_t)std::max(td.T - 1, 0) * S * (S - 1);
    // the table is an optimisation: not for models with so many states that building it costs more than it can save
    if (nb == 0 || nb > ((int64_t)4 << 20)) return BILD_OK;
    const size_t bytes = (size_t)ts.trans_entries * sizeof(TransEntry);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > free_b / 4) return BILD_OK;
    std::vector<int32_t> host((size_t)5 * nb); // seg_start (2 per sample) | seg_state (2 per sample) | traj_id
    int64_t r = 0;
    for (int j = 0; j < ts.n_traj; ++j)
        for (int s = 0; s < S; ++s)
            for (int sn = 0; sn < S; ++sn) {
                if (sn == s) continue;
                for (int t = 1; t < ts.descs[j].T; ++t, ++r) {
                    host[(size_t)2 * r] = 0;
                    host[(size_t)2 * r + 1] = t;
                    host[(size_t)2 * nb + 2 * r] = s;
                    host[(size_t)2 * nb + 2 * r + 1] = sn;
                    host[(size_t)4 * nb + r] = j;
                }
            }
    int32_t *d_desc = nullptr;
    double *d_sink = nullptr;
    TransEntry *d_tab = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = tab_malloc((void **)&d_desc, host.size() * sizeof(int32_t)) == hipSuccess &&
              tab_malloc((void **)&d_sink, (size_t)nb * sizeof(double)) == hipSuccess &&
              tab_malloc((void **)&d_tab, bytes) == hipSuccess && hipMemsetAsync(d_tab, 0, bytes, st) == hipSuccess &&
              hipMemcpy(d_desc, host.data(), host.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    // First launch: the entries alone.  How long transients last is not known before it has run, and the state table beside
    // the entries (common.h: a chain of close switches starts at its second switch) needs a record for every gap a chain can
    // START with -- gaps shorter than the first switch's transient, i.e. up to the longest converged transient of THIS set,
    // not a compile-time 64: the default model's longest is 45 frames (141 -> ~100 MB per 1000-frame trajectory).
    double *d_states = nullptr;
    float ms_total = 0.f;
    auto build_pass = [&](double *states) {
        ts.d_trans = d_tab; // launch_batch passes it on as the table to FILL (trans_state is still -1)
        ts.d_strans = states;
        (void)hipEventRecord(e0, st);
        bool good;
        {
            BuildingScope scope(1);
            good = launch_batch(m, ts, nb, 2, d_desc, d_desc + 2 * nb, d_desc + 4 * nb, nullptr, BILD_PATH_MODAL, st, d_sink) == BILD_OK;
        }
        (void)hipEventRecord(e1, st);
        good = good && hipStreamSynchronize(st) == hipSuccess;
        float ms = 0.f;
        if (good && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ms_total += ms;
        ts.d_trans = nullptr;
        ts.d_strans = nullptr;
        return good;
    };
    if (ok) ok = build_pass(nullptr);
    if (ok) {
        std::vector<TransEntry> all((size_t)ts.trans_entries);
        ok = hipMemcpy(all.data(), d_tab, bytes, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok) {
            std::vector<int32_t> ms;
            for (const TransEntry &en : all)
                if (en.m > 0) ms.push_back(en.m);
            // longest transient that converged (entries that ran into the trajectory's end say nothing about the filter)
            {
                int64_t i0 = 0;
                for (const TrajDesc &td : ts.descs) {
                    const int64_t cnt = (int64_t)td.T * S * S * ts.dstar_max;
                    for (int64_t i = 0; i < cnt; ++i) {
                        const TransEntry &en = all[(size_t)(i0 + i)];
                        const int t = (int)(i % td.T);
                        if (en.m > 0 && t + en.m < td.T) ts.trans_m_max = std::max(ts.trans_m_max, (int)en.m);
                    }
                    i0 += cnt;
                }
            }
            if (!ms.empty()) {
                std::nth_element(ms.begin(), ms.begin() + ms.size() * 9 / 10, ms.end());
                ts.trans_m_typ = ms[ms.size() * 9 / 10];
            }
        }
    }
    // Second launch: the same candidates once more, now leaving their states -- an optimisation with a budget.  The table
    // costs what its allocation and its fill cost (tens of GB per second: 26 GB, 0.8 s, for the 256 trajectories of BASELINE
    // configs[2]) and saves ~10 us per launch on the chains of close switches: worth it for sets of a few trajectories that
    // see batch after batch (one trajectory: 2 ms against 9 us per AMIS step), not for hundreds of them.  4 GB unless the
    // caller has declared >= 1e8 evaluations on the set (then 64 GB) or BILD_STATES_MAX_BYTES says otherwise; always at most
    // a third of the free memory.  Which tables exist depends on the set and that declaration alone (reproducibility).
    if (ok && !config().no_states && !(ts.expected_evals >= 0 && ts.expected_evals < kExpectPairs) && ts.trans_m_max >= 2) {
        int sgap = std::min<int>(std::max(2, std::min(config().states_max_gap, 255)), ts.trans_m_max + 1);
        const int sstride = std::max(1, std::min(config().states_stride, 8));
        int snq = (sgap - 2) / sstride + 1; // records for g = 1, 1 + sstride, ... <= sgap - 1
        const size_t per_q = (size_t)ts.strans_entries * prefix_record_doubles(m.NPm[kModal]) * sizeof(double);
        size_t budget = (size_t)std::max<int64_t>(config().states_max_bytes, 0);
        if (config().states_max_bytes < 0) budget = ts.expected_evals >= (int64_t)100000000 ? ((size_t)64 << 30) : ((size_t)4 << 30);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, free_b / 3);
        else budget = 0;
        // a table that would not fit covers the SHORT gaps (every chain saves its basis change, the frames saved grow with the
        // gap): as many records per switch as the budget holds, none below gaps of ~16 -- for sets of up to 32 trajectories:
        // what the table saves is the latency of a launch's longest chain, which does not grow with the number of
        // trajectories, while its cost does (configs[2], 256 trajectories: -11 % per step for 8.7 GB and 0.3 s)
        if ((size_t)snq * per_q > budget && per_q > 0) {
            snq = ts.n_traj <= 32 ? (int)(budget / per_q) : 0;
            sgap = snq * sstride + 1;
            if (snq * sstride < 16) snq = 0;
        }
        const size_t sbytes = (size_t)snq * per_q;
        if (snq > 0) {
            if (tab_malloc((void **)&d_states, sbytes) != hipSuccess) {
                d_states = nullptr;
                (void)hipGetLastError();
            }
        }
        if (d_states) {
            ts.sgap = sgap;
            ts.sstride = sstride;
            ts.snq = snq;
            ts.strans_records = ts.strans_entries * snq;
            if (!build_pass(d_states)) { // (the entries are complete; only the state table is lost)
                tab_free(d_states);
                d_states = nullptr;
                ts.strans_records = 0;
                (void)hipGetLastError();
            }
        }
    }
    ts.trans_build_ms = ms_total;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d_desc) tab_free(d_desc);
    if (d_sink) tab_free(d_sink);
    if (ok) {
        ts.d_trans = d_tab;
        ts.d_strans = d_states;
        ts.trans_state = 1;
    } else {
        if (d_tab) tab_free(d_tab);
        if (d_states) tab_free(d_states);
        (void)hipGetLastError();
    }
    return BILD_OK;
}

// The pair table (common.h): two switches closer together than the first one's transient, as one entry.  Built like the
// transient table, by the kernel itself, from candidates with two switches: one per (trajectory, s -> sn -> sm, frame, gap).
// Only for trajectory sets where it can pay: the build is a launch of (T - 1) (gap_max - 1) S (S-1)^2 short tasks per
// trajectory, worth it for sets that see batch after batch of candidates (one trajectory, ten thousand candidates per AMIS
// step), not for hundreds of trajectories with a few candidates each -- decided by the size of the build alone, so that a
// result never depends on what was evaluated before.
int ensure_pairs(const bild_model &m, const bild_trajset &ts, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(ts.prefix_mu);
    if (ts.trans2_state != 0) return BILD_OK;
    ts.trans2_state = -1;
    if (ts.trans_state != 1 || config().no_pairs || m.S < 2) return BILD_OK;
    if (ts.expected_evals >= 0 && ts.expected_evals < kExpectPairs) return BILD_OK; // (the second-level tables pay from a few thousand evaluations on)
    // gaps the table covers: up to the longest converged transient of the single table, 128 at most (BILD_PAIRS_MAX_GAP: another cap --
    // slow chains, whose transients last longer, leave more pairs to the frame loop)
    const int gap_cap = config().pairs_max_gap;
    const int S = m.S, G = std::min(gap_cap, ts.trans_m_max);
    if (G < 2) return BILD_OK;
    int64_t nb = 0;
    for (const TrajDesc &td : ts.descs) nb += (int64_t)std::max(td.T - 1, 0) * (G - 1) * S * (S - 1) * (S - 1);
    // (BILD_PAIRS_MAX_TASKS=<n>: another budget, for sets of many trajectories that will see hundreds of batches)
    const int64_t budget = config().pairs_max_tasks;
    if (nb == 0 || nb > budget) return BILD_OK;
    const int64_t entries = ts.trans_entries * S * G;
    const size_t bytes = (size_t)entries * sizeof(TransEntry);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > free_b / 4) return BILD_OK;
    // the build candidates -- seg_start (3 per task) | seg_state (3 per task) | traj_id -- are written on the device
    // (schedule.hip: pair_tasks_kernel): up to tens of millions of them, nothing the host should fill and send
    std::vector<int64_t> first((size_t)ts.n_traj + 1, 0);
    for (int j = 0; j < ts.n_traj; ++j)
        first[(size_t)j + 1] = first[j] + (int64_t)std::max(ts.descs[j].T - 1, 0) * (G - 1) * S * (S - 1) * (S - 1);
    int32_t *d_desc = nullptr;
    int64_t *d_first = nullptr;
    double *d_sink = nullptr;
    TransEntry *d_tab = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = tab_malloc((void **)&d_desc, (size_t)7 * nb * sizeof(int32_t)) == hipSuccess &&
              tab_malloc((void **)&d_first, first.size() * sizeof(int64_t)) == hipSuccess &&
              tab_malloc((void **)&d_sink, (size_t)nb * sizeof(double)) == hipSuccess &&
              tab_malloc((void **)&d_tab, bytes) == hipSuccess && hipMemsetAsync(d_tab, 0, bytes, st) == hipSuccess &&
              hipMemcpyAsync(d_first, first.data(), first.size() * sizeof(int64_t), hipMemcpyHostToDevice, st) == hipSuccess &&
              launch_pair_tasks(d_first, ts.n_traj, ts.d_descs, S, G, nb, d_desc, d_desc + 3 * nb, d_desc + 6 * nb, (void *)st) == 0 &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    if (ok) {
        ts.d_trans2 = d_tab; // launch_batch passes it on as the table to FILL (trans2_state is still -1)
        ts.gap_max = G;
        (void)hipEventRecord(e0, st);
        {
            BuildingScope scope(2);
            ok = launch_batch(m, ts, nb, 3, d_desc, d_desc + 3 * nb, d_desc + 6 * nb, nullptr, BILD_PATH_MODAL, st, d_sink) == BILD_OK;
        }
        (void)hipEventRecord(e1, st);
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        float ms = 0.f;
        if (ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ts.trans2_build_ms = ms;
        ts.d_trans2 = nullptr;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (!ok) (void)hipStreamSynchronize(st); // (`first` is read by an asynchronous copy)
    if (d_desc) tab_free(d_desc);
    if (d_first) tab_free(d_first);
    if (d_sink) tab_free(d_sink);
    if (ok) {
        // do the tables cover every candidate of at most two switches?  (schedule.hip: two_switch_cover_kernel)
        std::vector<int64_t> ent((size_t)ts.n_traj + 1, 0);
        for (int j = 0; j < ts.n_traj; ++j)
            ent[(size_t)j + 1] = ent[j] + (int64_t)ts.descs[j].dstar * S * (S - 1) * std::max(ts.descs[j].T - 1, 0);
        int64_t *d_ent = nullptr;
        int *d_cov = nullptr;
        int cov = 1;
        const bool good = tab_malloc((void **)&d_ent, ent.size() * sizeof(int64_t)) == hipSuccess && tab_malloc((void **)&d_cov, sizeof(int)) == hipSuccess &&
                          hipMemcpy(d_ent, ent.data(), ent.size() * sizeof(int64_t), hipMemcpyHostToDevice) == hipSuccess &&
                          hipMemcpy(d_cov, &cov, sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
                          launch_two_switch_cover(ts.d_descs, d_ent, ent.back(), ts.n_traj, S, ts.d_trans, d_tab, G, d_cov, (void *)st) == 0 &&
                          hipStreamSynchronize(st) == hipSuccess && hipMemcpy(&cov, d_cov, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
        if (d_ent) tab_free(d_ent);
        if (d_cov) tab_free(d_cov);
        if (!good) (void)hipGetLastError();
        ts.two_switch_covered = good && cov == 1 ? 1 : 0;
        ts.d_trans2 = d_tab;
        ts.trans2_entries = entries;
        ts.trans2_state = 1;
    } else {
        if (d_tab) tab_free(d_tab);
        (void)hipGetLastError();
    }
    return BILD_OK;
}

int launch_batch(const bild_model &m, const bild_trajset &ts, int64_t n, int K1, const int32_t *d_seg_start,
                 const int32_t *d_seg_state, const int32_t *d_traj_id, const int32_t *d_order, unsigned flags,
                 hipStream_t st, double *d_out, const SplitIn *sp)
{
    int mode;
    int rc = pick_mode(m, flags, &mode);
    if (rc) return rc;
    // which kernel family serves this model and path:
    //   kWide       41-128 modes, LDS-resident state (wide.hip)                      modal path only
    //   kModalTiles 33-40 modes, modal recursion on tile registers (modal_mfma.hip)  modal path only
    //   kDenseTiles dense recursion on the matrix pipe (dense_mfma.hip): symmetric models of up to 24 modes
    //               (BILD_DENSE_VALU=1: the LDS-fed vector formulation instead)
    //   kVector     the register-resident vector kernels of kernels.hip, geometry chosen per batch
    enum Family { kVector, kDenseTiles, kModalTiles, kWide };
    Family fam = kVector;
    if (m.wide || m.mid) {
        if (mode != kModal)
            return fail(BILD_ERR_UNSUPPORTED, "chains of more than %d effective modes (here %d) run on the modal path only%s%s", kMaxNP,
                        m.n, m.modal_ok ? "" : ", which is unavailable: ", m.modal_ok ? "" : m.modal_why.c_str());
        fam = m.wide ? kWide : kModalTiles;
    } else if (mode == kDense && m.symmetric && dense_mfma_supported(m.NPm[kDense]) && !config().dense_valu) {
        fam = kDenseTiles;
    }
    Geometry geom{};
    size_t lds = 0;
    if (fam == kVector) {
        // A split launch (below) sends only its chains of close switches through the frame loop -- a few per cent of a
        // batch with few switches per candidate, a third at k = 8 -- and deals them out itself, heaviest first.
        const bool no_split_env0 = config().no_split;
        // (every condition of `split` below that is known here: a launch that takes the geometry of the listed frame loop and
        // then runs the WHOLE batch with it would run at one or two waves per SIMD)
        const bool may_split = mode == kModal && K1 <= kSplitMaxK1 && !tl_building && !no_split_env0 && !(flags & (BILD_NO_SPLIT | BILD_NO_JUMP | BILD_NO_PREFIX)) &&
                               ts.trans_state == 1 && ts.d_prefix_L != nullptr && n * ts.dstar_max <= (int64_t)INT_MAX && !config().no_walk_plan;
        // (the first geometry of the chain length: fewest tasks per wave, and the one whose LDS leaves room for the walk plan)
        const int64_t tasks_for_geometry = may_split ? 1 : n * ts.dstar_max;
        // (the frame loop over the work lists is latency-bound: the row layout -- three mean slots, the shortest frame for a lone
        // wave -- also where fewer mean vectors would allow more tasks per wave; all geometries of a chain length agree bit for bit)
        const int means_for_geometry = may_split ? std::max(ts.means_max, (int)kDMax) : ts.means_max;
        if (!geometry_for(m.NPm[mode], mode, tasks_for_geometry, means_for_geometry, &geom) &&
            !geometry_for(m.NPm[mode], mode, tasks_for_geometry, ts.means_max, &geom))
            return fail(BILD_ERR_UNSUPPORTED, "no kernel for %d rows", m.NPm[mode]);
        if (may_split) {
            Geometry lg{};
            // (... including the room for the walk plan in the workgroup's share of the LDS)
            if (listed_geometry(geom, &lg) &&
                lds_bytes(m, lg, mode) + (size_t)lg.W * (64 / lg.G) * kWalkDoubles * sizeof(double) <= (size_t)160 * 1024 / (size_t)std::max(1, (4 * lg.OCC + lg.W - 1) / lg.W)) {
                // (Rounds 2-3 took a geometry at ONE wave per SIMD only while the estimated list fitted the chip once.  Since the lean
                // frame loop -- no spills at 16 / 20 modes, against 270 / 600 spilled registers of the batch geometries -- it wins at
                // every list length measured: chains of 32 / 40 beads, 10 000 ... 100 000 candidates, k = 4 / 8: 1.3-2.0x / 2.5x;
                // BASELINE configs[3] at k = 8: 335 -> 196 us.  tools/listed_rule.py, BILD_NO_LISTED_GEOMETRY for the comparison.)
                geom = lg;
            }
        }
        lds = lds_bytes(m, geom, mode);
        if (lds > 160 * 1024)
            return fail(BILD_ERR_UNSUPPORTED, "model tables need %zu bytes of LDS (> 160 KiB): too many states (%d) for chain length %d", lds, m.S, m.n);
    }
    // room for the walk plan (the table entries of all switches of a task, fetched at once) where it does not cost occupancy
    const size_t walk_bytes = fam == kVector ? (size_t)geom.W * (64 / geom.G) * kWalkDoubles * sizeof(double) : 0;
    // (a CU holds OCC waves per SIMD = 4 OCC / W workgroups of this geometry, and 160 KiB of LDS for them)
    const size_t lds_per_workgroup = fam == kVector ? (size_t)160 * 1024 / (size_t)std::max(1, (4 * geom.OCC + geom.W - 1) / geom.W) : 0;
    const bool walk_fits = fam == kVector && K1 <= kSegLds && lds + walk_bytes <= lds_per_workgroup && !config().no_walk_plan;

    bool timing; // this launch is bracketed by events (bild_kernel_timing: every p-th one) and counts the frames it runs
    {
        std::lock_guard<std::mutex> lk(g_time_mu);
        timing = g_time_on > 0 && !tl_building && (g_time_count++ % (uint64_t)g_time_on) == 0;
    }
    KParams p{};
    fill_params(m, ts, mode, p);
    p.ntasks = n * ts.dstar_max;
    p.K1 = K1;
    p.seg_start = d_seg_start;
    p.seg_state = d_seg_state;
    p.traj_id = d_traj_id;
    if (fam == kVector) {
        p.order = d_order;
        p.no_jump = (flags & BILD_NO_JUMP) || config().no_jump ? 1 : 0;
        if (mode == kModal && K1 > 0 && !(flags & BILD_NO_PREFIX)) {
            const int64_t seen = tl_building ? 0 : (ts.evals_seen += n);
            const bool after_env = config().tables_after >= 0; // experiments only: delay the tables
            const int64_t prefix_after = after_env ? config().tables_after : kPrefixAfter;
            const int64_t transients_after = after_env ? config().tables_after : kTransientsAfter;
            if (ts.prefix_state == 0 && seen >= prefix_after) ensure_prefix(m, ts, st);
            if (ts.prefix_state == 1) p.prefix = ts.d_prefix;
            if (ts.prefix_state == 1 && !tl_building && !(flags & BILD_NO_TAIL)) p.tail_g = ts.d_tail_g;
            p.tail_tol = std::ldexp(1.0, -std::max(8, std::min(config().tail_tol_bits, 43)));
            p.tail_margin = std::max(0, config().tail_margin);
            const bool no_states = config().no_states;
            if (tl_building == 1) {
                p.trans_dump = ts.d_trans;
                p.strans_dump = ts.d_strans;
                p.sgap = ts.sgap;
                p.sstride = ts.sstride;
                p.snq = ts.snq;
            } else if (tl_building == 2) {
                p.trans2_dump = ts.d_trans2;
                p.gap_max = ts.gap_max;
                if (!no_states && !(flags & BILD_NO_STATES)) {
                    p.strans = ts.d_strans;
                    p.sgap = ts.sgap;
                    p.sstride = ts.sstride;
                    p.snq = ts.snq;
                }
            } else if (p.prefix && !p.no_jump) {
                if (ts.trans_state == 1 && !no_states && !(flags & BILD_NO_STATES)) {
                    p.strans = ts.d_strans;
                    p.sgap = ts.sgap;
                    p.sstride = ts.sstride;
                    p.snq = ts.snq;
                }
                if (ts.trans_state == 0 && seen >= transients_after) ensure_transients(m, ts, st);
                if (ts.trans_state == 1) {
                    p.trans = ts.d_trans;
                    p.m_typ = ts.trans_m_typ;
                    if (ts.trans2_state == 0) ensure_pairs(m, ts, st);
                    if (ts.trans2_state == 1) {
                        p.trans2 = ts.d_trans2;
                        p.gap_max = ts.gap_max;
                    }
                    if (walk_fits) {
                        p.walk_lds = 1;
                        lds += walk_bytes;
                    }
                }
            }
        }
        if (timing) p.frames_run = m.d_frames;
        p.frames_task = g_frames_task.load();
    }
    // d* > 1: one partial result per (sample, covariance chain), summed by a second kernel.  The buffer belongs to
    // THIS call (stream-ordered allocation, released behind the reduction): launches of one model on different
    // streams, or a host-buffer call beside a device-buffer call, share nothing.
    double *target = d_out;
    if (ts.dstar_max > 1) {
        HIP_TRY(hipMallocAsync((void **)&target, (size_t)p.ntasks * sizeof(double), st));
    }
    p.out = target;

    // ---- the table walk in front of the frame loop (walk.hip) -----------------------------------------------------
    // With all tables in place a task is a handful of lookups unless it holds a chain of three or more close switches:
    // one lane per task walks the tables and writes the result, or hands the task on through the work lists; the frame
    // loop then runs the listed tasks only.  Same numbers added in the same order: bit-identical to the single launch
    // (BILD_NO_SPLIT=1).  Also the place where (s, theta) input becomes segment lists.
    const bool st_in = sp && sp->d_ss;
    if (st_in && K1 > kSplitMaxK1) return fail(BILD_ERR_INVALID, "internal: (s, theta) input with %d segments", K1);
    const bool no_split_env = config().no_split;
    const bool no_split = no_split_env || (flags & BILD_NO_SPLIT);
    const bool split = fam == kVector && mode == kModal && K1 <= kSplitMaxK1 && !tl_building && !no_split && p.trans != nullptr &&
                       p.walk_lds && ts.d_prefix_L != nullptr && p.ntasks <= (int64_t)INT_MAX;
    int32_t *work_alloc = nullptr, *lists_alloc = nullptr;
    bild_model::WorkSlot *used_slot = nullptr; // the persistent work-list block this launch alternates the counter set of
    std::unique_lock<std::mutex> slot_order;   // (WorkSlot::launch_mu: released when this function returns, by whatever path)
    auto release = [&]() {
        if (ts.dstar_max > 1) (void)hipFreeAsync(target, st);
        if (work_alloc) (void)hipFreeAsync(work_alloc, st);
        if (lists_alloc) (void)hipFreeAsync(lists_alloc, st);
    };
    if (st_in && (!d_seg_start || !d_seg_state)) {
        // (s, theta) rows resident in HBM and no room given for the lists the frame loop reads: the model's block on the
        // stream that owns it, else an allocation of this call
        const size_t bytes = 2 * (size_t)n * K1 * sizeof(int32_t);
        int32_t *lists = nullptr;
        {
            std::lock_guard<std::mutex> lk(m.mu);
            bild_model::WorkSlot *slot = m.slot_for(st);
            if (slot && slot->ws_lists.reserve(bytes) == BILD_OK) lists = (int32_t *)slot->ws_lists.ptr;
        }
        if (!lists) {
            if (hipMallocAsync((void **)&lists_alloc, bytes, st) != hipSuccess) {
                release();
                return fail(BILD_ERR_NOMEM, "segment lists: out of device memory");
            }
            lists = lists_alloc;
        }
        d_seg_start = lists;
        d_seg_state = lists + (size_t)n * K1;
        p.seg_start = d_seg_start;
        p.seg_state = d_seg_state;
    }
    if (split || st_in) {
        WalkParams w{};
        w.trajs = ts.d_descs;
        w.S = m.S;
        w.dstar_max = ts.dstar_max;
        w.K1 = K1;
        w.n = n;
        w.traj_id = d_traj_id;
        if (st_in) {
            w.ss = sp->d_ss;
            w.thetas = sp->d_thetas;
            w.seg_out_start = const_cast<int32_t *>(d_seg_start);
            w.seg_out_state = const_cast<int32_t *>(d_seg_state);
            w.status = sp->status;
        } else {
            w.seg_start = d_seg_start;
            w.seg_state = d_seg_state;
        }
        w.convert_all = split ? 0 : 1;
        // lists of <= 3 segments hold at most two switches: on a set whose tables cover every such candidate the walk finishes
        // the whole batch, and the frame loop -- which would find its lists empty -- is not launched
        w.no_lists = (split && K1 <= 3 && ts.two_switch_covered && p.trans2 != nullptr && !config().no_fused_launch) ? 1 : 0;
        if (split) {
            int32_t *d_work = nullptr, *d_lists = nullptr;
            const size_t list_bytes = (size_t)kWorkBuckets * (size_t)p.ntasks * sizeof(int32_t);
            {
                bild_model::WorkSlot *slot;
                {
                    std::lock_guard<std::mutex> lk(m.mu);
                    slot = m.slot_for(st);
                }
                if (slot) slot_order = std::unique_lock<std::mutex>(slot->launch_mu); // (never taken under m.mu: no lock order to get wrong)
                std::lock_guard<std::mutex> lk(m.mu);
                if (slot) {
                    DeviceBuf &ws_work = slot->ws_work;
                    if (ws_work.cap < kWorkHeader + list_bytes) {
                        // (hipFree inside waits for the device: nothing still reads the old block)
                        // (the memset on the launch's own stream: a plain hipMemset is not ordered against a non-blocking stream)
                        if (ws_work.reserve(kWorkHeader + list_bytes) != BILD_OK || hipMemsetAsync(ws_work.ptr, 0, kWorkHeader, st) != hipSuccess) {
                            release();
                            return fail(BILD_ERR_NOMEM, "work lists: allocation of %zu bytes failed", kWorkHeader + list_bytes);
                        }
                    }
                    d_work = (int32_t *)ws_work.ptr + kWorkBuckets * slot->work_set;
                    w.work_counts_next = (int32_t *)ws_work.ptr + kWorkBuckets * (1 - slot->work_set);
                    slot->work_set = 1 - slot->work_set;
                    used_slot = slot;
                    d_lists = (int32_t *)((char *)ws_work.ptr + kWorkHeader);
                }
            }
            if (!d_work) {
                hipError_t he = hipMallocAsync((void **)&work_alloc, kWorkHeader + list_bytes, st);
                if (he == hipSuccess) he = hipMemsetAsync(work_alloc, 0, kWorkHeader, st);
                if (he != hipSuccess) {
                    release();
                    return fail(BILD_ERR_NOMEM, "work lists: %s", hipGetErrorString(he));
                }
                d_work = work_alloc;
                d_lists = (int32_t *)((char *)work_alloc + kWorkHeader);
            }
            w.Lc = ts.d_prefix_L;
            w.trans = p.trans;
            w.trans2 = p.trans2;
            w.gap_max = p.gap_max;
            w.m_typ = p.m_typ;
            w.out = target;
            w.work_counts = d_work;
            w.work = d_lists;
            w.work_cap = p.ntasks;
            w.frames_task = p.frames_task;
            p.work_counts = w.work_counts;
            p.work = w.work;
            p.work_cap = w.work_cap;
            p.order = nullptr; // the work lists ARE the launch order
        }
        hipEvent_t w0 = nullptr, w1 = nullptr;
        if (timing && (hipEventCreate(&w0) != hipSuccess || hipEventCreate(&w1) != hipSuccess)) {
            // (behind the flip: the walk will not run, so the other counter set is not zeroed -- the next launch must not take it)
            if (used_slot) {
                std::lock_guard<std::mutex> lk(m.mu);
                used_slot->work_set = 1 - used_slot->work_set;
            }
            if (w0) (void)hipEventDestroy(w0);
            release();
            return fail(BILD_ERR_HIP, "timing events: hipEventCreate failed");
        }
        const int wrc = launch_walk(w, (void *)st, (void *)w0, (void *)w1); // (timed: the events ride on the dispatch)
        if (wrc != 0) {
            if (used_slot) { // the walk never ran: the other counter set was not zeroed -- the next launch must not take it
                std::lock_guard<std::mutex> lk(m.mu);
                used_slot->work_set = 1 - used_slot->work_set;
            }
            release();
            return fail(BILD_ERR_HIP, "walk kernel launch failed: %s", hipGetErrorString((hipError_t)wrc));
        }
        if (timing) {
            std::lock_guard<std::mutex> lk(g_time_mu);
            g_walk_events.emplace_back(w0, w1);
        }
    }

    // tasks per workgroup (the tile kernels size their own grid: 4 waves x 4 tasks)
    const int64_t tasks_per_block = fam == kWide ? 1 : fam != kVector ? 16 : (int64_t)geom.W * geom.tasks_per_wave();
    int64_t blocks = (p.ntasks + tasks_per_block - 1) / tasks_per_block;
    // (work lists: one residency of the chip at most -- most of the tasks never reach the frame loop)
    const int work_blocks = config().work_blocks;
    const int64_t max_blocks = split ? (work_blocks > 0 ? work_blocks : 256 * std::max(geom.OCC, 1)) : 256 * 16;
    const int grid = (int)std::min<int64_t>(std::max<int64_t>(blocks, 1), max_blocks);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // timed launches of the vector kernels carry their events on the dispatch (start / end of the kernel itself); the tile
    // kernels are bracketed by recorded events (milliseconds long: the brackets' own latency does not matter there)
    const bool ride = fam == kVector;
    if (timing) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        if (!ride) HIP_TRY(hipEventRecord(e0, st));
    }
    const bool frame_loop_needed = !(split && K1 <= 3 && ts.two_switch_covered && p.trans2 != nullptr && !config().no_fused_launch);
    int lrc = !frame_loop_needed   ? 0
              : fam == kWide       ? launch_logl_wide(m.NP, p, grid, (void *)st)
              : fam == kModalTiles ? launch_logl_modal_mfma(m.NPm[kModal], p, (void *)st)
              : fam == kDenseTiles ? launch_logl_dense_mfma(m.NPm[kDense], p, (void *)st)
                                   : launch_logl(geom, mode, p, grid, lds, (void *)st, timing ? (void *)e0 : nullptr, timing ? (void *)e1 : nullptr);
    if (lrc != 0) {
        release();
        return fail(BILD_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
    }
    if (work_alloc) (void)hipFreeAsync(work_alloc, st);
    if (lists_alloc) (void)hipFreeAsync(lists_alloc, st);
    if (timing) {
        if (!frame_loop_needed) HIP_TRY(hipEventRecord(e0, st)); // (no dispatch for the events to ride on: an empty bracket)
        if (!ride || !frame_loop_needed) HIP_TRY(hipEventRecord(e1, st));
        std::lock_guard<std::mutex> lk(g_time_mu);
        g_time_events.emplace_back(e0, e1);
        g_time_name = fam == kWide ? "logl_wide_kernel" : fam == kModalTiles ? "logl_modal_mfma_kernel" : fam == kDenseTiles ? "logl_dense_mfma_kernel" : kernel_name(geom, mode);
    }
    if (ts.dstar_max > 1) {
        lrc = launch_reduce_partials(target, d_out, n, ts.dstar_max, (void *)st);
        (void)hipFreeAsync(target, st);
        if (lrc != 0) return fail(BILD_ERR_HIP, "reduce launch failed: %s", hipGetErrorString((hipError_t)lrc));
    }
    if (st_in && !split) {
        // every row went through the frame loop, the refused ones with a marked list of no switch: NaN for them, as in a split
        // launch (entries that leave their results on the device cannot refuse a row otherwise)
        lrc = launch_mark_refused_rows(d_seg_start, K1, n, d_out, (void *)st);
        if (lrc != 0) return fail(BILD_ERR_HIP, "launch failed: %s", hipGetErrorString((hipError_t)lrc));
    }
    return BILD_OK;
}

// Launch order of a batch (vector kernels, modal path, tables in use).  Candidates differ in the number of frames they run
// themselves; the four (or so) tasks of a wavefront are independent rows of one instruction stream, so the wave lives as long
// as its busiest row.
//  1. sort by the work a candidate will do, most first: a wave's rows then finish together, and long work is dispatched
//     first.  The work is estimated from the candidate's switches and the transient table's frames-to-convergence (with
//     BILD_NO_JUMP: the remaining length behind the first switch);
//  2. when the whole grid is resident at once -- at most OCC workgroups per CU -- nothing is ever re-balanced at run time:
//     the dispatcher deals workgroups to the 256 CUs round-robin (workgroups b, b + 256, b + 512 share a CU;
//     measured: profiles/r02_placement.txt), so the sorted workgroups are dealt to CUs longest-processing-time-first
//     with the CU's number of workgroups as capacity, and written out in that dealing order.
// Purely a matter of speed: results do not depend on the order, and nothing relies on the dispatcher behaving so.
// order[slot] = sample.  Returns false when the identity is as good (nothing written).
bool schedule(const bild_model &m, const bild_trajset &ts, int64_t n, int K1, const int32_t *seg_start, const int32_t *seg_state,
              const int32_t *traj_id, unsigned flags, int32_t *order, bool inside_a_call)
{
    int mode;
    if (n < 2 || K1 < 2 || n > INT_MAX || pick_mode(m, flags, &mode) || mode != kModal || m.wide || m.mid) return false;
    if ((flags & BILD_NO_PREFIX) || ts.prefix_state < 0 || config().no_prefix || config().no_schedule) return false;
    const bool jumps = !(flags & BILD_NO_JUMP) && !config().no_jump;
    // with jumps but without a transient table every switch costs about the same wherever it is: nothing to sort by
    if (jumps && ts.trans_state != 1) return false;
    // Measured on the 10k batch (profiles/r02_transients.txt): with the tables the order is worth 22 us of kernel time and
    // costs 62 us of host time -- inside a host-buffer call it does not pay; a caller with resident candidates computes
    // it once (bild_schedule_segments) and reuses it.
    if (jumps && inside_a_call) return false;
    Geometry geom{};
    if (!geometry_for(m.NPm[mode], mode, n * ts.dstar_max, ts.means_max, &geom)) return false;
    const int Tmax = ts.Tmax, m_typ = ts.trans_m_typ;
    const bool pairs = ts.trans2_state == 1;
    std::vector<int32_t> count((size_t)Tmax + 2, 0), work((size_t)n);
    for (int64_t r = 0; r < n; ++r) {
        const TrajDesc &td = ts.descs[traj_id ? traj_id[r] : 0];
        const int T = td.T;
        const int32_t *a = seg_start + r * K1;
        int w = 0;
        if (!jumps) {
            int t0 = a[1];
            t0 = t0 < 1 ? 1 : (t0 > T ? T : t0);
            w = T - t0;
        } else {
            // Frames the candidate will run itself, estimated from its switch frames alone: a switch whose segment is at
            // least m_typ frames long (the table's typical frames-to-convergence) comes out of the transient table, unless a
            // run is in progress, which then ends m_typ frames behind it; shorter segments chain into one run.  (Boundaries
            // that switch nothing are rare and only blur the estimate.)
            int run_from = -1, links = 0;
            for (int i = 1; i < K1; ++i) {
                const int t = a[i];
                if (t >= T) break;
                const int gap = ((i + 1 < K1 && a[i + 1] < T) ? a[i + 1] : T) - t;
                if (run_from < 0) {
                    if (gap < m_typ) {
                        run_from = t;
                        links = 1;
                    }
                } else {
                    ++links;
                    if (gap >= m_typ) {
                        if (!(pairs && links == 2)) w += t + m_typ - run_from; // two switches: out of the pair table
                        run_from = -1;
                    }
                }
            }
            if (run_from >= 0 && !(links == 1 || (pairs && links == 2))) w += T - run_from;
            w = w > Tmax ? Tmax : w;
        }
        w = w < 0 ? 0 : w; // (a row of decreasing starts can make the estimate negative; host entries reject such rows beforehand)
        work[r] = w;
        ++count[Tmax - w + 1];
    }
    for (int i = 1; i <= Tmax + 1; ++i) count[i] += count[i - 1];
    std::vector<int32_t> sorted((size_t)n);
    for (int64_t r = 0; r < n; ++r) sorted[count[Tmax - work[r]]++] = (int32_t)r;
    if (jumps) {
        // Most candidates of a batch run no frame at all, a few run hundreds, and the rows of a wave share one instruction
        // stream: every event of a row (basis change, comparison, jump) is paid by the whole wave.  So the busy candidates
        // are SPREAD: the heaviest go one per wave, the next heaviest fill the second rows, and so on -- a wave then holds
        // one long row and light ones instead of four long rows whose events add up.
        const int64_t rpw = geom.tasks_per_wave() % ts.dstar_max == 0 ? geom.tasks_per_wave() / ts.dstar_max : 1;
        // ... round by round: the waves the chip holds at once (256 CUs x OCC workgroups x W waves) take the heaviest
        // candidates that fit into them, spread as above; the next round the next heaviest, and so on.  A batch that fits
        // the chip once is one round (plain spreading: the 10k batch, latency-bound by its longest chain); a batch many
        // times that size with few busy candidates has them all in its first round.
        // A batch of several rounds with more busy candidates than one round has waves is throughput-bound whatever the
        // order: then candidates of equal work share a wave (the sorted order as it is), heaviest waves first
        // (`profiles/r02_launch_order.txt`: 80 000 candidates 220 -> 157 us, configs[2]'s 256 000 1.77 -> 0.92 ms; a batch
        // that fits the chip once is better off spread even when every wave has busy rows: 10 000 x k = 8, 123 vs 152 us).
        const char *mode_env = config().sched_mode.empty() ? nullptr : config().sched_mode.c_str(); // experiments: "spread" / "sorted" whatever the batch
        const int64_t slots = (int64_t)256 * geom.OCC * geom.W * rpw;
        int64_t busy = 0;
        while (busy < n && work[sorted[busy]] > 0) ++busy;
        const bool packed = mode_env ? mode_env[1] == 'o' : (n > slots && busy > slots / rpw);
        if (packed) {
            std::copy(sorted.begin(), sorted.end(), order);
            return true;
        }
        const int64_t round = (mode_env && mode_env[1] == 'p') ? n : std::max<int64_t>(slots, rpw);
        for (int64_t base = 0; base < n; base += round) {
            const int64_t cnt = std::min(round, n - base), nw = cnt / rpw, n_full = nw * rpw;
            for (int64_t w = 0; w < nw; ++w)
                for (int64_t j = 0; j < rpw; ++j) order[base + w * rpw + j] = sorted[base + j * nw + w];
            for (int64_t i = n_full; i < cnt; ++i) order[base + i] = sorted[base + i]; // the lightest few: a last, partial wave
        }
        return true;
    }
    const int64_t per_block = std::max<int64_t>(1, (int64_t)geom.W * geom.tasks_per_wave() / ts.dstar_max);
    const int64_t nb = (n + per_block - 1) / per_block;
    const int kCUs = 256;
    if (nb <= kCUs || nb > (int64_t)kCUs * geom.OCC || (int64_t)geom.W * geom.tasks_per_wave() % ts.dstar_max != 0) {
        std::copy(sorted.begin(), sorted.end(), order);
        return true;
    }
    // blocks of the sorted list, longest first; block length = its first (longest) sample
    const int rounds = (int)((nb + kCUs - 1) / kCUs);
    const int extra = (int)(nb - (int64_t)(rounds - 1) * kCUs); // CUs 0 .. extra-1 take `rounds` workgroups, the others one less
    typedef std::pair<int64_t, int> Load; // (work so far, CU)
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int c = 0; c < kCUs; ++c) heap.push(Load(0, c));
    std::vector<int> filled(kCUs, 0);
    std::vector<int64_t> at((size_t)nb, -1); // launch position -> block of the sorted list
    for (int64_t b = 0; b < nb; ++b) {
        const Load top = heap.top();
        heap.pop();
        const int c = top.second;
        at[(size_t)c + (size_t)kCUs * filled[c]] = b;
        ++filled[c];
        const int cap = c < extra ? rounds : rounds - 1;
        if (filled[c] < cap) heap.push(Load(top.first + work[sorted[b * per_block]] + 1, c));
    }
    // only the last block of the sorted list can be short: it goes to the last position, so that blocks of samples and
    // workgroups stay aligned
    for (int64_t pos = 0; pos < nb; ++pos)
        if (at[(size_t)pos] == nb - 1) {
            std::swap(at[(size_t)pos], at[(size_t)nb - 1]);
            break;
        }
    int64_t w = 0;
    for (int64_t pos = 0; pos < nb; ++pos) {
        const int64_t b = at[(size_t)pos];
        const int64_t lo = b * per_block, hi = std::min(n, lo + per_block);
        for (int64_t i = lo; i < hi; ++i) order[w++] = sorted[i];
    }
    return true;
}

// Launch order for a host-buffer call, computed on the device behind the upload (schedule.hip): only where it pays -- a
// batch of several rounds on a trajectory set whose tables exist (from its second evaluation on); 1: no order (array order)
int device_order(const bild_model &m, const bild_trajset &ts, int64_t n, int K1, const int32_t *d_start, const int32_t *d_tid,
                 unsigned flags, hipStream_t st, const int32_t **d_order)
{
    if (n < 2 || K1 < 2 || n > INT_MAX || m.wide || m.mid || !m.modal_ok) return 1;
    const unsigned path = flags & 0xFu;
    if (path != BILD_PATH_AUTO && path != BILD_PATH_MODAL) return 1;
    if ((flags & (BILD_NO_PREFIX | BILD_NO_JUMP)) || config().no_prefix || config().no_jump || config().no_schedule) return 1;
    if (ts.prefix_state != 1 || ts.trans_state != 1) return 1;
    // a split launch orders its frame loop itself (work lists by expected work)
    if (K1 <= kSplitMaxK1 && !config().no_split) return 1;
    Geometry geom{};
    if (!geometry_for(m.NPm[kModal], kModal, n * ts.dstar_max, ts.means_max, &geom)) return 1;
    if (geom.tasks_per_wave() % ts.dstar_max != 0) return 1;
    const int rpw = geom.tasks_per_wave() / ts.dstar_max;
    const int64_t slots = (int64_t)256 * geom.OCC * geom.W * rpw;
    if (n <= slots) return 1; // one round: the order does not matter (10k batch: 80 vs 81 us)
    const size_t bytes = device_schedule_bytes(n);
    {
        std::lock_guard<std::mutex> lk(m.mu);
        if (m.ws_sched.reserve(bytes)) return 1;
    }
    return device_schedule(d_start, d_tid, ts.d_descs, K1, n, ts.trans_m_typ, ts.trans2_state == 1 ? 1 : 0, ts.Tmax, rpw, slots, m.ws_sched.ptr,
                           m.ws_sched.cap, d_order, (void *)st);
}

// Host buffers in, host buffer out.  `fill(h_start, h_state)` writes the n x K1 run-length segments straight into
// pinned staging memory (and validates them: these indices drive device addressing); then ONE host-to-device copy of
// the packed block [seg_start | seg_state | traj_id], the launch, one device-to-host copy of the results, one
// synchronisation -- all on the model's own stream.
// BILD_TRACE_STAGED=1: print where a host-buffer call spends its time (microseconds), every 64th call
struct StageClock {
    bool on;
    std::chrono::steady_clock::time_point t0;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    StageClock() : on(config().trace_staged), t0(std::chrono::steady_clock::now()) {}
    void lap(int i)
    {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        acc[i] += std::chrono::duration<double, std::micro>(t1 - t0).count();
        t0 = t1;
    }
};

// Device block of a host-buffer call (ws_in), filled by ONE copy out of the pinned block of the same layout:
//   [ header: reserved, zero ]                                                               kStagedHeader bytes
//   [ payload: segment lists (seg_start | seg_state), or (s, theta) rows (ss float64 | thetas uint8, padded to 8 bytes) ]
//   [ traj_id (n int32), when given ] [ launch order (n int32), when the host scheduled ]
// and behind what is copied, device only:
//   [ (s, theta) input: the segment lists the walk kernel writes for the frame loop: seg_start | seg_state ]
constexpr size_t kStagedHeader = 128;

template <typename Fill>
int run_staged(const bild_model *m, const bild_trajset *ts, int64_t n, int K1, const int32_t *traj_id, unsigned flags,
               double *out, double *d_out_user, hipStream_t st_user, bool st_payload, Fill &&fill)
{
    int rc;
    if (traj_id)
        for (int64_t r = 0; r < n; ++r)
            if (traj_id[r] < 0 || traj_id[r] >= ts->n_traj)
                return fail(BILD_ERR_INVALID, "traj_id[%lld]=%d out of range", (long long)r, traj_id[r]);
    std::lock_guard<std::mutex> call_lock(m->call_mu);
    const size_t nseg = (size_t)n * K1;
    const size_t payload = st_payload ? nseg * sizeof(double) + ((nseg + 7) & ~(size_t)7) : 2 * nseg * sizeof(int32_t);
    const size_t copy_cap = kStagedHeader + payload + 2 * (size_t)n * sizeof(int32_t); // room for traj_id and the launch order
    const size_t lists = st_payload ? 2 * nseg * sizeof(int32_t) : 0;
    // A previous call that left its results on the device (bild_logl_st_to_device: nothing waited for) may still be
    // running: its kernels read the device block and the work lists, its copy reads the pinned block.  The event was
    // recorded behind its last kernel.
    if (m->h_in_busy) HIP_TRY(hipEventSynchronize(m->h_in_event));
    m->h_in_busy = false;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        if ((rc = m->h_in.reserve(copy_cap))) return rc;
        if ((rc = m->ws_in.reserve(copy_cap + lists))) return rc;
        if ((rc = m->h_out.reserve((size_t)n * sizeof(double) + 64))) return rc; // (+ the status word of (s, theta) input)
        if (!d_out_user && (rc = m->ws_out.reserve((size_t)n * sizeof(double)))) return rc;
    }
    StageClock clk;
    char *h_base = (char *)m->h_in.ptr, *d_base = (char *)m->ws_in.ptr;
    std::memset(h_base, 0, kStagedHeader);
    int32_t *h_tid = (int32_t *)(h_base + kStagedHeader + payload);
    int32_t *h_order = h_tid + (traj_id ? n : 0);
    if ((rc = fill(h_base + kStagedHeader))) return rc;
    if (traj_id) std::memcpy(h_tid, traj_id, (size_t)n * sizeof(int32_t));
    clk.lap(0);
    // the host scheduler serves launches that will NOT be split (no tables, BILD_NO_JUMP, ...): see schedule()
    const bool ordered = !st_payload && schedule(*m, *ts, n, K1, (const int32_t *)(h_base + kStagedHeader),
                                                 (const int32_t *)(h_base + kStagedHeader) + nseg, traj_id, flags, h_order, true);
    clk.lap(1);
    const size_t in_bytes = kStagedHeader + payload + ((traj_id ? (size_t)n : 0) + (ordered ? (size_t)n : 0)) * sizeof(int32_t);
    int32_t *d_tid = traj_id ? (int32_t *)(d_base + kStagedHeader + payload) : nullptr;
    const int32_t *d_order = ordered ? (int32_t *)(d_base + kStagedHeader + payload) + (traj_id ? n : 0) : nullptr;
    int32_t *d_start, *d_state;
    SplitIn sp;
    // the status word lives in pinned host memory the device writes to directly: the host reads it after its synchronisation
    int32_t *h_status = (int32_t *)((char *)m->h_out.ptr + (size_t)n * sizeof(double));
    // Batches of up to 50 000 rows on one trajectory: the walk kernel reads the (s, theta) rows straight out of the pinned
    // block over PCIe (450 KB for the 10k batch) -- no host-to-device copy to enqueue and wait for: 136 -> 130 us per call.
    // BILD_IN_VIA_COPY=1: always through a copy in HBM.
    const bool in_copy = config().in_via_copy;
    const bool direct_in = !in_copy && st_payload && !traj_id && n <= 50000;
    if (st_payload) {
        d_start = (int32_t *)(d_base + copy_cap);
        d_state = d_start + nseg;
        const char *in_base = direct_in ? h_base : d_base;
        sp.d_ss = (const double *)(in_base + kStagedHeader);
        sp.d_thetas = (const int8_t *)(in_base + kStagedHeader + nseg * sizeof(double));
        if (d_out_user) { // nobody waits: the verdict stays with the model until bild_logl_st_status asks
            if (!m->h_status.ptr) {
                std::lock_guard<std::mutex> lk(m->mu);
                if ((rc = m->h_status.reserve(64))) return rc;
                std::memset(m->h_status.ptr, 0, 64);
            }
            h_status = (int32_t *)m->h_status.ptr;
        } else {
            h_status[0] = h_status[1] = 0;
        }
        sp.status = h_status;
    } else {
        d_start = (int32_t *)(d_base + kStagedHeader);
        d_state = d_start + nseg;
    }
    // Results of a host-buffer call: the kernels write them straight into the pinned block (80 KB of posted writes over
    // PCIe for the 10k batch) -- no device-to-host copy to launch and wait for.  BILD_OUT_VIA_COPY=1: through HBM and a copy.
    const bool out_via_copy = config().out_via_copy;
    double *d_out = d_out_user ? d_out_user : (out_via_copy ? (double *)m->ws_out.ptr : (double *)m->h_out.ptr);
    hipStream_t st = d_out_user ? st_user : m->stream;
    if (!direct_in) HIP_TRY(hipMemcpyAsync(d_base, h_base, in_bytes, hipMemcpyHostToDevice, st));
    if (!ordered && !st_payload) {
        const int32_t *on_device = nullptr;
        if (device_order(*m, *ts, n, K1, d_start, d_tid, flags, st, &on_device) == 0) d_order = on_device;
    }
    rc = launch_batch(*m, *ts, n, K1, d_start, d_state, d_tid, d_order, flags, st, d_out, &sp);
    if (rc) {
        (void)hipStreamSynchronize(st);
        return rc;
    }
    if (d_out_user) {
        // results stay in HBM, ordered on the caller's stream; the event covers everything this call reads
        HIP_TRY(hipEventRecord(m->h_in_event, st));
        m->h_in_busy = true;
        return BILD_OK;
    }
    clk.lap(2);
    if (out_via_copy) HIP_TRY(hipMemcpyAsync(m->h_out.ptr, d_out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    clk.lap(3);
    if (st_payload && h_status[0] != 0)
        return fail(BILD_ERR_INVALID, "interval lengths of sample %d are not non-negative finite numbers (of a point on the simplex)", h_status[1]);
    std::memcpy(out, m->h_out.ptr, (size_t)n * sizeof(double));
    clk.lap(4);
    if (clk.on) {
        static int calls = 0;
        if ((calls++ & 63) == 0)
            fprintf(stderr, "[bild staged n=%lld] fill %.1f  schedule %.1f  enqueue(H2D+launch) %.1f  wait(kernel+D2H) %.1f  copy out %.1f us\n",
                    (long long)n, clk.acc[0], clk.acc[1], clk.acc[2], clk.acc[3], clk.acc[4]);
    }
    return BILD_OK;
}

} // namespace

void *bild::internal_model_stream(const bild_model *m)
{
    if (!m || ensure_device(*m) != BILD_OK) return nullptr;
    return (void *)m->stream;
}

// (s, theta) rows that are resident in HBM already (internal.h): the fused AMIS step (amis_host.cpp) keeps its pooled samples
// there and hands the newest batch over where it lies
int bild::internal_logl_st_resident(const bild_model *m, const bild_trajset *ts, int64_t n, int K1, const double *d_ss,
                                    const uint8_t *d_thetas, unsigned flags, double *d_out, int32_t *status, void **stream)
{
    if (!m || !ts || ts->model != m || n < 1 || K1 < 1 || K1 > kSplitMaxK1 || !d_ss || !d_thetas || !d_out || !status)
        return fail(BILD_ERR_INVALID, "internal_logl_st_resident: bad arguments");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != ts->device) return fail(BILD_ERR_INVALID, "trajectory set lives on device %d, current device is %d", ts->device, dev);
    std::lock_guard<std::mutex> call_lock(m->call_mu);
    if (m->h_in_busy) HIP_TRY(hipEventSynchronize(m->h_in_event));
    m->h_in_busy = false;
    const size_t nseg = (size_t)n * K1;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        if (int rc = m->ws_in.reserve(2 * nseg * sizeof(int32_t))) return rc;
    }
    int32_t *d_start = (int32_t *)m->ws_in.ptr, *d_state = d_start + nseg; // the lists the walk kernel writes for the frame loop
    SplitIn sp;
    sp.d_ss = d_ss;
    sp.d_thetas = (const int8_t *)d_thetas;
    sp.status = status;
    if (stream) *stream = (void *)m->stream;
    int rc = launch_batch(*m, *ts, n, K1, d_start, d_state, nullptr, nullptr, flags, m->stream, d_out, &sp);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(m->h_in_event, m->stream)); // (the next host-buffer call must not reuse the lists before this one is through)
    m->h_in_busy = true;
    return BILD_OK;
}

// ------------------------------------------------------------------------------------
// exported
// ------------------------------------------------------------------------------------
extern "C" {

int bild_abi_version(void) { return BILD_AMD_ABI_VERSION; }

const char *bild_last_error(void) { return g_err.c_str(); }

void bild_set_last_error(const char *msg) { g_err = msg ? msg : ""; }

int bild_device_count(int *count)
{
    if (!count) return fail(BILD_ERR_INVALID, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *count = (e == hipSuccess) ? c : 0;
    return BILD_OK;
}

int bild_model_create(int N, int d, int S, const double *B, const double *G, const double *Sig, const double *M0,
                      const double *C0, const double *w, unsigned flags, bild_model **out)
{
    if (!out) return fail(BILD_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!B || !G || !Sig || !M0 || !C0 || !w) return fail(BILD_ERR_INVALID, "NULL model array");
    if (N < 1 || S < 1) return fail(BILD_ERR_INVALID, "need N >= 1 and S >= 1 (got N=%d, S=%d)", N, S);
    if (d < 1 || d > kDStore) return fail(BILD_ERR_UNSUPPORTED, "spatial dimension d=%d outside 1..%d", d, kDStore);
    if (S > 255) return fail(BILD_ERR_UNSUPPORTED, "S=%d states exceed 255", S);
    const size_t nn = (size_t)S * N * N, nd = (size_t)S * N * d;
    if (!all_finite(B, nn) || !all_finite(Sig, nn) || !all_finite(C0, nn) || !all_finite(G, nd) || !all_finite(M0, nd) ||
        !all_finite(w, (size_t)N))
        return fail(BILD_ERR_INVALID, "model arrays contain NaN or Inf");
    bild_model *m = new (std::nothrow) bild_model;
    if (!m) return fail(BILD_ERR_NOMEM, "out of memory");
    m->N = N;
    m->d = d;
    m->S = S;
    m->flags = flags;
    m->B.assign(B, B + nn);
    m->Sig.assign(Sig, Sig + nn);
    m->C0.assign(C0, C0 + nn);
    m->G.assign(G, G + nd);
    m->M0.assign(M0, M0 + nd);
    m->w.assign(w, w + N);
    int rc = analyse(*m);
    if (rc) {
        delete m;
        return rc;
    }
    *out = m;
    return BILD_OK;
}

int bild_model_destroy(bild_model *m)
{
    if (!m) return BILD_OK;
    for (int mode = 0; mode < 2; ++mode) {
        if (m->d_states[mode]) (void)hipFree(m->d_states[mode]);
        if (m->d_tab[mode]) (void)hipFree(m->d_tab[mode]);
    }
    if (m->h_in_busy) (void)hipEventSynchronize(m->h_in_event); // (a call nobody waited for still reads these blocks)
    if (m->h_in_event) (void)hipEventDestroy(m->h_in_event);
    m->ws_in.release();
    m->ws_out.release();
    m->ws_sched.release();
    for (bild_model::WorkSlot &sl : m->slots) {
        sl.ws_work.release();
        sl.ws_lists.release();
    }
    m->h_in.release();
    m->h_out.release();
    m->h_status.release();
    if (m->d_frames) (void)hipFree(m->d_frames);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return BILD_OK;
}

int bild_model_query(const bild_model *m, int what, int64_t *value)
{
    if (!m || !value) return fail(BILD_ERR_INVALID, "NULL argument");
    switch (what) {
    case BILD_Q_N: *value = m->N; break;
    case BILD_Q_D: *value = m->d; break;
    case BILD_Q_S: *value = m->S; break;
    case BILD_Q_MODAL_OK: *value = m->modal_ok; break;
    case BILD_Q_NP: *value = m->NP; break;
    case BILD_Q_NEFF: *value = m->n; break;
    case BILD_Q_HAS_G: *value = m->has_G; break;
    default: return fail(BILD_ERR_INVALID, "unknown query %d", what);
    }
    return BILD_OK;
}

int bild_model_export(const bild_model *m, int what, int s, int s2, double *buf, int64_t buf_len)
{
    if (!m || !buf) return fail(BILD_ERR_INVALID, "NULL argument");
    const int n = m->n, S = m->S;
    if (s < 0 || s >= S || s2 < 0 || s2 >= S) return fail(BILD_ERR_INVALID, "state index out of range");
    const double *src = nullptr;
    int64_t len = 0;
    switch (what) {
    case BILD_X_LAMBDA: src = m->lam.data() + (size_t)s * n; len = n; break;
    case BILD_X_SIGMA: src = m->sigd.data() + (size_t)s * n; len = n; break;
    case BILD_X_Q: src = m->Q.data() + (size_t)s * n * n; len = (int64_t)n * n; break;
    case BILD_X_WQ: src = m->wq.data() + (size_t)s * n; len = n; break;
    case BILD_X_R: src = m->R.data() + ((size_t)s2 * S + s) * n * n; len = (int64_t)n * n; break;
    case BILD_X_C0Q: src = m->C0q.data() + (size_t)s * n * n; len = (int64_t)n * n; break;
    case BILD_X_V: src = m->V.data(); len = (int64_t)m->N * n; break;
    default: return fail(BILD_ERR_INVALID, "unknown export %d", what);
    }
    if (what != BILD_X_V && !m->modal_ok) return fail(BILD_ERR_UNSUPPORTED, "modal analysis unavailable: %s", m->modal_why.c_str());
    if (buf_len < len) return fail(BILD_ERR_INVALID, "buffer too small: need %lld doubles", (long long)len);
    std::memcpy(buf, src, (size_t)len * sizeof(double));
    return BILD_OK;
}

int bild_trajset_create(const bild_model *m, int n_traj, const int32_t *T, const double *x, const double *loc_err,
                        bild_trajset **out)
{
    if (!out) return fail(BILD_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!m || !T || !x || !loc_err) return fail(BILD_ERR_INVALID, "NULL argument");
    if (n_traj < 1) return fail(BILD_ERR_INVALID, "need at least one trajectory");
    const int d = m->d;
    int64_t total = 0;
    for (int j = 0; j < n_traj; ++j) {
        if (T[j] < 1) return fail(BILD_ERR_INVALID, "trajectory %d has length %d < 1", j, T[j]);
        total += T[j];
    }
    for (int64_t i = 0; i < (int64_t)n_traj * d; ++i)
        if (!(loc_err[i] >= 0.0) || !std::isfinite(loc_err[i]))
            return fail(BILD_ERR_INVALID, "localization error must be finite and >= 0");

    int rc = ensure_device(*m);
    if (rc) return rc;

    bild_trajset *ts = new (std::nothrow) bild_trajset;
    if (!ts) return fail(BILD_ERR_NOMEM, "out of memory");
    ts->model = m;
    ts->n_traj = n_traj;
    ts->d = d;
    ts->device = m->device;
    ts->descs.resize(n_traj);

    // device copy of the data: a frame with any NaN coordinate is missing (pyx:178) -> all NaN
    // layout: per trajectory T rows + kPadRows padding rows (the kernels fetch up to kPadRows frames ahead), then zeros
    std::vector<double> xd((size_t)(total + (int64_t)kPadRows * n_traj) * d + kZeroPad, 0.0);
    const double qnan = std::nan("");
    int64_t off = 0;
    auto cleanup = [&](int code) {
        if (ts->d_x) (void)hipFree(ts->d_x);
        if (ts->d_descs) (void)hipFree(ts->d_descs);
        delete ts;
        return code;
    };
    // the largest steady-state variance of the observable w.x over the states: with s2 the scale of an innovation
    double wCw_max = 0.0;
    {
        const int N = m->N;
        for (int s_ = 0; s_ < m->S; ++s_) {
            const double *C0 = m->C0.data() + (size_t)s_ * N * N;
            double q = 0.0;
            for (int i = 0; i < N; ++i)
                for (int jj = 0; jj < N; ++jj) q += m->w[i] * C0[(size_t)i * N + jj] * m->w[jj];
            wCw_max = std::max(wCw_max, q);
        }
    }
    std::vector<double> xscales((size_t)n_traj, 0.0);
    hipError_t he = hipMalloc((void **)&ts->d_x, xd.size() * sizeof(double));
    if (he != hipSuccess) return cleanup(fail(BILD_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(he)));
    ts->d_zeros = ts->d_x + (size_t)(total + (int64_t)kPadRows * n_traj) * d;
    for (int j = 0; j < n_traj; ++j) {
        TrajDesc &td = ts->descs[j];
        std::memset(&td, 0, sizeof td);
        td.T = T[j];
        const int64_t doff = off + (int64_t)kPadRows * j; // device row offset: the padding rows of trajectories 0..j-1 precede
        td.x = ts->d_x + doff * d;
        int nvalid = 0;
        double xscale = 0.0;
        for (int t = 0; t < T[j]; ++t) {
            bool valid = true;
            for (int k = 0; k < d; ++k) valid &= !std::isnan(x[(off + t) * d + k]);
            for (int k = 0; k < d; ++k) {
                xd[(doff + t) * d + k] = valid ? x[(off + t) * d + k] : qnan;
                if (valid && std::isfinite(x[(off + t) * d + k])) xscale = std::max(xscale, std::fabs(x[(off + t) * d + k]));
            }
            nvalid += valid;
        }
        td.nvalid = nvalid;
        td.xscale = xscale;
        xscales[(size_t)j] = xscale;
        ts->all_valid = ts->all_valid && nvalid == T[j];
        // np.unique(err, return_inverse=True): sorted unique values (pyx:145)
        double uniq[kDStore];
        int nu = 0;
        for (int k = 0; k < d; ++k) {
            const double e = loc_err[(size_t)j * d + k];
            bool seen = false;
            for (int u = 0; u < nu; ++u) seen |= uniq[u] == e;
            if (!seen) uniq[nu++] = e;
        }
        std::sort(uniq, uniq + nu);
        // one covariance chain per distinct error; a task carries at most kDMax mean vectors, so an error shared by
        // more dimensions than that gets several chains (the covariance recursion is simply repeated)
        int nchains = 0;
        for (int u = 0; u < nu; ++u) {
            int in_chain = kDMax; // forces a new chain at the first dimension
            for (int k = 0; k < d; ++k) {
                if (loc_err[(size_t)j * d + k] != uniq[u]) continue;
                if (in_chain == kDMax) {
                    td.s2[nchains] = uniq[u] * uniq[u];
                    td.ndims[nchains] = 0;
                    ++nchains;
                    in_chain = 0;
                }
                td.dims[nchains - 1][td.ndims[nchains - 1]++] = k;
                ++in_chain;
            }
        }
        td.dstar = nchains;
        td.nuniq = nu;
        for (int u = 0; u < nchains; ++u) td.mscale[u] = std::min(xscales[(size_t)j], 6.0 * std::sqrt(td.s2[u] + wCw_max));
        nu = nchains;
        ts->dstar_max = std::max(ts->dstar_max, nu);
        for (int u = 0; u < nu; ++u) ts->means_max = std::max(ts->means_max, (int)td.ndims[u]);
        ts->Tmax = std::max(ts->Tmax, (int)T[j]);
        off += T[j];
    }
    {
        int64_t rec = 0;
        for (int j = 0; j < n_traj; ++j) {
            ts->descs[j].prefix_rec0 = rec;
            ts->descs[j].trans0 = rec * m->S; // S entries (one per new state) for every prefix record
            ts->descs[j].strans0 = rec * (m->S - 1); // S - 1 switches (one per OTHER state) for every prefix record
            rec += (int64_t)T[j] * m->S * ts->dstar_max;
        }
        ts->strans_entries = rec * (m->S - 1);
        ts->strans_records = 0;
        ts->prefix_records = rec;
        ts->trans_entries = rec * m->S;
    }
    he = hipMemcpy(ts->d_x, xd.data(), xd.size() * sizeof(double), hipMemcpyHostToDevice);
    if (he != hipSuccess) return cleanup(fail(BILD_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(he)));
    he = hipMalloc((void **)&ts->d_descs, (size_t)n_traj * sizeof(TrajDesc));
    if (he != hipSuccess) return cleanup(fail(BILD_ERR_NOMEM, "hipMalloc failed: %s", hipGetErrorString(he)));
    he = hipMemcpy(ts->d_descs, ts->descs.data(), (size_t)n_traj * sizeof(TrajDesc), hipMemcpyHostToDevice);
    if (he != hipSuccess) return cleanup(fail(BILD_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(he)));
    *out = ts;
    return BILD_OK;
}

int bild_trajset_expect(bild_trajset *ts, int64_t evaluations)
{
    if (!ts) return fail(BILD_ERR_INVALID, "NULL handle");
    if (evaluations < 0) return fail(BILD_ERR_INVALID, "a negative number of expected evaluations");
    if (ts->prefix_state != 0) return fail(BILD_ERR_INVALID, "the trajectory set has been evaluated on already: declare the expected use first");
    ts->expected_evals = evaluations;
    return BILD_OK;
}

int bild_trajset_destroy(bild_trajset *ts)
{
    if (!ts) return BILD_OK;
    if (ts->d_x) (void)hipFree(ts->d_x);
    if (ts->d_descs) (void)hipFree(ts->d_descs);
    tab_free(ts->d_prefix);
    tab_free(ts->d_prefix_L);
    tab_free(ts->d_tail_g);
    tab_free(ts->d_trans);
    tab_free(ts->d_trans2);
    tab_free(ts->d_strans);
    delete ts;
    return BILD_OK;
}

static int check_eval_args(const bild_model *m, const bild_trajset *ts, int64_t n, int K1)
{
    if (!m || !ts) return fail(BILD_ERR_INVALID, "NULL handle");
    if (ts->model != m) return fail(BILD_ERR_INVALID, "trajectory set belongs to a different model");
    if (n < 0) return fail(BILD_ERR_INVALID, "negative batch size");
    if (K1 < 1) return fail(BILD_ERR_INVALID, "need at least one segment per sample");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != ts->device) return fail(BILD_ERR_INVALID, "trajectory set lives on device %d, current device is %d", ts->device, dev);
    return BILD_OK;
}

int bild_logl_segments_device_ordered(const bild_model *m, const bild_trajset *ts, int64_t n, int K1,
                                      const int32_t *d_seg_start, const int32_t *d_seg_state, const int32_t *d_traj_id,
                                      const int32_t *d_order, unsigned flags, void *hip_stream, double *d_out)
{
    int rc = check_eval_args(m, ts, n, K1);
    if (rc) return rc;
    if (n == 0) return BILD_OK;
    if (!d_seg_start || !d_seg_state || !d_out) return fail(BILD_ERR_INVALID, "NULL device buffer");
    hipStream_t st = (hipStream_t)hip_stream;
    if (flags & BILD_VALIDATE_DEVICE) {
        // checked on the device, verdict read back BEFORE anything is launched: this variant of the call waits
        int *d_err = nullptr;
        int h_err[2] = {0, 0};
        const size_t err_bytes = (2 + (d_order ? (size_t)n : 0)) * sizeof(int);
        HIP_TRY(hipMallocAsync((void **)&d_err, err_bytes, st));
        HIP_TRY(hipMemsetAsync(d_err, 0, err_bytes, st));
        int lrc = launch_validate(d_seg_start, d_seg_state, d_traj_id, d_order, n, K1, m->S, ts->n_traj, d_err, (void *)st);
        hipError_t ce = lrc == 0 ? hipMemcpyAsync(h_err, d_err, sizeof h_err, hipMemcpyDeviceToHost, st) : (hipError_t)lrc;
        (void)hipFreeAsync(d_err, st);
        if (ce != hipSuccess) return fail(BILD_ERR_HIP, "descriptor check failed to run: %s", hipGetErrorString(ce));
        HIP_TRY(hipStreamSynchronize(st));
        static const char *const what[] = {"", "traj_id out of range", "first segment does not start at frame 0",
                                           "segment starts are decreasing", "state out of range",
                                           "launch order is not a permutation of the samples"};
        if (h_err[0] != 0)
            return fail(BILD_ERR_INVALID, "device descriptors rejected: %s (e.g. sample %d)", what[h_err[0] & 7], h_err[1]);
    }
    return launch_batch(*m, *ts, n, K1, d_seg_start, d_seg_state, d_traj_id, d_order, flags, st, d_out);
}

int bild_logl_st_device(const bild_model *m, const bild_trajset *ts, int64_t n, int K1, const double *d_ss, const uint8_t *d_thetas,
                        const int32_t *d_traj_id, unsigned flags, void *hip_stream, double *d_out, int32_t *d_status)
{
    int rc = check_eval_args(m, ts, n, K1);
    if (rc) return rc;
    if (n == 0) return BILD_OK;
    if (!d_ss || !d_thetas || !d_out) return fail(BILD_ERR_INVALID, "NULL device buffer");
    if (K1 > kSplitMaxK1) return fail(BILD_ERR_UNSUPPORTED, "(s, theta) rows resident in HBM: at most %d segments per candidate (got %d)", kSplitMaxK1, K1);
    SplitIn sp;
    sp.d_ss = d_ss;
    sp.d_thetas = (const int8_t *)d_thetas;
    // the verdict on the rows: the caller's word, or a scratch word of the model nobody reads (rows that are no points on
    // the simplex still get NaN)
    if (d_status) {
        sp.status = d_status;
    } else {
        std::lock_guard<std::mutex> lk(m->mu);
        if (!m->h_status.ptr) {
            if ((rc = m->h_status.reserve(64))) return rc;
            std::memset(m->h_status.ptr, 0, 64);
        }
        sp.status = (int32_t *)m->h_status.ptr + 8;
    }
    return launch_batch(*m, *ts, n, K1, nullptr, nullptr, d_traj_id, nullptr, flags, (hipStream_t)hip_stream, d_out, &sp);
}

int bild_logl_segments_device(const bild_model *m, const bild_trajset *ts, int64_t n, int K1, const int32_t *d_seg_start,
                              const int32_t *d_seg_state, const int32_t *d_traj_id, unsigned flags, void *hip_stream,
                              double *d_out)
{
    return bild_logl_segments_device_ordered(m, ts, n, K1, d_seg_start, d_seg_state, d_traj_id, nullptr, flags, hip_stream, d_out);
}

int bild_schedule_segments(const bild_model *m, const bild_trajset *ts, int64_t n, int K1, const int32_t *seg_start,
                           const int32_t *seg_state, const int32_t *traj_id, unsigned flags, int32_t *order)
{
    if (!m || !ts || !order) return fail(BILD_ERR_INVALID, "NULL argument");
    if (ts->model != m) return fail(BILD_ERR_INVALID, "trajectory set belongs to a different model");
    if (n < 0 || K1 < 1) return fail(BILD_ERR_INVALID, "bad sizes");
    if (n > 0 && !seg_start) return fail(BILD_ERR_INVALID, "NULL buffer");
    if (traj_id)
        for (int64_t r = 0; r < n; ++r)
            if (traj_id[r] < 0 || traj_id[r] >= ts->n_traj) return fail(BILD_ERR_INVALID, "traj_id out of range");
    // the same checks as bild_logl_segments: the estimate of a candidate's work indexes arrays with these numbers
    for (int64_t r = 0; r < n; ++r) {
        const int32_t *a = seg_start + r * K1;
        if (a[0] != 0) return fail(BILD_ERR_INVALID, "seg_start[%lld][0] must be 0", (long long)r);
        for (int i = 1; i < K1; ++i)
            if (a[i] < a[i - 1] || a[i] < 1)
                return fail(BILD_ERR_INVALID, "segment starts of sample %lld are decreasing (or a later segment starts at frame 0)", (long long)r);
        if (seg_state)
            for (int i = 0; i < K1; ++i)
                if (seg_state[r * K1 + i] < 0 || seg_state[r * K1 + i] >= m->S)
                    return fail(BILD_ERR_INVALID, "state %d out of range at sample %lld", seg_state[r * K1 + i], (long long)r);
    }
    if (!schedule(*m, *ts, n, K1, seg_start, seg_state, traj_id, flags, order, false))
        for (int64_t r = 0; r < n; ++r) order[r] = (int32_t)r;
    return BILD_OK;
}

int bild_frames_executed(const bild_model *m, const bild_trajset *ts, int64_t n, int K1, const int32_t *seg_start,
                         const int32_t *traj_id, const int32_t *order, unsigned flags, double *frames_total, double *frames_run)
{
    if (!m || !ts || !frames_total || !frames_run || (n > 0 && !seg_start)) return fail(BILD_ERR_INVALID, "NULL argument");
    int mode;
    int rc = pick_mode(*m, flags, &mode);
    if (rc) return rc;
    const bool prefix = ts->prefix_state == 1 && mode == kModal && !m->wide && !m->mid && !(flags & BILD_NO_PREFIX);
    Geometry geom{};
    int tpw = 1;
    if (prefix && geometry_for(m->NPm[mode], mode, n * ts->dstar_max, ts->means_max, &geom)) tpw = geom.tasks_per_wave();
    double total = 0.0, run = 0.0;
    const int ds = ts->dstar_max;
    // tasks are (slot, chain) pairs, tpw consecutive tasks share a wave and start at the earliest first switch among them
    const int64_t ntasks = n * ds;
    for (int64_t w0 = 0; w0 < ntasks; w0 += tpw) {
        int tw = INT_MAX;
        for (int64_t t = w0; t < std::min(ntasks, w0 + tpw); ++t) {
            const int64_t r = order ? order[t / ds] : t / ds;
            const TrajDesc &td = ts->descs[traj_id ? traj_id[r] : 0];
            if ((int)(t % ds) >= td.dstar) continue;
            int t0 = K1 > 1 ? seg_start[r * K1 + 1] : td.T;
            t0 = t0 < 1 ? 1 : (t0 > td.T ? td.T : t0);
            tw = std::min(tw, t0);
        }
        for (int64_t t = w0; t < std::min(ntasks, w0 + tpw); ++t) {
            const int64_t r = order ? order[t / ds] : t / ds;
            const TrajDesc &td = ts->descs[traj_id ? traj_id[r] : 0];
            if ((int)(t % ds) >= td.dstar) continue;
            total += td.T;
            run += prefix ? td.T - tw : td.T;
        }
    }
    *frames_total = total;
    *frames_run = run;
    return BILD_OK;
}

int bild_debug_frames_per_task(int32_t *d_buffer)
{
    g_frames_task.store(d_buffer);
    return BILD_OK;
}

int bild_frames_run_read(const bild_model *m, int64_t *frames)
{
    if (!m || !frames) return fail(BILD_ERR_INVALID, "NULL argument");
    *frames = 0;
    if (!m->d_frames) return BILD_OK;
    unsigned long long v[kFrameCounters];
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(v, m->d_frames, sizeof v, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(m->d_frames, 0, sizeof v));
    unsigned long long tot = 0;
    for (unsigned long long w : v) tot += w;
    *frames = (int64_t)tot;
    return BILD_OK;
}

int bild_prefix_info(const bild_trajset *ts, int64_t *bytes, double *build_ms)
{
    if (!ts) return fail(BILD_ERR_INVALID, "NULL handle");
    const bool built = ts->prefix_state == 1, trans = ts->trans_state == 1, pairs = ts->trans2_state == 1;
    if (bytes)
        *bytes = (built ? ts->prefix_records * prefix_record_doubles(ts->model->NPm[kModal]) * (int64_t)sizeof(double) : 0) +
                 (trans ? ts->trans_entries * (int64_t)sizeof(TransEntry) : 0) +
                 (pairs ? ts->trans2_entries * (int64_t)sizeof(TransEntry) : 0) +
                 (trans && ts->d_strans ? ts->strans_records * prefix_record_doubles(ts->model->NPm[kModal]) * (int64_t)sizeof(double) : 0);
    if (build_ms)
        *build_ms = (built ? ts->prefix_build_ms : 0.0) + (trans ? ts->trans_build_ms : 0.0) + (pairs ? ts->trans2_build_ms : 0.0);
    return BILD_OK;
}

int bild_logl_segments(const bild_model *m, const bild_trajset *ts, int64_t n, int K1, const int32_t *seg_start,
                       const int32_t *seg_state, const int32_t *traj_id, unsigned flags, double *out)
{
    int rc = check_eval_args(m, ts, n, K1);
    if (rc) return rc;
    if (n == 0) return BILD_OK;
    if (!seg_start || !seg_state || !out) return fail(BILD_ERR_INVALID, "NULL buffer");
    const int S = m->S;
    return run_staged(m, ts, n, K1, traj_id, flags, out, nullptr, nullptr, false, [&](char *payload) -> int {
        int32_t *h_start = (int32_t *)payload, *h_state = h_start + (size_t)n * K1;
        for (int64_t r = 0; r < n; ++r) {
            const int32_t *a = seg_start + r * K1, *b = seg_state + r * K1;
            if (a[0] != 0) return fail(BILD_ERR_INVALID, "seg_start[%lld][0] must be 0", (long long)r);
            for (int i = 0; i < K1; ++i) {
                if (b[i] < 0 || b[i] >= S) return fail(BILD_ERR_INVALID, "state %d out of range at sample %lld", b[i], (long long)r);
                if (i > 0 && (a[i] < a[i - 1] || a[i] < 1))
                    return fail(BILD_ERR_INVALID, "segment starts of sample %lld are decreasing (or a later segment starts at frame 0)", (long long)r);
                h_start[r * K1 + i] = a[i];
                h_state[r * K1 + i] = b[i];
            }
        }
        return BILD_OK;
    });
}

// The (s, theta) parametrisation of the sampler itself.  Switch frames as reference bild/amis.py:685-688 computes
// them, in the same floating-point operations:  np.cumsum(s)[:-1] is a sequential sum, times (T - 1) is one
// multiplication, floor, + 1.  No contraction of the multiply into the add: x86-64 baseline code has no fused
// instruction, and the pragma keeps it that way on any other target (file scope: it covers the lambda below).
#pragma clang fp contract(off)
static int st_row(const double *s, const int64_t *th, int K1, int S, double Tm1, int64_t r, int32_t *a, int32_t *b)
{
    a[0] = 0;
    double acc = 0.0;
    int32_t prev = 0;
    bool ok = true;
    for (int i = 0; i < K1; ++i) {
        ok &= (uint64_t)th[i] < (uint64_t)S;
        b[i] = (int32_t)th[i];
        if (i + 1 < K1) {
            acc = acc + s[i];
            const double pos = acc * Tm1;
            // floor(pos) for 0 <= pos < 2^31 is the truncating conversion (one SSE2 instruction, no libm call).  The
            // intrinsic is defined for every input: NaN and out-of-range values give INT64_MIN, which the unsigned range
            // test below rejects.  A position in (-1, 0) would truncate to 0 where np.floor gives -1 (the reference then
            // builds a profile whose FIRST interval is empty, amis.py:685-693): refused, like every negative position.
            const int64_t fl = _mm_cvttsd_si64(_mm_set_sd(pos));
            ok &= (uint64_t)fl < 2147483646ull;
            ok &= !(pos < 0.0);
            const int32_t idx = (int32_t)fl + 1;
            ok &= idx >= prev;
            prev = idx;
            a[i + 1] = idx;
        }
    }
    if (ok) return BILD_OK;
    for (int i = 0; i < K1; ++i)
        if (th[i] < 0 || th[i] >= S)
            return fail(BILD_ERR_INVALID, "state %lld out of range at sample %lld", (long long)th[i], (long long)r);
    return fail(BILD_ERR_INVALID, "interval lengths of sample %lld are not non-negative finite numbers (of a point on the simplex)", (long long)r);
}

int bild_segments_from_st(int64_t n, int K1, int n_states, const int32_t *T, int64_t T_stride, const double *ss,
                          const int64_t *thetas, int32_t *seg_start, int32_t *seg_state)
{
    if (n < 0 || K1 < 1 || n_states < 1) return fail(BILD_ERR_INVALID, "bad sizes");
    if (n == 0) return BILD_OK;
    if (!T || !ss || !thetas || !seg_start || !seg_state) return fail(BILD_ERR_INVALID, "NULL buffer");
    for (int64_t r = 0; r < n; ++r) {
        const int32_t Tr = T[r * T_stride];
        if (Tr < 1) return fail(BILD_ERR_INVALID, "trajectory length %d < 1", Tr);
        int rc = st_row(ss + r * K1, thetas + r * K1, K1, n_states, (double)(Tr - 1), r, seg_start + r * K1, seg_state + r * K1);
        if (rc) return rc;
    }
    return BILD_OK;
}

static int logl_st_impl(const bild_model *m, const bild_trajset *ts, int64_t n, int K1, const double *ss, const int64_t *thetas,
                        const int32_t *traj_id, unsigned flags, double *out, double *d_out, void *hip_stream)
{
    int rc = check_eval_args(m, ts, n, K1);
    if (rc) return rc;
    if (n == 0) return BILD_OK;
    if (!ss || !thetas || (!out && !d_out)) return fail(BILD_ERR_INVALID, "NULL buffer");
    const int S = m->S;
    // The rows go up as they are -- float64 interval lengths, states narrowed to one byte -- and the walk kernel turns
    // them into switch frames on the device (walk.hip; same operations as st_row below, bit for bit).  Lists of more
    // segments than that kernel holds in registers are converted here.
    const bool host_convert = config().st_on_host;
    if (K1 <= kSplitMaxK1 && !host_convert)
        return run_staged(m, ts, n, K1, traj_id, flags, out, d_out, (hipStream_t)hip_stream, true, [&](char *payload) -> int {
            const size_t nseg = (size_t)n * K1;
            std::memcpy(payload, ss, nseg * sizeof(double));
            uint8_t *th8 = (uint8_t *)(payload + nseg * sizeof(double));
            uint64_t bad = 0;
            for (size_t i = 0; i < nseg; ++i) {
                bad |= (uint64_t)((uint64_t)thetas[i] >= (uint64_t)S);
                th8[i] = (uint8_t)thetas[i];
            }
            if (bad)
                for (size_t i = 0; i < nseg; ++i)
                    if (thetas[i] < 0 || thetas[i] >= S)
                        return fail(BILD_ERR_INVALID, "state %lld out of range at sample %lld", (long long)thetas[i], (long long)(i / K1));
            return BILD_OK;
        });
    return run_staged(m, ts, n, K1, traj_id, flags, out, d_out, (hipStream_t)hip_stream, false, [&](char *payload) -> int {
        int32_t *h_start = (int32_t *)payload, *h_state = h_start + (size_t)n * K1;
        for (int64_t r = 0; r < n; ++r) {
            const double Tm1 = (double)(ts->descs[traj_id ? traj_id[r] : 0].T - 1);
            int rc2 = st_row(ss + r * K1, thetas + r * K1, K1, S, Tm1, r, h_start + r * K1, h_state + r * K1);
            if (rc2) return rc2;
        }
        return BILD_OK;
    });
}

int bild_logl_st(const bild_model *m, const bild_trajset *ts, int64_t n, int K1, const double *ss, const int64_t *thetas,
                 const int32_t *traj_id, unsigned flags, double *out)
{
    if (!out) return fail(BILD_ERR_INVALID, "NULL buffer");
    return logl_st_impl(m, ts, n, K1, ss, thetas, traj_id, flags, out, nullptr, nullptr);
}

int bild_logl_st_to_device(const bild_model *m, const bild_trajset *ts, int64_t n, int K1, const double *ss,
                           const int64_t *thetas, const int32_t *traj_id, unsigned flags, void *hip_stream, double *d_out)
{
    if (!d_out) return fail(BILD_ERR_INVALID, "NULL device buffer");
    return logl_st_impl(m, ts, n, K1, ss, thetas, traj_id, flags, nullptr, d_out, hip_stream);
}

int bild_logl_profiles(const bild_model *m, const bild_trajset *ts, int64_t n, int64_t ld, const int32_t *states,
                       const int32_t *traj_id, unsigned flags, double *out)
{
    if (!m || !ts) return fail(BILD_ERR_INVALID, "NULL handle");
    if (n < 0) return fail(BILD_ERR_INVALID, "negative batch size");
    if (n == 0) return BILD_OK;
    if (!states || !out) return fail(BILD_ERR_INVALID, "NULL buffer");
    // run-length encode
    std::vector<int> nseg((size_t)n);
    int K1 = 1;
    for (int64_t r = 0; r < n; ++r) {
        const int tj = traj_id ? traj_id[r] : 0;
        if (tj < 0 || tj >= ts->n_traj) return fail(BILD_ERR_INVALID, "traj_id[%lld]=%d out of range", (long long)r, tj);
        const int T = ts->descs[tj].T;
        if (ld < T) return fail(BILD_ERR_INVALID, "profile row stride %lld shorter than trajectory length %d", (long long)ld, T);
        int c = 1;
        for (int t = 1; t < T; ++t) c += states[r * ld + t] != states[r * ld + t - 1];
        nseg[r] = c;
        K1 = std::max(K1, c);
    }
    std::vector<int32_t> seg_start((size_t)n * K1), seg_state((size_t)n * K1);
    for (int64_t r = 0; r < n; ++r) {
        const int tj = traj_id ? traj_id[r] : 0;
        const int T = ts->descs[tj].T;
        int c = 0;
        seg_start[r * K1] = 0;
        seg_state[r * K1] = states[r * ld];
        for (int t = 1; t < T; ++t)
            if (states[r * ld + t] != states[r * ld + t - 1]) {
                ++c;
                seg_start[r * K1 + c] = t;
                seg_state[r * K1 + c] = states[r * ld + t];
            }
        for (int i = c + 1; i < K1; ++i) {
            seg_start[r * K1 + i] = INT_MAX;
            seg_state[r * K1 + i] = seg_state[r * K1 + c];
        }
    }
    return bild_logl_segments(m, ts, n, K1, seg_start.data(), seg_state.data(), traj_id, flags, out);
}

int bild_flop_count(const bild_model *m, const bild_trajset *ts, int64_t n, const int32_t *traj_id, unsigned flags,
                    double *canonical, double *executed)
{
    if (!m || !ts || !canonical || !executed) return fail(BILD_ERR_INVALID, "NULL argument");
    int mode;
    int rc = pick_mode(*m, flags, &mode);
    if (rc) return rc;
    const double N = m->N, d = m->d, nr = m->n;
    double can = 0.0, exe = 0.0;
    for (int64_t r = 0; r < n; ++r) {
        const int tj = traj_id ? traj_id[r] : 0;
        if (tj < 0 || tj >= ts->n_traj) return fail(BILD_ERR_INVALID, "traj_id out of range");
        const TrajDesc &td = ts->descs[tj];
        const double T = td.T, Tv = td.nvalid, ds = td.dstar, du = td.nuniq;
        can += (T - 1) * (4 * N * N * N * du + 2 * N * N * d) + Tv * ((4 * N * N + 3 * N) * du + 4 * N * d);
        if (mode == kDense)
            exe += (T - 1) * (4 * nr * nr * nr * ds + 2 * nr * nr * d) + Tv * ((4 * nr * nr + 3 * nr) * ds + 4 * nr * d);
        else // elementwise predict (2 mul / entry) + update; basis changes at state switches not counted
            exe += (T - 1) * (2 * nr * nr * ds + nr * ds + 2 * nr * d) + Tv * ((4 * nr * nr + 3 * nr) * ds + 4 * nr * d);
    }
    *canonical = can;
    *executed = exe;
    return BILD_OK;
}

int bild_logl_st_status(const bild_model *m, int64_t *bad_row)
{
    if (!m) return fail(BILD_ERR_INVALID, "NULL handle");
    std::lock_guard<std::mutex> call_lock(m->call_mu);
    if (m->h_in_busy) HIP_TRY(hipEventSynchronize(m->h_in_event));
    m->h_in_busy = false;
    if (bad_row) *bad_row = -1;
    if (!m->h_status.ptr) return BILD_OK;
    int32_t *h = (int32_t *)m->h_status.ptr;
    if (h[0] == 0) return BILD_OK;
    const int row = h[1];
    h[0] = h[1] = 0;
    if (bad_row) *bad_row = row;
    return fail(BILD_ERR_INVALID, "interval lengths of sample %d are not non-negative finite numbers (of a point on the simplex)", row);
}

int bild_kernel_timing_read_walk(double *total_ms, int64_t *launches)
{
    if (!total_ms || !launches) return fail(BILD_ERR_INVALID, "NULL argument");
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    {
        std::lock_guard<std::mutex> lk(g_time_mu);
        ev.swap(g_walk_events);
    }
    double tot = 0.0;
    for (auto &pr : ev) {
        HIP_TRY(hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        tot += ms;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    *total_ms = tot;
    *launches = (int64_t)ev.size();
    return BILD_OK;
}

int bild_kernel_timing(int enable)
{
    std::lock_guard<std::mutex> lk(g_time_mu);
    g_time_on = enable < 0 ? 0 : enable;
    g_time_count = 0;
    return BILD_OK;
}

int bild_kernel_timing_read(double *total_ms, int64_t *launches, char *name, int name_len)
{
    if (!total_ms || !launches) return fail(BILD_ERR_INVALID, "NULL argument");
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::string nm;
    {
        std::lock_guard<std::mutex> lk(g_time_mu);
        ev.swap(g_time_events);
        nm = g_time_name;
    }
    double tot = 0.0;
    for (auto &pr : ev) {
        HIP_TRY(hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        tot += ms;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    *total_ms = tot;
    *launches = (int64_t)ev.size();
    if (name && name_len > 0) {
        std::snprintf(name, (size_t)name_len, "%s", nm.c_str());
    }
    return BILD_OK;
}

} // extern "C"
