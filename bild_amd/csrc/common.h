// Shared host/device definitions for the Rouse Kalman-filter log-likelihood kernels.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace bild {

constexpr int kDMax = 3;      // mean vectors ONE task carries (the reference default is d = 3, models.py:222)
constexpr int kDStore = 8;    // spatial dimensions supported; with d > 3 a trajectory's dimensions are spread over
constexpr int kChains = 8;    // several "covariance chains" of <= kDMax dimensions each (chains with equal localization
                              // error repeat the covariance recursion: d = 4 costs two tasks per sample, d = 7 three)
constexpr int kPadRows = 4;   // rows behind each trajectory on the device: the frame loops fetch that many frames ahead
constexpr int kMaxWaves = 4;  // wavefronts per workgroup (fewer for long chains: LDS capacity)
constexpr int kMaxNP = 32;    // largest padded chain length with a register-resident kernel (kernels.hip)
constexpr int kWideMaxNP = 128; // largest padded chain length at all: LDS-resident state (wide.hip), modal path only

enum Mode : int { kDense = 0, kModal = 1 };

// Per-state block of the packed model ("state blob"), in doubles.  Vectors are padded to NP
// rows with zeros, matrices to NP x NP.
//   dense: lam/sig unused, wq = w,      G/M0/C0 as given
//   modal: lam = eig(B), wq = Q^T w, sig = diag(Q^T Sig Q), G = Q^T G, M0 = Q^T M0, C0 = Q^T C0 Q
struct StateBlock {
    static constexpr int lam(int)     { return 0; }
    static constexpr int wq(int NP)   { return NP; }
    static constexpr int sig(int NP)  { return 2 * NP; }
    static constexpr int G(int NP)    { return 3 * NP; }                 // [kDStore][NP]
    static constexpr int M0(int NP)   { return (3 + kDStore) * NP; }     // [kDStore][NP]
    static constexpr int C0(int NP)   { return (3 + 2 * kDStore) * NP; } // [NP][NP]
    static constexpr int size(int NP) { return (3 + 2 * kDStore) * NP + NP * NP; }
};

// Matrices that live in LDS for the whole kernel ("table"): stride padded by 2 doubles so the
// same element of different matrices falls into different LDS banks.
//   dense: B[s] at slot s, Sig[s] at slot S + s                     (2 S matrices)
//   modal: R[s2][s] = Q[s2]^T Q[s] at slot s2 * S + s               (S*S matrices), or, for many states,
//          Q[s] at slot s and Q[s]^T at slot S + s                  (2 S matrices, basis change in two steps)
constexpr int table_stride(int NP) { return NP * NP + 2; }

struct TrajDesc {
    const double *x; // device, (T + kPadRows) x d: padding rows behind the data; every coordinate of a missing frame is NaN
    int32_t T;
    int32_t dstar;           // number of covariance chains: distinct localization errors (pyx:145), split to <= kDMax dims
    double s2[kChains];      // squared localization error of chain e (ascending; repeated where a chain was split)
    int32_t ndims[kChains];  // dims that use covariance chain e (<= kDMax)
    int32_t dims[kChains][kDMax];
    int32_t nvalid;          // frames with data
    int32_t nuniq;           // distinct localization errors (the reference's d*, pyx:145), <= dstar
    int64_t prefix_rec0;     // first record of this trajectory in the prefix table (records, see prefix_record_doubles)
    double xscale;           // largest |coordinate| of the data
    double mscale[kChains];  // absolute floor of the mean-vector comparison of chain e: min(xscale, 6 sqrt(S_e)), S_e the largest
                             // steady-state innovation variance of the chain (see logl_kernel: convergence check)
    int64_t trans0;          // first entry of this trajectory in the transient table
    int64_t strans0;         // first record of this trajectory in the transient state table
};

// Transient table (vector kernels, modal path, beside the prefix table): entry of (trajectory, chain e, old state s, new
// state sn, frame t) at  trans0 + ((e * S + s) * S + sn) * T + t  -- a filter that sits on the switch-free filter of s in
// front of frame t and switches to sn there needs m frames to converge onto the switch-free filter of sn, and over those
// frames its log-likelihood exceeds that filter's by c.  m = T - t for a transient that runs to the end of the trajectory
// without converging (usable by a last switch only); m <= 0: no entry.
struct TransEntry {
    double c;
    int32_t m;
    int32_t pad;
};
// Pair table (second level): two switches s -> sn at frame t and sn -> sm at frame t + g, g < gap_max, closer together than
// the first transient needs -- entry at  (trans0 * S + (((e * S + s) * S + sn) * S + sm) * T + t) * gap_max + g:  m frames from
// t until the filter sits on the switch-free filter of sm, and c = what those frames add beyond that filter's sums from
// t on.  Chains of three and more switches are run frame by frame.

// Prefix table (vector kernels, modal path): the filter state after frame t of a task that has not switched yet
// depends only on (trajectory, covariance chain e, initial state s, t) -- not on the candidate profile.  It is
// computed once per trajectory set by the likelihood kernel itself (one task per (trajectory, e, s) that stores its
// state after every frame) and every candidate starts from the record in front of its first switch.
// Record of frame t, in doubles: NP + kDMax columns of NP rows ([C | M] in the modal basis of s, column layout of
// the kernels), then sum e^2/S per mean column (kDMax), P, E (mantissa / exponent of the running product of S), the running
// log-likelihood L of the frames so far, the number of observed frames so far, pad.
// Records of trajectory j: prefix_rec0 + ((e * S + s) * T + t).
constexpr int prefix_record_doubles(int NP) { return (NP + kDMax) * NP + kDMax + 2 + 3; }
// Transient state table (vector kernels, modal path).  The candidates that build the transient table run, for every
// (trajectory, chain e, old state s, new state sn != s, frame t), the frames t, t + 1, ... in state sn from the switch-free
// filter of s -- exactly what a candidate with a switch s -> sn at t runs until its NEXT switch.  They leave their state
// after g = 1 .. kStateGap - 1 frames (same record layout as the prefix table: [C | M], sums of e^2/S, mantissa / exponent
// of the product of S, observed frames; the running log-likelihood is not used).  A chain of close switches -- a
// transient that has not converged when the next switch comes -- then starts at its second switch from the record
// (t, g = distance of the two switches) and continues the record's accumulators: the same numbers the candidate itself
// would have produced, bit for bit, without running the g frames in between.
// The gaps are covered up to the set's longest converged transient (a chain of close switches begins with a gap shorter
// than its first transient), and only every `sstride`-th gap has a record (g = 1, 1 + sstride, ...: a chain starts from the
// last record at or in front of its second switch and runs the <= sstride - 1 frames in between itself -- a third of the
// memory for 0.25 us more per chain at the default stride of 3).  Records of (trajectory, e, s, sn, t), snq per entry:
//   (strans0 + (((e * S + s) * (S - 1) + (sn - (sn > s))) * T + t)) * snq + (g - 1) / sstride.
constexpr int kStateGap = 255;  // largest gap that can ever be covered (BILD_STATES_MAX_GAP: 128 by default)
constexpr int kStateStride = 3; // BILD_STATES_STRIDE

struct KParams {
    const double *states; // S state blocks
    const double *tab;    // table matrices
    int32_t tab_doubles;
    int32_t S, d, has_G;
    int32_t all_valid; // no trajectory of the set has a missing frame
    const TrajDesc *trajs;
    int64_t ntasks; // n samples * dstar_max
    int32_t dstar_max;
    int32_t K1;
    const int32_t *seg_start;
    const int32_t *seg_state;
    const int32_t *traj_id; // may be null
    const double *zeros;    // a few zero doubles (stride-0 source for non-mean columns)
    double *out;            // ntasks partial results
    // vector kernels only (kernels.hip); the others ignore them
    const int32_t *order;   // slot -> sample (launch order chosen by the host scheduler), null: identity
    const double *prefix;   // prefix table to start from, null: every task starts at frame 0
    double *prefix_dump;    // non-null: this launch BUILDS the table (one task per (trajectory, e, s), K1 = 1)
    int32_t no_jump;        // with a prefix table: run every frame behind the first switch (no convergence jumps)
    int32_t tab_factored;   // modal table = Q[s] at slot s, Q[s]^T at slot S + s (many states) instead of R[s2][s] at s2*S + s
    unsigned long long *frames_run; // non-null: tasks add the number of frames they ran themselves (bench accounting)
    int32_t *frames_task;   // non-null (diagnostics): frames run by each task, indexed like `out`
    const TransEntry *trans; // transient table to take whole transients from, null: every transient is run
    TransEntry *trans_dump;  // non-null: this launch BUILDS the transient table (one task per entry, K1 = 2)
    int32_t m_typ;           // with `trans`: typical frames-to-convergence of a transient (wave priority by expected work)
    const TransEntry *trans2; // pair table: two switches less than gap_max frames apart, taken as one transient
    TransEntry *trans2_dump;  // non-null: this launch BUILDS the pair table (one task per entry, K1 = 3)
    int32_t gap_max;          // gaps 1 .. gap_max - 1 have entries in the pair table
    int32_t walk_lds;         // the launch has kWalkDoubles of LDS per task behind the segment lists (see logl_kernel: walk plan)
    double *prefix_L_dump;    // with prefix_dump: the running log-likelihood of every record once more, densely (8 B per record)
    // transient STATE table (see below): a chain of close switches starts at its SECOND switch, from the state the first
    // transient has reached there
    const double *strans;
    double *strans_dump;      // non-null: this launch (the one that builds the transient table) also fills the state table
    int32_t sgap;             // gaps 1 .. sgap - 1 behind a switch are covered ...
    int32_t sstride, snq;     // ... by a record for every sstride-th of them (g = 1 + q * sstride, q < snq records per switch)
    // first-order tail (tail.hip): per prefix record kDMax x NP doubles g -- what a deviation of the means from the record does to
    // the log-likelihood of all later frames; null: a transient is run until its means have converged too
    const double *tail_g;
    double tail_tol;     // a mean column may be this far (relative) from the table's when the tail is taken
    int32_t tail_margin; // frames beyond the table's own transient before the next switch may come
    // work lists (walk.hip): this launch runs only the tasks the table walk could not finish -- kWorkBuckets lists of
    // `work_cap` task indices (index into `out`) each, heaviest bucket last, with their lengths in work_counts
    const int32_t *work;
    const int32_t *work_counts;
    int64_t work_cap;
};

// Work lists between the table walk (walk.hip) and the frame loop (kernels.hip).  A task lands in the bucket of the
// frames it is expected to run (estimate / kWorkBucketFrames, capped): the frame-loop launch deals the buckets out
// heaviest first -- a counting sort that costs nothing.
constexpr int kWorkBuckets = 16;
constexpr int kWorkBucketFrames = 20;

struct WalkParams {
    const TrajDesc *trajs;
    int32_t S, dstar_max, K1;
    int64_t n; // candidates; tasks = n * dstar_max, task = candidate * dstar_max + chain
    // the candidates: run-length segments (seg_start / seg_state, n x K1) ...
    const int32_t *seg_start;
    const int32_t *seg_state;
    // ... or the sampler's own (s, theta) (bild/amis.py:717-739): ss n x K1 float64, thetas n x K1 int8.  The switch
    // frames are computed here as FixedkSampler.st2profile does (amis.py:685-688) and the lists of the tasks that go on
    // to the frame loop (all lists with `convert_all`) are written to seg_out_start / seg_out_state (n x K1).
    const double *ss;
    const int8_t *thetas;
    int32_t *seg_out_start;
    int32_t *seg_out_state;
    int32_t convert_all; // no tables to walk: convert every list, finish nothing
    int32_t no_lists;    // the host has proved that no task of this launch can need a frame (lists of <= 3 segments on a set whose
                         // tables cover every candidate of two switches): no frame loop follows; a task that would be listed gets NaN
    int32_t *status;     // (s, theta) input: [0] != 0 when a row is not a point on the simplex / a state is out of range, [1] such a row
    const int32_t *traj_id; // may be null
    const double *Lc;       // running log-likelihood of the switch-free filters, one double per prefix record
    const TransEntry *trans;
    const TransEntry *trans2; // may be null
    int32_t gap_max, m_typ;
    double *out;            // results of the tasks that need no frame
    int32_t *work;          // kWorkBuckets x work_cap
    int32_t *work_counts;   // kWorkBuckets, zero at launch
    int32_t *work_counts_next; // non-null: the set of counters the next launch on this workspace will use; zeroed here
    int64_t work_cap;
    unsigned long long *tasks_done; // non-null (bench accounting): tasks finished by the walk
    int32_t *frames_task;           // non-null (diagnostics): a task finished here ran no frame
    int32_t debug;                  // timing experiments only (BILD_WALK_DEBUG; wrong results): 1 no pair loads, 2 no append, 4 no loads at all
};
int launch_walk(const WalkParams &p, void *stream, void *ev_start = nullptr, void *ev_stop = nullptr);
int launch_tail(const TrajDesc *d_trajs, int n_traj, int S, int NP, int d, int dstar_max, const double *d_states, const double *d_prefix,
                const int64_t *d_first, int64_t total, double *d_gain, double *d_tail_g, void *stream);
int launch_mark_refused_rows(const int32_t *seg_start, int K1, int64_t n, double *out, void *stream);

// launch geometry for a padded chain length
struct Geometry {
    int NP, CPL, G; // padded rows, columns per lane, lanes per group
    int W;          // wavefronts per workgroup
    int OCC;        // wavefronts per SIMD the register allocation is bounded for
    int id;         // index into the compiled-kernel table
    int modes;      // paths that may pick it automatically: 1 dense, 2 modal; 0: only through BILD_GEOM
    int tasks_per_wave() const { return 64 / G; }
    // mean vectors a group can carry next to the NP covariance columns (a covariance chain needs one per
    // dimension that shares its localization error: d for d* = 1, fewer when the errors differ)
    int mean_slots() const { return CPL * G - NP < kDMax ? CPL * G - NP : kDMax; }
};

// doubles of LDS one group needs: image of X*[C|M], NP+kDMax columns of NP rows
constexpr int group_image_doubles(int NP) { return (NP + kDMax) * NP; }
// LDS layout of the vector kernels, in doubles: [matrix tables (tab_doubles)] [per state: lam | wq | sig (3 NP each)]
// [per group: product image] [per group: the task's segment list, kSegLds (start, state) pairs]
constexpr int kFrameCounters = 256; // words of the frames-run counter (KParams::frames_run), summed by the host
constexpr int kSegLds = 16;
constexpr int kRowConsts = 2; // per-task constants the frame loop reads from LDS instead of holding (or spilling) registers: s2
constexpr int group_seg_doubles() { return kSegLds + kRowConsts; } // 2 * kSegLds int32 (the list is cleaned in place) + constants
constexpr int state_header_doubles(int NP) { return 3 * NP; }
// optional (KParams::walk_lds): the task's table entries, fetched for all its switches at once -- 3 doubles per switch
constexpr int kWalkDoubles = 3 * kSegLds;

// host-callable launchers implemented in kernels.hip
// (ev_start / ev_stop: hipEvent_t attached to the dispatch of a timed launch, or null)
int launch_logl(const Geometry &g, int mode, const KParams &p, int grid, size_t lds_bytes, void *stream, void *ev_start = nullptr,
                void *ev_stop = nullptr);
int launch_reduce_partials(const double *partial, double *out, int64_t n, int dstar_max, void *stream);
int launch_prefix_L(const TrajDesc *d_trajs, int n_traj, int S, int NP, int dstar_max, int Tmax, double *d_prefix, double *d_prefix_L, void *stream);
// d_err: 2 ints (verdict, a sample that shows it) followed, when `order` is given, by n hit counters; zeroed by the caller
int launch_validate(const int32_t *seg_start, const int32_t *seg_state, const int32_t *traj_id, const int32_t *order, int64_t n,
                    int K1, int S, int n_traj, int *d_err, void *stream);
// smallest compiled row count >= n_rows, or 0
int padded_rows(int n_rows);
// launch geometry for `ntasks` recursions of a chain padded to NP rows, each with up to `means` mean vectors
// (several are compiled per NP: few tasks per wave for small batches, many for throughput, fewer lanes per
// task when fewer mean vectors are needed); env BILD_GEOM=<id> overrides.
bool geometry_for(int NP, int mode, int64_t ntasks, int means, Geometry *g);
const char *kernel_name(const Geometry &g, int mode);
// the geometry of the frame loop over the work lists of a split launch, where it differs from `from` (same layout, same
// arithmetic, another register budget); false: keep `from`
bool listed_geometry(const Geometry &from, Geometry *g);
// the geometry whose kernel is also compiled as the builder of the prefix table (KParams::prefix_dump)
bool builder_geometry(int NP, Geometry *g);
// dense recursion on the fp64 matrix pipe (dense_mfma.hip): NP a multiple of 4, <= 24
bool dense_mfma_supported(int NP);
int launch_logl_dense_mfma(int NP, const KParams &p, void *stream);
// modal recursion on tile registers (modal_mfma.hip): NP = 36, 40
constexpr int kMidMaxNP = 40;
bool modal_mfma_supported(int NP);
int launch_logl_modal_mfma(int NP, const KParams &p, void *stream);
// chains of more than kMidMaxNP modes (wide.hip): one task per workgroup, state in LDS
size_t wide_lds_bytes(int NP);
// schedule.hip: launch order computed on the device (workspace of device_schedule_bytes(n); everything on `stream`)
int launch_pair_tasks(const int64_t *d_first_task, int n_traj, const TrajDesc *d_trajs, int S, int G, int64_t nb, int32_t *d_seg_start,
                      int32_t *d_seg_state, int32_t *d_traj_id, void *stream);
int launch_two_switch_cover(const TrajDesc *d_trajs, const int64_t *d_first, int64_t total, int n_traj, int S, const TransEntry *trans,
                            const TransEntry *trans2, int gap_max, int *d_covered, void *stream);
size_t device_schedule_bytes(int64_t n);
int device_schedule(const int32_t *d_seg_start, const int32_t *d_traj_id, const TrajDesc *d_trajs, int K1, int64_t n, int m_typ, int pairs,
                    int Tmax, int rpw, int64_t slots, void *ws, size_t ws_bytes, const int32_t **d_order, void *stream);
int launch_logl_wide(int NP, const KParams &p, int grid, void *stream);

} // namespace bild
