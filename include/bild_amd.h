/*
 * bild_amd -- C ABI of the MI355X (gfx950) Rouse Kalman-filter log-likelihood.
 *
 * This is the drop-in boundary for the one hot path of BILD
 * (OpenTrajectoryAnalysis/bild): everything the reference does per candidate looping
 * profile inside
 *
 *     bild/src/MSRouse_logL.pyx:95-256   MSRouse_logL(model, profile, traj)        (native)
 *     bild/src/MSRouse_logL_py.py:54-121 MSRouse_logL(model, profile, traj)        (fallback)
 *
 * batched over the samples of one AMIS step, i.e. the loop of
 *
 *     bild/amis.py:717-739               FixedkSampler.logL(ss, thetas) -> (N,) float64
 *
 * Plain C: opaque handles, raw pointers and sizes.  No torch / numpy / HIP types appear
 * in any signature (a HIP stream is passed as void*).  All matrices are row-major
 * float64.  Every function returns a bild_status; bild_last_error() gives the message of
 * the calling thread's most recent failure.  There is NO CPU fallback: evaluation
 * functions fail with BILD_ERR_NO_DEVICE when no gfx950 device is usable.
 *
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef BILD_AMD_H
#define BILD_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: round 4 -- bild_config_string / bild_config_reload, the inference driver (bild_run_*), the direct exchange
 *    (bild_exchange_*); negative cumulative positions of an (s, theta) row are refused on every path.
 * 1: rounds 1-3 (the version constant was not bumped while entry points were added). */
#define BILD_AMD_ABI_VERSION 2

typedef enum bild_status {
    BILD_OK              = 0,
    BILD_ERR_INVALID     = 1, /* bad argument: NULL, shape, range, non-finite model       */
    BILD_ERR_HIP         = 2, /* HIP runtime call failed                                  */
    BILD_ERR_NO_DEVICE   = 3, /* no usable GPU                                            */
    BILD_ERR_UNSUPPORTED = 4, /* model outside the compiled kernel envelope (N, d, S)     */
    BILD_ERR_NOMEM       = 5
} bild_status;

/* evaluation path, low 4 bits of `flags` of the bild_logl_* calls */
#define BILD_PATH_AUTO  0u /* modal when the model admits it (symmetric B), else dense    */
#define BILD_PATH_DENSE 1u /* canonical recursion  C <- B C B + Sig  every frame
                              (pyx:220-241)                                               */
#define BILD_PATH_MODAL 2u /* same recursion carried in each state's eigenbasis of B      */

/* bild_logl_segments_device only: check the device-resident descriptors with a small kernel first (traj_id range,
 * first start 0, non-decreasing starts, states < S) and refuse the batch with BILD_ERR_INVALID instead of letting a bad
 * index drive an out-of-range access.  The verdict has to be read back, so with this flag the call WAITS for the stream
 * once before launching.  Without it the device entry trusts its input (the host-buffer entries always validate). */
#define BILD_VALIDATE_DEVICE 0x10u

/* Do not start candidates from the prefix table (see "prefix table" below): every task runs from frame 0.  Results are
 * bit-identical either way; the flag exists for A/B measurements and tests.  (Environment BILD_NO_PREFIX=1: never build one.) */
#define BILD_NO_PREFIX 0x20u

/* With a prefix table: do not take the table's sums where a candidate's filter has converged onto the switch-free one
 * (see "prefix table" below); every frame behind the first switch is run.  Bit-identical to BILD_NO_PREFIX.
 * (Environment BILD_NO_JUMP=1: the same for every call.) */
#define BILD_NO_JUMP 0x40u

/* Split launches.  With all tables of a trajectory set in place a batch is launched as two kernels: a table walk with
 * one LANE per task (csrc/walk.hip), which finishes every task that needs no Kalman frame -- 96 % of the headline batch --
 * and appends the others to work lists ordered by expected work, and the frame loop (csrc/kernels.hip) over the listed
 * tasks only.  The walk adds the same numbers in the same order as the frame-loop kernel would: results are bit-identical
 * to the single launch, which this flag (or BILD_NO_SPLIT=1 in the environment) restores for A/B measurements and tests.
 * Lists of up to 16 segments per candidate are split; longer ones always take the single launch. */
#define BILD_NO_SPLIT 0x80u

/* Do not start chains of close switches from the transient state table (see "prefix table" below: the state a transient
 * has reached when the next switch comes is a record of that table): every chain runs from its first switch.  Results are
 * bit-identical either way.  (Environment BILD_NO_STATES=1: never build the table.  The table holds a record for every gap
 * a chain can start with -- up to the set's longest converged transient, measured when the transient table is built -- and
 * costs ~100 MB per 1000 frames of a 2-state trajectory: it is built for sets that need at most 4 GB of it, 64 GB for sets
 * declared for >= 1e8 evaluations with bild_trajset_expect, BILD_STATES_MAX_BYTES=<n> otherwise; at most a third of the free
 * memory in any case.) */
#define BILD_NO_STATES 0x100u

/* Do not leave a transient on the first-order tail (csrc/tail.hip: once the covariance of a candidate's filter has converged
 * onto the switch-free filter's, the effect of the remaining deviation of the means on all later frames is a dot product with
 * a vector kept beside the prefix table): every transient runs until its means have converged too, as until round 3.  The
 * results differ by ~1e-12 (second-order terms of a deviation below 2^-20).  (Environment BILD_NO_TAIL=1: never build the vectors.) */
#define BILD_NO_TAIL 0x200u

/* bild_model_create flags */
#define BILD_MODEL_NO_REDUCE 1u /* keep all N modes: skip the invariant-subspace reduction */

typedef struct bild_model   bild_model;
typedef struct bild_trajset bild_trajset;

int         bild_abi_version(void);
const char *bild_last_error(void);
void        bild_set_last_error(const char *msg); /* for the library's own translation units */

/* The experiment switches of the library are environment variables (BILD_NO_SPLIT, BILD_NO_STATES, BILD_PAIRS_MAX_TASKS ...:
 * csrc/config.h lists them all).  They are read ONCE, at the first use of the library in a process.
 * bild_config_string(): the switches that are set, as "NAME=value NAME=value" ("" when the defaults are in force) -- what
 * a bug report should carry.  bild_config_reload(): read the environment again (tests and tools that flip a switch inside
 * one process; must not run concurrently with evaluations; tables a trajectory set has built already stay as they are). */
const char *bild_config_string(void);
int         bild_config_reload(void);

/* Number of visible GPUs (0 without a device; never fails on a CPU-only host). */
int bild_device_count(int *count);

/* ------------------------------------------------------------------ model ----------
 * Replaces the per-call model setup of the reference kernel
 * (bild/src/MSRouse_logL.pyx:150-166): measurement vector w, and for each of the S
 * states the propagator (B, G, Sig) and steady state (M0, C0) that the reference pulls
 * out of rouse.Model (._dynamics['B'|'G'|'Sig'], .steady_state()).
 *
 *   N  monomers, d spatial dimensions (1..8), S states
 *   B, Sig, C0 : S x N x N     G, M0 : S x N x d     w : N
 *
 * Envelope: d <= 8 (a task carries up to three mean vectors; more dimensions with one localization error repeat
 * the covariance recursion: d = 4..6 costs two tasks per sample, d = 7..8 three), S <= 255, and at most 128 modes left by the (exact) invariant-subspace
 * reduction -- N <= 256 monomers for the default end-to-end measurement, N <= 128 for an
 * arbitrary w; above 32 modes only the modal path exists.  Outside: BILD_ERR_UNSUPPORTED.
 *
 * Host analysis only (eigenbases for the modal path, packing); no GPU is needed to
 * create a model.  Device copies are made lazily on the device that is current when an
 * evaluation first uses the model.  The caller keeps ownership of all inputs.
 */
int bild_model_create(int N, int d, int S,
                      const double *B, const double *G, const double *Sig,
                      const double *M0, const double *C0, const double *w,
                      unsigned flags, bild_model **out);
int bild_model_destroy(bild_model *m);

/* integer properties of a model */
#define BILD_Q_N            0
#define BILD_Q_D            1
#define BILD_Q_S            2
#define BILD_Q_MODAL_OK     3 /* 1 if the modal path is available                         */
#define BILD_Q_NP           4 /* padded row count the kernels run with                    */
#define BILD_Q_NEFF         5 /* modes kept after invariant-subspace reduction            */
#define BILD_Q_HAS_G        6 /* 1 if any G entry is non-zero                             */
int bild_model_query(const bild_model *m, int what, int64_t *value);

/* Host-analysis export (for tests that run without a GPU).  `what`:
 *   BILD_X_LAMBDA : n       eigenvalues of B[s] in the modal basis
 *   BILD_X_SIGMA  : n       diagonal of Q^T Sig[s] Q
 *   BILD_X_Q      : Nr x n  modal basis of state s (columns), in reduced coordinates
 *   BILD_X_WQ     : n       Q^T w
 *   BILD_X_R      : n x n   basis change  Q[s2]^T Q[s]   (state s -> s2)
 *   BILD_X_C0Q    : n x n   Q^T C0[s] Q
 *   BILD_X_V      : N x Nr  reduction basis (identity when no reduction applied)
 * n = Nr = BILD_Q_NEFF.  `buf` must hold the stated number of doubles. */
#define BILD_X_LAMBDA 0
#define BILD_X_SIGMA  1
#define BILD_X_Q      2
#define BILD_X_WQ     3
#define BILD_X_R      4
#define BILD_X_C0Q    5
#define BILD_X_V      6
int bild_model_export(const bild_model *m, int what, int s, int s2, double *buf, int64_t buf_len);

/* ------------------------------------------------------------- trajectories --------
 * Replaces the per-call trajectory setup (pyx:144-147, 174-178): the (T, d) data with
 * NaN marking missing frames (a frame is missing iff any coordinate is NaN), and the
 * per-dimension localization error from which the reference derives
 * s2 = unique(err)^2 and Cind (np.unique, sorted).
 *
 *   n_traj trajectories, lengths T[j] >= 1
 *   x        : sum(T) x d, trajectories concatenated in order
 *   loc_err  : n_traj x d, standard deviations (> 0 is not required, >= 0 is)
 *
 * Needs a GPU: the data are uploaded once and stay resident for all AMIS steps.
 */
int bild_trajset_create(const bild_model *m, int n_traj, const int32_t *T,
                        const double *x, const double *loc_err, bild_trajset **out);
/* (The device memory of a set's tables -- blocks of 128 KiB and more -- is not returned to the driver but kept, in size classes, for the
 * tables of the next set: up to BILD_TABLE_CACHE_BYTES, default 4 GB per process; 0 turns that off.  No evaluation on the set may be in
 * flight when it is destroyed.) */
int bild_trajset_destroy(bild_trajset *ts);
/* Optional, before the first evaluation on the set: how many evaluations the caller expects to run on it in total.  The
 * tables of a set (see "prefix table" below) are built at its first evaluation and cost about 1.5 ms per trajectory of
 * 1000 frames; they pay from a few hundred evaluations on (prefix + transient tables: >= 300) resp. a few thousand (pair
 * and state tables: >= 3000).  Without a declaration everything is built -- right for an AMIS run, wasteful for a
 * handful of single evaluations per trajectory.  Which tables exist depends on the set and on this declaration alone,
 * never on the call history: results stay reproducible (and agree between declarations to rounding, like between sets). */
int bild_trajset_expect(bild_trajset *ts, int64_t evaluations);

/* ---------------------------------------------------------------- evaluation -------
 * One call = one AMIS batch (bild/amis.py:717-739): n independent evaluations
 * logL(profile_r, traj[traj_id[r]]) -> out[r].
 *
 * Profiles come run-length encoded, K1 segments per sample:
 *   segment i of sample r is in state seg_state[r*K1+i] and starts at frame
 *   seg_start[r*K1+i]; seg_start[r*K1+0] must be 0, later starts >= 1 and non-decreasing (segment 0 owns
 *   frame 0: it selects the steady state the filter starts from and is never empty).
 *   Empty later segments (equal starts, or a start >= T) are legal and skipped -- this is what
 *   FixedkSampler.st2profile (bild/amis.py:685-693) produces for
 *   seg_start[1:] = floor(cumsum(s)[:-1]*(T-1)) + 1, seg_state = theta.
 *   state[0] selects the steady state the filter starts from, state[t] the propagator
 *   into frame t (bild/util.py:15-23).
 *   traj_id may be NULL (all samples refer to trajectory 0).
 *
 * All-missing trajectory -> 0.0 (semantics of MSRouse_logL_py.py:90-94).  NaN/Inf in a
 * result is passed through, never trapped.
 *
 * REPRODUCIBILITY CONTRACT (differs from the reference, whose MSRouse_logL is a pure function of its three arguments):
 *   - on ONE trajectory set a result is a pure function of (candidate, trajectory): bit-identical whatever the batch it
 *     is part of, the order of the batch, the entry point (host buffers, device buffers, (s, theta) rows or segment
 *     lists), the stream, the launch geometry, what was evaluated before, and with or without the split launch / the
 *     transient state table (BILD_NO_SPLIT, BILD_NO_STATES);
 *   - the same (candidate, trajectory) pair on two DIFFERENT trajectory sets (the trajectory alone / among hundreds of
 *     others) may take its sums from different tables -- which of them are built depends on the size of the set -- and
 *     agrees to rounding, |delta| ~ 1e-11 at |logL| ~ 3e4, not to the bit;
 *   - BILD_NO_JUMP (or BILD_NO_PREFIX) gives the frame-by-frame result, which IS a pure function of its inputs on any
 *     set; the default differs from it by ~1e-11 (bound and measurements: DESIGN.md section 2, tests/test_gpu_adversarial.py).
 * Against the reference's Cython kernel every path agrees to |delta| < 1e-8 (observed 1e-10; the reference's two own
 * kernels differ by 2.6e-10 among themselves).
 */

/* host buffers in, host buffer out; synchronous */
int bild_logl_segments(const bild_model *m, const bild_trajset *ts, int64_t n, int K1,
                       const int32_t *seg_start, const int32_t *seg_state,
                       const int32_t *traj_id, unsigned flags, double *out);

/* The sampler's own parametrisation, i.e. the arguments of FixedkSampler.logL(ss, thetas) (bild/amis.py:717-739)
 * as they are: ss (n x K1 float64, rows on the simplex), thetas (n x K1 int64).  The switch frames are computed here
 * exactly as FixedkSampler.st2profile does (bild/amis.py:685-688: sequential cumsum, times (T-1), floor, +1; T is the
 * length of the sample's trajectory), written straight into pinned staging memory and shipped with one copy.
 * A row is refused (BILD_ERR_INVALID) when one of its cumulative positions cumsum(s)[i] * (T-1) is negative, non-finite,
 * >= 2^31 or smaller than the one before it -- i.e. whenever it is not a point on the simplex in a way that would change
 * the profile (the reference does not check: np.floor of a negative position gives switch index 0 and a profile whose
 * first interval is empty).  Host buffers, synchronous. */
int bild_logl_st(const bild_model *m, const bild_trajset *ts, int64_t n, int K1,
                 const double *ss, const int64_t *thetas, const int32_t *traj_id,
                 unsigned flags, double *out);

/* Same input, but the n results stay in HBM: d_out (device, n doubles) is written asynchronously on `hip_stream`
 * (a hipStream_t, NULL = default stream), nothing is copied back and nothing is waited for.  This is what a
 * multi-GPU AMIS step uses: the shard's results go straight into the all-gather (dist.ShardedModel). */
int bild_logl_st_to_device(const bild_model *m, const bild_trajset *ts, int64_t n, int K1,
                           const double *ss, const int64_t *thetas, const int32_t *traj_id,
                           unsigned flags, void *hip_stream, double *d_out);

/* bild_logl_st_to_device waits for nothing, so it cannot refuse a row that is not a point on the simplex (negative or
 * non-finite interval lengths): such a row gets NaN as its result and the model remembers it.  This call waits for the
 * model's pending to_device calls, returns BILD_ERR_INVALID if any row was refused since the last query (*bad_row: one
 * of them, else -1; may be NULL) and forgets the verdict.  bild_logl_st itself reports such rows directly. */
int bild_logl_st_status(const bild_model *m, int64_t *bad_row);

/* The conversion alone (host only, no GPU): FixedkSampler.st2profile's switch frames (bild/amis.py:685-693) as
 * run-length segments.  Sample r belongs to a trajectory of T[r * T_stride] frames (T_stride 0: one length for all).
 * seg_start / seg_state: n x K1 int32 out. */
int bild_segments_from_st(int64_t n, int K1, int n_states, const int32_t *T, int64_t T_stride,
                          const double *ss, const int64_t *thetas,
                          int32_t *seg_start, int32_t *seg_state);

/* expanded profiles (MSRouse_logL semantics, pyx:95-98): states is n rows of length
 * ld >= max T, row r holds T[traj_id[r]] valid entries.  Run-length encoded on the host,
 * then as above. */
int bild_logl_profiles(const bild_model *m, const bild_trajset *ts, int64_t n, int64_t ld,
                       const int32_t *states, const int32_t *traj_id, unsigned flags,
                       double *out);

/* device buffers in, device buffer out; asynchronous on `hip_stream` (a hipStream_t, or
 * NULL for the default stream).  This is the entry point timed by bench.py: nothing
 * crosses PCIe.  d_out must hold n doubles.  Re-entrant: concurrent launches of one model on
 * different streams share no scratch (partial results of d* > 1 live in a per-call, stream-ordered
 * allocation).  The descriptors are NOT validated unless BILD_VALIDATE_DEVICE is set in `flags`. */
int bild_logl_segments_device(const bild_model *m, const bild_trajset *ts, int64_t n, int K1,
                              const int32_t *d_seg_start, const int32_t *d_seg_state,
                              const int32_t *d_traj_id, unsigned flags, void *hip_stream,
                              double *d_out);

/* The sampler's own (s, theta) rows, RESIDENT IN HBM: d_ss (n x K1 float64), d_thetas (n x K1 uint8), d_traj_id (n int32
 * or NULL).  Everything FixedkSampler.logL does for a batch happens on the device, asynchronously on `hip_stream`: switch
 * frames as st2profile computes them (bild/amis.py:685-688), cleaning, the table walk, the frame loop for the chains of
 * close switches, results to d_out (n doubles).  K1 <= 16.  Rows that are no points on the simplex (negative / non-finite
 * interval lengths, a state >= S) get NaN; d_status (2 ints in device-visible memory, zeroed by the caller, or NULL)
 * then holds {1, such a row}.  This is the entry bench.py times as `value`: candidates in HBM as the sampler produced
 * them, nothing converted, ordered or scheduled beforehand. */
int bild_logl_st_device(const bild_model *m, const bild_trajset *ts, int64_t n, int K1,
                        const double *d_ss, const uint8_t *d_thetas, const int32_t *d_traj_id,
                        unsigned flags, void *hip_stream, double *d_out, int32_t *d_status);

/* ------------------------------------------------- prefix table and launch order -------
 * Until its first switch a candidate's filter state depends only on (trajectory, localization-error chain, initial
 * state, frame) -- not on the candidate.  At the first evaluation of a trajectory set on the modal path (chains of up to
 * 32 modes) the library therefore runs the recursion once per (trajectory, chain, state) WITHOUT switches, keeps the state
 * after every frame in HBM (T x S x d* records of ~1 KB; skipped when it would take more than a quarter of the free
 * memory), and every candidate starts from the record in front of its first switch.  Same kernel, same arithmetic:
 * results are bit-identical to running every candidate from frame 0 (BILD_NO_PREFIX).  One-time cost per trajectory
 * set: one launch of about the duration of a single candidate, reported by bild_prefix_info.
 *
 * Behind a switch a Kalman filter forgets its starting point at a geometric rate: after a few tens of frames the
 * candidate's state [C | M] agrees with the table's record of the same frame and state to rounding.  The kernel checks
 * exactly that (whole state, first 24 frames behind the switch and then after as many frames as the measured deviation
 * still needs at the usual rate of decay; relative tolerance 2^-43 per column) and,
 * once it holds, takes the table's sums up to the candidate's next switch and continues from the record in front of
 * it: equal state + same propagator + same data = same future.  Nothing is assumed about stationarity; a candidate
 * that does not converge runs every frame.  This changes results by ~1e-12 (differences of running sums, tolerance of
 * the comparison); BILD_NO_JUMP switches it off.  A result never depends on the other candidates of the batch.
 *
 * Transient tables.  A transient that starts on the switch-free filter of the old state depends on (trajectory, chain,
 * old state, new state, frame) only.  Right behind the prefix table the library lets the kernel evaluate one candidate
 * per such switch and keeps how many frames the transient took to converge and what it added to the log-likelihood
 * beyond the new state's own sums (16 bytes per entry).  A candidate whose next switch is at least that many frames
 * away takes the entry and runs nothing.  Second level, the pair table: two switches closer together than the first
 * one's transient, keyed by (old, middle, new state, frame, gap <= 64) -- built for trajectory sets where the build (a
 * launch of T x gaps x S(S-1)^2 short tasks per trajectory) needs at most 40 M tasks; chains of three and more close
 * switches are run frame by frame from the table's state in front of them.  All tables are built at the FIRST evaluation
 * on a trajectory set and never later, and whether they are built depends on the set alone: results are reproducible
 * bit for bit for a given trajectory set whatever was evaluated before, in whatever batches and order.  The same candidate
 * evaluated on two different sets (one trajectory alone / among hundreds) may take its sums from different tables and
 * agrees to ~1e-11.  BILD_NO_TRANSIENTS=1 / BILD_NO_PAIRS=1 (environment) switch a level off for experiments;
 * BILD_PAIRS_MAX_TASKS=<n> replaces the 40 M budget (sets of many trajectories that will see hundreds of batches).
 *
 * Candidates then differ in length, so the order in which they are dealt to wavefronts matters for speed (never for
 * results).  The host-buffer entry points schedule internally (batches larger than the chip holds at once: on the device,
 * behind the upload; BILD_NO_SCHEDULE=1 switches every scheduling off).  For device-resident candidates the caller may obtain
 * the launch order once (bild_schedule_segments, host arrays) and pass it, device-resident, to
 * bild_logl_segments_device_ordered: order[slot] = index of the sample evaluated in slot `slot`; a permutation of
 * 0..n-1 (checked only with BILD_VALIDATE_DEVICE).  NULL = the order of the arrays. */
int bild_schedule_segments(const bild_model *m, const bild_trajset *ts, int64_t n, int K1,
                           const int32_t *seg_start, const int32_t *seg_state,
                           const int32_t *traj_id /* may be NULL */,
                           unsigned flags, int32_t *order /* out, n */);
int bild_logl_segments_device_ordered(const bild_model *m, const bild_trajset *ts, int64_t n, int K1,
                                      const int32_t *d_seg_start, const int32_t *d_seg_state,
                                      const int32_t *d_traj_id, const int32_t *d_order,
                                      unsigned flags, void *hip_stream, double *d_out);
/* frames (task x frame pairs) of a batch, and how many of them the launch really runs when candidates start from the
 * prefix table in the given launch order (host arrays; order may be NULL): the executed-operation count of a launch is
 * bild_flop_count's `executed` times frames_run / frames_total */
int bild_frames_executed(const bild_model *m, const bild_trajset *ts, int64_t n, int K1,
                         const int32_t *seg_start, const int32_t *traj_id, const int32_t *order,
                         unsigned flags, double *frames_total, double *frames_run);
/* frames the tasks of all launches of this model ran themselves since the last call (counted on the device while
 * bild_kernel_timing is enabled; the rest came out of the prefix table); resets the counter; synchronises the device */
int bild_frames_run_read(const bild_model *m, int64_t *frames);
/* diagnostics: while d_buffer (device, one int32 per task = sample x localization-error chain) is set, every launch of
 * the vector kernels records how many frames each task ran itself; NULL switches it off */
int bild_debug_frames_per_task(int32_t *d_buffer);
/* size of the tables (prefix, transient, pair) in bytes (0: none built) and the device time their construction took */
int bild_prefix_info(const bild_trajset *ts, int64_t *bytes, double *build_ms);

/* ---------------------------------------------------------- several GPUs ------------
 * One process per GPU.  The evaluations of a batch are independent, so ranks evaluate disjoint shards with no
 * communication inside the likelihood; the ONE exchange of an AMIS step is an all-gather of the shards' results
 * (every rank forms the importance weights from the full vector, bild/amis.py:843-845).  These calls put that
 * collective behind the C ABI, on RCCL over xGMI, so that a multi-GPU host program needs nothing but this library:
 *
 *   rank 0:      bild_comm_unique_id(id)            -> 128 bytes, to be handed to every rank by ANY channel the host
 *                                                      program has (file, socket, MPI, torch.distributed store ...)
 *   every rank:  bild_comm_create(id, world, rank)  on the device that is current (one per rank)
 *   per step:    bild_logl_st_to_device(... stream, d_local)   results of the rank's shard stay in HBM
 *                bild_comm_allgather(comm, d_local, d_all, n_per_rank, stream)   same stream: ordered behind the kernel
 *                one device-to-host copy of d_all
 *
 * Shards must be padded to a common length n_per_rank.  RCCL is located at run time (an RCCL already loaded into the
 * process is used; else librccl.so.1; bild_comm_library(path) or BILD_AMD_RCCL name another one); without one the
 * calls fail with BILD_ERR_UNSUPPORTED.  bild_amd/dist.py (LibraryComm, ShardedModel) is the Python binding. */
#define BILD_COMM_ID_BYTES 128
typedef struct bild_comm bild_comm;
int bild_comm_library(const char *path);
int bild_comm_unique_id(char *id, int id_len);
int bild_comm_create(const char *id, int world, int rank, bild_comm **out);
int bild_comm_allgather(bild_comm *c, const double *d_send, double *d_recv, int64_t n_per_rank, void *hip_stream);
int bild_comm_destroy(bild_comm *c);
/* device buffers for a host program that brings no GPU framework of its own (shard and gathered vector above);
 * bild_device_to_host copies on `hip_stream` and waits for it */
int bild_device_alloc(int64_t bytes, void **out);
int bild_device_free(void *ptr);
int bild_device_to_host(void *dst, const void *d_src, int64_t bytes, void *hip_stream);

/* The direct exchange: the same all-gather as ONE kernel per rank that stores the rank's shard straight into every
 * peer's receive block (mapped through an IPC handle; over xGMI between GPUs) and waits for the peers' shards there --
 * for the 10-256 KB a rank contributes per AMIS step a ring collective is latency-bound (7 dependent hops on 8 GPUs), a
 * one-shot peer write is one hop.  Replaces nothing in the reference (bild/amis.py:732-733 declines to
 * parallelise); the vector it assembles is the one bild/amis.py:843-845 consumes.
 *
 *   every rank:  bild_exchange_create(world, rank, slot_doubles, &x)      on its current device; slot_doubles >= longest shard
 *                bild_exchange_handle(x, handle)                          64 bytes, to be handed to every rank by any channel
 *                bild_exchange_connect(x, handles)                        all ranks' handles in rank order (world x 64 bytes)
 *   per step:    bild_logl_st_to_device(... stream, d_local)
 *                bild_exchange_allgather(x, d_local, n, stream, d_all)    same stream; d_all: world x n doubles on the device
 *                (consume d_all on that stream, or copy it to the host: bild_device_to_host)
 *                bild_exchange_status(x, &peer)                           after the stream has been waited for: a peer that
 *                                                                         did not deliver within the timeout (5 s) is an error
 * Every rank must call bild_exchange_allgather the same number of times with the same n.  world <= 16.  The waits are
 * bounded: a rank whose peer never arrives gets BILD_ERR_HIP from bild_exchange_status instead of a hung GPU.
 * bild_exchange_set_step (tests: rehearse the wrap of the 32-bit step counter; the same call on every rank, with no
 * exchange in flight) and bild_exchange_set_timeout are for tests and tools. */
#define BILD_EXCHANGE_HANDLE_BYTES 64
typedef struct bild_exchange bild_exchange;
int bild_exchange_create(int world, int rank, int64_t slot_doubles, bild_exchange **out);
int bild_exchange_handle(const bild_exchange *x, char *handle);
int bild_exchange_connect(bild_exchange *x, const char *handles);
int bild_exchange_allgather(bild_exchange *x, const double *d_send, int64_t n, void *hip_stream, double *d_recv);
int bild_exchange_status(bild_exchange *x, int *peer);
int bild_exchange_set_step(bild_exchange *x, uint32_t step);
int bild_exchange_set_timeout(bild_exchange *x, double seconds);
int bild_exchange_destroy(bild_exchange *x);

/* canonical floating-point operations of one batch,
 *   F = (T-1)(4 N^3 d* + 2 N^2 d) + Tv((4 N^2 + 3 N) d* + 4 N d)      
 * summed over samples, and the operations the selected path actually executes. */
int bild_flop_count(const bild_model *m, const bild_trajset *ts, int64_t n,
                    const int32_t *traj_id /* host, may be NULL */, unsigned flags,
                    double *canonical, double *executed);

/* name and accumulated device time (ms, HIP events on the launch stream) of the kernel the
 * most recent evaluations ran; resets the accumulator.  Used by bench.py for the roofline
 * figure.  Timing is off unless enabled. */
/* enable = 0: off; 1: every launch is bracketed by events; p > 1: every p-th launch (the events themselves cost a few
 * microseconds per launch: sampling keeps them out of most steps of a timed region) */
int bild_kernel_timing(int enable);
int bild_kernel_timing_read(double *total_ms, int64_t *launches, char *name, int name_len);
/* the same for the table-walk kernel that precedes the frame loop in a split launch (see "split launches" above) */
int bild_kernel_timing_read_walk(double *total_ms, int64_t *launches);

/* ------------------------------------------------ host-side AMIS bookkeeping -------
 * SURVEY section 8, row f-1.  Plain host code (no GPU): everything reference
 * bild/amis.py FixedkSampler.step (amis.py:805-906) does once the likelihood of the new
 * batch is known -- mixture denominators of all samples drawn so far, weights, refit of
 * the Dirichlet (amis.py:110-151) and CFC (amis.py:284-399) proposals, brakes
 * (amis.py:856-874), evidence / standard error / KL (amis.py:876-903) -- in one pass over
 * the pooled samples.  Random numbers are drawn by the caller (NumPy stream, reference
 * order).  bild_amd/amis.py holds the same bookkeeping in NumPy as the specification.
 *
 *   k1 = k + 1 intervals, n states, transitions[from*n + to] != 0 where allowed
 *   a0 (k1), logp0 (n x k1, [state][slot]): the initial proposal
 */
typedef struct bild_amis bild_amis;
int bild_amis_create(int k1, int n, const uint8_t *transitions, double concentration_brake,
                     double polarization_brake, double logprior, const double *a0,
                     const double *logp0, bild_amis **out);
int bild_amis_destroy(bild_amis *m);
const char *bild_amis_error(const bild_amis *m);
int64_t bild_amis_pool_size(const bild_amis *m);
int64_t bild_amis_num_proposals(const bild_amis *m);
/* proposal `which` (negative: from the end); either output may be NULL */
int bild_amis_params(const bild_amis *m, int64_t which, double *a, double *logp);
/* pooled per-sample arrays: 0 logLs, 1 log mixture denominators, 2 log density under the
 * current proposal, 3 log weights */
int bild_amis_pool(const bild_amis *m, int what, double *out);
/* rebuild a freshly created sampler from saved state: Q_extra further proposals (a: Q_extra x k1,
 * logp: Q_extra x n x k1) and P pooled samples with their per-sample arrays */
int bild_amis_restore(bild_amis *m, int64_t Q_extra, const double *a, const double *logp, int64_t P,
                      const double *ss, const int64_t *thetas, const double *logLs,
                      const double *logd, const double *cur, const double *logw);
/* traces from the current proposal (amis.py:223-256); u: k1 blocks of N uniform numbers */
int bild_amis_sample_traces(const bild_amis *m, int64_t N, const double *u, int64_t *thetas);
/* one iteration: ss (N x k1), thetas (N x k1), logLs (N) -> evidence[3] = (logev, dlogev, KL);
 * a CFC fit that does not converge returns BILD_ERR_INVALID with
 * bild_amis_error() == "Iteration did not converge" (the reference raises RuntimeError) */
int bild_amis_step(bild_amis *m, int64_t N, const double *ss, const int64_t *thetas,
                   const double *logLs, double *evidence);
/* The fused step (pooled samples on the device, see bild_amis_use_device below; K1 = k + 1 <= 16): the new samples go up
 * ONCE, as the (s, theta) rows the likelihood kernels read, straight into the pool; their log-likelihood on `ts` (one
 * trajectory) is computed there (as bild_logl_st would) and written into the pool; the passes over the pool follow on the
 * same stream.  Nothing but a few hundred partial sums comes down; the host's copy of the pool (bild_amis_pool) catches up
 * when it is asked for.  Same results as bild_logl_st followed by bild_amis_step. */
int bild_amis_step_fused(bild_amis *m, const bild_model *model, const bild_trajset *ts, int64_t N,
                         const double *ss, const int64_t *thetas, unsigned flags, double *evidence);
/* The same with the N samples DRAWN on the device from the current proposal -- a counter-based generator (Philox-4x32-10)
 * keyed by `seed`, one stream per (step, sample); gamma variates by Marsaglia-Tsang, traces slot by slot from the CFC
 * weights.  Opt-in: not the reference's NumPy random stream (bild/amis.py:831-832), the same sampler in distribution.
 * Nothing goes up but the proposal's parameters.  bild_amis_pool_samples brings the pooled samples to the host. */
int bild_amis_step_device_rng(bild_amis *m, const bild_model *model, const bild_trajset *ts, int64_t N,
                              uint64_t seed, unsigned flags, double *evidence);
int bild_amis_pool_samples(const bild_amis *m, double *ss /* P x k1 */, int64_t *thetas /* P x k1 */);
/* keep the pooled samples in HBM and run the passes of bild_amis_step over them on the GPU (enable != 0), or return
 * to the host implementation (0).  Same arithmetic per sample (csrc/amis_math.h); sums are formed per block in a fixed
 * order, so results are reproducible and agree with the host's to rounding.  For batches of thousands of samples per
 * step; BILD_ERR_UNSUPPORTED when n_states * (k + 1) > 64, BILD_ERR_NO_DEVICE without a GPU. */
int bild_amis_use_device(bild_amis *m, int enable);

/* Histograms behind the choice of the next k in the adaptive-k loop (reference
 * bild/choicesampler.py:115-210): rvs (samplesize x kmax) common random sample, mu (kmax)
 * evidence estimates (NaN = ignored), dmu (kmax) natural step sizes, omit (kmax) flags or
 * NULL.  Outputs (any may be NULL): n0[kmax], dn[kmax][kmax], n_omit[kmax]. */
int bild_choice_counts(int64_t samplesize, int kmax, const double *rvs, const double *mu,
                       const double *dmu, double dE, const uint8_t *omit,
                       int64_t *n0, int64_t *dn, int64_t *n_omit);

/* ------------------------------------------------ the inference driver ------------
 * the adaptive-k loops of bild.core.sample (bild/core.py:138-227) for MANY trajectories
 * as one native state machine, advanced one ROUND at a time.  A round is one iteration of that loop for every trajectory
 * that is still running: at most one AMIS step (bild/amis.py:805-906) per trajectory, the exhaustive enumerations of the
 * samplers it opens (bild/amis.py:741-803), the choice sampler that picks the next k (bild/choicesampler.py:83-210) and
 * the stop rules -- with ONE likelihood call for all candidate rows of the round, and the per-trajectory bookkeeping on a
 * pool of host threads (BILD_HOST_THREADS, default min(8, cores)).
 *
 * Random numbers are NOT drawn by the library: bild_run_plan says how many the round needs, the caller draws them in
 * bulk from whatever stream it keeps (bild_amd.core.sample_many: three vectorised NumPy calls per round) --
 *     gammas    counts[0] standard gamma variates with the shape parameters *gamma_shapes  (np.random.standard_gamma)
 *     uniforms  counts[1] numbers in [0, 1)                                                  (np.random.random_sample)
 *     normals   counts[2] standard normal numbers                                            (np.random.standard_normal)
 * and per trajectory they are consumed in the reference's order: the gamma variates of the Dirichlet draw (normalised
 * exactly as np.random.dirichlet does: sequential sum, one reciprocal), the uniforms of the state traces (k + 1 blocks
 * of N, bild_amis_sample_traces), the samplesize x kmax normals of the choice sampler.  A run of ONE trajectory therefore
 * walks through the random numbers bild.core.sample would and reproduces it bit for bit.
 *
 * Per-k constants that do not depend on the trajectory are the caller's to provide, for k = 0 .. n_k - 1 (n_k >= k_max + 1):
 *     logp0     the CFC weights of the uniform distribution over traces (CFC.logp_uniform, amis.py:455-476),
 *               concatenated: n_states x (k + 1) doubles for each k
 *     logprior  log k! - log N_total(k)                                  (amis.py:654-659)
 *     n_total   N_total(k), the number of valid traces                   (amis.py:478-497)
 *     n_traces / traces   for every k that may be enumerated: all valid traces (CFC.full_sample, amis.py:499-536),
 *               n_traces[k] x (k + 1) int32, concatenated over k; n_traces[k] = 0: none given
 */
typedef struct bild_run bild_run;
typedef struct bild_run_settings {
    int32_t init_runs;            /* AMIS steps a freshly opened sampler takes at once (core.py:24: 20)   */
    int32_t k_lookahead;          /* core.py:26: 2                                                          */
    int32_t k_max;                /* core.py:27: 20                                                         */
    int32_t reserved;
    double  certainty_in_k;       /* stop when max p(k) reaches it (core.py:25: 0.99)                      */
    double  dE;                   /* evidence margin (core.py:23: 0)                                        */
    int64_t N;                    /* candidates per AMIS step (amis.py:624: 100)                            */
    double  concentration_brake;  /* amis.py:625: 1e-2                                                      */
    double  polarization_brake;   /* amis.py:626: 1e-3                                                      */
    int64_t max_fev;              /* amis.py:627: 20000                                                     */
    int64_t max_fcomplete;        /* amis.py:628: 1000                                                      */
    int64_t choice_samplesize;    /* choicesampler.py:83: 10000                                             */
} bild_run_settings;
int bild_run_create(int n_traj, const int32_t *T, int n_states, const uint8_t *transitions,
                    const bild_run_settings *settings, int n_k, const double *logp0, const double *logprior,
                    const double *n_total, const int64_t *n_traces, const int32_t *traces, bild_run **out);
int bild_run_destroy(bild_run *r);
const char *bild_run_error(const bild_run *r);
/* counts[8]: gammas, uniforms, normals, candidate rows of the round, trajectories still running, AMIS steps in the round,
 * trajectories that have failed so far (bild_run_traj_info), reserved.
 * counts[3] == 0 and counts[4] == 0: the run is over (no stage / finish for this plan). */
int bild_run_plan(bild_run *r, int64_t *counts, const double **gamma_shapes);
/* the whole round on the GPU: rows from the random numbers, bild_logl_st over them (traj_id = index of the trajectory in
 * the set `ts`, which must hold the run's trajectories in order), bookkeeping */
int bild_run_round(bild_run *r, const bild_model *m, const bild_trajset *ts, unsigned flags,
                   const double *gammas, const double *uniforms, const double *normals);
/* the same in three steps for a caller that evaluates the rows itself (CPU tests, models that are not this library's):
 * stage, look at the rows (ss: n x K1 float64, thetas: n x K1 int64, traj_id: n; rows shorter than K1 are padded with
 * empty intervals in their last state), finish with their log-likelihoods */
int bild_run_stage(bild_run *r, const double *gammas, const double *uniforms);
int bild_run_rows(const bild_run *r, int64_t *n, int *K1, const double **ss, const int64_t **thetas,
                  const int32_t **traj_id);
int bild_run_finish(bild_run *r, const double *logLs, const double *normals);
/* results.  traj_info: info[5] = state (0 running, 1 done, 2 failed), kind of failure (1: "Iteration did not converge",
 * a RuntimeError in the reference; 2: ValueError; 3: other), samplers, log rows, widest pk / KLD row.
 * traj_log: k, flags (bit 0: pk given, 1: KLD given, 2: I_la given; bits 8-15 / 16-23: entries of the pk / KLD row), I_la,
 * pk and KLD (rows x width, NaN-padded).
 * sampler_info: info[5] = kind (0: k >= T, 1: enumerated, 2: AMIS), exhausted, AMIS steps, evidences, enumerated rows.
 * sampler_data: evidences (x 3); for an enumerated sampler its rows.  take_core: ownership of an AMIS sampler's native
 * bookkeeping passes to the caller (bild_amis_destroy).  totals[3]: rounds, likelihood evaluations, host threads. */
int bild_run_traj_info(const bild_run *r, int j, int64_t *info, const char **message);
int bild_run_traj_log(const bild_run *r, int j, int32_t *k, int32_t *flags, double *i_la, double *pk, double *kld);
int bild_run_sampler_info(const bild_run *r, int j, int k, int64_t *info);
int bild_run_sampler_data(const bild_run *r, int j, int k, double *evidences, double *ss, int64_t *thetas,
                          double *logLs);
int bild_run_take_core(bild_run *r, int j, int k, bild_amis **out);
int bild_run_totals(const bild_run *r, int64_t *totals);

/* Weighted state occupancy per frame over a set of profiles (reference bild/amis.py:945-972):
 * post[s*T + t] = sum of w[p] over the profiles p that are in state s at frame t.  Profiles as
 * in bild_logl_segments (P x k1).  Sums of non-negative terms only. */
int bild_interval_marginals(int64_t P, int k1, int n, int64_t T, const int32_t *seg_start,
                            const int32_t *seg_state, const double *w, double *post);

#ifdef __cplusplus
}
#endif
#endif /* BILD_AMD_H */
