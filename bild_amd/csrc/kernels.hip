// Rouse Kalman-filter log-likelihood kernels for gfx950 (MI355X, CDNA4).
//
// What is computed (reference bild/src/MSRouse_logL.pyx:186-256, MSRouse_logL_py.py:96-121):
// for every task = (sample, covariance chain e) a T-step Kalman filter over the N-monomer
// chain, observed through y = w.x with localization noise s2[e]; the task's result is the
// sum of the per-frame Gaussian log-densities of the dimensions that use chain e.
//
// Mapping to the machine
// ----------------------
// * One task runs on a *group* of G lanes of a wavefront (G = 13 for the default 2-state model,
//   which reduces to 10 modes: one lane per column of [C | M]), floor(64/G) groups per wave,
//   4 waves per workgroup.  Recursions are independent, so there is no inter-workgroup
//   traffic and no grid-level synchronisation.
// * The filter state is the augmented matrix  A = [ C | M ]  (NP x (NP + d)): covariance
//   and the d mean vectors.  A is distributed BY COLUMN: lane gl of the group owns columns
//   gl*CPL .. gl*CPL+CPL-1 in registers (CPL*NP doubles).  With C symmetric,
//       Cw_j = sum_i w_i C_ij                       is a lane-local dot product,
//       yhat_dim = sum_i w_i M_i,dim                is the same dot product on an M column,
//       C_:j -= Cw (Cw_j / S),  M_:dim += Cw (nu/S) are the same lane-local rank-1 update,
//   so covariance and mean columns are processed by identical code and the only
//   cross-lane step per frame is an all-gather of the NP numbers Cw through LDS.
// * kModal: the recursion is carried in the eigenbasis Q_s of the current state's
//   (symmetric) propagator B_s, where the predict  C <- B C B + Sig  is elementwise,
//   C'_ij <- lam_i lam_j C'_ij + sig_i delta_ij.  A state switch s -> s2 applies the
//   orthogonal basis change R = Q_s2^T Q_s:  A <- R A (all columns), C <- C R^T.
// * kDense: the canonical recursion, C <- B C B + Sig every frame (pyx:220-241), done as
//   two lane-local mat-vecs per column with an in-group transpose through LDS between
//   them.  Same code path as the modal basis change.
// * Propagator / basis-change matrices are staged once per workgroup into LDS and read
//   as group-wide broadcasts (ds_read_b128); one LDS operand pair feeds 2*CPL FMAs per row.
// * fp64 throughout (v_fma_f64).  The modal predict is elementwise, so there is no GEMM to move to the matrix
//   pipe here (the one use tried in this kernel -- the cross-lane sum S = s2 + w.(Cw) as two
//   v_mfma_f64_4x4x4_4b, "block layout" below -- gains 2-5 % and is an opt-in geometry: the f64 matrix
//   and vector instructions draw on the same FMA throughput).  The DENSE recursion, which IS two small GEMMs
//   per frame, runs on the matrix pipe in dense_mfma.hip whenever the chain tiles into 4x4 blocks; the kDense
//   code below serves the other chain lengths.
// * sum_t log S_t is accumulated as a running product with exponent extraction
//   (frexp) and one log per task, instead of one log per frame.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <hip/hip_ext.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>

#include "config.h"
#include "common.h"

// expected frames of a wave's busiest row from which it asks for issue priority 1 / 2 / 3 (logl_kernel, "wave priority")
#ifndef BILD_PRIO_T1
#ifndef BILD_SNAKE
#define BILD_SNAKE 1 // (0: every layer of a spread work list in the same direction -- A/B builds)
#endif
#ifndef BILD_SPREAD_ALWAYS
#define BILD_SPREAD_ALWAYS 1 // (0: work lists longer than the grid has rows are packed, neighbouring slots to one wave)
#endif
#define BILD_PRIO_T1 100
#define BILD_PRIO_T2 130
#define BILD_PRIO_T3 160
#endif

namespace bild {
namespace {

constexpr double kLog2Pi = 1.8378770664093453;
constexpr double kLn2 = 0.69314718055994531;

// Order LDS traffic between lanes of ONE wavefront: DS instructions of a wave execute in
// issue order, so only the compiler has to be kept from reordering.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sum of one value per lane over the 16 lanes {4b + c + 16k : c, k < 4} that share lane bits 2-3
// ("block" b of v_mfma_f64_4x4x4_4b: operands are laid out as lane = row-or-column + 4*block + 16*k,
// measured with tools/probe/mfma_blocksum.hip), plus `add`; every lane of the block gets the result.
// First product: rows of A summed over k against B = 1; second: the four row sums summed again.
// Two issue slots on the matrix pipe instead of a chain of NP FMAs on the vector pipe.
__device__ __forceinline__ double block_sum16(double v, double add)
{
    const double rows = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
    return __builtin_amdgcn_mfma_f64_4x4x4f64(rows, 1.0, add, 0, 0, 0);
}

// ROW layout: a task occupies one 16-lane row of the wavefront and lane j of the row owns column j of [C | M].
// gfx950 lets a double-precision VOP2 instruction read its first operand through DPP with the row_newbcast
// controls (lane I of the row broadcast to all 16 lanes), so "acc += x[lane I] * w" is ONE instruction and the
// all-gather of C w through LDS (a write, a wave barrier, NP/2 reads and their latency) disappears from the
// frame loop.  hipcc does not fold a v_mov_b64_dpp into the FMA and does not pad the hazards of an asm
// statement: a DPP source needs two wait states after the VALU instruction that wrote it (dpp_ready).
template <int I>
__device__ __forceinline__ void fmac_bcast(double &acc, double x, double w)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w), "n"(I));
}
// orders every later reader of x behind two wait states after its producer
__device__ __forceinline__ void dpp_ready(double &x) { asm volatile("s_nop 1" : "+v"(x)); }

template <int NP, int I = 0>
struct RowOps {
    // sa += sum over even i of x[lane i] w[i], sb likewise over odd i (the order of the packed layouts)
    static __device__ __forceinline__ void dot(double &sa, double &sb, double x, const double (&w)[NP])
    {
        fmac_bcast<I>(sa, x, w[I]);
        fmac_bcast<I + 1>(sb, x, w[I + 1]);
        RowOps<NP, I + 2>::dot(sa, sb, x, w);
    }
    // col[i] += x[lane i] * c
    static __device__ __forceinline__ void rank1(double (&col)[NP], double x, double c)
    {
        fmac_bcast<I>(col[I], x, c);
        fmac_bcast<I + 1>(col[I + 1], x, c);
        RowOps<NP, I + 2>::rank1(col, x, c);
    }
};
template <int NP>
struct RowOps<NP, NP> {
    static __device__ __forceinline__ void dot(double &, double &, double, const double (&)[NP]) {}
    static __device__ __forceinline__ void rank1(double (&)[NP], double, double) {}
};

// log() is needed once per task; out of line, so that its polynomial constants are not hoisted
// out of the task loop into registers the frame loop then has to spill around.
__device__ __attribute__((noinline)) double log_once(double x) { return log(x); }

// The log-likelihood of a piece from its sums: tot = sum over the dimensions of sum_t e_t^2 / S_t, the product of the S_t as P x 2^E, nv
// observed frames, nd dimensions (reference bild/src/MSRouse_logL.pyx:250-254 summed over frames).  One place for the arithmetic of
// the task loop (piece_value) and of the pass that fills the running log-likelihoods of the prefix table (prefix_L_kernel).
__device__ __forceinline__ double piece_from_sums(double tot, double P, int E, int nv, int nd)
{
    const double logS = log_once(P) + (double)E * kLn2;
    tot += (double)nd * (logS + (double)nv * kLog2Pi);
    return -0.5 * tot;
}

template <int NP, int CPL>
struct Cols {
    double v[CPL][NP];
};

// Tg[c][i] = sum_k X[i][k] * in[q][k]  for the own columns c = cidx[q]: the product X * A is
// streamed column-major into the group's LDS image Tg (NC columns of NP doubles), two rows at
// a time, so no second register image of A is needed.  X is row-major in LDS (dense propagator
// or modal basis change); one X operand pair feeds 2*CPL FMAs.
template <int NP, int CPL, bool ROOMY = false, typename XPtr>
__device__ __forceinline__ void matvec_to_lds(XPtr X, const Cols<NP, CPL> &in, double *__restrict__ Tg,
                                              const int (&cidx)[CPL], const bool (&store)[CPL])
{
    // rows per trip: two give each X operand pair 2*CPL independent FMAs; with a single column per
    // lane one row at a time keeps the operand registers of this (rare) path out of the budget of
    // the frame loop -- unless the geometry has registers to spare (ROOMY: two waves per SIMD, the frame loop over the
    // work lists, where a basis change is 2.6 us = nine frames of a chain): then two rows per trip, two trips in flight
    if constexpr (ROOMY && CPL == 1) {
        // Hand-scheduled (round 4).  Left to itself the compiler reads the five 16-byte pieces of a matrix row, WAITS for all
        // of them, runs the row's ten FMAs, and only then asks for the next row: ten LDS latencies per product, 2.2 us per
        // basis change -- ten frames' worth, paid two or three times by every chain that ends a launch.  Here the two rows of
        // trip i + 1 are asked for BEFORE the FMAs of trip i (two register sets, 40 VGPRs: the frame loop's L / sgd / wq are
        // dead across a basis change), and scheduling barriers keep the order: the product is then bound by its hundred FMAs.
        // Every accumulator sees its terms in the order of the loop below (k ascending, even and odd k apart): bit-identical.
        constexpr int H = NP / 2;
        double2 cur[2][H], nxt[2][H];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k2 = 0; k2 < H; ++k2) cur[r][k2] = *reinterpret_cast<const double2 *>(X + r * NP + 2 * k2);
#pragma unroll
        for (int i = 0; i < NP; i += 2) {
            if (i + 2 < NP) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k2 = 0; k2 < H; ++k2) nxt[r][k2] = *reinterpret_cast<const double2 *>(X + (i + 2 + r) * NP + 2 * k2);
            }
            __builtin_amdgcn_sched_barrier(0);
            double a[2] = {0.0, 0.0}, a2[2] = {0.0, 0.0};
#pragma unroll
            for (int k2 = 0; k2 < H; ++k2)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    a[r] = fma(cur[r][k2].x, in.v[0][2 * k2], a[r]);
                    a2[r] = fma(cur[r][k2].y, in.v[0][2 * k2 + 1], a2[r]);
                }
            a[0] += a2[0];
            a[1] += a2[1];
            if (store[0]) *reinterpret_cast<double2 *>(Tg + cidx[0] * NP + i) = make_double2(a[0], a[1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k2 = 0; k2 < H; ++k2) cur[r][k2] = nxt[r][k2];
        }
        return;
    }
    constexpr int R = (CPL == 1 && !ROOMY) ? 1 : 2;
    constexpr int kTrips = ROOMY ? NP / 2 : 1; // (ROOMY: the whole product unrolled, so that the LDS reads of the matrix run ahead of the FMAs)
#pragma unroll kTrips
    for (int i = 0; i < NP; i += R) {
        double a[R][CPL], a2[R][CPL]; // even / odd k: two dependent chains per output instead of one
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < CPL; ++q) a[r][q] = a2[r][q] = 0.0;
#pragma unroll
        for (int k = 0; k < NP; k += 2) {
            double2 x[R];
#pragma unroll
            for (int r = 0; r < R; ++r) x[r] = *reinterpret_cast<const double2 *>(X + (i + r) * NP + k);
#pragma unroll
            for (int q = 0; q < CPL; ++q)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    a[r][q] = fma(x[r].x, in.v[q][k], a[r][q]);
                    a2[r][q] = fma(x[r].y, in.v[q][k + 1], a2[r][q]);
                }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < CPL; ++q) a[r][q] += a2[r][q];
#pragma unroll
        for (int q = 0; q < CPL; ++q)
            if (store[q]) {
                if (R == 2)
                    *reinterpret_cast<double2 *>(Tg + cidx[q] * NP + i) = make_double2(a[0][q], a[R - 1][q]);
                else
                    Tg[cidx[q] * NP + i] = a[0][q];
            }
    }
}

// FLAVOR selects what the frame loop has to carry:
//   0: external force (G != 0) and missing frames   1: missing frames   2: neither (every frame valid)
// LAY: 0 packed groups of G consecutive lanes; 1 a group is a 16-lane block of the f64 4x4x4 matrix instruction
// (S as a block sum on the matrix pipe); 2 a group is a 16-lane row, cross-lane operands by DPP row broadcast;
// 3 / 4: the row / the packed layout in the instantiation that serves the frame loop over the work lists only (kLean)
// DUMP: this instantiation builds the prefix table (one per chain length is compiled, see launch_geom)
// JUMP: this instantiation carries the machinery that takes frames out of the tables (convergence checks, transient table);
// the frame-by-frame instantiation stays lean (the jump code costs registers the frame loop then spills around)
// BUILD: the JUMP instantiation that builds the transient / pair / state tables (KParams::trans_dump, trans2_dump) -- and the only one
// that can: what a builder needs (entries, state records) stays out of the evaluating kernels, and what they need (tails, walk plan,
// landing on tables that do not exist yet) out of the builders.  (The tail code of round 4 cost the builders 70 % while it was
// compiled into both: configs[3] tables 234 -> 406 ms.)
template <int NP, int CPL, int G, int W, int OCC, int LAY, int MODE, int FLAVOR, bool DUMP, bool JUMP, bool BUILD>
__device__ __forceinline__ void logl_body(const KParams &p)
{
    constexpr bool HASG = FLAVOR == 0;
    constexpr bool ALLVALID = FLAVOR == 2;
    constexpr int kThreads = 64 * W;
    constexpr int kWaves = W;
    constexpr int GPW = 64 / G;          // groups (= tasks in flight) per wavefront
    // block layout: a group is one 16-lane block of the f64 4x4x4 matrix instruction (lanes that share
    // bits 2-3), so that S = s2 + w.(Cw) is a block sum on the matrix pipe
    constexpr bool BLK = LAY == 1;
    constexpr bool ROW = LAY == 2 || LAY == 3;
    // the frame loop over the work lists ONLY (geometries 23 / 24 / 25): never builds a table, so the instantiation carries none
    // of the builders' tests in its frame loop, and has a frame loop of its own (below)
    constexpr bool kLean = LAY == 3 || LAY == 4; // (4: the packed layout, likewise for the listed launch only)
    static_assert(LAY == 0 || LAY == 4 || (G == 16 && CPL == 1), "block / row layouts: 16 lanes per task, one column per lane");
    constexpr int MS = table_stride(NP); // LDS matrix stride
    constexpr int SB = StateBlock::size(NP);
    static_assert(NP % 2 == 0, "rows are read in pairs");
    // mean columns a group has room for; the host only selects a geometry whose MC covers the largest
    // number of dimensions any covariance chain of the trajectory set carries (Geometry::mean_slots)
    constexpr int MC = (CPL * G - NP < kDMax) ? CPL * G - NP : kDMax;
    static_assert(MC >= 1, "not enough columns for [C | M]");

    extern __shared__ __align__(16) double smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int grp = BLK ? ((lane >> 2) & 3) : lane / G;
    const int gl = BLK ? ((lane & 3) | ((lane >> 4) << 2)) : lane - grp * G; // ROW: lane / 16, lane % 16

    // Work lists (walk.hip): this launch runs only the tasks the table walk handed on -- kWorkBuckets lists by expected
    // work, dealt out heaviest first.  Slot q of that order goes to (wave, row) so that the heaviest tasks get a wave
    // each while they fit (the other rows of the wave then idle: a row's events -- basis change, comparison, jump --
    // are paid by the whole wave), the next heaviest the second rows, and so on; lists longer than the grid has rows
    // are throughput-bound and tasks of similar work share a wave.
    const bool listed = JUMP && p.work != nullptr;
    int64_t n_tasks = p.ntasks;
    if (listed) {
        int64_t tot = 0;
        for (int bq = 0; bq < kWorkBuckets; ++bq) tot += p.work_counts[bq];
        n_tasks = tot;
        if ((int64_t)blockIdx.x * kWaves >= tot) return; // (all rows of this workgroup would idle)
    }

    // Matrix tables live in LDS for the whole kernel: the dense propagators are needed every frame,
    // and a modal basis change walks its matrix row by row in a dependent loop -- from L2 that was
    // ~500 cycles per row, 10 us per switch (measured: +89 % kernel time at k = 20).
    for (int i = tid; i < p.tab_doubles; i += kThreads) smem[i] = p.tab[i];
    // ... and so do the per-state vectors a state switch reloads (lam | wq | sig: the head of each state block)
    constexpr int HDR = state_header_doubles(NP);
    for (int i = tid; i < p.S * HDR; i += kThreads) smem[p.tab_doubles + i] = p.states[(size_t)(i / HDR) * StateBlock::size(NP) + i % HDR];
    __syncthreads();
    // per-group scratch: image of X*A, NP + kDMax columns of NP doubles; its first NP doubles
    // double as the all-gather buffer of the update
    if (grp >= GPW) return; // lanes beyond the last whole group (64 % G != 0) idle
    const double *const lds_hdr = smem + p.tab_doubles;
    const int lds_tab = p.tab_doubles + p.S * HDR;
    double *const scratch = smem + lds_tab + (size_t)(wv * GPW + grp) * group_image_doubles(NP);
    // the task's segment list (K1 <= kSegLds): a switch then costs two LDS reads instead of two dependent L2 round trips
    int32_t *const seg_lds = reinterpret_cast<int32_t *>(smem + lds_tab + (size_t)(kWaves * GPW) * group_image_doubles(NP)) +
                             (size_t)(wv * GPW + grp) * (2 * group_seg_doubles());
    // ... and the task's constants of the frame loop.  In a register such a value is live across the whole loop and the first
    // to be spilled -- the reload (scratch memory, vmcnt) then sits in every frame and waits for the trajectory prefetch too
    typedef __attribute__((address_space(3))) double lds_double_t;
    lds_double_t *const row_const = (lds_double_t *)reinterpret_cast<double *>(seg_lds + 2 * kSegLds);
    // walk plan (p.walk_lds): per switch of the task, what the tables hold for it -- see `land`
    lds_double_t *const walk =
        (lds_double_t *)(smem + lds_tab + (size_t)(kWaves * GPW) * (group_image_doubles(NP) + group_seg_doubles())) +
        (size_t)(wv * GPW + grp) * kWalkDoubles;

    int cidx[CPL];
    bool isC[CPL], hasImg[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        cidx[q] = gl * CPL + q;
        isC[q] = cidx[q] < NP;
        hasImg[q] = cidx[q] < NP + MC; // spare columns have no LDS image (and stay zero)
    }

    const int S = p.S;
    const int d = p.d;
    const int K1 = p.K1;
    const int64_t gstride = (int64_t)gridDim.x * (kWaves * GPW);
    const int64_t n_waves = (int64_t)gridDim.x * kWaves, wave_id = (int64_t)blockIdx.x * kWaves + wv;
    // (listed, and the list fits the rows of the grid: spread -- row j of wave w takes slot j * n_waves + w, every second
    // layer in reverse: the waves that carry the heaviest tasks of one layer get the lightest of the next, or none)
    const bool spread = listed && (BILD_SPREAD_ALWAYS || n_tasks <= gstride);

    for (int64_t it = spread ? (int64_t)grp : wave_id * GPW + grp;; it += spread ? (int64_t)GPW : gstride) {
        int64_t task = it;
        if (spread) { // `it` counts layers of n_waves slots
            if (it * n_waves >= n_tasks) break;
            task = it * n_waves + (BILD_SNAKE && (it & 1) ? n_waves - 1 - wave_id : wave_id);
            if (task >= n_tasks) continue;
        } else if (task >= n_tasks) {
            break;
        }
        int64_t r, otask;
        int e;
        if (listed) {
            // slot `task` of the buckets laid end to end, heaviest (last) bucket first
            int64_t q = task;
            int bq = kWorkBuckets - 1;
            for (; bq > 0; --bq) {
                const int c = p.work_counts[bq];
                if (q < c) break;
                q -= c;
            }
            otask = p.work[(int64_t)bq * p.work_cap + q];
            r = otask / p.dstar_max;
            e = (int)(otask - r * p.dstar_max);
        } else {
            const int64_t slot = task / p.dstar_max;
            e = (int)(task - slot * p.dstar_max);
            r = p.order ? p.order[slot] : slot; // launch order is a scheduling matter only
            otask = r * p.dstar_max + e;
        }
#ifdef BILD_TASK_CLOCK
        const unsigned long long clock_begin = wall_clock64(); // diagnostics build only: tools/task_clock.py
        unsigned long long clock_events = 0;                   // (== 2: ticks inside comparisons / jumps, and how many)
        int n_events = 0;
#endif
        // (the listed frame loop: a lone task's prologue is a string of dependent memory round trips -- the task's list is asked
        // for HERE, beside the trajectory descriptor, not behind it; lane i holds entry i, lists of up to kSegLds segments)
        int32_t pre_start = 0, pre_state = 0;
        if constexpr (kLean) {
            if (K1 <= kSegLds && gl < K1) {
                pre_start = p.seg_start[r * K1 + gl];
                pre_state = p.seg_state[r * K1 + gl];
            }
        }
        const int tj = p.traj_id ? p.traj_id[r] : 0;
        const TrajDesc *__restrict__ td = p.trajs + tj;
        if (e >= td->dstar) {
            if (gl == 0) p.out[otask] = 0.0;
            continue;
        }
        const int T = td->T;
        const double s2 = td->s2[e];
        if (ROW) {
            if (gl == 0) row_const[0] = s2;
            wave_lds_fence();
        }
        double s2_now = s2; // ROW: read again from LDS at the top of every frame (see row_const)
        const int nd = td->ndims[e];

        bool isM[CPL];
        int xoff[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int mi = cidx[q] - NP;
            isM[q] = (mi >= 0) && (mi < nd);
            xoff[q] = isM[q] ? td->dims[e][mi] : 0;
        }

        // Trajectory stream.  The pointers come out of memory: tell the compiler they are global,
        // or every load becomes a flat_load (which also ties up lgkmcnt).  Each own column walks
        // its own coordinate with a running pointer; columns that are not mean columns read a
        // zero word with stride 0, so that  e = w.col - x  is the innovation sign-flipped for a
        // mean column and plain Cw_j for a covariance column -- one formula, no selects.  The
        // device copy has one padding row per trajectory: fetching one frame ahead never leaves it.
        typedef const __attribute__((address_space(1))) double *gptr_t;
        gptr_t px[CPL];
        int xstep[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            px[q] = isM[q] ? (gptr_t)td->x + xoff[q] : (gptr_t)p.zeros;
            xstep[q] = isM[q] ? d : 0;
        }
        gptr_t pprobe = (gptr_t)td->x; // coordinate 0 is NaN <=> frame missing
        auto fetch = [&](double (&xv)[CPL], double &probe) {
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
#ifdef BILD_X_NOLOAD // timing experiment only (wrong results): what the trajectory loads cost
                xv[q] = 0.25;
#else
                xv[q] = *px[q];
#endif
                px[q] += xstep[q];
            }
            if (!ALLVALID) {
                probe = *pprobe;
                pprobe += d;
            } else {
                probe = 0.0;
            }
        };

        const int32_t *__restrict__ sst = p.seg_start + r * K1;
        const int32_t *__restrict__ ssv = p.seg_state + r * K1;
        // The segment list as given may hold boundaries that are no switches (a segment in the state of its predecessor),
        // empty segments (equal starts) and segments beyond the trajectory.  Lists of up to kSegLds segments are copied
        // to LDS and cleaned there, in place: what remains are the real switches inside the trajectory, strictly
        // increasing -- so that a result depends on the expanded profile only, not on how it was encoded, and a switch
        // costs two LDS reads instead of two dependent L2 round trips.  Longer lists are walked in global memory.
        const bool seg_in_lds = K1 <= kSegLds;
        int nseg = K1;
        if (seg_in_lds) {
            volatile int32_t *const sl = seg_lds;
            if constexpr (kLean && (BLK || ROW)) { // (16 lanes per task: one entry per lane, loaded at the top of the task)
                if (gl < K1) {
                    sl[gl] = pre_start;
                    sl[kSegLds + gl] = pre_state;
                }
            } else {
                for (int i = gl; i < K1; i += (BLK || ROW) ? 16 : G) {
                    sl[i] = sst[i];
                    sl[kSegLds + i] = ssv[i];
                }
            }
            wave_lds_fence();
            int cnt = 1, prev = sl[kSegLds];
            for (int i = 1; i < K1; ++i) {
                const int st0 = sl[i], sv = sl[kSegLds + i];
                const int end = (i + 1 < K1) ? sl[i + 1] : INT_MAX;
                if (st0 >= T) break;                    // starts are sorted: nothing behind this one is inside either
                if (end <= st0 || sv == prev) continue; // empty, or not a change of state
                sl[cnt] = st0;                          // cnt <= i: entries still to be read are not touched
                sl[kSegLds + cnt] = sv;
                prev = sv;
                ++cnt;
            }
            nseg = cnt;
            wave_lds_fence();
        }
        auto seg_start_of = [&](int i) { return seg_in_lds ? seg_lds[i] : sst[i]; };
        auto seg_state_of = [&](int i) { return seg_in_lds ? seg_lds[kSegLds + i] : ssv[i]; };
        // Wave priority.  The launch ends when its longest chain of short segments has been run, frame by frame, by ONE
        // row -- on a SIMD shared with two other waves that mostly have slack.  A wave that expects a long run asks the
        // issue arbiter for priority (frames to run estimated from the switch frames, as the host scheduler does):
        // 136 -> 119 us for the 10k batch in array order, 116 -> 114 us in the scheduler's order
        // (profiles/r02_wave_priority.txt).  A matter of speed only.
        if (JUMP && seg_in_lds && p.trans != nullptr) {
            int w = 0, run_from = -1, links = 0; // a chain: switches less than m_typ frames apart, run as one piece
            const int mt = p.m_typ;
            const bool pairs = p.trans2 != nullptr;
            for (int i = 1; i < nseg; ++i) {
                const int t1 = seg_lds[i];
                const int gap = ((i + 1 < nseg) ? seg_lds[i + 1] : T) - t1;
                if (run_from < 0) {
                    if (gap < mt) {
                        run_from = t1;
                        links = 1;
                    }
                } else {
                    ++links;
                    if (gap >= mt) { // the chain ends m_typ frames behind this switch
                        if (!(pairs && links == 2)) w += t1 + mt - run_from;
                        run_from = -1;
                    }
                }
            }
            // a chain that reaches the end of the trajectory: one switch, or two with the pair table, are table entries too
            if (run_from >= 0 && !(links == 1 || (pairs && links == 2))) w += T - run_from;
            if (__ballot(w >= BILD_PRIO_T3)) __builtin_amdgcn_s_setprio(3);
            else if (__ballot(w >= BILD_PRIO_T2)) __builtin_amdgcn_s_setprio(2);
            else if (__ballot(w >= BILD_PRIO_T1)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        int seg = 0;
        int s = seg_state_of(0);
#if defined(BILD_TASK_CLOCK) && BILD_TASK_CLOCK == 3
        const unsigned long long clock_a = wall_clock64(); // descriptor read, segment list cleaned
#endif
        int next_start = (nseg > 1) ? seg_start_of(1) : INT_MAX;

        // ---- per-state registers -------------------------------------------------
        // measurement vector in the current basis.  Row layout, jump instantiation: NOT kept in registers -- the same for
        // all lanes of a row, it is read from the state header in LDS at the top of every update (five 16-byte reads, behind
        // the predict), and the twenty registers go to the event paths, which otherwise spill around the frame loop
        // (at two waves per SIMD -- the geometry of the frame loop over the work lists, id 23 -- there are registers enough:
        // a lone wave then does not wait for five LDS reads in front of every update)
        constexpr bool kWqFromLds = ROW && JUMP && OCC >= 3;
        double wq[NP];
        // modal predict  C'_ij <- lam_i lam_j C'_ij + sig_i delta_ij,  M'_i <- lam_i M'_i  as ONE
        // fma per entry: L[q][i] = lam_i * (lam_c or 1), sgd[i] = sig_c on the own diagonal entry
        // (row i == column c of column slot q = i % CPL) and 0 elsewhere.  Rebuilt at switches.
        Cols<NP, CPL> L;
        double sgd[NP];
        double wown = 0.0; // block layout: w_c of the own covariance column, 0 for the other lanes
        auto load_state = [&](int st) {
            const double *__restrict__ sb = lds_hdr + (size_t)st * HDR; // lam | wq | sig at the offsets of the state block
            if constexpr (!kWqFromLds) {
#pragma unroll
                for (int i = 0; i < NP; ++i) wq[i] = sb[StateBlock::wq(NP) + i];
            }
            if (BLK) wown = isC[0] ? sb[StateBlock::wq(NP) + cidx[0]] : 0.0;
            if (MODE == kModal) {
                double mu[CPL], sgc[CPL];
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    const int cc = isC[q] ? cidx[q] : 0;
                    mu[q] = isC[q] ? sb[StateBlock::lam(NP) + cc] : (hasImg[q] ? 1.0 : 0.0);
                    sgc[q] = isC[q] ? sb[StateBlock::sig(NP) + cc] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const double li = sb[StateBlock::lam(NP) + i];
#pragma unroll
                    for (int q = 0; q < CPL; ++q) L.v[q][i] = li * mu[q];
                    sgd[i] = (gl == i / CPL) ? sgc[i % CPL] : 0.0;
                }
            }
        };
        load_state(s);

        // ---- initial condition: steady state of state profile[0] (pyx:160-163) ----
        // (a task that starts from a table -- every task of the jump instantiation -- loads its state there: start_from)
        Cols<NP, CPL> col;
        if (JUMP && p.prefix != nullptr && !p.no_jump) {
#pragma unroll
            for (int q = 0; q < CPL; ++q)
#pragma unroll
                for (int i = 0; i < NP; ++i) col.v[q][i] = 0.0;
        } else {
            const double *__restrict__ sb = p.states + (size_t)s * SB;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const double *src = isC[q] ? sb + StateBlock::C0(NP) + cidx[q] * NP
                                           : sb + StateBlock::M0(NP) + xoff[q] * NP;
                const double live = (isC[q] || isM[q]) ? 1.0 : 0.0;
#pragma unroll
                for (int i = 0; i < NP; ++i) col.v[q][i] = live * src[i];
            }
        }

        double accq[CPL]; // per own column: sum of e^2 / S (meaningful for mean columns)
#pragma unroll
        for (int q = 0; q < CPL; ++q) accq[q] = 0.0;
        double P = 1.0; // running product of S (mantissa), exponent in E
        int E = 0;
        int nv = 0;     // observed frames behind these accumulators

        // ---- Kalman update (pyx:19-90) ---------------------------------------------
        auto fetch_wq = [&]() { // (kWqFromLds) at the top of every update: issued behind the predict by the scheduler
            if constexpr (kWqFromLds) {
                typedef double __attribute__((ext_vector_type(2))) d2_t;
                typedef const __attribute__((address_space(3))) d2_t *wq_ptr_t; // (the address moves with s: nothing to hoist)
                const wq_ptr_t wp = (wq_ptr_t)(lds_hdr + (size_t)s * HDR + StateBlock::wq(NP));
#pragma unroll
                for (int i = 0; i < NP; i += 2) {
                    const d2_t w2 = wp[i / 2];
                    wq[i] = w2.x;
                    wq[i + 1] = w2.y;
                }
            }
        };
        auto update = [&](const double (&xv)[CPL]) {
            fetch_wq();
            double ev[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int i = 0; i < NP; i += 2) {
                    a0 = fma(wq[i], col.v[q][i], a0);
                    a1 = fma(wq[i + 1], col.v[q][i + 1], a1);
                }
                ev[q] = (a0 + a1) - xv[q]; // covariance column: (C w)_c ; mean column: -(x - w.M)
            }
            if constexpr (ROW) {
                // C w is never gathered: lane i of the row holds (C w)_i in ev and every use reads it by row broadcast
                dpp_ready(ev[0]);
                double sa = s2_now, sb2 = 0.0;
                RowOps<NP>::dot(sa, sb2, ev[0], wq);
                const double Sv = sa + sb2;
                double Sinv = __builtin_amdgcn_rcp(Sv);
                Sinv = fma(fma(-Sv, Sinv, 1.0), Sinv, Sinv);
                Sinv = fma(fma(-Sv, Sinv, 1.0), Sinv, Sinv);
                const double coef = ev[0] * Sinv;
                accq[0] = fma(ev[0], coef, accq[0]);
                RowOps<NP>::rank1(col.v[0], ev[0], -coef); // col[i] += (C w)_i * (-coef) = fma(-coef, cw[i], col[i])
                int ex;
                P = frexp(P * Sv, &ex);
                E += ex;
                ++nv;
                return;
            }
            // every lane publishes (no exec masking): mean / spare columns land behind the NP
            // gathered entries (cidx < CPL*G <= group image size)
#pragma unroll
            for (int q = 0; q < CPL; ++q) scratch[cidx[q]] = ev[q];
            wave_lds_fence();
            double cw[NP];
#pragma unroll
            for (int i = 0; i < NP; i += 2) {
                const double2 t2 = *reinterpret_cast<const double2 *>(scratch + i);
                cw[i] = t2.x;
                cw[i + 1] = t2.y;
            }
            wave_lds_fence();
            double Sv;
            if (BLK) {
                Sv = block_sum16(wown * ev[0], s2);
            } else {
                double sa = s2, sb2 = 0.0;
#pragma unroll
                for (int i = 0; i < NP; i += 2) {
                    sa = fma(wq[i], cw[i], sa);
                    sb2 = fma(wq[i + 1], cw[i + 1], sb2);
                }
                Sv = sa + sb2;
            }
            // 1/S: hardware reciprocal seed + two Newton steps (full double precision for the
            // normal, positive S a covariance produces; NaN/Inf/0 propagate as such)
            double Sinv = __builtin_amdgcn_rcp(Sv);
            Sinv = fma(fma(-Sv, Sinv, 1.0), Sinv, Sinv);
            Sinv = fma(fma(-Sv, Sinv, 1.0), Sinv, Sinv);
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const double coef = ev[q] * Sinv; // K_c * S for a covariance column, -nu/S for a mean column
                accq[q] = fma(ev[q], coef, accq[q]); // e^2 / S
#pragma unroll
                for (int i = 0; i < NP; ++i) col.v[q][i] = fma(-coef, cw[i], col.v[q][i]);
            }
            int ex;
            P = frexp(P * Sv, &ex);
            E += ex;
            ++nv;
        };

        // A <- X A for all columns, then C <- C X^T for the covariance part: two lane-local
        // multiplies, each streamed through the group's LDS image; the covariance columns are
        // read back TRANSPOSED after the first one (row c of X*C is column c of (X*C)^T, and
        // X (X C)^T = (X C X^T)^T = X C X^T).  Used for the dense predict (X = B_s) and the
        // modal basis change (X = R).  `after_left` runs on the mean columns between the
        // two multiplies (adds G in the dense predict).
        auto sandwich = [&](auto X, auto &&after_left) {
            matvec_to_lds<NP, CPL, (JUMP && ((ROW && OCC <= 2) || kLean))>(X, col, scratch, cidx, hasImg);
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                // covariance: walk row c of the image; mean: own column; spare: anything in range
                // (tried: storing the covariance block ACROSS the image, so that this transposed read becomes five 16-byte
                // reads and one wait instead of ten 8-byte reads each waited for -- the basis change gained 0.1 us and the
                // frame loop, re-allocated by the compiler, lost 0.04 us per frame: 53.7 -> 56.9 us at k = 4; dropped)
                const int c = isC[q] ? cidx[q] : (hasImg[q] ? cidx[q] * NP : 0);
                const int st = isC[q] ? NP : 1;
                const double keep = hasImg[q] ? 1.0 : 0.0;
                if constexpr (JUMP && ((ROW && OCC <= 2) || kLean)) {
                    // (all reads in flight, ONE wait: the compiler's own order is read, wait, multiply, ten times over)
                    double tmp[NP];
#pragma unroll
                    for (int i = 0; i < NP; ++i) tmp[i] = scratch[c + i * st];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NP; ++i) col.v[q][i] = keep * tmp[i];
                } else {
#pragma unroll
                    for (int i = 0; i < NP; ++i) col.v[q][i] = keep * scratch[c + i * st];
                }
            }
            wave_lds_fence();
            after_left();
            matvec_to_lds<NP, CPL, (JUMP && ((ROW && OCC <= 2) || kLean))>(X, col, scratch, cidx, isC);
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < CPL; ++q)
                if (isC[q]) {
#pragma unroll
                    for (int i = 0; i < NP; i += 2) {
                        const double2 t2 = *reinterpret_cast<const double2 *>(scratch + cidx[q] * NP + i);
                        col.v[q][i] = t2.x;
                        col.v[q][i + 1] = t2.y;
                    }
                }
            wave_lds_fence();
        };

#ifndef BILD_JUMP_FIRST
#define BILD_JUMP_FIRST 24
#endif
        constexpr int kJumpFirst = BILD_JUMP_FIRST; // first comparison this many frames behind a switch
        int t_check = 0; // frame index (frames < t_check are processed) at which the next convergence check is due
        int s_loaded = s; // state whose vectors (wq, L, sgd) are in registers
        // the switch the current segment began with -- its frame, and the state in front of it (first-order tail) -- lives in the
        // task's second LDS constant: two registers live across the frame loop cost the geometries that are short of them
        // another round of spills (the pair table's builder, (2, 7): 420 -> 700 us when they were registers)
        auto note_switch = [&](int s_from, int t_seg) {
            row_const[1] = __hiloint2double(s_from, t_seg);
        };
        note_switch(s, 0);
        // frame t is the first of a new segment: state bookkeeping, and the basis change of a real switch
        auto enter_segment = [&](int t) {
            {
                do {
                    ++seg;
                    next_start = (seg + 1 < nseg) ? seg_start_of(seg + 1) : INT_MAX;
                } while (t >= next_start); // (never loops on a cleaned list)
                const int sn = seg_state_of(seg);
                if (sn != s) {
                    if (MODE == kModal) {
                        // basis change: one matrix R[sn][s], or -- many states -- out of the old basis (Q[s]) and
                        // into the new one (Q[sn]^T)
                        const int steps = p.tab_factored ? 2 : 1;
                        for (int st = 0; st < steps; ++st) {
                            const int slot = p.tab_factored ? (st == 0 ? s : S + sn) : sn * S + s;
#ifndef BILD_EXPERIMENT_NO_SANDWICH // timing experiment only (wrong results): what the basis change itself costs
                            sandwich(const_cast<const double *>(smem) + (size_t)slot * MS, [] {});
#endif
                        }
                    }
                    note_switch(s, t);
                    s = sn;
                    load_state(s);
                    s_loaded = s;
                    t_check = t + kJumpFirst;
                }
            }
        };
        // one frame t >= 1: predict (pyx:206-241), masked update (pyx:244-248)
        auto frame = [&](const double (&xv)[CPL], double probe) {
            if (ROW && kLean) s2_now = row_const[0];
            if (MODE == kModal) {
#pragma unroll
                for (int q = 0; q < CPL; ++q)
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        col.v[q][i] = (q == i % CPL) ? fma(L.v[q][i], col.v[q][i], sgd[i]) : L.v[q][i] * col.v[q][i];
                if (HASG) {
                    const double *__restrict__ gb = p.states + (size_t)s * SB + StateBlock::G(NP);
#pragma unroll
                    for (int q = 0; q < CPL; ++q)
                        if (isM[q]) {
#pragma unroll
                            for (int i = 0; i < NP; ++i) col.v[q][i] += gb[xoff[q] * NP + i];
                        }
                }
            } else {
                const double *__restrict__ sb = p.states + (size_t)s * SB;
                constexpr bool has_G = HASG;
                sandwich(const_cast<const double *>(smem) + (size_t)s * MS, [&] {
                    if (has_G) {
#pragma unroll
                        for (int q = 0; q < CPL; ++q)
                            if (isM[q]) {
#pragma unroll
                                for (int i = 0; i < NP; ++i) col.v[q][i] += sb[StateBlock::G(NP) + xoff[q] * NP + i];
                            }
                    }
                });
                // + Sig[:, c] (symmetric: row c of the LDS copy)
                const double *__restrict__ sg = smem + (size_t)(S + s) * MS;
#pragma unroll
                for (int q = 0; q < CPL; ++q)
                    if (isC[q]) {
#pragma unroll
                        for (int i = 0; i < NP; i += 2) {
                            const double2 t2 = *reinterpret_cast<const double2 *>(sg + cidx[q] * NP + i);
                            col.v[q][i] += t2.x;
                            col.v[q][i + 1] += t2.y;
                        }
                    }
            }
            if (ALLVALID || !isnan(probe)) update(xv);
        };

        // ---- which frames to run at all ------------------------------------------------------------------------------
        // Without tables: frame 0 is an update on the steady state without a predict (pyx:186-190), then every frame.
        // With the tables of the trajectory set (common.h), a task runs only where its filter differs from the switch-free
        // filter of its current state:
        //  * the part in front of its first switch is a difference of the running log-likelihood L_s(t) of that filter;
        //  * behind a switch the filter forgets where it came from at a geometric rate (a few tens of frames for a Rouse
        //    chain).  From kJumpFirst frames behind the switch on -- and then after as many frames as the measured deviation
        //    still needs at the usual rate of decay -- the task compares its whole state [C | M] with the table's record of
        //    the same frame and state.  Equal states, same propagator and same data give equal futures: once they agree to
        //    kJumpTol (relative to the largest entry of each column), the rest of the segment is again a difference of
        //    L_s(t), and the task is back at a "synchronised point" in front of its next switch.  No stationarity is
        //    assumed: a task that never converges (long gaps, slow modes) simply runs every frame;
        //  * a transient that starts at a synchronised point depends on (trajectory, chain, frame, old state, new state)
        //    only -- not on the candidate.  The TRANSIENT TABLE holds, for every such switch, how many frames m it takes to
        //    converge and what it adds to the log-likelihood beyond the table's own sums.  A task whose next switch is at
        //    least m frames away takes that entry and runs nothing.
        // The pieces are sums of O(1e4) numbers taken apart and put together again: a result deviates from the
        // frame-by-frame one by ~1e-11 (BILD_NO_JUMP runs every frame behind the first switch, bit-identical to
        // BILD_NO_PREFIX).  Rows of a wavefront are independent: each has its own frame counter and trajectory pointers.
        constexpr int NC = NP + kDMax;
        constexpr int REC = prefix_record_doubles(NP);
        constexpr int kRecP = NC * NP + kDMax, kRecE = kRecP + 1, kRecL = kRecP + 2, kRecNv = kRecP + 3;
        constexpr double kJumpTol = 1.1368683772161603e-13; // 2^-43
        const bool restore = !DUMP && p.prefix != nullptr;
        const bool jumping = JUMP && restore && !p.no_jump;
        // (launch_geom: a BUILD instantiation is launched when, and only when, p.trans_dump or p.trans2_dump is set -- with the prefix
        // table in place and jumps allowed)
        constexpr bool building_transients = BUILD;
        const bool use_transients = jumping && p.trans != nullptr && !building_transients;
        auto record_of = [&](int st, int t) { return p.prefix + (td->prefix_rec0 + ((int64_t)e * S + st) * T + t) * REC; };
        auto record = [&](int t) { return record_of(s, t); };
        auto load_cols = [&](const double *__restrict__ rec) {
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const double keep = (isC[q] || isM[q]) ? 1.0 : 0.0;
                const double *src = rec + (hasImg[q] ? cidx[q] : 0) * NP;
#pragma unroll
                for (int i = 0; i < NP; i += 2) {
                    const double2 t2 = *reinterpret_cast<const double2 *>(src + i);
                    col.v[q][i] = keep * t2.x;
                    col.v[q][i + 1] = keep * t2.y;
                }
            }
        };
        // lanes of this task, as a mask over the wavefront (for the row-wide verdict of a comparison)
        unsigned long long group_mask;
        if (BLK) group_mask = 0x000F000F000F000Full << (4 * grp);
        else group_mask = (G == 64 ? ~0ull : ((1ull << G) - 1)) << (grp * G);
        // log-likelihood of the frames behind the accumulators (pyx:88, 251-256), the same in every lane of the task
        auto piece_value = [&]() {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) acc += isM[q] ? accq[q] : 0.0;
            scratch[gl] = acc;
            wave_lds_fence();
            double tot = 0.0;
#pragma unroll
            for (int g2 = 0; g2 < G; ++g2) tot += scratch[g2];
            wave_lds_fence();
            return piece_from_sums(tot, P, E, nv, nd);
        };
        // sum of one value per lane over the lanes of the task, in lane order (the same in every lane of the task)
        auto group_sum = [&](double v) {
            scratch[gl] = v;
            wave_lds_fence();
            double tot = 0.0;
#pragma unroll
            for (int g2 = 0; g2 < G; ++g2) tot += scratch[g2];
            wave_lds_fence();
            return tot;
        };
        auto reset_piece = [&]() {
#pragma unroll
            for (int q = 0; q < CPL; ++q) accq[q] = 0.0;
            P = 1.0;
            E = 0;
            nv = 0;
        };

        int t = 1;
        int nrun = 0; // frames run: minus the first frame of the open run while it lasts, plus its last frame + 1 when it ends
        double extra = 0.0; // finished pieces: table differences, transient entries, own pieces that have ended
        // the accumulators hold a piece that is not in `extra` yet (lean: an int -- a loop-carried bool of divergent lanes is a lane
        // mask in scalar registers, merged at the bottom of EVERY frame)
        typename std::conditional<kLean, int, bool>::type open_run = 1;
        double xc[CPL], xn[CPL], pc, pn;
        // start a run of own frames at frame t0 from the table's state in front of it; the trajectory pointers stand at
        // frame t_ptr (0 at the start of a task, t + 1 inside the frame loop)
        auto start_from = [&](const double *__restrict__ rec, int t0, bool cumulative, int t_ptr) {
            load_cols(rec);
            if (cumulative) { // accumulators continue the table's (BILD_NO_JUMP: bit-identical to the run from frame 0)
#pragma unroll
                for (int q = 0; q < CPL; ++q) accq[q] = isM[q] ? rec[NC * NP + (cidx[q] - NP)] : 0.0;
                P = rec[kRecP];
                E = (int)rec[kRecE];
                nv = (int)rec[kRecNv];
            } else {
                reset_piece();
            }
            if (s_loaded != s) {
                load_state(s);
                s_loaded = s;
            }
#pragma unroll
            for (int q = 0; q < CPL; ++q) px[q] += (int64_t)xstep[q] * (t0 - t_ptr);
            if (!ALLVALID) pprobe += (int64_t)d * (t0 - t_ptr);
            fetch(xn, pn); // frame t0 (or the first padding row)
            nrun -= t0;
            t_check = t0 + 8; // (a switch at t0 sets its own; lists too long to be cleaned may hold boundaries that switch nothing)
            open_run = 1;
        };
        auto start_run = [&](int t0, bool cumulative, int t_ptr) { start_from(record(t0 - 1), t0, cumulative, t_ptr); };
        // A chain of close switches begins at the synchronised point in front of frame t (the start of segment seg + 1).  With
        // the transient state table (common.h) it begins at its SECOND switch instead: the state the first transient has
        // reached there, accumulators included, is a record of the table -- written by the candidate that built the transient
        // table's entry for this switch, which ran the very same frames from the very same record.
        auto begin_chain = [&](int t_ptr) {
            if constexpr (JUMP) {
                if (p.strans != nullptr && seg_in_lds && seg + 2 < nseg) {
                    const int sn = seg_state_of(seg + 1), t2 = seg_start_of(seg + 2), g1 = t2 - t;
                    if (sn != s && g1 >= 1 && g1 < p.sgap) {
                        // the table keeps every `sstride`-th gap (g = 1, 1 + sstride, ...): the chain starts from the last
                        // record at or in front of its second switch and runs the frames in between itself (at most
                        // sstride - 1 of them, in the state of the gap) -- the frames the candidate would have run anyway
                        const int q = (g1 - 1) / p.sstride, g0 = 1 + q * p.sstride;
                        const int64_t entry = (((int64_t)e * S + s) * (S - 1) + (sn - (sn > s ? 1 : 0))) * T + t;
                        const double *__restrict__ rec = p.strans + ((td->strans0 + entry) * p.snq + q) * REC;
                        ++seg; // the segment of sn: the frame at t2 moves on to the next one
                        note_switch(s, t);
                        s = sn;
                        next_start = t2;
                        t += g0;
                        start_from(rec, t, true, t_ptr);
                        return;
                    }
                }
            }
            start_run(t, false, t_ptr);
        };
        // The walk plan.  Which table entries a task may need is known from its (cleaned) segment list alone: for the switch
        // into segment i, the transient entry of (state i-1 -> state i, frame start_i) and, where segment i is shorter than the
        // pair table's range, the pair entry of (state i-1 -> i -> i+1, start_i, length of segment i) -- each with the running
        // sums of the new state's filter up to the switch behind.  Lane i - 1 of the task fetches those of switch i, all at
        // once, and leaves the two sums and the two frame counts in LDS; `land` then walks from switch to switch without
        // waiting for memory (four switches: 6.6 -> ~2 us of a task's life; the same numbers, added in the same order).
        const bool planned = use_transients && p.walk_lds != 0 && seg_in_lds;
        if (planned) {
            for (int i = 1 + gl; i < nseg; i += (BLK || ROW) ? 16 : G) {
                const int ti = seg_lds[i], s0 = seg_lds[kSegLds + i - 1], s1 = seg_lds[kSegLds + i];
                const int n2 = (i + 1 < nseg) ? seg_lds[i + 1] : INT_MAX;
                const int t3 = n2 < T ? n2 : T;
                const TransEntry en = p.trans[td->trans0 + (((int64_t)e * S + s0) * S + s1) * T + ti];
                const double la = record_of(s1, ti - 1)[kRecL], lb = record_of(s1, t3 - 1)[kRecL];
                double v2 = 0.0;
                int m2 = 0;
                if constexpr (kLean) {
                    // every load of the plan in ONE round trip: the pair entry and its running sums are asked for
                    // unconditionally, with indices of entry 0 where there is no pair (walk.hip does the same)
                    const bool pair_ok = p.trans2 != nullptr && n2 < T && n2 - ti < p.gap_max;
                    const int sm = pair_ok ? seg_lds[kSegLds + i + 1] : 0;
                    const int n4 = (pair_ok && i + 2 < nseg) ? seg_lds[i + 2] : INT_MAX;
                    const int t4 = n4 < T ? n4 : T;
                    const TransEntry *tab2 = p.trans2 != nullptr ? p.trans2 : p.trans;
                    const int64_t i2 = pair_ok ? (td->trans0 * S + ((((int64_t)e * S + s0) * S + s1) * S + sm) * T + ti) * p.gap_max + (n2 - ti) : td->trans0;
                    const TransEntry e2 = tab2[i2];
                    const double l4 = record_of(sm, t4 - 1)[kRecL], l2 = record_of(sm, ti - 1)[kRecL];
                    if (pair_ok) {
                        v2 = e2.c + (l4 - l2);
                        m2 = e2.m;
                    }
                } else if (p.trans2 != nullptr && n2 < T && n2 - ti < p.gap_max) {
                    const int sm = seg_lds[kSegLds + i + 1];
                    const int n4 = (i + 2 < nseg) ? seg_lds[i + 2] : INT_MAX;
                    const int t4 = n4 < T ? n4 : T;
                    const TransEntry e2 =
                        p.trans2[(td->trans0 * S + ((((int64_t)e * S + s0) * S + s1) * S + sm) * T + ti) * p.gap_max + (n2 - ti)];
                    v2 = e2.c + (record_of(sm, t4 - 1)[kRecL] - record_of(sm, ti - 1)[kRecL]);
                    m2 = e2.m;
                }
                walk[3 * i] = en.c + (lb - la);
                walk[3 * i + 1] = v2;
                walk[3 * i + 2] = __hiloint2double(m2, en.m);
            }
            // ... and, by the task's last lane (slot 0 belongs to no switch), the running sum in front of the first switch
            if (gl == ((BLK || ROW) ? 15 : G - 1)) {
                const int t1 = (nseg > 1 && seg_lds[1] < T) ? seg_lds[1] : T;
                walk[0] = record_of(seg_lds[kSegLds], t1 - 1)[kRecL];
            }
            wave_lds_fence();
        }
        // at a synchronised point in front of frame t (== next_start, or T): take whatever the tables hold
        auto land = [&]() {
            while (planned && t < T) { // t == start of segment seg + 1
                const int i = seg + 1;
                const double v1 = walk[3 * i], v2 = walk[3 * i + 1], mm = walk[3 * i + 2];
                const int m1 = __double2loint(mm), m2 = __double2hiint(mm);
                const int nn = (seg + 2 < nseg) ? seg_start_of(seg + 2) : INT_MAX;
                const int t3 = nn < T ? nn : T;
                if (m1 > 0 && t + m1 <= t3) {
                    extra += v1;
                    ++seg;
                    s = seg_state_of(seg);
                    next_start = nn;
                    t = t3;
                    continue;
                }
                if (m2 <= 0) break;
                const int n4 = (seg + 3 < nseg) ? seg_start_of(seg + 3) : INT_MAX;
                const int t4 = n4 < T ? n4 : T;
                if (t + m2 > t4) break;
                extra += v2;
                seg += 2;
                s = seg_state_of(seg);
                next_start = n4;
                t = t4;
            }
            while (!planned && t < T && use_transients) {
                const int sn = seg_state_of(seg + 1);
                if (sn == s) break; // (uncleaned list) not a switch: run on
                const int nn = (seg + 2 < nseg) ? seg_start_of(seg + 2) : INT_MAX;
                const int t3 = nn < T ? nn : T;
                const TransEntry en = p.trans[td->trans0 + (((int64_t)e * S + s) * S + sn) * T + t];
                const double la = record_of(sn, t - 1)[kRecL], lb = record_of(sn, t3 - 1)[kRecL];
                if (en.m > 0 && t + en.m <= t3) {
                    extra += en.c + (lb - la);
                    ++seg;
                    s = sn;
                    next_start = nn;
                    t = t3;
                    continue;
                }
                // the transient reaches into the next switch: the two together may be in the pair table
                if (p.trans2 == nullptr || nn >= T || nn - t >= p.gap_max || nn <= t) break;
                const int sm = seg_state_of(seg + 2);
                if (sm == sn) break; // (uncleaned list)
                const int n4 = (seg + 3 < nseg) ? seg_start_of(seg + 3) : INT_MAX;
                const int t4 = n4 < T ? n4 : T;
                const TransEntry e2 =
                    p.trans2[(td->trans0 * S + ((((int64_t)e * S + s) * S + sn) * S + sm) * T + t) * p.gap_max + (nn - t)];
                if (e2.m <= 0 || t + e2.m > t4) break; // a chain of three or more: run it
                extra += e2.c + (record_of(sm, t4 - 1)[kRecL] - record_of(sm, t - 1)[kRecL]);
                seg += 2;
                s = sm;
                next_start = n4;
                t = t4;
            }
        };
#if defined(BILD_TASK_CLOCK) && BILD_TASK_CLOCK == 3
        const unsigned long long clock_b = wall_clock64(); // state vectors loaded, lambdas set up
        unsigned long long clock_c = clock_b;
#endif
        if (jumping) {
            open_run = 0;
            t = next_start < 1 ? 1 : (next_start < T ? next_start : T); // first switch (>= 1: segment 0 owns frame 0), or T
            if (!building_transients) {
                extra = planned ? (double)walk[0] : record(t - 1)[kRecL];
                land();
            }
#if defined(BILD_TASK_CLOCK) && BILD_TASK_CLOCK == 3
            clock_c = wall_clock64(); // tables walked
#endif
            if (t < T) begin_chain(0);
        } else if (restore) {
            t = next_start < 1 ? 1 : (next_start < T ? next_start : T);
            start_run(t, true, 0);
        } else {
            nrun = -1; // the run starts at frame 1
            fetch(xc, pc); // frame 0
            fetch(xn, pn); // frame 1 (or the first padding row)
            if (ALLVALID || !isnan(pc)) update(xc);
        }
        // this launch builds the prefix table: the state after every frame goes to its record (tasks have K1 = 1, s is fixed)
        auto dump = [&](int tt) {
            double *__restrict__ rec = p.prefix_dump + (td->prefix_rec0 + ((int64_t)e * S + s) * T + tt) * REC;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                if (hasImg[q]) {
#pragma unroll
                    for (int i = 0; i < NP; i += 2)
                        *reinterpret_cast<double2 *>(rec + cidx[q] * NP + i) = make_double2(col.v[q][i], col.v[q][i + 1]);
                }
                if (isM[q]) rec[NC * NP + (cidx[q] - NP)] = accq[q];
            }
            // (the running log-likelihood of the switch-free filter, kRecL and its dense copy, is filled in behind this launch from
            // the sums stored here -- prefix_L_kernel, one thread per record: computed here it was a logarithm and an LDS round
            // trip in every frame of a launch that is two wavefronts per trajectory, a quarter of the builder's 0.94 us per frame)
            if (gl == 0) {
                rec[kRecP] = P;
                rec[kRecE] = (double)E;
                rec[kRecNv] = (double)nv;
            }
        };
        // this launch builds the transient table and, beside it, the transient state table (common.h): the state after
        // every frame of the transient goes to its record (tasks have two segments: the switch is at sst[1])
        const bool dump_states = JUMP && !kLean && building_transients && p.strans_dump != nullptr && K1 == 2 && nseg == 2;
        const int t_switch = dump_states ? seg_start_of(1) : 0;
        const int64_t state_rec0 = dump_states ? (td->strans0 + (((int64_t)e * S + seg_state_of(0)) * (S - 1) +
                                                                 (seg_state_of(1) - (seg_state_of(1) > seg_state_of(0) ? 1 : 0))) * T + t_switch) * p.snq
                                               : 0;
        auto dump_state = [&](int g) {
            double *__restrict__ rec = p.strans_dump + (state_rec0 + g) * REC;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                if (hasImg[q]) {
#pragma unroll
                    for (int i = 0; i < NP; i += 2)
                        *reinterpret_cast<double2 *>(rec + cidx[q] * NP + i) = make_double2(col.v[q][i], col.v[q][i + 1]);
                }
                if (isM[q]) rec[NC * NP + (cidx[q] - NP)] = accq[q];
            }
            if (gl == 0) {
                rec[kRecP] = P;
                rec[kRecE] = (double)E;
                rec[kRecNv] = (double)nv;
            }
        };
        if constexpr (DUMP) dump(0);
        // A comparison with the table is due at frame t (the state after frame t - 1 against the table's record): either the
        // next look is scheduled, or the own piece ends here and the task goes on at its next synchronised point.
        // Returns true when the task is finished with its frames (a table-building launch whose transient has converged).
        auto compare_with_table = [&]() -> bool {
#if defined(BILD_TASK_CLOCK) && BILD_TASK_CLOCK == 2
            const unsigned long long ev0 = wall_clock64();
#endif
            const double *__restrict__ rec = record(t - 1);
            // per lane: largest deviation of the own column(s) from the table's, in units of the tolerance
            double excess = 0.0;
            bool same = true;
            // first-order tail (tail.hip): the covariance columns must have converged as before; a mean column may still be
            // kTailTol = 2^-20 away -- what that deviation does to every later frame is a dot product with the table's g
            const double kTailTol = p.tail_tol;  // (2^-20 unless BILD_TAIL_TOL_BITS says otherwise: tools/tail_tolerance.py)
            const int kTailMargin = p.tail_margin; // frames beyond the table's own transient before the next switch may come (8)
            const bool tails = JUMP && p.tail_g != nullptr && !building_transients;
            bool near = true;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                if (!hasImg[q]) continue;
                // (mean columns: the floor is the scale of the DATA as far as the model can explain it -- for data drawn
                // from the model the largest coordinate itself, 4-5 standard deviations of the innovation; for data the model
                // did not produce (an offset, outliers) the innovations e grow with the data, the log-likelihood error of a
                // deviation d is ~ 20 |e| d / S, and the floor must not grow with them: TrajDesc::mscale)
                double dev = 0.0, ref = isM[q] ? td->mscale[e] : 0.0;
#pragma unroll
                for (int i = 0; i < NP; i += 2) {
                    const double2 r2 = *reinterpret_cast<const double2 *>(rec + cidx[q] * NP + i);
                    dev = fmax(dev, fmax(fabs(col.v[q][i] - r2.x), fabs(col.v[q][i + 1] - r2.y)));
                    ref = fmax(ref, fmax(fabs(r2.x), fabs(r2.y)));
                }
                const double bar = kJumpTol * ref;
                same = same && (dev <= bar); // a NaN anywhere never compares equal
                excess = fmax(excess, dev > bar ? dev / fmax(bar, 1e-300) : 0.0);
                const double bar_t = isM[q] ? kTailTol * ref : bar;
                near = near && (dev <= bar_t);
            }
            const unsigned long long agree = __ballot(same);
            bool converged = (agree & group_mask) == group_mask;
            double tail_term = 0.0;
            if (tails && !converged) {
                // (The comparisons stay where the table BUILDERS made theirs -- the next look is scheduled by the full criterion,
                // tails or not: a transient must never be found converged at a frame its builder did not look at, or a chain
                // that starts from the state table -- which skips the frames in front of its second switch, comparisons
                // included -- and the same chain run from its first switch would part ways by a tolerance.)
                if ((__ballot(near) & group_mask) == group_mask) {
                    // may the rest of the segment come out of the table?  Only if the means will have converged by the next
                    // switch: the table knows how long a transient from a synchronised start takes at this very place (the
                    // entry of the switch this segment began with), a chain's last transient gets kTailMargin frames more
                    const int t2 = next_start < T ? next_start : T;
                    bool far = t2 >= T;
                    wave_lds_fence();
                    const double noted = row_const[1];
                    const int t_seg = __double2loint(noted), s_from = __double2hiint(noted);
                    if (!far && p.trans != nullptr && t_seg >= 1) {
                        const int m_ref = p.trans[td->trans0 + (((int64_t)e * S + s_from) * S + s) * T + t_seg].m;
                        // ... and by what the means still have to lose: `excess` is their deviation in units of the full
                        // tolerance; a transient from a synchronised start lost ~30 bits (a deviation of 2^-13 of the scale
                        // down to 2^-43) in m_ref frames, so `bits` more take bits * m_ref / 30 frames -- a chain's last
                        // transient starts further off than a single switch's and is asked for more (soak: 3-state chains)
                        int bits = 40;
                        if (!(__ballot(excess >= 0x1p36) & group_mask)) bits = 36;
                        if (!(__ballot(excess >= 0x1p32) & group_mask)) bits = 32;
                        if (!(__ballot(excess >= 0x1p28) & group_mask)) bits = 28;
                        if (!(__ballot(excess >= 0x1p24) & group_mask)) bits = 24;
                        if (!(__ballot(excess >= 0x1p20) & group_mask)) bits = 20;
                        if (!(__ballot(excess >= 0x1p16) & group_mask)) bits = 16;
                        if (!(__ballot(excess >= 0x1p12) & group_mask)) bits = 12;
                        if (!(__ballot(excess >= 0x1p8) & group_mask)) bits = 8;
                        far = m_ref > 0 && t2 - t_seg >= m_ref + kTailMargin && t2 - t >= (bits * m_ref + 29) / 30 + 4;
                    }
                    if (far) {
                        converged = true;
                        double mine = 0.0;
#pragma unroll
                        for (int q = 0; q < CPL; ++q)
                            if (isM[q]) {
                                const double *__restrict__ gq =
                                    p.tail_g + ((td->prefix_rec0 + ((int64_t)e * S + s) * T + (t - 1)) * kDMax + (cidx[q] - NP)) * NP;
                                double acc = 0.0;
#pragma unroll
                                for (int i = 0; i < NP; ++i) acc = fma(gq[i], col.v[q][i] - rec[cidx[q] * NP + i], acc);
                                mine += acc;
                            }
                        tail_term = group_sum(mine);
                    }
                }
            }
            if (!converged) {
                // not yet: the deviation shrinks geometrically (the default Rouse model: 0.73 bits per frame), so the
                // next look comes after about the frames the worst column still needs at 1.3 frames per bit -- by the
                // lower edge of its bucket, i.e. rather too early than too late; a slower filter is simply asked again
                int wait = 4;
                if (__ballot(excess >= 0x1p4) & group_mask) wait = 8;
                if (__ballot(excess >= 0x1p8) & group_mask) wait = 12;
                if (__ballot(excess >= 0x1p16) & group_mask) wait = 24;
                if (__ballot(excess >= 0x1p24) & group_mask) wait = 32;
                if (__ballot(excess >= 0x1p32) & group_mask) wait = 44;
                if (__ballot(!(excess < 0x1p60)) & group_mask) wait = 64; // far off, or not a number
                t_check = t + wait;
            } else {
                // converged at frame t: the own piece ends here (with what the remaining deviation of the means still adds)
                extra += piece_value();
                extra += tail_term;
                open_run = 0;
                nrun += t;
                if (building_transients) return true;
                // (the lean loop has asked for frame t + 1 already, the other one has not)
                const int t_ptr = kLean ? t + 2 : t + 1;
                const int t2 = next_start < T ? next_start : T;
                extra += record(t2 - 1)[kRecL] - rec[kRecL];
                t = t2;
                land();
                if (t < T) {
                    begin_chain(t_ptr); // leaves frame t in xn
                    if constexpr (kLean) {
#pragma unroll
                        for (int q = 0; q < CPL; ++q) xc[q] = xn[q];
                        pc = pn;
                        fetch(xn, pn);
                    }
                }
            }
#if defined(BILD_TASK_CLOCK) && BILD_TASK_CLOCK == 2
            clock_events += wall_clock64() - ev0;
            ++n_events;
#endif
            return false;
        };
        if constexpr (kLean) {
            // The frame loop of the launch over the work lists.  A lone wave -- the chains that end a launch -- pays for every
            // instruction around the frame: ONE test per frame for everything that is not a frame (`t_event`: the next frame at
            // which a segment begins or a comparison is due), no tests of the table builders, no loop-carried flag in the scalar
            // masks: 53 -> 48 us for the 10k x k = 4 launch, 97 -> 85 us at k = 8.  (The same structure costs the geometries that
            // are short of registers more spills than it saves instructions -- (16, 1, 19) went from two reloads per frame to
            // seven, 3x slower --, so they keep the loop below.)
            auto next_event = [&]() {
                const int tc = (jumping && t_check > t) ? t_check : INT_MAX;
                return next_start < tc ? next_start : tc;
            };
            int t_event = next_event();
            while (t < T) {
                // invariant: xn holds frame t, the pointers stand at frame t + 1.  The next frame's data are asked for FIRST, in
                // front of the branch: in one basic block with the frame the scheduler sinks the load behind the last use of the
                // current frame's data (same register), and the delivery below then waits for L2 in every frame
#pragma unroll
                for (int q = 0; q < CPL; ++q) xc[q] = xn[q];
                pc = pn;
                fetch(xn, pn);
                if (t >= t_event) {
                    if (jumping && t == t_check) {
                        (void)compare_with_table();
                        if (t >= T) break;
                    }
                    if (t >= next_start) enter_segment(t);
                    t_event = next_event();
                }
                frame(xc, pc);
                ++t;
                // The next frame's data were asked for at the top of this one: take delivery HERE, a whole frame later, and
                // not where the register allocator happens to copy them (mid-frame: a wave with the SIMD to itself -- the
                // long chains that end a launch -- then waited for L2 in every frame: 0.45 us per frame instead of 0.24)
#pragma unroll
                for (int q = 0; q < CPL; ++q) asm volatile("" : "+v"(xn[q]));
            }
        } else {
            while (t < T) {
                // invariant: xn holds frame t, the pointers stand at frame t + 1
#pragma unroll
                for (int q = 0; q < CPL; ++q) xc[q] = xn[q];
                pc = pn;
                fetch(xn, pn);
                if (ROW) s2_now = row_const[0];
                if (t >= next_start) enter_segment(t);
                frame(xc, pc);
                if constexpr (DUMP) dump(t);
                ++t;
                if constexpr (JUMP) {
#pragma unroll
                    for (int q = 0; q < CPL; ++q) asm volatile("" : "+v"(xn[q])); // (delivery of the next frame's data: see above)
                }
                if constexpr (JUMP) {
                    if (dump_states && t < T && t - t_switch < p.sgap && (t - t_switch - 1) % p.sstride == 0)
                        dump_state((t - t_switch - 1) / p.sstride);
                }
                if (JUMP && jumping && t == t_check && t < T) {
                    if (compare_with_table()) break;
                }
            }
        }
        if (open_run != 0) {
            extra += piece_value(); // a run that reached the end of the trajectory (all of it, without tables)
            nrun += t;
        }
        if (building_transients) {
            // what this transient adds beyond the running sums of the new state's own filter over the same frames, and how
            // many frames it took (frames < t are processed; it started at the switch, frame t0)
            // (the descriptor as given: a second switch behind the trajectory's end has been cleaned away, its entry is void)
            const int t0 = sst[1], s_old = ssv[0], tb = t < T ? t : T;
            const bool all_switched = nseg == K1 && seg == nseg - 1; // converged (or ended) behind the LAST switch
            const double c = extra - (record_of(s, tb - 1)[kRecL] - record_of(s, t0 - 1)[kRecL]);
            if (gl == 0) {
                TransEntry en;
                en.c = all_switched ? c : 0.0;
                en.m = all_switched ? tb - t0 : 0;
                en.pad = 0;
                if (p.trans2_dump)
                    p.trans2_dump[(td->trans0 * S + ((((int64_t)e * S + s_old) * S + ssv[1]) * S + ssv[2]) * T + t0) * p.gap_max +
                                  (sst[2] - t0)] = en;
                else
                    p.trans_dump[td->trans0 + (((int64_t)e * S + s_old) * S + ssv[1]) * T + t0] = en;
                p.out[otask] = 0.0;
            }
        } else if (gl == 0) {
            p.out[otask] = extra;
        }
        // bench accounting: one of kFrameCounters words per workgroup slot (a single word would serialise ten thousand
        // atomics that all arrive at the end of a short launch)
        if (p.frames_run && gl == 0) atomicAdd(p.frames_run + (blockIdx.x % kFrameCounters), (unsigned long long)nrun);
#ifdef BILD_TASK_CLOCK
        if (p.frames_task && gl == 0)
#if BILD_TASK_CLOCK == 3
            p.frames_task[otask] = (int32_t)(((((clock_a - clock_begin) / 2) & 0x3ff) << 20) | ((((clock_b - clock_begin) / 2) & 0x3ff) << 10) |
                                             (((clock_c - clock_begin) / 2) & 0x3ff));
        else if (false)
#endif
            p.frames_task[otask] = BILD_TASK_CLOCK == 2 ? (int32_t)(((clock_events & 0xffffull) << 16) | (unsigned)n_events)
                                                        : (int32_t)(((clock_begin & 0xffffull) << 16) | (wall_clock64() & 0xffffull));
#else
        if (p.frames_task && gl == 0) p.frames_task[otask] = nrun;
#endif
        wave_lds_fence();
    }
}

template <int NP, int CPL, int G, int W, int OCC, int LAY, int MODE, int FLAVOR, bool DUMP = false, bool JUMP = false, bool BUILD = false>
__global__ void __launch_bounds__(64 * W, OCC) logl_kernel(const KParams p)
{
    logl_body<NP, CPL, G, W, OCC, LAY, MODE, FLAVOR, DUMP, JUMP, BUILD>(p);
}

// The running log-likelihood of every record of the prefix table (common.h), from the sums its builder left in the record: one thread
// per (task, frame); the sums of the dimensions are added in the order piece_value adds them (lane order: dimension 0, 1, 2).
__global__ void prefix_L_kernel(const TrajDesc *__restrict__ trajs, int n_traj, int S, int NP, int dstar_max, int blocks_per_task,
                                double *__restrict__ prefix, double *__restrict__ prefix_L)
{
    const int task = blockIdx.x / blocks_per_task; // (j * dstar_max + e) * S + s
    const int s = task % S, e = (task / S) % dstar_max, j = task / (S * dstar_max);
    if (j >= n_traj) return;
    const TrajDesc &td = trajs[j];
    const int t = (blockIdx.x % blocks_per_task) * blockDim.x + threadIdx.x;
    if (e >= td.dstar || t >= td.T) return;
    const int NC = NP + kDMax, REC = prefix_record_doubles(NP);
    const int kRecP = NC * NP + kDMax, kRecE = kRecP + 1, kRecL = kRecP + 2, kRecNv = kRecP + 3;
    const int64_t r = td.prefix_rec0 + ((int64_t)e * S + s) * td.T + t;
    double *rec = prefix + r * REC;
    const int nd = td.ndims[e];
    double tot = 0.0;
    for (int m = 0; m < nd; ++m) tot += rec[NC * NP + m];
    const double L = piece_from_sums(tot, rec[kRecP], (int)rec[kRecE], (int)rec[kRecNv], nd);
    rec[kRecL] = L;
    if (prefix_L) prefix_L[r] = L;
}

__global__ void reduce_partials_kernel(const double *__restrict__ partial, double *__restrict__ out, int64_t n,
                                       int dstar_max)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double tot = 0.0;
    for (int e = 0; e < dstar_max; ++e) tot += partial[r * dstar_max + e];
    out[r] = tot;
}

// Optional check of device-resident descriptors (BILD_VALIDATE_DEVICE): the indices below drive LDS and HBM
// addressing in the likelihood kernels.  err[0] = first kind of violation seen (1 traj_id, 2 first start, 3 order of
// starts, 4 state), err[1] = a sample that shows it.
__global__ void validate_kernel(const int32_t *__restrict__ seg_start, const int32_t *__restrict__ seg_state,
                                const int32_t *__restrict__ traj_id, const int32_t *__restrict__ order, int64_t n, int K1,
                                int S, int n_traj, int *err)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int bad = 0;
    // order must be a permutation: in range, and every sample hit exactly once (err[2 + sample] counts the hits)
    if (order) {
        if (order[r] < 0 || order[r] >= n) bad = 5;
        else if (atomicAdd(err + 2 + order[r], 1) != 0) bad = 5;
    }
    if (traj_id && (traj_id[r] < 0 || traj_id[r] >= n_traj)) bad = 1;
    const int32_t *a = seg_start + r * K1, *b = seg_state + r * K1;
    if (!bad && a[0] != 0) bad = 2;
    for (int i = 0; i < K1 && !bad; ++i) {
        if (i > 0 && (a[i] < a[i - 1] || a[i] < 1)) bad = 3;
        else if (b[i] < 0 || b[i] >= S) bad = 4;
    }
    if (bad && atomicCAS(err, 0, bad) == 0) err[1] = (int)(r < INT_MAX ? r : INT_MAX);
}

template <int NP, int CPL, int G, int W, int OCC, int LAY>
int launch_geom(int mode, const KParams &p, int grid, size_t lds, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1)
{
    constexpr int kThreads = 64 * W;
    hipError_t err;
    auto go = [&](auto k) -> int {
        err = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return (int)err;
        // timed launches: the events are attached to the dispatch itself (start and end of the kernel as the profiler sees
        // them), not recorded around it -- events AROUND a launch add the latency of two barrier packets, 12-14 us, to a
        // kernel of this size
        if (ev0 && ev1) hipExtLaunchKernelGGL(k, dim3(grid), dim3(kThreads), (unsigned)lds, st, ev0, ev1, 0, p);
        else hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), lds, st, p);
        return (int)hipGetLastError();
    };
    // G != 0 (an external force on the chain) never occurs through the reference's entry points
    // (rouse.Model is built with F = 0, models.py:246): keep it out of the common kernels
    // (and trajectories without a single missing frame out of the masked ones)
    const int flavor = p.has_G ? 0 : (p.all_valid ? 2 : 1);
    if (LAY >= 3 && (p.prefix_dump || p.trans_dump || p.trans2_dump)) return (int)hipErrorInvalidValue; // (see kLean)
    if (p.prefix_dump) {
        // the table is built by the packed one-column geometry with room for kDMax mean vectors
        if constexpr (LAY == 0 && CPL == 1 && G == NP + kDMax) {
            if (mode != kModal) return (int)hipErrorInvalidValue;
            if (flavor == 0) return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 0, true>);
            if (flavor == 1) return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 1, true>);
            return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 2, true>);
        } else {
            return (int)hipErrorInvalidValue;
        }
    }
    if (p.trans_dump || p.trans2_dump) {
        // the transient / pair / state tables: the BUILD instantiation of the geometry (never the lean ones, see above)
        if constexpr (LAY <= 2) {
            if (mode != kModal || !p.prefix || p.no_jump) return (int)hipErrorInvalidValue;
            if (flavor == 0) return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 0, false, true, true>);
            if (flavor == 1) return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 1, false, true, true>);
            return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 2, false, true, true>);
        } else {
            return (int)hipErrorInvalidValue;
        }
    }
    if (mode == kModal && p.prefix && !p.no_jump) {
        if (flavor == 0) return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 0, false, true>);
        if (flavor == 1) return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 1, false, true>);
        return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 2, false, true>);
    }
    if (mode == kModal) {
        if (flavor == 0) return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 0>);
        if (flavor == 1) return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 1>);
        return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kModal, 2>);
    }
    if (flavor == 0) return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kDense, 0>);
    if (flavor == 1) return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kDense, 1>);
    return go(logl_kernel<NP, CPL, G, W, OCC, LAY, kDense, 2>);
}

// (id, rows, columns per lane, lanes per group, waves per workgroup, min waves per SIMD, paths);
// CPL * G >= NP + 1: a group carries min(3, CPL * G - NP) mean vectors (G = NP + 1 / + 2 serve chains of one /
// two dimensions -- d < 3, or localization errors that differ between dimensions -- with more tasks per wave).  W is chosen so that W * (64/G) group images + the dense tables fit in
// 160 KiB of LDS; OCC bounds the register allocation (512 / OCC VGPRs per lane).  The table
// may hold several geometries per NP, listed by increasing tasks per wave: geometry_for takes
// the first one whose waves are all resident at once, else the last (densest) one.
// Measured on MI355X (profiles/r01_geometry_sweep.txt):
//  * NP = 10: (CPL, G) = (3, 5) and (5, 3) -- fewer lanes per task, fewer instructions per
//    task -- lose to 2 columns per lane at every batch size: they run at one wave per SIMD
//    with a longer dependent chain per frame.  (2, 7) beats (2, 8): 9 instead of 8 tasks per
//    wave at the same instruction count (13 of 14 column slots used).
//    One column per lane (1, 13) at three waves per SIMD has the shortest frame latency
//    (0.28 us vs 0.46 us per frame for a lone wave) and wins up to ~12k tasks; (2, 7) beyond.
//  * NP = 16 / 20: one column per lane at two waves per SIMD beats 2 and 3 columns per lane (one
//    wave per SIMD) at every batch size from 3 000 to 160 000 tasks; forcing two waves per SIMD
//    onto the multi-column geometries spills inside the frame loop (4-6x slower).
//  * NP >= 24: the multi-column geometries spill in the frame loop even at one wave per SIMD
//    (10-20x slower); one column per lane (1-2 tasks per wave) does not.
//  * NP = 10 / 12, one column per lane, G = 16 (ids 15 / 16, only through BILD_GEOM): the block layout -- a
//    task is one 16-lane block of the f64 4x4x4 matrix instruction and S comes from block_sum16.  2-5 %
//    faster than the packed (1, 13) / (1, 15) at every batch size, 7 % with missing frames: the f64 matrix
//    and vector pipes share their FMA throughput, so only the redundancy of the per-lane S chain is saved.
//    Not selected automatically: its S is summed in another order, and with it the results of one profile
//    would differ in the last bits between batch sizes (all other geometries of an NP agree bit for bit).
//  * NP = 10 / 12, one column per lane, G = 16, row layout (ids 21 / 22): a task is one 16-lane row and every use of
//    (C w)_i reads lane i of the row by DPP row broadcast inside the FMA itself (v_fmac_f64_dpp ... row_newbcast) --
//    no LDS all-gather in the frame loop, 20 registers less.  Same operations in the same order as the packed
//    layouts, hence bit-identical results (checked: max |diff| = 0 against (1, 13) on every batch size tried).
//    10 000 x T=1000: 485 -> 448 us; a lone wave (3 000 tasks) 285 -> 237 us; with missing frames 522 -> 484 us
//    Id 23 is the same layout bounded for TWO waves per SIMD: the frame loop over the work lists of a split launch
//    (walk.hip), which is latency-bound -- a few hundred tasks, one per wave, the launch as long as its longest chain --
//    and gains 6-9 % from the larger register budget (no spills around the frame loop, the measurement vector in
//    registers instead of five LDS reads per frame: 63.6 -> 59.6 us at k = 4, 116 -> 105 us at k = 8); for whole batches
//    run frame by frame three waves per SIMD stay the better geometry (r02_occ2_again.txt).
//    (profiles/r02_row_layout.txt).  Listed before the packed geometry with the same number of tasks per wave, so the
//    modal path picks it whenever three mean vectors are needed; with fewer, (1, 11) / (1, 12) carry 5 tasks per wave.
// seventh field: layout (0 packed, 1 matrix-instruction block, 2 row; 3 / 4: row / packed for the frame loop over the work lists only); last field: which paths may select the geometry
// automatically (1 = dense, 2 = modal, 3 = both).
// The dense recursion is FMA-bound with one LDS operand feeding 2*CPL FMAs, so it wants several
// columns per lane where the modal one wants a single column.
#ifndef BILD_ROW_OCC
#define BILD_ROW_OCC 3
#endif
#define BILD_GEOMETRIES(X)     \
    X(0, 4, 1, 7, 4, 3, 0, 3)     \
    X(1, 4, 2, 4, 4, 2, 0, 3)     \
    X(17, 8, 1, 9, 4, 3, 0, 3)    \
    X(18, 8, 1, 10, 4, 3, 0, 3)   \
    X(2, 8, 1, 11, 4, 3, 0, 3)    \
    X(19, 10, 1, 11, 4, 3, 0, 3)  \
    X(20, 10, 1, 12, 4, 3, 0, 3)  \
    X(21, 10, 1, 16, 4, BILD_ROW_OCC, 2, 2)  \
    X(23, 10, 1, 16, 4, 2, 3, 0)  \
    X(3, 10, 1, 13, 4, 3, 0, 3)   \
    X(4, 10, 2, 7, 4, 2, 0, 3)    \
    X(22, 12, 1, 16, 4, 2, 2, 2)  \
    X(5, 12, 1, 15, 4, 2, 0, 3)   \
    X(6, 12, 2, 8, 4, 2, 0, 3)    \
    X(7, 16, 1, 19, 4, 2, 0, 2)   \
    X(24, 16, 1, 19, 4, 1, 4, 0)  \
    X(8, 16, 3, 7, 4, 1, 0, 1)    \
    X(9, 20, 1, 23, 4, 2, 0, 2)   \
    X(25, 20, 1, 23, 4, 1, 4, 0)  \
    X(10, 20, 2, 12, 4, 1, 0, 1)  \
    X(11, 20, 3, 8, 4, 1, 0, 1)   \
    X(12, 24, 1, 27, 4, 1, 0, 3)  \
    X(13, 28, 1, 31, 4, 1, 0, 3)  \
    X(14, 32, 1, 35, 4, 1, 0, 3)  \
    X(15, 10, 1, 16, 4, 3, 1, 0)  \
    X(16, 12, 1, 16, 4, 2, 1, 0)

constexpr Geometry kGeoms[] = {
#define X(ID, NP, CPL, G, W, OCC, LAY, MODES) {NP, CPL, G, W, OCC, ID, MODES},
    BILD_GEOMETRIES(X)
#undef X
};

} // namespace

int padded_rows(int n_rows)
{
    for (const Geometry &c : kGeoms)
        if (c.NP >= n_rows) return c.NP;
    return 0;
}

bool geometry_for(int NP, int mode, int64_t ntasks, int means, Geometry *g)
{
    if (bild::config().geom >= 0) {
        const int id = bild::config().geom;
        for (const Geometry &c : kGeoms)
            if (c.id == id && c.NP == NP && c.mean_slots() >= means) {
                *g = c;
                return true;
            }
    }
    // candidates are listed by increasing lanes per task, then increasing tasks per wave.  Of those with room
    // for `means` mean vectors, one column per lane and the fewest lanes come first; take the first one whose
    // waves are all resident at once (256 CUs x 4 SIMDs x OCC waves); if none is, the one with the smallest
    // (rounds of resident waves) x (cost of a frame ~ CPL + c0), c0 = per-frame overhead in units of
    // one column's work (dense: FMA-bound, ~0; modal: ~2)
    const Geometry *best = nullptr;
    int64_t best_cost = 0;
    for (const Geometry &c : kGeoms) {
        if (c.NP != NP || c.mean_slots() < means || !(c.modes & (mode == kDense ? 1 : 2))) continue;
        // of the one-column geometries only the tightest fit is a candidate (the wider ones idle lanes)
        if (best && best->CPL == 1 && c.CPL == 1) continue;
        const int64_t waves = (ntasks + c.tasks_per_wave() - 1) / c.tasks_per_wave();
        const int64_t slots = 1024 * (int64_t)c.OCC;
        if (waves <= slots) {
            best = &c;
            break;
        }
        const int64_t cost = ((waves + slots - 1) / slots) * (c.CPL + (mode == kDense ? 0 : 2));
        if (!best || cost < best_cost) {
            best = &c;
            best_cost = cost;
        }
    }
    if (!best) return false;
    *g = *best;
    return true;
}

// The frame loop over the work lists is latency-bound (a few hundred tasks, the launch as long as its longest chain): it
// takes the geometry of the chain length with the LARGEST register budget -- for 10 modes the row layout at two waves per SIMD
// (id 23), for 16 / 20 modes the packed layout at one wave per SIMD (ids 24 / 25: what spills to scratch memory at two waves
// per SIMD -- two reloads per frame, dozens per comparison -- stays in registers there).
bool listed_geometry(const Geometry &from, Geometry *g)
{
    if (bild::config().geom >= 0 || bild::config().no_listed_geometry) return false;
    const int to = from.id == 21 ? 23 : from.id == 7 ? 24 : from.id == 9 ? 25 : -1;
    for (const Geometry &c : kGeoms)
        if (c.id == to) {
            *g = c;
            return true;
        }
    return false;
}

bool builder_geometry(int NP, Geometry *g)
{
    for (const Geometry &c : kGeoms)
        if (c.NP == NP && c.CPL == 1 && c.G == NP + kDMax && c.id != 15 && c.id != 16 && c.id != 21 && c.id != 22 && c.id != 23) {
            *g = c;
            return true;
        }
    return false;
}

const char *kernel_name(const Geometry &, int mode) { return mode == kModal ? "logl_kernel<modal>" : "logl_kernel<dense>"; }

int launch_logl(const Geometry &g, int mode, const KParams &p, int grid, size_t lds, void *stream, void *ev_start, void *ev_stop)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t ev0 = reinterpret_cast<hipEvent_t>(ev_start), ev1 = reinterpret_cast<hipEvent_t>(ev_stop);
    switch (g.id) {
#define X(ID, NP, CPL, G, W, OCC, LAY, MODES) \
    case ID: return launch_geom<NP, CPL, G, W, OCC, LAY>(mode, p, grid, lds, st, ev0, ev1);
        BILD_GEOMETRIES(X)
#undef X
    default: return -1;
    }
}

int launch_validate(const int32_t *seg_start, const int32_t *seg_state, const int32_t *traj_id, const int32_t *order, int64_t n,
                    int K1, int S, int n_traj, int *d_err, void *stream)
{
    const int bs = 256;
    hipLaunchKernelGGL(validate_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, reinterpret_cast<hipStream_t>(stream),
                       seg_start, seg_state, traj_id, order, n, K1, S, n_traj, d_err);
    return (int)hipGetLastError();
}

int launch_prefix_L(const TrajDesc *d_trajs, int n_traj, int S, int NP, int dstar_max, int Tmax, double *d_prefix, double *d_prefix_L, void *stream)
{
    const int64_t tasks = (int64_t)n_traj * dstar_max * S;
    if (tasks <= 0 || Tmax <= 0) return 0;
    const int bpt = (Tmax + 255) / 256;
    if (tasks * bpt >= ((int64_t)1 << 31)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(prefix_L_kernel, dim3((unsigned)(tasks * bpt)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_trajs, n_traj, S, NP,
                       dstar_max, bpt, d_prefix, d_prefix_L);
    return (int)hipGetLastError();
}

int launch_reduce_partials(const double *partial, double *out, int64_t n, int dstar_max, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int bs = 256;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, st, partial, out, n,
                       dstar_max);
    return (int)hipGetLastError();
}

} // namespace bild
