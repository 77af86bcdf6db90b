// Table walk: the part of an AMIS batch that needs no Kalman frame (gfx950).
//
// With the tables of a trajectory set in place (common.h: prefix / transient / pair table) most candidates of a batch
// -- 96 % of the headline batch -- are a handful of table lookups: the running log-likelihood of the switch-free
// filter up to the first switch, then one transient (or pair) entry plus a difference of running sums per switch
// (reference: the loop bild/amis.py:735-739 over MSRouse_logL, bild/src/MSRouse_logL.pyx:95-256, of which this is an
// exact-to-rounding restatement).  That arithmetic needs no filter state, so it does not belong
// on a 16-lane row of the frame-loop kernel (84 registers per lane, LDS tables, spills): here ONE LANE takes one task
//
//   * reads its segment list -- or the sampler's own (s, theta) row, which it converts to switch frames exactly as
//     FixedkSampler.st2profile does (bild/amis.py:685-688: sequential cumsum, one multiplication by T - 1, floor, + 1;
//     no contraction) --,
//   * cleans it (kernels.hip: boundaries that switch nothing, empty segments and segments beyond the trajectory go),
//   * fetches, for all switches at once, the entries the walk may need (independent loads: one round trip),
//   * walks from synchronised point to synchronised point adding the same numbers in the same order as the frame-loop
//     kernel's `land`, and
//   * either writes the result, or -- at the first chain of switches the tables do not cover -- appends the task to
//     a work list for the frame-loop kernel, which runs only those (kernels.hip, KParams::work).
//
// Everything is kept in registers: the lists are indexed with compile-time constants only (loops over KMAX slots,
// fully unrolled), so nothing goes to scratch memory.  Results are bit-identical to the single-kernel launch
// (BILD_NO_SPLIT) for every task -- those finished here by construction of the walk, the others because the frame-loop
// kernel starts them from scratch.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <limits.h>
#include <stdlib.h>

#include "config.h"
#include "common.h"

namespace bild {
namespace {

// K1 = segments per candidate, a compile-time constant: every loop over the list unrolls without a trip-count test in
// between, so that the loads of the list go out back to back (with the test, each pair of loads was waited for before
// the next was issued: K1 memory round trips instead of one).
// returns the work-list bucket of a task that goes on to the frame loop, -1 for a task that is done
template <int KMAX, bool ST>
__device__ __forceinline__ int walk_task(const WalkParams &p, const int64_t task)
{
    constexpr int K1 = KMAX;
    const int64_t r = task / p.dstar_max;
    const int e = (int)(task - r * p.dstar_max);
    const int S = p.S;
    const int tj = p.traj_id ? p.traj_id[r] : 0;
    const TrajDesc *__restrict__ td = p.trajs + tj;
    const int T = td->T;

    // ---- the segment list -----------------------------------------------------------------------------------------
    int a[KMAX], b[KMAX];
    bool ok = true;
    if constexpr (ST) {
        const double *__restrict__ s = p.ss + r * K1;
        const uint8_t *__restrict__ th = reinterpret_cast<const uint8_t *>(p.thetas) + r * K1;
        const double Tm1 = (double)(T - 1);
        double acc = 0.0;
        int prev = 0;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            a[i] = INT_MAX;
            b[i] = 0;
        }
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            if (i < K1) {
                const int st = th[i];
                ok = ok && st < S;
                b[i] = st;
                if (i == 0) a[0] = 0;
                if (i + 1 < K1) {
                    acc = __dadd_rn(acc, s[i]);            // np.cumsum: sequential
                    const double pos = __dmul_rn(acc, Tm1); // one multiplication, not fused with the sum
                    // floor for 0 <= pos < 2^31 is the truncating conversion; as on the host (api.cpp: st_row) a negative
                    // position -- where truncation and np.floor differ -- and NaN are refused
                    const bool in_range = pos >= 0.0 && pos < 2147483646.0;
                    const int idx = in_range ? (int)pos + 1 : INT_MAX;
                    ok = ok && in_range && idx >= prev;
                    prev = idx;
                    if (i + 1 < KMAX) a[i + 1] = idx;
                }
            }
        }
        if (!ok) {
            // not a point on the simplex (or a state out of range): nothing of this row drives an address
            if (atomicCAS(p.status, 0, 1) == 0) p.status[1] = (int)(r < INT_MAX ? r : INT_MAX);
            if (!p.convert_all) {
                p.out[task] = __longlong_as_double(0x7ff8000000000000ll);
            } else if (e == 0) {
                // every list goes on to the frame loop: the refused row gets one without a switch (nothing of the row drives an
                // address there either), marked by a negative first start -- a start no kernel reads, segment 0 owns frame 0 --,
                // and its result is replaced by NaN behind the frame loop (mark_refused_rows_kernel)
#pragma unroll
                for (int i = 0; i < KMAX; ++i)
                    if (i < K1) {
                        p.seg_out_start[r * K1 + i] = i == 0 ? -1 : INT_MAX;
                        p.seg_out_state[r * K1 + i] = 0;
                    }
            }
            return -1;
        }
    } else {
        const int32_t *__restrict__ sst = p.seg_start + r * K1;
        const int32_t *__restrict__ ssv = p.seg_state + r * K1;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
            a[i] = INT_MAX;
            b[i] = 0;
            if (i < K1) {
                a[i] = sst[i];
                b[i] = ssv[i];
            }
        }
    }
    // (every chain of a candidate that goes on to the frame loop writes the candidate's list: the same values)
    auto write_list = [&]() {
        if constexpr (ST) {
#pragma unroll
            for (int i = 0; i < KMAX; ++i)
                if (i < K1) {
                    p.seg_out_start[r * K1 + i] = a[i];
                    p.seg_out_state[r * K1 + i] = b[i];
                }
        }
    };
    if (p.convert_all) {
        if (e == 0) write_list();
        return -1;
    }
    if (e >= td->dstar) {
        p.out[task] = 0.0;
        return -1;
    }

    // ---- cleaned list, without moving anything: which entries are switches, and what lies behind each ---------------
    // (the cleaning rule of kernels.hip: an entry at or beyond the trajectory's end ends the list; an empty segment and
    // a segment in the state of its predecessor are no switches)
    unsigned keep = 1u;   // bit i: entry i is a real switch (bit 0: the initial segment)
    int sprev[KMAX];      // state in front of entry i
    {
        int prev = b[0];
        bool dead = false;
#pragma unroll
        for (int i = 1; i < KMAX; ++i) {
            sprev[i] = prev;
            if (i < K1) {
                const int end = (i + 1 < K1 && i + 1 < KMAX) ? a[i + 1] : INT_MAX;
                dead = dead || a[i] >= T;
                if (!dead && end > a[i] && b[i] != prev) {
                    keep |= 1u << i;
                    prev = b[i];
                }
            }
        }
    }
    int n2[KMAX], sm[KMAX], n4[KMAX]; // behind switch i: start and state of the next switch, start of the one after
    int first = T;                    // first switch, or T
    {
        int nxt_t = INT_MAX, nxt_s = 0, nxt2_t = INT_MAX;
#pragma unroll
        for (int i = KMAX - 1; i >= 1; --i) {
            n2[i] = nxt_t;
            sm[i] = nxt_s;
            n4[i] = nxt2_t;
            if (keep & (1u << i)) {
                nxt2_t = nxt_t;
                nxt_t = a[i];
                nxt_s = b[i];
            }
        }
        if (nxt_t < T) first = nxt_t;
    }

    // ---- everything the walk may need, for all switches at once ------------------------------------------------------
    // Loads of a group of switches are issued together, unconditionally and without a branch in between (a switch that
    // is none reads entry 0 of the tables): one memory round trip per group instead of one per switch.
    const int64_t rec_e = td->prefix_rec0 + (int64_t)e * S * T; // records of chain e: + state * T + frame
    const int64_t tr_e = td->trans0 + (int64_t)e * S * S * T;   // entries of chain e: + (s * S + sn) * T + frame
    const double *__restrict__ Lc = p.Lc + rec_e;
    const TransEntry *__restrict__ tr1 = p.trans + tr_e;
    const bool pairs = p.trans2 != nullptr && !(p.debug & 1);
    const TransEntry *__restrict__ tr2 = pairs ? p.trans2 : p.trans; // (no pair table: the loads still need an address)
    const int64_t tr2_e = pairs ? (td->trans0 * S + (int64_t)e * S * S * S * T) * p.gap_max : 0;
    double v1[KMAX], v2[KMAX];
    int m1[KMAX], m2[KMAX];
    double extra = Lc[(int64_t)b[0] * T + (first - 1)];
    constexpr int kGroup = 5;
    if (p.debug & 4) {
        p.out[task] = extra;
        return -1;
    }
#pragma unroll
    for (int g0 = 1; g0 < KMAX; g0 += kGroup) {
        TransEntry en[kGroup], e2[kGroup];
        double la[kGroup], lb[kGroup], l2[kGroup], l4[kGroup];
        bool pair_ok[kGroup];
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
            const int i = g0 + j;
            if (i < KMAX) {
                const bool kp = (keep >> i) & 1u;
                const int ti = kp ? a[i] : 1, s0 = kp ? sprev[i] : 0, s1 = kp ? b[i] : 0;
                const int t3 = (kp && n2[i] < T) ? n2[i] : T;
                pair_ok[j] = kp && pairs && n2[i] < T && n2[i] - ti < p.gap_max;
                const int smj = pair_ok[j] ? sm[i] : 0;
                const int t4 = (pair_ok[j] && n4[i] < T) ? n4[i] : T;
                const int64_t i2 = pair_ok[j] ? tr2_e + ((((int64_t)s0 * S + s1) * S + smj) * T + ti) * p.gap_max + (n2[i] - ti) : 0;
                en[j] = tr1[((int64_t)s0 * S + s1) * T + ti];
                la[j] = Lc[(int64_t)s1 * T + (ti - 1)];
                lb[j] = Lc[(int64_t)s1 * T + (t3 - 1)];
                e2[j] = tr2[i2];
                l2[j] = Lc[(int64_t)smj * T + (ti - 1)];
                l4[j] = Lc[(int64_t)smj * T + (t4 - 1)];
            }
        }
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
            const int i = g0 + j;
            if (i < KMAX) {
                const bool kp = (keep >> i) & 1u;
                v1[i] = en[j].c + (lb[j] - la[j]);
                m1[i] = kp ? en[j].m : 0;
                v2[i] = e2[j].c + (l4[j] - l2[j]);
                m2[i] = pair_ok[j] ? e2[j].m : 0;
            }
        }
    }

    // ---- the walk (kernels.hip: land) ---------------------------------------------------------------------------------
    bool heavy = false, skip = false;
#pragma unroll
    for (int i = 1; i < KMAX; ++i) {
        if (!(keep & (1u << i)) || heavy) continue;
        if (skip) { // second switch of a pair that came out of the pair table
            skip = false;
            continue;
        }
        const int t3 = n2[i] < T ? n2[i] : T;
        const int t4 = n4[i] < T ? n4[i] : T;
        if (m1[i] > 0 && a[i] + m1[i] <= t3) {
            extra += v1[i];
        } else if (m2[i] > 0 && a[i] + m2[i] <= t4) {
            extra += v2[i];
            skip = true;
        } else {
            heavy = true;
        }
    }
    if (!heavy) {
        p.out[task] = extra;
        if (p.frames_task) p.frames_task[task] = 0;
        if (p.tasks_done) {
            const unsigned long long done = __ballot(1);
            if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)done) - 1)
                atomicAdd(p.tasks_done, (unsigned long long)__popcll(done));
        }
        return -1;
    }

    // ---- a chain the tables do not cover: hand the task to the frame loop, in the bucket of its expected work ----------
    // (the estimate of the host scheduler, api.cpp: schedule -- chains of switches less than m_typ frames apart)
    int w = 0;
    {
        int run_from = -1, links = 0;
        const int mt = p.m_typ;
        const bool pairs = p.trans2 != nullptr;
#pragma unroll
        for (int i = 1; i < KMAX; ++i) {
            if (!(keep & (1u << i))) continue;
            const int t1 = a[i];
            const int gap = (n2[i] < T ? n2[i] : T) - t1;
            if (run_from < 0) {
                if (gap < mt) {
                    run_from = t1;
                    links = 1;
                }
            } else {
                ++links;
                if (gap >= mt) {
                    if (!(pairs && links == 2)) w += t1 + mt - run_from;
                    run_from = -1;
                }
            }
        }
        if (run_from >= 0 && !(links == 1 || (pairs && links == 2))) w += T - run_from;
    }
    if (p.no_lists) { // (cannot happen: the host checked the tables entry by entry -- a visible NaN rather than a stale result if it did)
        p.out[task] = __longlong_as_double(0x7ff8000000000000ll);
        return -1;
    }
    int bucket = w / kWorkBucketFrames;
    bucket = bucket < 0 ? 0 : (bucket >= kWorkBuckets ? kWorkBuckets - 1 : bucket);
    write_list();
    return bucket;
}

constexpr int kWalkThreads = 256;

template <int KMAX, bool ST>
__global__ void __launch_bounds__(kWalkThreads) walk_kernel(const WalkParams p)
{
    // Appending to the work lists: positions are handed out per workgroup in LDS, then ONE global atomic per bucket and
    // workgroup reserves the group's stretch of the list.  (Per-task atomics on sixteen words serialise in L2; so do
    // per-wave ones once a launch has thousands of waves: 25 us of a 200 000-candidate launch.)
    __shared__ int cnt[kWorkBuckets], base[kWorkBuckets];
    const int tid = threadIdx.x;
    const int64_t task = (int64_t)blockIdx.x * kWalkThreads + tid;
    if (tid < kWorkBuckets) cnt[tid] = 0;
    // the counters of the NEXT launch on this workspace (the other of two sets): nobody reads or writes them now
    if (p.work_counts_next != nullptr && task < kWorkBuckets) p.work_counts_next[task] = 0;
    __syncthreads();
    int bucket = -1;
    if (task < p.n * p.dstar_max) bucket = walk_task<KMAX, ST>(p, task);
    if (p.debug & 2) bucket = -1;
    int pos = 0;
    if (bucket >= 0) pos = atomicAdd(&cnt[bucket], 1);
    __syncthreads();
    if (tid < kWorkBuckets && cnt[tid] > 0) base[tid] = atomicAdd(p.work_counts + tid, cnt[tid]);
    __syncthreads();
    if (bucket >= 0) p.work[(int64_t)bucket * p.work_cap + base[bucket] + pos] = (int32_t)task;
}

template <bool ST>
int launch_st(const WalkParams &p, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1)
{
    const int64_t ntasks = p.n * p.dstar_max;
    const unsigned grid = (unsigned)((ntasks + kWalkThreads - 1) / kWalkThreads);
    // one instantiation per list length (k = K1 - 1 switches per candidate)
#define BILD_WALK_CASE(KMAX)                                                                  \
    if (p.K1 == KMAX) {                                                                       \
        if (ev0 && ev1) hipExtLaunchKernelGGL((walk_kernel<KMAX, ST>), dim3(grid), dim3(kWalkThreads), 0, st, ev0, ev1, 0, p); \
        else hipLaunchKernelGGL((walk_kernel<KMAX, ST>), dim3(grid), dim3(kWalkThreads), 0, st, p);      \
        return (int)hipGetLastError();                                                        \
    }
    BILD_WALK_CASE(1)
    BILD_WALK_CASE(2)
    BILD_WALK_CASE(3)
    BILD_WALK_CASE(4)
    BILD_WALK_CASE(5)
    BILD_WALK_CASE(6)
    BILD_WALK_CASE(7)
    BILD_WALK_CASE(8)
    BILD_WALK_CASE(9)
    BILD_WALK_CASE(10)
    BILD_WALK_CASE(11)
    BILD_WALK_CASE(12)
    BILD_WALK_CASE(13)
    BILD_WALK_CASE(14)
    BILD_WALK_CASE(15)
    BILD_WALK_CASE(16)
#undef BILD_WALK_CASE
    return (int)hipErrorInvalidValue;
}

} // namespace

namespace {
__global__ void mark_refused_rows_kernel(const int32_t *__restrict__ seg_start, int K1, int64_t n, double *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && seg_start[r * K1] < 0) out[r] = __longlong_as_double(0x7ff8000000000000ll);
}
} // namespace

// (s, theta) rows whose lists were all converted for a single launch of the frame loop: NaN for the rows the conversion refused
int launch_mark_refused_rows(const int32_t *seg_start, int K1, int64_t n, double *out, void *stream)
{
    if (n <= 0) return 0;
    const int bs = 256;
    hipLaunchKernelGGL(mark_refused_rows_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, reinterpret_cast<hipStream_t>(stream),
                       seg_start, K1, n, out);
    return (int)hipGetLastError();
}

int launch_walk(const WalkParams &p, void *stream, void *ev_start, void *ev_stop)
{
    if (p.n <= 0) return 0;
    if (bild::config().walk_debug) const_cast<WalkParams &>(p).debug = bild::config().walk_debug;
    if (p.n * p.dstar_max > (int64_t)INT_MAX) return (int)hipErrorInvalidValue; // task indices in the work lists are int32
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t ev0 = reinterpret_cast<hipEvent_t>(ev_start), ev1 = reinterpret_cast<hipEvent_t>(ev_stop);
    return p.ss ? launch_st<true>(p, st, ev0, ev1) : launch_st<false>(p, st, ev0, ev1);
}

} // namespace bild
