// Launch order of a batch whose candidates are not yet known to the host scheduler: the host-buffer entry points
// (bild_logl_st, bild_logl_segments, ...) hand their descriptors to the device and launch at once.  For batches of
// several rounds with many busy candidates the order is worth a factor of two in kernel time (api.cpp: schedule;
// profiles/r02_launch_order.txt), and walking the lists on the host would cost more than that -- so the same estimate
// (frames a candidate will run itself, from its switch frames and the tables' typical transient length) and the same
// rule (sorted when throughput-bound, spread otherwise) run here, on the stream of the launch: one pass over the
// segment lists, a radix sort of (work, index) pairs (hipCUB; stable, so the order is reproducible), one pass that
// writes the order.  A matter of speed only: results never depend on the launch order (tests assert it bit for bit).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace bild {
namespace {

__global__ void __launch_bounds__(256) work_kernel(const int32_t *__restrict__ seg_start, const int32_t *__restrict__ traj_id,
                                                   const TrajDesc *__restrict__ trajs, int K1, int64_t n, int m_typ, int pairs, int Tmax,
                                                   unsigned *__restrict__ keys, int32_t *__restrict__ idx,
                                                   unsigned long long *__restrict__ busy)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int w = 0;
    if (r < n) {
        const int T = trajs[traj_id ? traj_id[r] : 0].T;
        const int32_t *a = seg_start + r * K1;
        // (as api.cpp: schedule) a chain = switches less than m_typ frames apart; one switch, or two with the pair table,
        // come out of the tables
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
                    if (!(pairs && links == 2)) w += t + m_typ - run_from;
                    run_from = -1;
                }
            }
        }
        if (run_from >= 0 && !(links == 1 || (pairs && links == 2))) w += T - run_from;
        w = w > Tmax ? Tmax : (w < 0 ? 0 : w);
        keys[r] = (unsigned)(Tmax - w); // ascending keys = descending work
        idx[r] = (int32_t)r;
    }
    const unsigned long long any = __ballot(w > 0);
    if ((threadIdx.x & 63) == 0 && any) atomicAdd(busy, (unsigned long long)__popcll(any));
}

__global__ void __launch_bounds__(256) order_kernel(const int32_t *__restrict__ sorted, int64_t n, int rpw, int64_t slots,
                                                    const unsigned long long *__restrict__ busy, int32_t *__restrict__ order)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool packed = n > slots && (int64_t)*busy > slots / rpw;
    if (packed) {
        order[i] = sorted[i];
        return;
    }
    const int64_t round = slots > rpw ? slots : rpw;
    const int64_t base = (i / round) * round;
    const int64_t cnt = (n - base < round) ? n - base : round, nw = cnt / rpw, li = i - base;
    order[i] = li < nw * rpw ? sorted[base + (li % rpw) * nw + li / rpw] : sorted[i];
}

// Do the tables cover EVERY candidate of at most two switches?  (api.cpp: a split launch of lists of <= 3 segments then hands
// nothing to the frame loop, and its launch -- empty, but five microseconds of a twelve-microsecond step -- is not made.)
// One thread per entry of the transient table (trajectory j, chain e, s -> sn, frame t): a single switch there must have an
// entry (m > 0), and for every gap g in front of that transient's convergence a second switch sn -> sm at t + g must have its
// pair entry.  Mirrors walk.hip's walk for K1 <= 3 term by term; any violation clears *covered.
__global__ void __launch_bounds__(256) two_switch_cover_kernel(const TrajDesc *__restrict__ trajs, const int64_t *__restrict__ first,
                                                               int n_traj, int S, const TransEntry *__restrict__ trans,
                                                               const TransEntry *__restrict__ trans2, int gap_max, int *covered)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= first[n_traj]) return;
    int j = 0;
    for (int hi = n_traj; hi - j > 1;) { // first[j] <= r < first[j + 1]
        const int mid = (j + hi) / 2;
        if (first[mid] <= r) j = mid;
        else hi = mid;
    }
    const TrajDesc &td = trajs[j];
    const int T = td.T;
    int64_t q = r - first[j]; // ((e * S + s) * (S - 1) + sn') * (T - 1) + (t - 1)
    const int t = 1 + (int)(q % (T - 1));
    q /= (T - 1);
    const int snp = (int)(q % (S - 1));
    q /= (S - 1);
    const int s = (int)(q % S), e = (int)(q / S);
    const int sn = snp + (snp >= s ? 1 : 0);
    bool ok = true;
    const int m1 = trans[td.trans0 + (((int64_t)e * S + s) * S + sn) * T + t].m;
    if (m1 <= 0 || t + m1 > T) ok = false;
    const int g_end = ok ? ((m1 < T - t ? m1 : T - t)) : 0; // gaps 1 .. g_end - 1: the second switch comes before the first has converged
    for (int g = 1; g < g_end && ok; ++g) {
        if (g >= gap_max) {
            ok = false;
            break;
        }
        for (int sm = 0; sm < S; ++sm) {
            if (sm == sn) continue;
            const int m2 = trans2[(td.trans0 * S + ((((int64_t)e * S + s) * S + sn) * S + sm) * T + t) * gap_max + g].m;
            if (m2 <= 0 || t + m2 > T) ok = false;
        }
    }
    if (!ok) atomicAnd(covered, 0);
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// The candidates that build the pair table (api.cpp: ensure_pairs), written where they are needed: task r is
// (trajectory j, s -> sn -> sm, first switch at t, second at t + g) in the order j, s, sn, sm, t, g.
__global__ void __launch_bounds__(256) pair_tasks_kernel(const int64_t *__restrict__ first_task, int n_traj, const TrajDesc *__restrict__ trajs,
                                                         int S, int G, int64_t nb, int32_t *__restrict__ seg_start,
                                                         int32_t *__restrict__ seg_state, int32_t *__restrict__ traj_id)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nb) return;
    int lo = 0, hi = n_traj; // first_task[lo] <= r < first_task[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (first_task[mid] <= r) lo = mid;
        else hi = mid;
    }
    const int64_t local = r - first_task[lo];
    const int T = trajs[lo].T;
    const int64_t per_combo = (int64_t)(T - 1) * (G - 1);
    const int combo = (int)(local / per_combo);
    const int64_t rem = local - (int64_t)combo * per_combo;
    const int t = 1 + (int)(rem / (G - 1)), g = 1 + (int)(rem % (G - 1));
    const int s = combo / ((S - 1) * (S - 1)), rest = combo % ((S - 1) * (S - 1));
    const int a = rest / (S - 1), b = rest % (S - 1);
    const int sn = a < s ? a : a + 1, sm = b < sn ? b : b + 1;
    seg_start[3 * r] = 0;
    seg_start[3 * r + 1] = t;
    seg_start[3 * r + 2] = t + g; // beyond the end: the kernel voids the entry
    seg_state[3 * r] = s;
    seg_state[3 * r + 1] = sn;
    seg_state[3 * r + 2] = sm;
    traj_id[r] = lo;
}

} // namespace

int launch_pair_tasks(const int64_t *d_first_task, int n_traj, const TrajDesc *d_trajs, int S, int G, int64_t nb, int32_t *d_seg_start,
                      int32_t *d_seg_state, int32_t *d_traj_id, void *stream)
{
    hipLaunchKernelGGL(pair_tasks_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_first_task, n_traj, d_trajs, S, G,
                       nb, d_seg_start, d_seg_state, d_traj_id);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// d_first: n_traj + 1 prefix sums of dstar_j * S * (S - 1) * (T_j - 1) (device); d_covered: one int, set to 1 by the caller
int launch_two_switch_cover(const TrajDesc *d_trajs, const int64_t *d_first, int64_t total, int n_traj, int S, const TransEntry *trans,
                            const TransEntry *trans2, int gap_max, int *d_covered, void *stream)
{
    if (total <= 0) return 0;
    hipLaunchKernelGGL(two_switch_cover_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       d_trajs, d_first, n_traj, S, trans, trans2, gap_max, d_covered);
    return (int)hipGetLastError();
}

size_t device_schedule_bytes(int64_t n)
{
    size_t temp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const unsigned *)nullptr, (unsigned *)nullptr, (const int32_t *)nullptr,
                                             (int32_t *)nullptr, (int)n, 0, 32, nullptr);
    return 256 + 5 * align256((size_t)n * 4) + align256(temp) + 256;
}

int device_schedule(const int32_t *d_seg_start, const int32_t *d_traj_id, const TrajDesc *d_trajs, int K1, int64_t n, int m_typ, int pairs,
                    int Tmax, int rpw, int64_t slots, void *ws, size_t ws_bytes, const int32_t **d_order, void *stream)
{
    if (n < 1 || n > 0x7fffffff || ws_bytes < device_schedule_bytes(n)) return 1;
    hipStream_t st = (hipStream_t)stream;
    char *base = (char *)ws;
    unsigned long long *busy = (unsigned long long *)base;
    const size_t stride = align256((size_t)n * 4);
    unsigned *keys_in = (unsigned *)(base + 256), *keys_out = (unsigned *)(base + 256 + stride);
    int32_t *idx_in = (int32_t *)(base + 256 + 2 * stride), *idx_out = (int32_t *)(base + 256 + 3 * stride);
    int32_t *order = (int32_t *)(base + 256 + 4 * stride);
    void *temp = base + 256 + 5 * stride;
    size_t temp_bytes = ws_bytes - (256 + 5 * stride);
    if (hipMemsetAsync(busy, 0, sizeof(unsigned long long), st) != hipSuccess) return 1;
    const int blocks = (int)((n + 255) / 256);
    hipLaunchKernelGGL(work_kernel, dim3(blocks), dim3(256), 0, st, d_seg_start, d_traj_id, d_trajs, K1, n, m_typ, pairs, Tmax, keys_in, idx_in,
                       busy);
    int bits = 1;
    while ((1u << bits) <= (unsigned)Tmax && bits < 32) ++bits;
    if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, (int)n, 0, bits, st) != hipSuccess) return 1;
    hipLaunchKernelGGL(order_kernel, dim3(blocks), dim3(256), 0, st, idx_out, n, rpw, slots, busy, order);
    if (hipGetLastError() != hipSuccess) return 1;
    *d_order = order;
    return 0;
}

} // namespace bild
