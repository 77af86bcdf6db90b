// The first-order tail of a transient (round 4).
//
// Behind a switch a candidate's filter converges onto the switch-free filter of its new state (common.h: prefix table).  The
// COVARIANCE forgets twice as fast as the means (its deviation is quadratic in what the means' is linear in): for the
// default Rouse model C agrees with the table's to 2^-43 after 16-20 frames, the means after 36-46 -- and until round 4 a
// candidate ran all of them.  Once C has converged, K_t and S_t of the candidate ARE the table's, and a mean deviation
// delta at frame t0 evolves linearly, delta_t = (I - K_t w^T) Lambda delta_{t-1} (Lambda delta_{t-1} over a missing frame), and
// shifts every later innovation by -w^T Lambda delta_{t-1}.  What that does to the log-likelihood of the frames behind t0 is,
// per dimension,
//     sum_t [ e_t (w^T Lambda delta_{t-1}) / S_t  -  (w^T Lambda delta_{t-1})^2 / (2 S_t) ]  =  g_{t0} . delta_{t0}  +  O(delta^2),
// with a vector g_{t0} that depends on the TABLE's filter alone (e_t: its innovations):
//     g_{t-1} = [frame t observed] (e_t / S_t) Lambda w  +  A_t^T g_t,     A_t^T g = Lambda (g - w (K_t . g))  resp.  Lambda g,     g_{T-1} = 0.
// The recursion is run once per (trajectory, chain, state), right behind the prefix table, and leaves g for every record
// (kDMax x NP doubles, beside the table: KParams::tail_g) -- in two kernels: what it needs of the table's filter at every frame
// (gain, innovations) is recomputed for all frames in parallel, the pass that is sequential in t then moves one double per
// lane and frame (3.1 -> ~0.5 ms per T = 1000 trajectory against one kernel that did both).  The frame loop then leaves a transient as
// soon as its covariance has converged and its means are within 2^-24 of the table's (delta^2 terms < 1e-13), adds
// g . delta, and takes the table's sums for the rest of the segment -- if the next switch is far enough away for the means
// to have converged by then (kernels.hip: compare_with_table).  NumPy experiment behind it (default model, N = 20 / 32): jump
// after 17-21 / 75 frames instead of 36-46 / 89-111, remaining log-likelihood reproduced to 2-4e-13.
//
// g needs no more than a few digits (it multiplies a delta of 1e-7): the filter quantities are recomputed here from the
// records in plain arithmetic, not in the bit-exact order of the frame loop.
#include <hip/hip_runtime.h>

#include "common.h"

namespace bild {
namespace {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Phase 1, parallel over frames: one wavefront per (task, frame t) recomputes what the table's filter did AT frame t from the
// record of frame t - 1 -- gain K_t (NP doubles) and, per dimension, e_t / S_t -- into a scratch array of NP + 4 doubles per
// record: [K (NP) | e_0/S, e_1/S, e_2/S | observed].  (task = (trajectory j, chain e, state s); lane i < NP owns row i)
__global__ void __launch_bounds__(64) tail_gain_kernel(const TrajDesc *__restrict__ trajs, int n_traj, int S, int NP, int d, int dstar_max,
                                                       const double *__restrict__ states, const double *__restrict__ prefix,
                                                       const int64_t *__restrict__ first, double *__restrict__ gain)
{
    // blocks are numbered trajectory by trajectory: first[j] = number of (chain, state, frame) blocks in front of trajectory j
    const int64_t b = blockIdx.x;
    int j = 0;
    for (int hi = n_traj; hi - j > 1;) { // first[j] <= b < first[j + 1]
        const int mid = (j + hi) / 2;
        if (first[mid] <= b) j = mid;
        else hi = mid;
    }
    const TrajDesc &td = trajs[j];
    const int T = td.T;
    int64_t q = b - first[j]; // (e * S + s) * T + t
    const int t = (int)(q % T);
    q /= T;
    const int s = (int)(q % S), e = (int)(q / S);
    if (e >= td.dstar || t < 1) return;
    const int lane = threadIdx.x, nd = td.ndims[e];
    const bool row = lane < NP;
    const int i = row ? lane : 0;
    const int REC = prefix_record_doubles(NP);
    const double *sb = states + (size_t)s * StateBlock::size(NP);
    const double lam = row ? sb[StateBlock::lam(NP) + i] : 0.0, wi = row ? sb[StateBlock::wq(NP) + i] : 0.0,
                 sig = row ? sb[StateBlock::sig(NP) + i] : 0.0;
    const int64_t rec0 = td.prefix_rec0 + ((int64_t)e * S + s) * T;
    const double *rec = prefix + (rec0 + (t - 1)) * REC; // state after frame t - 1
    const double *x = td.x + (size_t)t * d;
    double *out = gain + (rec0 + t) * (NP + 4);
    const bool observed = !isnan(x[0]);
    if (!observed) {
        if (lane < NP + 4) out[lane] = 0.0;
        return;
    }
    // predicted covariance times w:  (C- w)_i = lam_i sum_c lam_c w_c C_ic + sig_i w_i   (row i of C = column i, symmetric)
    double cw = 0.0;
    if (row) {
        for (int c = 0; c < NP; ++c) cw = fma(sb[StateBlock::lam(NP) + c] * sb[StateBlock::wq(NP) + c], rec[(size_t)i * NP + c], cw);
        cw = fma(lam, cw, sig * wi);
    }
    double part[1 + kDMax];
    part[0] = row ? wi * cw : 0.0;
    for (int m = 0; m < kDMax; ++m) part[1 + m] = (row && m < nd) ? wi * lam * rec[(size_t)(NP + m) * NP + i] : 0.0; // (no external force: the host checks has_G)
    for (int v = 0; v < 1 + kDMax; ++v) part[v] = wave_sum(part[v]);
    const double Sv = td.s2[e] + part[0];
    if (row) out[i] = cw / Sv;
    if (lane < kDMax) out[NP + lane] = lane < nd ? (x[td.dims[e][lane]] - part[1 + lane]) / Sv : 0.0;
    if (lane == kDMax) out[NP + kDMax] = 1.0;
}

// Phase 2, sequential over frames: one wavefront per task runs  g_{t-1} = (e_t / S_t) Lambda w + Lambda (g_t - w (K_t . g_t))  backwards
// (Lambda g_t over a missing frame); what it needs per frame is one double per lane and three scalars, asked for a frame ahead
__global__ void __launch_bounds__(64) tail_scan_kernel(const TrajDesc *__restrict__ trajs, int n_traj, int S, int NP, int dstar_max,
                                                       const double *__restrict__ states, const double *__restrict__ gain,
                                                       double *__restrict__ tail_g)
{
    const int task = blockIdx.x; // (j * dstar_max + e) * S + s
    const int s = task % S, e = (task / S) % dstar_max, j = task / (S * dstar_max);
    if (j >= n_traj) return;
    const TrajDesc &td = trajs[j];
    if (e >= td.dstar) return;
    const int lane = threadIdx.x, T = td.T;
    const bool row = lane < NP;
    const int i = row ? lane : 0;
    const double *sb = states + (size_t)s * StateBlock::size(NP);
    const double lam = row ? sb[StateBlock::lam(NP) + i] : 0.0, wi = row ? sb[StateBlock::wq(NP) + i] : 0.0;
    const int64_t rec0 = td.prefix_rec0 + ((int64_t)e * S + s) * T;
    const int W = NP + 4;
    double g[kDMax] = {0.0, 0.0, 0.0};
    if (row)
        for (int m = 0; m < kDMax; ++m) tail_g[((rec0 + (T - 1)) * kDMax + m) * NP + i] = 0.0; // g_{T-1} = 0
    if (T < 2) return;
    const double *row_t = gain + (rec0 + (T - 1)) * W;
    double K = row ? row_t[i] : 0.0, c0 = row_t[NP], c1 = row_t[NP + 1], c2 = row_t[NP + 2], obs = row_t[NP + kDMax];
    for (int t = T - 1; t >= 1; --t) {
        // (the numbers of frame t - 1 are on their way while frame t is worked on)
        const double *nx = gain + (rec0 + (t > 1 ? t - 1 : 1)) * W;
        const double Kn = row ? nx[i] : 0.0, c0n = nx[NP], c1n = nx[NP + 1], c2n = nx[NP + 2], obsn = nx[NP + kDMax];
        if (obs != 0.0) {
            double kg[kDMax];
            for (int m = 0; m < kDMax; ++m) kg[m] = row ? K * g[m] : 0.0;
            for (int m = 0; m < kDMax; ++m) kg[m] = wave_sum(kg[m]);
            const double c[kDMax] = {c0, c1, c2};
            for (int m = 0; m < kDMax; ++m) g[m] = lam * (g[m] - wi * kg[m]) + c[m] * lam * wi;
        } else {
            for (int m = 0; m < kDMax; ++m) g[m] *= lam;
        }
        if (row)
            for (int m = 0; m < kDMax; ++m) tail_g[((rec0 + (t - 1)) * kDMax + m) * NP + i] = g[m];
        K = Kn;
        c0 = c0n;
        c1 = c1n;
        c2 = c2n;
        obs = obsn;
    }
}

} // namespace

// tail_g: prefix_records x kDMax x NP doubles; gain: scratch of prefix_records x (NP + 4) doubles; first: n_traj + 1 prefix sums of
// dstar_max * S * T_j (device), total = first[n_traj]
int launch_tail(const TrajDesc *d_trajs, int n_traj, int S, int NP, int d, int dstar_max, const double *d_states, const double *d_prefix,
                const int64_t *d_first, int64_t total, double *d_gain, double *d_tail_g, void *stream)
{
    const int64_t tasks = (int64_t)n_traj * dstar_max * S;
    if (tasks <= 0 || total <= 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(tail_gain_kernel, dim3((unsigned)total), dim3(64), 0, st, d_trajs, n_traj, S, NP, d, dstar_max, d_states, d_prefix, d_first, d_gain);
    hipLaunchKernelGGL(tail_scan_kernel, dim3((unsigned)tasks), dim3(64), 0, st, d_trajs, n_traj, S, NP, dstar_max, d_states, d_gain, d_tail_g);
    return (int)hipGetLastError();
}

} // namespace bild
