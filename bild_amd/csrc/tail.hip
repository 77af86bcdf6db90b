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
// lane and frame (3.1 ms per T = 1000 trajectory in one kernel that did both -> 0.8 ms in two -> see tail_scan_kernel).  The frame loop then leaves a transient as
// soon as its covariance has converged and its means are within 2^-20 of the table's (delta^2 terms ~ 1e-11 at the bound, 1e-12 in the runs), adds
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

// Phase 2, sequential over frames:  g_{t-1} = (e_t / S_t) Lambda w + Lambda (g_t - w (K_t . g_t))  backwards (Lambda g_t over a missing frame).
// The three dimensions of a task sit side by side in one wavefront -- rows of W = 16 lanes (W = 32 above 16 modes: two wavefronts
// per task), lane i of a row owns mode i of its dimension --, so that K_t . g_t is ONE butterfly over a row for all of them, in
// DPP moves (no LDS round trips: four ds_bpermute steps per dimension were 0.79 us of every frame, 0.8 ms of a T = 1000
// trajectory's first evaluation); what a frame needs from memory -- one double per lane and two scalars -- is asked for
// kTailChunk frames ahead.
constexpr int kTailChunk = 16;

template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// sum over the 16 lanes of a DPP row, the same in every lane of the row
__device__ __forceinline__ double row_sum16(double v)
{
    v += dpp_move<0xB1>(v);  // quad_perm [1, 0, 3, 2]
    v += dpp_move<0x4E>(v);  // quad_perm [2, 3, 0, 1]
    v += dpp_move<0x141>(v); // row_half_mirror
    v += dpp_move<0x140>(v); // row_mirror
    return v;
}

template <int W>
__global__ void __launch_bounds__(64) tail_scan_kernel(const TrajDesc *__restrict__ trajs, int n_traj, int S, int NP, int dstar_max,
                                                       const double *__restrict__ states, const double *__restrict__ gain,
                                                       double *__restrict__ tail_g)
{
    constexpr int kDimsPerWave = 64 / W, kParts = (kDMax + kDimsPerWave - 1) / kDimsPerWave;
    const int task = blockIdx.x / kParts, part = blockIdx.x % kParts; // task = (j * dstar_max + e) * S + s
    const int s = task % S, e = (task / S) % dstar_max, j = task / (S * dstar_max);
    if (j >= n_traj) return;
    const TrajDesc &td = trajs[j];
    if (e >= td.dstar) return;
    const int lane = threadIdx.x, T = td.T;
    const int m = part * kDimsPerWave + lane / W, il = lane % W;
    const bool row = il < NP && m < kDMax;
    const int i = il < NP ? il : 0, mm = m < kDMax ? m : 0;
    const double *sb = states + (size_t)s * StateBlock::size(NP);
    const double lam = row ? sb[StateBlock::lam(NP) + i] : 0.0, wi = row ? sb[StateBlock::wq(NP) + i] : 0.0;
    const int64_t rec0 = td.prefix_rec0 + ((int64_t)e * S + s) * T;
    const int Wd = NP + 4;
    double g = 0.0;
    if (row) tail_g[((rec0 + (T - 1)) * kDMax + m) * NP + i] = 0.0; // g_{T-1} = 0
    if (T < 2) return;
    // frames T - 1 ... 1 in chunks of kTailChunk; the chunk after this one is on its way while this one is worked on
    double Kb[kTailChunk], cb[kTailChunk], ob[kTailChunk];
    auto ask = [&](int t_hi, double *K, double *c, double *o) {
#pragma unroll
        for (int u = 0; u < kTailChunk; ++u) {
            const int t = t_hi - u > 1 ? t_hi - u : 1;
            const double *r = gain + (rec0 + t) * Wd;
            K[u] = r[i];
            c[u] = r[NP + mm];
            o[u] = r[NP + kDMax];
        }
    };
    ask(T - 1, Kb, cb, ob);
    for (int t_hi = T - 1; t_hi >= 1; t_hi -= kTailChunk) {
        double Kn[kTailChunk], cn[kTailChunk], on[kTailChunk];
        ask(t_hi - kTailChunk, Kn, cn, on);
#pragma unroll
        for (int u = 0; u < kTailChunk; ++u) {
            const int t = t_hi - u;
            if (t >= 1) {
                if (ob[u] != 0.0) {
                    double kg = row ? Kb[u] * g : 0.0;
                    kg = row_sum16(kg);
                    if constexpr (W == 32) kg += __shfl_xor(kg, 16, 64);
                    g = lam * (g - wi * kg) + cb[u] * lam * wi;
                } else {
                    g *= lam;
                }
                if (row) tail_g[((rec0 + (t - 1)) * kDMax + m) * NP + i] = g;
            }
        }
#pragma unroll
        for (int u = 0; u < kTailChunk; ++u) {
            Kb[u] = Kn[u];
            cb[u] = cn[u];
            ob[u] = on[u];
        }
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
    if (NP <= 16)
        hipLaunchKernelGGL(tail_scan_kernel<16>, dim3((unsigned)tasks), dim3(64), 0, st, d_trajs, n_traj, S, NP, dstar_max, d_states, d_gain, d_tail_g);
    else // (two dimensions per wavefront: two wavefronts per task)
        hipLaunchKernelGGL(tail_scan_kernel<32>, dim3((unsigned)(2 * tasks)), dim3(64), 0, st, d_trajs, n_traj, S, NP, dstar_max, d_states, d_gain, d_tail_g);
    return (int)hipGetLastError();
}

} // namespace bild
