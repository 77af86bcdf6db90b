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
// This kernel runs that backward recursion once per (trajectory, chain, state), right behind the prefix table, and leaves
// g for every record (kDMax x NP doubles, beside the table: KParams::tail_g).  The frame loop then leaves a transient as
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

// one wavefront per (trajectory, chain e, state s); lane i < NP owns row i of everything
__global__ void __launch_bounds__(64) tail_kernel(const TrajDesc *__restrict__ trajs, int n_traj, int S, int NP, int d, int dstar_max,
                                                  const double *__restrict__ states, const double *__restrict__ prefix,
                                                  double *__restrict__ tail_g)
{
    const int task = blockIdx.x; // (j * dstar_max + e) * S + s
    const int s = task % S, e = (task / S) % dstar_max, j = task / (S * dstar_max);
    if (j >= n_traj) return;
    const TrajDesc &td = trajs[j];
    if (e >= td.dstar) return;
    const int lane = threadIdx.x, T = td.T, nd = td.ndims[e];
    const bool row = lane < NP;
    const int i = row ? lane : 0;
    const int REC = prefix_record_doubles(NP);
    const double *sb = states + (size_t)s * StateBlock::size(NP);
    const double lam = row ? sb[StateBlock::lam(NP) + i] : 0.0, wi = row ? sb[StateBlock::wq(NP) + i] : 0.0,
                 sig = row ? sb[StateBlock::sig(NP) + i] : 0.0;
    const double s2 = td.s2[e];
    const int64_t rec0 = td.prefix_rec0 + ((int64_t)e * S + s) * T;
    auto wave_sum = [](double v) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    };
    double g[kDMax] = {0.0, 0.0, 0.0};
    // g_{T-1} = 0
    if (row)
        for (int m = 0; m < kDMax; ++m) tail_g[((rec0 + (T - 1)) * kDMax + m) * NP + i] = 0.0;
    for (int t = T - 1; t >= 1; --t) {
        const double *rec = prefix + (rec0 + (t - 1)) * REC; // state after frame t - 1
        const double *x = td.x + (size_t)t * d;
        const bool observed = !isnan(x[0]);
        if (observed) {
            // predicted covariance times w:  (C- w)_i = lam_i sum_j lam_j w_j C_ij + sig_i w_i   (row i of C = column i, symmetric)
            double cw = 0.0;
            if (row) {
                for (int c = 0; c < NP; ++c) {
                    const double lc = sb[StateBlock::lam(NP) + c], wc = sb[StateBlock::wq(NP) + c];
                    cw = fma(lc * wc, rec[(size_t)i * NP + c], cw);
                }
                cw = fma(lam, cw, sig * wi);
            }
            const double Sv = s2 + wave_sum(row ? wi * cw : 0.0);
            const double K = cw / Sv;
            for (int m = 0; m < kDMax; ++m) {
                if (m >= nd) continue;
                const int dim = td.dims[e][m];
                const double mi = row ? rec[(size_t)(NP + m) * NP + i] : 0.0;
                const double innov = x[dim] - wave_sum(row ? wi * lam * mi : 0.0); // (no external force: the host checks has_G)
                const double kg = wave_sum(row ? K * g[m] : 0.0);
                g[m] = lam * (g[m] - wi * kg) + (innov / Sv) * lam * wi;
            }
        } else {
            for (int m = 0; m < kDMax; ++m) g[m] *= lam;
        }
        if (row)
            for (int m = 0; m < kDMax; ++m) tail_g[((rec0 + (t - 1)) * kDMax + m) * NP + i] = m < nd ? g[m] : 0.0;
    }
}

} // namespace

// tail_g: prefix_records x kDMax x NP doubles; one wavefront per (trajectory, chain, state)
int launch_tail(const TrajDesc *d_trajs, int n_traj, int S, int NP, int d, int dstar_max, const double *d_states, const double *d_prefix,
                double *d_tail_g, void *stream)
{
    const int64_t tasks = (int64_t)n_traj * dstar_max * S;
    if (tasks <= 0) return 0;
    hipLaunchKernelGGL(tail_kernel, dim3((unsigned)tasks), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), d_trajs, n_traj, S, NP, d,
                       dstar_max, d_states, d_prefix, d_tail_g);
    return (int)hipGetLastError();
}

} // namespace bild
