// Long chains: the Rouse Kalman-filter log-likelihood for models whose reduced chain has more
// modes than fit the register-resident kernels of kernels.hip (32 < n <= kWideMaxNP).
//
// Same recursion, same packed model (common.h), modal path only -- the predict is elementwise in the
// eigenbasis of the current state's propagator, a state switch is the basis change  A <- R A,
// C <- C R^T  (reference bild/src/MSRouse_logL.pyx:186-256 in that basis; see kernels.hip).
//
// Mapping: ONE task per workgroup of 256-1024 lanes (by chain length); the filter state A = [C | M] (NP x (NP + 3)) lives in
// LDS, column-major with an odd leading dimension (conflict-free both along a column and across
// columns).  A lane owns a fixed (column, row slice) of A for the whole recursion, so the per-frame
// passes over A need no synchronisation among themselves; the cross-lane steps of a frame are the
// reduction of the partial dot products, the sum S = s2 + w.(Cw) (one wavefront, shuffles) and the
// broadcast of 1/S -- three barriers per observed frame.  Basis changes stream the matrix from L2
// (its transpose is the table entry of the opposite switch, so the loads coalesce) and work in place:
// left-multiply all columns, transpose C, left-multiply the covariance columns again
// (R (R C)^T = R C R^T for symmetric C).
//
// This is the fallback that keeps every chain length of the reference usable; it is HBM/L2- and
// LDS-latency bound, not tuned like the n <= 32 kernels.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>

#include "config.h"
#include "common.h"

namespace bild {
namespace {

constexpr double kLog2Pi = 1.8378770664093453;
constexpr double kLn2 = 0.69314718055994531;

template <int kThreads>
__global__ void __launch_bounds__(kThreads) logl_wide_kernel(const KParams p, const int NP)
{
    extern __shared__ __align__(16) double sm[];
    const int NC = NP + kDMax;
    const int LD = NP + 1; // NP is even: odd leading dimension
    const int R = kThreads / NC;             // row slices per column (>= 1: NC <= 131)
    const int RS = (NP + R - 1) / R;         // rows per slice
    double *const A = sm;                    // NC columns of LD
    double *const lam = A + (size_t)NC * LD; // NP
    double *const sig = lam + NP;            // NP
    double *const wq = sig + NP;             // NP
    double *const ev = wq + NP;              // NC: e_j; its first NP entries are u = C w
    double *const part = ev + NC;            // R * NC partial dot products
    double *const misc = part + (size_t)R * NC; // [0] S, [1..3] per-dimension sums at the end

    const int tid = threadIdx.x;
    const int r = tid / NC;
    const int j = tid - r * NC;
    const bool active = r < R;
    const int i0 = active ? r * RS : 0;
    const int i1 = active ? min(NP, i0 + RS) : 0;
    double *const col = A + (size_t)j * LD;

    const int S = p.S, d = p.d, K1 = p.K1;
    const int SB = StateBlock::size(NP);
    const int MS = table_stride(NP);

    for (int64_t task = blockIdx.x; task < p.ntasks; task += gridDim.x) {
        const int64_t smp = task / p.dstar_max;
        const int e = (int)(task - smp * p.dstar_max);
        const int tj = p.traj_id ? p.traj_id[smp] : 0;
        const TrajDesc *__restrict__ td = p.trajs + tj;
        if (e >= td->dstar) {
            if (tid == 0) p.out[task] = 0.0;
            continue;
        }
        const int T = td->T;
        const double s2 = td->s2[e];
        const int nd = td->ndims[e];
        const double *__restrict__ x = td->x;
        // own column: covariance column j < NP, or mean vector of dimension xdim
        const int mi = j - NP;
        const bool isM = active && mi >= 0 && mi < nd;
        const int xdim = isM ? td->dims[e][mi] : 0;

        const int32_t *__restrict__ sst = p.seg_start + smp * K1;
        const int32_t *__restrict__ ssv = p.seg_state + smp * K1;
        int seg = 0;
        int s = ssv[0];
        int next_start = (K1 > 1) ? sst[1] : INT_MAX;

        double muj = 1.0; // lam_j for a covariance column, 1 for a mean column
        auto load_state = [&](int st) { // callers synchronise afterwards
            const double *__restrict__ sb = p.states + (size_t)st * SB;
            for (int i = tid; i < NP; i += kThreads) {
                lam[i] = sb[StateBlock::lam(NP) + i];
                sig[i] = sb[StateBlock::sig(NP) + i];
                wq[i] = sb[StateBlock::wq(NP) + i];
            }
            muj = (active && j < NP) ? sb[StateBlock::lam(NP) + j] : 1.0;
        };

        __syncthreads(); // previous task is done with LDS
        load_state(s);
        {   // steady state of state profile[0] (pyx:160-163)
            const double *__restrict__ sb = p.states + (size_t)s * SB;
            if (active) {
                const double *src = (j < NP) ? sb + StateBlock::C0(NP) + (size_t)j * NP
                                             : sb + StateBlock::M0(NP) + (size_t)xdim * NP;
                const double live = (j < NP || isM) ? 1.0 : 0.0;
                for (int i = i0; i < i1; ++i) col[i] = live * src[i];
            }
        }
        __syncthreads();

        // X = R[sn][s] applied from the left to the first `ncols` columns, in place
        auto left_multiply = [&](const double *__restrict__ XT, int ncols) {
            const int cpp = kThreads / NP; // columns per pass
            const int cj = tid / NP, i = tid - cj * NP;
            for (int c0 = 0; c0 < ncols; c0 += cpp) {
                const int c = c0 + cj;
                double acc = 0.0;
                const bool mine = cj < cpp && c < ncols;
                if (mine) {
                    const double *a = A + (size_t)c * LD;
                    double acc2 = 0.0;
                    for (int k = 0; k < NP; k += 2) { // XT[k][i] = X[i][k]; NP is even
                        acc = fma(XT[(size_t)k * NP + i], a[k], acc);
                        acc2 = fma(XT[(size_t)(k + 1) * NP + i], a[k + 1], acc2);
                    }
                    acc += acc2;
                }
                __syncthreads();
                if (mine) A[(size_t)c * LD + i] = acc;
                __syncthreads();
            }
        };
        auto basis_change = [&](int sn, int so) {
            __syncthreads(); // frames without an observation have no barrier: lanes may still be predicting
            const double *XT = p.tab + (size_t)(so * S + sn) * MS; // R[so][sn] = R[sn][so]^T
            left_multiply(XT, NC);
            for (int idx = tid; idx < NP * NP; idx += kThreads) { // transpose C in place
                const int a = idx / NP, b = idx - a * NP;
                if (a < b) {
                    const double t = A[(size_t)a * LD + b];
                    A[(size_t)a * LD + b] = A[(size_t)b * LD + a];
                    A[(size_t)b * LD + a] = t;
                }
            }
            __syncthreads();
            left_multiply(XT, NP);
        };

        double accm = 0.0; // own mean column: sum of e^2 / S
        double P = 1.0;    // thread 0: running product of S (mantissa), exponent in E
        int E = 0;

        for (int t = 0; t < T; ++t) {
            if (t > 0 && t >= next_start) {
                do {
                    ++seg;
                    next_start = (seg + 1 < K1) ? sst[seg + 1] : INT_MAX;
                } while (t >= next_start);
                const int sn = ssv[seg];
                if (sn != s) {
                    basis_change(sn, s);
                    s = sn;
                    load_state(s);
                    __syncthreads();
                }
            }
            const bool valid = p.all_valid || !isnan(x[(size_t)t * d]);
            // predict (pyx:206-241, elementwise in the eigenbasis) fused with the dot product w.col
            double acc = 0.0;
            if (active) {
                if (t > 0) {
                    const double *__restrict__ gb =
                        (p.has_G && isM) ? p.states + (size_t)s * SB + StateBlock::G(NP) + (size_t)xdim * NP : nullptr;
                    for (int i = i0; i < i1; ++i) {
                        double a = col[i] * (lam[i] * muj);
                        if (i == j) a += sig[i];
                        if (gb) a += gb[i];
                        col[i] = a;
                        acc = fma(wq[i], a, acc);
                    }
                } else {
                    for (int i = i0; i < i1; ++i) acc = fma(wq[i], col[i], acc);
                }
            }
            if (!valid) continue; // block-uniform
            // ---- Kalman update (pyx:19-90) ------------------------------------------------------
            if (active) part[r * NC + j] = acc;
            __syncthreads();
            if (active && r == 0) {
                double tot = 0.0;
                for (int q = 0; q < R; ++q) tot += part[q * NC + j];
                ev[j] = tot - (isM ? x[(size_t)t * d + xdim] : 0.0); // (C w)_j, or -(x - w.M) for a mean column
            }
            __syncthreads();
            if (tid < 64) {
                double sp = 0.0;
                for (int i = tid; i < NP; i += 64) sp = fma(wq[i], ev[i], sp);
                for (int off = 32; off > 0; off >>= 1) sp += __shfl_xor(sp, off, 64);
                if (tid == 0) misc[0] = s2 + sp;
            }
            __syncthreads();
            const double Sv = misc[0];
            const double Sinv = 1.0 / Sv;
            if (active) {
                const double ej = ev[j];
                const double coef = ej * Sinv;
                if (isM && r == 0) accm = fma(ej, coef, accm);
                for (int i = i0; i < i1; ++i) col[i] = fma(-coef, ev[i], col[i]);
            }
            if (tid == 0) {
                int ex;
                P = frexp(P * Sv, &ex);
                E += ex;
            }
            // the next writes to part / ev / misc come after the next frame's barriers
        }

        // ---- sum of the per-frame log-densities (pyx:88, 251-256) ---------------------------------
        __syncthreads();
        if (active && r == 0 && j >= NP) misc[1 + mi] = isM ? accm : 0.0;
        __syncthreads();
        if (tid == 0) {
            double tot = misc[1] + misc[2] + misc[3];
            const double logS = log(P) + (double)E * kLn2;
            tot += (double)nd * (logS + (double)td->nvalid * kLog2Pi);
            p.out[task] = -0.5 * tot;
        }
    }
}

} // namespace

// lanes per workgroup, by chain length: more lanes shorten a lane's row slice but make the three barriers of a
// frame dearer.  Measured on 5 000-10 000 x T = 1000 batches (profiles/r01_wide.txt): 256 lanes win up to n = 64,
// 512 for n = 72 ... 88, 1024 from n = 100 (2.4x / 3.5x faster than 256 at n = 100 / 128).
// env BILD_WIDE_THREADS overrides (256 / 512 / 1024).
int wide_threads(int NP)
{
    if (bild::config().wide_threads) {
        const int t = bild::config().wide_threads;
        if (t == 256 || t == 512 || t == 1024) return t;
    }
    return NP <= 68 ? 256 : (NP <= 92 ? 512 : 1024);
}

size_t wide_lds_bytes(int NP)
{
    const int NC = NP + kDMax, LD = NP + 1, R = wide_threads(NP) / NC;
    return ((size_t)NC * LD + 3 * (size_t)NP + NC + (size_t)R * NC + 4) * sizeof(double);
}

int launch_logl_wide(int NP, const KParams &p, int grid, void *stream)
{
    const size_t lds = wide_lds_bytes(NP);
    const int threads = wide_threads(NP);
    auto go = [&](auto kernel) -> int {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return (int)err;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, reinterpret_cast<hipStream_t>(stream), p, NP);
        return (int)hipGetLastError();
    };
    if (threads == 1024) return go(logl_wide_kernel<1024>);
    if (threads == 512) return go(logl_wide_kernel<512>);
    return go(logl_wide_kernel<256>);
}

} // namespace bild
