// The modal recursion on tile registers, for chains of 33-40 effective modes.
//
// Same recursion as the kModal path of kernels.hip (elementwise predict in the eigenbasis of the current state's
// propagator, basis change R = Q_s'^T Q_s at a state switch; reference bild/src/MSRouse_logL.pyx:186-256 in that
// basis), in the data layout of dense_mfma.hip: four tasks per wavefront as the four blocks of
// v_mfma_f64_4x4x4_4b, lane = x + 4*task + 16*y, every matrix cut into 4x4 tiles with one register per tile and
// lane (element (row y, column x); the instruction's A operand reads a register as the transposed tile).
//
//   predict     C[i][j] <- lam_r lam_c C[i][j] (+ sig on the diagonal), M[i] <- lam_r M[i] (+ G): vector multiplies
//               on the tile registers with per-lane copies of lam along rows / columns
//   update      u^T = w^T C, S, C -= u u^T / S (tiles on and above the diagonal, the others transposed by one
//               instruction each, which keeps C exactly symmetric), M += u e^T / S: as in dense_mfma.hip
//   switch      C <- R C R^T, M <- R M as two tile GEMMs, Y = C R^T and X = R Y, with the tiles of R streamed from
//               L2 (both need the element R[4a + x][4b + y]).  Matrix instructions execute for all four tasks of a
//               wave, so whenever ANY of them switches the wave runs the sandwich, with the identity in place of R
//               for the tasks that do not.
//
// Why: the vector kernels of kernels.hip end at 32 modes (one task already fills a wavefront there) and beyond only
// the LDS-resident fallback of wide.hip existed; here four tasks share a wave and the O(n^2) work of the update sits
// on the matrix pipe: 1.7-2.2x faster than the fallback at 36 / 40 modes.  Registers: C and Y are NT^2 tiles each
// (NT = 10: 200 of the 256 register pairs of a lane at one wave per SIMD); no LDS.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>

#include "common.h"

namespace bild {
namespace {

constexpr double kLog2Pi = 1.8378770664093453;
constexpr double kLn2 = 0.69314718055994531;

__device__ __forceinline__ double mma(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double block_sum(double v, double add) { return mma(mma(v, 1.0, 0.0), 1.0, add); }

template <int NT, bool HASG>
__global__ void __launch_bounds__(256, 1) logl_modal_mfma_kernel(const KParams p)
{
    constexpr int NP = 4 * NT;
    constexpr int MS = table_stride(NP);
    constexpr int SB = StateBlock::size(NP);
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
    const int S = p.S, d = p.d, K1 = p.K1;
    const int64_t gstride = (int64_t)gridDim.x * 16;
    const double eye = x == y ? 1.0 : 0.0;

    for (int64_t base = ((int64_t)blockIdx.x * 4 + wv) * 4; base < p.ntasks; base += gstride) {
        const int64_t task = base + blk;
        const bool exists = task < p.ntasks;
        const int64_t tsafe = exists ? task : p.ntasks - 1;
        const int64_t smp = tsafe / p.dstar_max;
        const int e = (int)(tsafe - smp * p.dstar_max);
        const int tj = p.traj_id ? p.traj_id[smp] : 0;
        const TrajDesc *__restrict__ td = p.trajs + tj;
        const bool live = exists && e < td->dstar;
        const int ee = e < td->dstar ? e : 0;
        const int T = live ? td->T : 0;
        const double s2 = td->s2[ee];
        const int nd = live ? td->ndims[ee] : 0;
        const double *__restrict__ xt = td->x;
        const bool isE = live && y == 0 && x < nd;
        const int xdim = isE ? td->dims[ee][x] : 0;
        const int mdim = (live && x < nd) ? td->dims[ee][x] : -1;

        const int32_t *__restrict__ sst = p.seg_start + smp * K1;
        const int32_t *__restrict__ ssv = p.seg_state + smp * K1;
        int seg = 0;
        int s = ssv[0];
        int next_start = (K1 > 1) ? sst[1] : INT_MAX;

        // per-state, per-lane constants: lam along the rows / columns of a tile, sig on the diagonal of the diagonal
        // tiles, w as column 0 of a tile (read transposed: the row vector w^T) and along the columns (for u.w)
        double lrow[NT], lcol[NT], sgd[NT], Wc[NT], wx[NT], Gt[NT];
        auto load_state = [&](int st) {
            const double *__restrict__ sb = p.states + (size_t)st * SB;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                lrow[i] = sb[StateBlock::lam(NP) + 4 * i + y];
                lcol[i] = sb[StateBlock::lam(NP) + 4 * i + x];
                sgd[i] = x == y ? sb[StateBlock::sig(NP) + 4 * i + y] : 0.0;
                Wc[i] = x == 0 ? sb[StateBlock::wq(NP) + 4 * i + y] : 0.0;
                wx[i] = sb[StateBlock::wq(NP) + 4 * i + x];
                Gt[i] = (HASG && mdim >= 0) ? sb[StateBlock::G(NP) + (size_t)mdim * NP + 4 * i + y] : 0.0;
            }
        };
        load_state(s);

        double Ct[NT][NT], Mt[NT];
        {
            const double *__restrict__ sb = p.states + (size_t)s * SB;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j) Ct[i][j] = sb[StateBlock::C0(NP) + (size_t)(4 * i + y) * NP + 4 * j + x];
                Mt[i] = mdim >= 0 ? sb[StateBlock::M0(NP) + (size_t)mdim * NP + 4 * i + y] : 0.0;
            }
        }

        double acc = 0.0;
        double P = 1.0;
        int E = 0;
        int Tmax = T;
#pragma unroll
        for (int off = 4; off < 16; off <<= 1) Tmax = max(Tmax, __shfl_xor(Tmax, off, 64));

        for (int t = 0; t < Tmax; ++t) {
            const bool running = t < T;
            if (t > 0) {
                int sn = s;
                if (running && t >= next_start) {
                    do {
                        ++seg;
                        next_start = (seg + 1 < K1) ? sst[seg + 1] : INT_MAX;
                    } while (t >= next_start);
                    sn = ssv[seg];
                }
                const bool sw = sn != s;
                if (__any(sw)) {
                    // ---- basis change: C <- R C R^T, M <- R M with R = R[sn][s]; identity for the tasks that stay ----
                    const double *__restrict__ Rm = p.tab + (size_t)(sn * S + s) * MS + (size_t)x * NP + y; // &R[x][y]
                    double Y[NT][NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
#pragma unroll
                        for (int i = 0; i < NT; ++i) Y[i][j] = 0.0;
#pragma unroll
                        for (int k = 0; k < NT; ++k) {
                            // tile (k, j) of R^T: element (y, x) = R[4j + x][4k + y]
                            const double b = sw ? Rm[(size_t)(4 * j) * NP + 4 * k] : (k == j ? eye : 0.0);
#pragma unroll
                            for (int i = 0; i < NT; ++i) Y[i][j] = mma(Ct[k][i], b, Y[i][j]);
                        }
                    }
                    double Mn[NT];
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        double xrow[NT], mn = 0.0;
#pragma unroll
                        for (int j = 0; j < NT; ++j) xrow[j] = 0.0;
#pragma unroll
                        for (int k = 0; k < NT; ++k) {
                            // R(i, k) as an A operand: the register holds its transpose, element (y, x) = R[4i + x][4k + y]
                            const double a = sw ? Rm[(size_t)(4 * i) * NP + 4 * k] : (k == i ? eye : 0.0);
#pragma unroll
                            for (int j = 0; j < NT; ++j) xrow[j] = mma(a, Y[k][j], xrow[j]);
                            mn = mma(a, Mt[k], mn);
                        }
#pragma unroll
                        for (int j = 0; j < NT; ++j) Ct[i][j] = xrow[j];
                        Mn[i] = mn;
                    }
#pragma unroll
                    for (int i = 0; i < NT; ++i) Mt[i] = Mn[i];
                    s = sn;
                    load_state(s);
                }
                // ---- predict: elementwise in the eigenbasis ------------------------------------------------
#pragma unroll
                for (int i = 0; i < NT; ++i) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const double v = Ct[i][j] * lrow[i];
                        Ct[i][j] = (i == j) ? fma(v, lcol[j], sgd[i]) : v * lcol[j];
                    }
                    Mt[i] = HASG ? fma(Mt[i], lrow[i], Gt[i]) : Mt[i] * lrow[i];
                }
            }
            // ---- masked Kalman update (pyx:19-90, 244-248) -------------------------------------------
            const double probe = running ? xt[(size_t)t * d] : 0.0;
            const bool valid = running && !isnan(probe);
            double U[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < NT; ++k) a = mma(Wc[k], Ct[k][j], a);
                U[j] = a;
            }
            double Ew = 0.0;
#pragma unroll
            for (int k = 0; k < NT; ++k) Ew = mma(Wc[k], Mt[k], Ew);
            double part = 0.0;
#pragma unroll
            for (int j = 0; j < NT; ++j) part = fma(U[j], wx[j], part);
            const double Sv = block_sum(part, s2);
            double Sinv = __builtin_amdgcn_rcp(Sv);
            Sinv = fma(fma(-Sv, Sinv, 1.0), Sinv, Sinv);
            Sinv = fma(fma(-Sv, Sinv, 1.0), Sinv, Sinv);
            const double gate = valid ? Sinv : 0.0;
            const double xv = (isE && valid) ? xt[(size_t)t * d + xdim] : 0.0;
            const double innov = (isE && valid) ? xv - Ew : 0.0;
            const double Es = innov * gate;
            acc = fma(innov, Es, acc);
            double Us[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) Us[j] = -U[j] * gate;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
#pragma unroll
                for (int j = i; j < NT; ++j) Ct[i][j] = mma(U[i], Us[j], Ct[i][j]); // C -= u u^T / S, upper tiles
                Mt[i] = mma(U[i], Es, Mt[i]);
            }
#pragma unroll
            for (int i = 1; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < i; ++j) Ct[i][j] = mma(Ct[j][i], eye, 0.0);      // lower tiles: transposes
            if (valid) {
                int ex;
                P = frexp(P * Sv, &ex);
                E += ex;
            }
        }

        const double tot0 = block_sum(acc, 0.0);
        if (exists && x == 0 && y == 0) {
            double out = 0.0;
            if (live) {
                const double logS = log(P) + (double)E * kLn2;
                out = -0.5 * (tot0 + (double)nd * (logS + (double)td->nvalid * kLog2Pi));
            }
            p.out[task] = out;
        }
    }
}

template <int NT>
int launch_nt(const KParams &p, int grid, hipStream_t st)
{
    if (p.has_G)
        hipLaunchKernelGGL((logl_modal_mfma_kernel<NT, true>), dim3(grid), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((logl_modal_mfma_kernel<NT, false>), dim3(grid), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}

} // namespace

// Measured against the vector kernels (10 000 x T = 1000, profiles/r01_modal_mfma.txt): slower at 24 / 28 / 32 modes
// (4.7 / 6.2 / 7.3 ms against 2.4 / 3.0 / 6.3 ms), so only the sizes beyond their reach are compiled.
bool modal_mfma_supported(int NP) { return NP == 36 || NP == 40; }

int launch_logl_modal_mfma(int NP, const KParams &p, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t groups = (p.ntasks + 15) / 16;
    const int grid = (int)(groups < 1 ? 1 : (groups > 256 * 16 ? 256 * 16 : groups));
    switch (NP / 4) {
    case 9: return launch_nt<9>(p, grid, st);
    case 10: return launch_nt<10>(p, grid, st);
    default: return -1;
    }
}

} // namespace bild
