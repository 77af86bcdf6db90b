// The canonical (dense) recursion on the fp64 matrix pipe.
//
// What is computed: the reference's literal Kalman filter, C <- B C B + Sig and M <- B M + G every frame
// (bild/src/MSRouse_logL.pyx:206-241), masked rank-1 update (pyx:19-90) -- the same as the kDense path of
// kernels.hip, for chains whose padded length NP is a multiple of 4 up to 24.
//
// Mapping: v_mfma_f64_4x4x4_4b multiplies FOUR independent 4x4 blocks per instruction; here a block is a task:
// a wavefront carries 4 recursions, lane = x + 4*task + 16*y.  Every NP x NP matrix is cut into (NP/4)^2 tiles
// of 4x4; a tile lives in ONE register (pair) per lane, element (row y, column x) -- the layout the instruction
// writes its result in (D), which is also the layout of its B operand, while its A operand reads the same
// register as the TRANSPOSED tile (operand layouts measured with tools/probe/mfma_blocksum.hip).  With C, B and
// Sig symmetric that is all the data movement there is:
//     Y = C B        Y[i][j]  = sum_k  A<-C[k][i] (read transposed = C[i][k])   x  B<-B[k][j]
//     C' = B Y + Sig C'[i][j] = sum_k  A<-B[k][i] (= B[i][k])                   x  B<-Y[k][j]   + Sig[i][j]
//     M' = B M + G   the same with the 4-column tiles of M = [M_x M_y M_z 0]
//     u^T = w^T C    row form: A<-Wc[k] (a tile whose column 0 is w: read transposed, row 0 is w^T) x B<-C[k][j]
//     S = s2 + u.w   per-lane partial + the 16-lane block sum (two more matrix instructions)
//     C -= u u^T/S   A<-U[i] (row form read transposed: column 0 is u)  x  B<-U[j] * (-1/S)   accumulated into C
//     M += u e^T/S   A<-U[i]  x  B<-E/S   with E the row-form tile of the innovation x - w^T M
// so the state never leaves registers and no LDS is used at all.  Per frame and wavefront (4 tasks, NP = 20):
// 297 matrix instructions and ~40 vector ones, against ~2600 vector FMAs per task in the LDS-fed formulation.
// Tasks of a wave may be in different states (per-lane copies of the B / Sig tiles, reloaded at a switch), have
// different lengths and missing frames (their update is scaled by 0; finished tasks keep propagating harmlessly).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>

#include "common.h"

namespace bild {
namespace {

constexpr double kLog2Pi = 1.8378770664093453;
constexpr double kLn2 = 0.69314718055994531;

__device__ __forceinline__ double mma(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

// sum over the 16 lanes of a task's block (see kernels.hip: block_sum16), plus `add`; every lane gets it
__device__ __forceinline__ double block_sum(double v, double add) { return mma(mma(v, 1.0, 0.0), 1.0, add); }

template <int NT, bool HASG>
__global__ void __launch_bounds__(256, 1) logl_dense_mfma_kernel(const KParams p)
{
    constexpr int NP = 4 * NT;
    constexpr int MS = table_stride(NP);
    constexpr int SB = StateBlock::size(NP);
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int x = lane & 3, blk = (lane >> 2) & 3, y = lane >> 4;
    const int S = p.S, d = p.d, K1 = p.K1;
    const int64_t gstride = (int64_t)gridDim.x * 16;

    // grid-stride over groups of 4 tasks; a wave's loop runs as long as ANY of its four tasks exists
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
        // this lane's trajectory coordinate: row 0 of the innovation tile, column x = mean vector x
        const bool isE = live && y == 0 && x < nd;
        const int xdim = isE ? td->dims[ee][x] : 0;
        const int mdim = (live && x < nd) ? td->dims[ee][x] : -1; // dimension of mean column x (any row y)

        const int32_t *__restrict__ sst = p.seg_start + smp * K1;
        const int32_t *__restrict__ ssv = p.seg_state + smp * K1;
        int seg = 0;
        int s = ssv[0];
        int next_start = (K1 > 1) ? sst[1] : INT_MAX;

        double Bt[NT][NT], St[NT][NT], Gt[NT], Wc[NT], wx[NT];
        auto load_state = [&](int st) {
            const double *__restrict__ Bm = p.tab + (size_t)st * MS;
            const double *__restrict__ Sm = p.tab + (size_t)(S + st) * MS;
            const double *__restrict__ sb = p.states + (size_t)st * SB;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    Bt[i][j] = Bm[(size_t)(4 * i + y) * NP + 4 * j + x];
                    St[i][j] = Sm[(size_t)(4 * i + y) * NP + 4 * j + x];
                }
                if (HASG) Gt[i] = mdim >= 0 ? sb[StateBlock::G(NP) + (size_t)mdim * NP + 4 * i + y] : 0.0;
            }
        };
        {
            const double *__restrict__ sb = p.states + (size_t)s * SB;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                Wc[i] = x == 0 ? sb[StateBlock::wq(NP) + 4 * i + y] : 0.0; // column 0 of the tile is w (dense: same for all states)
                wx[i] = sb[StateBlock::wq(NP) + 4 * i + x];                 // w along the columns, for u.w
                if (!HASG) Gt[i] = 0.0;
            }
        }
        load_state(s);

        // steady state of state profile[0] (pyx:160-163)
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

        const double eye = x == y ? 1.0 : 0.0; // the 4x4 identity as a tile
        double acc = 0.0; // sum of e^2 / S of this lane's dimension
        double P = 1.0;   // running product of S (mantissa), exponent in E
        int E = 0;

        // frames to run: the longest of the four tasks (lanes of a block agree on T)
        int Tmax = T;
#pragma unroll
        for (int off = 4; off < 16; off <<= 1) Tmax = max(Tmax, __shfl_xor(Tmax, off, 64));

        for (int t = 0; t < Tmax; ++t) {
            const bool running = t < T;
            if (t > 0) {
                if (running && t >= next_start) {
                    do {
                        ++seg;
                        next_start = (seg + 1 < K1) ? sst[seg + 1] : INT_MAX;
                    } while (t >= next_start);
                    const int sn = ssv[seg];
                    if (sn != s) {
                        s = sn;
                        load_state(s);
                    }
                }
                // ---- predict (pyx:206-241) ------------------------------------------------------
                double Y[NT][NT];
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        double a = 0.0;
#pragma unroll
                        for (int k = 0; k < NT; ++k) a = mma(Ct[k][i], Bt[k][j], a);
                        Y[i][j] = a;
                    }
                // C' is symmetric: only the tiles on and above the diagonal are multiplied out, the others are their
                // transposes -- one instruction each (the tile read as A, i.e. transposed, times the identity) instead
                // of NT; it also keeps C exactly symmetric, which the A-operand trick above relies on
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = i; j < NT; ++j) {
                        double a = St[i][j];
#pragma unroll
                        for (int k = 0; k < NT; ++k) a = mma(Bt[k][i], Y[k][j], a);
                        Ct[i][j] = a;
                    }
#pragma unroll
                for (int i = 1; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < i; ++j) Ct[i][j] = mma(Ct[j][i], eye, 0.0);
                double Mn[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    double a = Gt[i];
#pragma unroll
                    for (int k = 0; k < NT; ++k) a = mma(Bt[k][i], Mt[k], a);
                    Mn[i] = a;
                }
#pragma unroll
                for (int i = 0; i < NT; ++i) Mt[i] = Mn[i];
            }
            // ---- masked Kalman update (pyx:19-90, 244-248) -------------------------------------------
            const double probe = running ? xt[(size_t)t * d] : 0.0;
            const bool valid = running && !isnan(probe); // agreed on by the 16 lanes of the task
            double U[NT];                                // row form: row 0 of tile j is u^T = (w^T C)[4j .. 4j+3]
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < NT; ++k) a = mma(Wc[k], Ct[k][j], a);
                U[j] = a;
            }
            double Ew = 0.0; // row 0, column x: w^T M_x
#pragma unroll
            for (int k = 0; k < NT; ++k) Ew = mma(Wc[k], Mt[k], Ew);
            double part = 0.0;
#pragma unroll
            for (int j = 0; j < NT; ++j) part = fma(U[j], wx[j], part); // rows y > 0 of U are zero
            const double Sv = block_sum(part, s2);
            double Sinv = __builtin_amdgcn_rcp(Sv);
            Sinv = fma(fma(-Sv, Sinv, 1.0), Sinv, Sinv);
            Sinv = fma(fma(-Sv, Sinv, 1.0), Sinv, Sinv);
            const double gate = valid ? Sinv : 0.0; // a missing frame (or a finished / absent task) leaves the state alone
            const double xv = (isE && valid) ? xt[(size_t)t * d + xdim] : 0.0;
            const double innov = (isE && valid) ? xv - Ew : 0.0; // x - w^T M, in row 0 of the tile
            const double Es = innov * gate;
            acc = fma(innov, Es, acc);
            double Us[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) Us[j] = -U[j] * gate;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j) Ct[i][j] = mma(U[i], Us[j], Ct[i][j]); // C -= u u^T / S
                Mt[i] = mma(U[i], Es, Mt[i]);                                        // M += u (x - w^T M)^T / S
            }
            if (valid) {
                int ex;
                P = frexp(P * Sv, &ex);
                E += ex;
            }
        }

        // ---- sum of the per-frame log-densities (pyx:88, 251-256) -------------------------------------
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
        hipLaunchKernelGGL((logl_dense_mfma_kernel<NT, true>), dim3(grid), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((logl_dense_mfma_kernel<NT, false>), dim3(grid), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}

} // namespace

bool dense_mfma_supported(int NP) { return NP % 4 == 0 && NP >= 4 && NP <= 24; }

int launch_logl_dense_mfma(int NP, const KParams &p, void *stream)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t groups = (p.ntasks + 15) / 16; // 16 tasks per workgroup of 4 waves
    const int grid = (int)(groups < 1 ? 1 : (groups > 256 * 16 ? 256 * 16 : groups));
    switch (NP / 4) {
    case 1: return launch_nt<1>(p, grid, st);
    case 2: return launch_nt<2>(p, grid, st);
    case 3: return launch_nt<3>(p, grid, st);
    case 4: return launch_nt<4>(p, grid, st);
    case 5: return launch_nt<5>(p, grid, st);
    case 6: return launch_nt<6>(p, grid, st);
    default: return -1;
    }
}

} // namespace bild
