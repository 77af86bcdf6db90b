// Micro-benchmark: fp64 issue rates on gfx950 -- v_fma_f64 (VALU) vs v_mfma_f64_16x16x4_f64 vs
// v_mfma_f64_4x4x4_4b_f64, one to four waves per SIMD, operands in registers.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/f64_rates.hip -o gpurun_out/f64_rates && gpurun_out/f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) probe(double *out, int iters, double seed)
{
    const int lane = threadIdx.x;
    double a = seed + lane * 1e-3, b = 1.0 + 1e-9 * lane;
    if (MODE == 0) {
        double acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fma(a, b, acc[j]);
        }
        double s = 0;
        for (int j = 0; j < 8; ++j) s += acc[j];
        out[blockIdx.x * blockDim.x + lane] = s;
    } else if (MODE == 1) {
        d4 acc[4];
        for (int j = 0; j < 4; ++j) acc[j] = (d4){0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        }
        d4 s = acc[0] + acc[1] + acc[2] + acc[3];
        out[blockIdx.x * blockDim.x + lane] = s[0] + s[1] + s[2] + s[3];
    } else {
        double acc[4];
        for (int j = 0; j < 4; ++j) acc[j] = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[j], 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + lane] = acc[0] + acc[1] + acc[2] + acc[3];
    }
}

template <int MODE>
void run(const char *name, double flop_per_inst_per_wave, int insts_per_iter)
{
    double *out;
    (void)hipMalloc(&out, sizeof(double) * 256 * 4096);
    const int iters = 20000;
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; blocks_per_cu *= 2) {
        const int grid = 256 * blocks_per_cu;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double waves = grid * 4.0;
        const double flop = waves * (double)iters * insts_per_iter * flop_per_inst_per_wave;
        // cycles per instruction per SIMD assuming 2.4 GHz is not known: report TFLOP/s and ns per inst per wave
        printf("%-28s waves/SIMD=%d  %.1f TFLOP/s   %.2f ns per instruction per wave\n", name, blocks_per_cu,
               flop / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * insts_per_iter) / blocks_per_cu);
    }
    (void)hipFree(out);
}

int main()
{
    run<0>("v_fma_f64", 64 * 2.0, 8);
    run<1>("v_mfma_f64_16x16x4_f64", 16 * 16 * 4 * 2.0, 4);
    run<2>("v_mfma_f64_4x4x4_4b_f64", 4 * 4 * 4 * 4 * 2.0, 4);
    return 0;
}
