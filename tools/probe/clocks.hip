// Does the shader clock depend on how much of the chip a short launch keeps busy?  Every wave runs a chain of
// dependent fp64 FMAs of fixed length; lane 0 of each wave records the shader-clock ticks (s_memtime) and the
// 100 MHz wall-clock ticks (s_memrealtime) the chain took.  Launches of 400 lone waves (what is left of the 10k
// likelihood launch after 15 us) against launches that fill every SIMD with 1 / 3 waves, back to back like bench steps.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/clocks.hip -o gpurun_out/clocks && gpurun_out/clocks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(64) chain(long long *rec, int n, double seed)
{
    double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) x = fma(x, y, 1e-9); // one dependent v_fma_f64 per iteration
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        rec[3 * blockIdx.x] = c1 - c0;
        rec[3 * blockIdx.x + 1] = w1 - w0;
        rec[3 * blockIdx.x + 2] = (long long)(x * 0.0);
    }
}

int main()
{
    const int n = 8000; // ~70 us for a lone wave at ~9 cycles per dependent FMA and 2.4 GHz
    long long *d;
    const int max_waves = 1024 * 3;
    hipMalloc(&d, sizeof(long long) * 3 * max_waves);
    std::vector<long long> h(3 * max_waves);
    for (int waves : {400, 1024, 3072, 400}) {
        for (int rep = 0; rep < 30; ++rep) hipLaunchKernelGGL(chain, dim3(waves), dim3(64), 0, 0, d, n, 1.0);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, sizeof(long long) * 3 * waves, hipMemcpyDeviceToHost);
        std::vector<double> mhz, us, cyc;
        for (int w = 0; w < waves; ++w) {
            mhz.push_back((double)h[3 * w] / ((double)h[3 * w + 1] / 100.0));
            us.push_back((double)h[3 * w + 1] / 100.0);
            cyc.push_back((double)h[3 * w] / n);
        }
        std::sort(mhz.begin(), mhz.end());
        std::sort(us.begin(), us.end());
        std::sort(cyc.begin(), cyc.end());
        printf("%4d waves of 64 lanes, %d dependent v_fma_f64 each (30th launch of a series): median %.1f us, shader clock %.0f MHz "
               "(s_memtime ticks per us; min %.0f max %.0f), %.2f clock ticks per FMA\n",
               waves, n, us[waves / 2], mhz[waves / 2], mhz.front(), mhz.back(), cyc[waves / 2]);
    }
    return 0;
}
