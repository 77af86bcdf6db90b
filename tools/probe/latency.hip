// Dependent-issue latencies of the instructions a frame of the likelihood kernel is made of, for a wave that has its SIMD
// to itself, and the accuracy of v_rcp_f64 (how many Newton steps does 1/S need?).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/latency.hip -o gpurun_out/latency && gpurun_out/latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>

template <int MODE>
__global__ void __launch_bounds__(64) chain(long long *rec, int n, double seed)
{
    double x = seed + threadIdx.x * 1e-9, x2 = x + 1e-3, x3 = x + 2e-3, x4 = x + 3e-3;
    const double y = 1.0000001, c = 1e-9;
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) x = fma(x, y, c);                       // one dependent chain of v_fma_f64
            if (MODE == 1) { x = fma(x, y, c); x2 = fma(x2, y, c); x3 = fma(x3, y, c); x4 = fma(x4, y, c); } // four chains
            if (MODE == 2) x = __builtin_amdgcn_rcp(x) + 0.5;      // v_rcp_f64 + v_add_f64, dependent
            if (MODE == 3) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y)); // dependent DPP fma (with its hazard nop)
            if (MODE == 4) x = x + y;                              // v_add_f64
        }
    }
    const long long c1 = clock64();
    if (threadIdx.x == 0) {
        rec[2 * blockIdx.x] = c1 - c0;
        rec[2 * blockIdx.x + 1] = (long long)((x + x2 + x3 + x4) * 0.0);
    }
}

__global__ void rcp_error(const double *in, double *err0, double *err1, double *err2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s = in[i];
    double r = __builtin_amdgcn_rcp(s);
    const double exact = 1.0 / s; // correctly rounded division
    err0[i] = fabs(r - exact) / exact;
    r = fma(fma(-s, r, 1.0), r, r);
    err1[i] = fabs(r - exact) / exact;
    r = fma(fma(-s, r, 1.0), r, r);
    err2[i] = fabs(r - exact) / exact;
}

template <int MODE>
double run(long long *d, int n, int per_iter)
{
    std::vector<long long> h(2 * 256);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(chain<MODE>, dim3(256), dim3(64), 0, 0, d, n, 1.0);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, sizeof(long long) * 2 * 256, hipMemcpyDeviceToHost);
    std::vector<double> t;
    for (int w = 0; w < 256; ++w) t.push_back((double)h[2 * w] / ((double)n * 16 * per_iter));
    std::sort(t.begin(), t.end());
    return t[128];
}

int main()
{
    long long *d;
    (void)hipMalloc(&d, sizeof(long long) * 2 * 256);
    const int n = 2000;
    printf("one wave per SIMD (256 waves), shader-clock ticks per instruction (s_memtime), loop of 16 per branch:\n");
    printf("  dependent v_fma_f64                 %6.2f\n", run<0>(d, n, 1));
    printf("  four independent v_fma_f64 chains   %6.2f per instruction\n", run<1>(d, n, 4));
    printf("  dependent v_rcp_f64 + v_add_f64     %6.2f per pair\n", run<2>(d, n, 1));
    printf("  dependent v_fmac_f64_dpp (+s_nop 1) %6.2f\n", run<3>(d, n, 1));
    printf("  dependent v_add_f64                 %6.2f\n", run<4>(d, n, 1));
    // accuracy of v_rcp_f64 on the range of innovation variances
    const int m = 1 << 20;
    std::vector<double> in(m), e0(m), e1(m), e2(m);
    unsigned long long state = 88172645463325252ull;
    for (int i = 0; i < m; ++i) {
        state ^= state << 13; state ^= state >> 7; state ^= state << 17;
        const double u = (double)(state >> 11) / 9007199254740992.0;
        in[i] = exp(-7.0 + 21.0 * u); // 1e-3 .. 1e6
    }
    double *din, *d0, *d1, *d2;
    (void)hipMalloc(&din, m * 8); (void)hipMalloc(&d0, m * 8); (void)hipMalloc(&d1, m * 8); (void)hipMalloc(&d2, m * 8);
    (void)hipMemcpy(din, in.data(), m * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(rcp_error, dim3(m / 256), dim3(256), 0, 0, din, d0, d1, d2, m);
    (void)hipMemcpy(e0.data(), d0, m * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(e1.data(), d1, m * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(e2.data(), d2, m * 8, hipMemcpyDeviceToHost);
    printf("v_rcp_f64 against the correctly rounded quotient, %d arguments in 1e-3 .. 1e6: largest relative error %.3e; "
           "after one Newton step %.3e; after two %.3e  (2^-53 = 1.11e-16)\n",
           m, *std::max_element(e0.begin(), e0.end()), *std::max_element(e1.begin(), e1.end()), *std::max_element(e2.begin(), e2.end()));
    return 0;
}
