// Probe: which operand arrangement of two v_mfma_f64_4x4x4_4b gives every lane of a 16-lane block the sum
// of one value per lane of that block?   hipcc --offload-arch=gfx950 -O2 mfma_blocksum.hip -o mfma_blocksum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void probe(const double *p, double *out)
{
    const int l = threadIdx.x;
    const double v = p[l], one = 1.0;
    double x1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v, one, 0.0, 0, 0, 0);   // A = p, B = 1
    double x2 = __builtin_amdgcn_mfma_f64_4x4x4f64(one, v, 0.0, 0, 0, 0);   // A = 1, B = p
    out[0 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(one, x1, 0.0, 0, 0, 0);
    out[1 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(x1, one, 0.0, 0, 0, 0);
    out[2 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(one, x2, 0.0, 0, 0, 0);
    out[3 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(x2, one, 0.0, 0, 0, 0);
    out[4 * 64 + l] = x1;
    out[5 * 64 + l] = x2;
}

int main()
{
    double hp[64], ho[6 * 64], *dp, *dout;
    for (int l = 0; l < 64; ++l) hp[l] = 1.0 + 0.37 * l + 0.011 * l * l;
    hipMalloc(&dp, sizeof hp); hipMalloc(&dout, sizeof ho);
    hipMemcpy(dp, hp, sizeof hp, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dp, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    const char *names[4] = {"mfma(1, mfma(p,1))", "mfma(mfma(p,1), 1)", "mfma(1, mfma(1,p))", "mfma(mfma(1,p), 1)"};
    for (int c = 0; c < 4; ++c) {
        double worst = 0;
        for (int l = 0; l < 64; ++l) {
            double s = 0;
            for (int j = 0; j < 16; ++j) s += hp[(l / 16) * 16 + j];
            worst = fmax(worst, fabs(ho[c * 64 + l] - s) / s);
        }
        printf("%-22s worst relative deviation from the block sum: %.3g\n", names[c], worst);
    }
    printf("x1 (A=p,B=1) lanes 0..15:"); for (int l = 0; l < 16; ++l) printf(" %.3f", ho[4 * 64 + l]); printf("\n");
    printf("x2 (A=1,B=p) lanes 0..15:"); for (int l = 0; l < 16; ++l) printf(" %.3f", ho[5 * 64 + l]); printf("\n");
    printf("p            lanes 0..15:"); for (int l = 0; l < 16; ++l) printf(" %.3f", hp[l]); printf("\n");
    return 0;
}
