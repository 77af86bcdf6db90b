// Where does the dispatcher put the workgroups of a grid that does not fill the chip a whole number of times?
// Every workgroup (256 threads, register budget of the likelihood kernels: three workgroups per CU at most)
// records the XCD / shader engine / CU it runs on and stays resident for ~100 us so that the whole grid is placed
// at once.  Prints the histogram of workgroups per CU and of waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/placement.hip -o gpurun_out/placement && gpurun_out/placement 625 782
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void __launch_bounds__(256, 3) __attribute__((amdgpu_num_vgpr(168))) census(unsigned *rec, long long spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        rec[2 * w] = hw;
        rec[2 * w + 1] = xcc;
    }
}

int main(int argc, char **argv)
{
    for (int a = 1; a < argc; ++a) {
        const int grid = atoi(argv[a]);
        unsigned *d;
        (void)hipMalloc(&d, sizeof(unsigned) * 8 * grid);
        hipLaunchKernelGGL(census, dim3(grid), dim3(256), 0, 0, d, 10000LL); // 100 MHz wall clock: 100 us
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(8 * grid);
        (void)hipMemcpy(h.data(), d, sizeof(unsigned) * 8 * grid, hipMemcpyDeviceToHost);
        std::map<unsigned, int> per_cu, per_simd;
        std::map<unsigned, std::map<unsigned, int>> blocks_cu;
        for (int w = 0; w < 4 * grid; ++w) {
            const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned cuid = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            per_simd[(cuid << 2) | simd]++;
            blocks_cu[cuid][w / 4]++;
        }
        std::map<int, int> hist_cu, hist_simd;
        for (auto &kv : blocks_cu) hist_cu[(int)kv.second.size()]++;
        for (auto &kv : per_simd) hist_simd[kv.second]++;
        printf("grid %d workgroups (%d waves): %zu CUs used;", grid, 4 * grid, blocks_cu.size());
        for (auto &kv : hist_cu) printf("  %d CUs with %d workgroups", kv.second, kv.first);
        printf(" |");
        for (auto &kv : hist_simd) printf("  %d SIMDs with %d waves", kv.second, kv.first);
        // which rounds of the launch order share a CU: first block index mod 256 of each CU's blocks
        int shown = 0;
        printf("\n   blocks of the first CUs:");
        for (auto &kv : blocks_cu) {
            if (shown++ >= 6) break;
            printf(" [");
            for (auto &b : kv.second) printf(" %u", b.first);
            printf(" ]");
        }
        printf("\n");
        (void)hipFree(d);
    }
    return 0;
}
