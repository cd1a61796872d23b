// Which SIMD does hardware wave k of a workgroup sit on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh [12], se [15:13])
// hipcc --offload-arch=gfx950 -O2 tools/wave_simd_probe.hip -o tools/wave_simd_probe.bin && tools/wave_simd_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out, int waves)
{
    unsigned hw = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * waves + (threadIdx.x >> 6)] = hw;
}
int main()
{
    for (int waves : {4, 8, 16}) {
        const int blocks = 1024;
        unsigned *d;
        hipMalloc(&d, blocks * waves * 4);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 0, 0, d, waves);
        std::vector<unsigned> h(blocks * waves);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        int hist[16][4] = {};
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < waves; ++w) hist[w][(h[b * waves + w] >> 4) & 3]++;
        printf("%d waves per workgroup: SIMD histogram per hardware wave (1024 workgroups)\n", waves);
        for (int w = 0; w < waves; ++w) printf("  wave %2d: %4d %4d %4d %4d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        printf("  first workgroups:");
        for (int b = 0; b < 4; ++b) { printf(" ["); for (int w = 0; w < waves; ++w) printf("%u", (h[b * waves + w] >> 4) & 3); printf("]"); }
        printf("\n");
        hipFree(d);
    }
    return 0;
}
