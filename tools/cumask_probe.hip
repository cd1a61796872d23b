// Probe: does hipExtStreamCreateWithCUMask restrict a stream's workgroups to the masked CUs on this box, and how do mask bits map
// to XCDs?   hipcc --offload-arch=gfx950 -o tools/cumask_probe.bin tools/cumask_probe.hip && tools/cumask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void census(unsigned *out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
}
int main()
{
    const int nb = 2048;
    unsigned *d;
    hipMalloc(&d, nb * 4);
    std::vector<unsigned> h(nb);
    struct T { int lo, hi; } tests[] = {{0, 256}, {0, 64}, {0, 160}, {160, 256}, {64, 256}, {0, 96}, {96, 256}, {32, 64}};
    std::vector<std::vector<int>> sets;
    for (auto tc : tests) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tc.lo; i < tc.hi; ++i) mask[i >> 5] |= 1u << (i & 31);
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask);
        if (e != hipSuccess) { printf("[%d,%d): hipExtStreamCreateWithCUMask failed: %s\n", tc.lo, tc.hi, hipGetErrorString(e)); continue; }
        (void)hipMemsetAsync(d, 0xff, nb * 4, st);
        hipLaunchKernelGGL(census, dim3(nb), dim3(256), 0, st, d, 20000);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
        int per_xcc[16] = {0};
        std::vector<int> seen(16 * 256, 0);
        int distinct = 0;
        for (int b = 0; b < nb; ++b) {
            const unsigned xcc = (h[b] >> 16) & 15, cu = (h[b] >> 8) & 0xff;   // cu_id[11:8] sh[12] se[15:13]
            per_xcc[xcc]++;
            if (!seen[xcc * 256 + cu]++) distinct++;
        }
        sets.push_back(seen);
        printf("bits [%3d,%3d): distinct CUs %3d ; per xcc:", tc.lo, tc.hi, distinct);
        for (int x = 0; x < 8; ++x) { int c = 0; for (int k = 0; k < 256; ++k) c += seen[x * 256 + k] > 0; printf(" %d", c); }
        printf("\n");
        (void)hipStreamDestroy(st);
    }
    auto overlap = [&](int a, int b) { int c = 0; for (int k = 0; k < 16 * 256; ++k) c += sets[a][k] > 0 && sets[b][k] > 0; return c; };
    printf("overlap [0,160) & [160,256): %d ; [0,64) & [64,256): %d ; [0,96) & [96,256): %d ; [0,64) & [32,64): %d\n", overlap(2, 3), overlap(1, 4),
           overlap(5, 6), overlap(1, 7));
    return 0;
}
