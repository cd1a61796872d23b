// Tuning aid (round 5): what does the shader clock run at while EVERY SIMD of the chip issues fp32 MFMAs back to back, what issue
// interval does a dependent-free stream of v_mfma_f32_16x16x4_f32 reach (the pipe's nominal: 32 cycles), and how does that depend on the
// number of independent accumulators a wave cycles through and on the waves per SIMD?  (The kernels' MFMA-busy figures are quoted
// against the nominal 2.4 GHz x 32 cycles.)  Shader clock (clock64) and the 100 MHz wall clock (wall_clock64) at both ends of every wave.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_clock_probe.hip -o tools/mfma_clock_probe.bin && tools/mfma_clock_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int DISTINCT>
__global__ void k(int iters, unsigned long long *out, float *sink)
{
    f32x4 a[NACC] = {};
    float x[8], y[8];
    for (int q = 0; q < 8; ++q) { x[q] = 1.0f + threadIdx.x * 1e-6f + q; y[q] = 0.5f + q; }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
            for (int q = 0; q < NACC; ++q)
                a[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[DISTINCT ? (r * NACC + q) & 7 : 0], y[DISTINCT ? (r * NACC + q) & 7 : 0], a[q], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.0f;
    for (int q = 0; q < NACC; ++q) s += a[q][0];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = w1 - w0;
    }
}

template <int NACC, int DISTINCT>
static void run(int cus, int wps, int bs, int iters)
{
    const int waves = cus * 4 * wps, blocks = waves / (bs / 64);
    unsigned long long *dout; float *dsink;
    hipMalloc(&dout, (size_t)waves * 2 * 8); hipMalloc(&dsink, (size_t)waves * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, DISTINCT>), dim3(blocks), dim3(bs), 0, 0, iters, dout, dsink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)waves * 2);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int w = 0; w < waves; ++w) { ghz.push_back((double)h[2 * w] / ((double)h[2 * w + 1] * 10.0)); cyc.push_back((double)h[2 * w] / (8.0 * iters)); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double tf = (double)waves * 8.0 * iters * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%d accumulators, %s operands, %d-thread blocks, %d wave(s) per SIMD, %d MFMAs per wave: %.2f ms; clock %.3f GHz (min %.3f); cycles per MFMA and wave "
           "median %.1f (min %.1f, max %.1f); %.1f TFLOP/s = %.3f of the nominal 157.3\n", NACC, DISTINCT ? "distinct" : "same", bs, wps, 8 * iters, ms,
           ghz[ghz.size() / 2], ghz.front(), cyc[cyc.size() / 2], cyc.front(), cyc.back(), tf, tf / 157.3);
    hipFree(dout); hipFree(dsink);
}

int main()
{
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int it = 100000;   // ~11 ms at 32 cycles per MFMA
    run<8, 0>(cus, 1, 256, it); run<4, 0>(cus, 1, 256, it); run<2, 0>(cus, 1, 256, it);
    run<8, 0>(cus, 1, 64, it);  run<4, 0>(cus, 1, 64, it);
    run<8, 1>(cus, 1, 256, it); run<4, 1>(cus, 1, 256, it);
    run<8, 0>(cus, 2, 256, it); run<4, 0>(cus, 2, 256, it); run<2, 0>(cus, 2, 256, it); run<4, 1>(cus, 2, 256, it);
    run<4, 0>(cus, 3, 256, it); run<2, 0>(cus, 3, 256, it); run<4, 0>(cus, 4, 256, it);
    return 0;
}
