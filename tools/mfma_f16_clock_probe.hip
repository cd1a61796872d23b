// Tuning aid (round 5): what does the chip sustain when EVERY SIMD issues v_mfma_f32_32x32x16_f16 back to back for tens of
// milliseconds -- shader clock (clock64 against the 100 MHz wall clock), cycles per MFMA, TFLOP/s -- with 4 independent accumulators
// per wave (what csrc/conv_f16x3.hip has), one or two waves per SIMD, operands that are all zero / small / random (the matrix
// pipe's power draw depends on the data).  The nominal rate is 32 cycles per MFMA at 2.4 GHz = 2.5 PFLOP/s.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_clock_probe.hip -o variants/mfma_f16_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(int iters, int mode, unsigned long long *out, float *sink)
{
    f32x16 a[4] = {};
    h8 x[2], y[2];
    unsigned r = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 2; ++q)
        for (int j = 0; j < 8; ++j) {
            r = r * 1664525u + 1013904223u;
            const float u = ((r >> 8) & 0xffff) / 65536.0f - 0.5f;
            x[q][j] = (_Float16)(mode == 0 ? 0.0f : mode == 1 ? 1.0f : u);
            r = r * 1664525u + 1013904223u;
            y[q][j] = (_Float16)(mode == 0 ? 0.0f : mode == 1 ? 0.5f : (((r >> 8) & 0xffff) / 65536.0f - 0.5f));
        }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                a[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[q & 1], y[q >> 1], a[q], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.0f;
    for (int q = 0; q < 4; ++q) s += a[q][0];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = w1 - w0;
    }
}

static void run(int cus, int wps, int iters, int mode)
{
    const int bs = 256 * wps, waves = cus * 4 * wps, blocks = cus;
    unsigned long long *dout; float *dsink;
    hipMalloc(&dout, (size_t)waves * 2 * 8); hipMalloc(&dsink, (size_t)waves * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(bs), 0, 0, iters, mode, dout, dsink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)waves * 2);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int w = 0; w < waves; ++w) { ghz.push_back((double)h[2 * w] / ((double)h[2 * w + 1] * 10.0)); cyc.push_back((double)h[2 * w] / (12.0 * iters)); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double tf = (double)waves * 12.0 * iters * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%s operands, %d wave(s) per SIMD, %d MFMAs per wave: %.2f ms; clock %.3f GHz (min %.3f); cycles per MFMA and wave median %.1f (min %.1f, max %.1f); "
           "%.0f TFLOP/s = %.3f of 2500\n", mode == 0 ? "zero" : mode == 1 ? "constant" : "random", wps, 12 * iters, ms, ghz[ghz.size() / 2], ghz.front(),
           cyc[cyc.size() / 2], cyc.front(), cyc.back(), tf, tf / 2500.0);
    hipFree(dout); hipFree(dsink);
}

int main()
{
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    for (int mode = 0; mode < 3; ++mode)
        for (int wps = 1; wps <= 2; ++wps) {
            run(cus, wps, 20000 / wps, mode);      // ~3 ms
            run(cus, wps, 200000 / wps, mode);     // ~30 ms
        }
    return 0;
}
