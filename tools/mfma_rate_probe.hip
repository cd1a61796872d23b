// How fast do four waves per SIMD issue v_mfma_f32_16x16x4_f32 in the shape k_gemm uses (10 independent accumulator
// chains, 40 MFMAs per chunk), alone and with the per-chunk / per-tap vector work around them?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate_probe.bin mfma_rate_probe.hip && ./mfma_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__attribute__((amdgpu_waves_per_eu(4, 4))) __global__ __launch_bounds__(64) void k(float *out, int iters, float seed)
{
    f32x4 acc[10], tot[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    const f32x4 z = {0, 0, 0, 0};
    f32x4 av[10], bv[5];
    for (int j = 0; j < 10; ++j) av[j] = f32x4{seed + j, seed, 1.0f, 0.5f} * (float)threadIdx.x;
    for (int j = 0; j < 5; ++j) bv[j] = f32x4{seed, 1.0f + j, seed, 0.25f};
    for (int it = 0; it < iters; ++it) {
        for (int j = 0; j < 10; ++j) acc[j] = z;
        for (int chunk = 0; chunk < 2; ++chunk) {
            if (MODE >= 1) {  // the mask multiply / select of a chunk
                for (int j = 0; j < 5; ++j) {
                    asm volatile("" : "+v"(bv[j]));
                    bv[j] = (threadIdx.x & 1) ? bv[j] * seed : z + bv[j];
                }
            }
            for (int u = 0; u < 2; ++u) {
                for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(av[u * 5 + j]));
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int j = 0; j < 5; ++j)
                        acc[u * 5 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u * 5 + j][c], bv[j][c], acc[u * 5 + j], 0, 0, 0);
            }
        }
        if (MODE >= 2) {  // chunk_total + tot
            for (int u = 0; u < 2; ++u)
                tot[u] = tot[u] + ((((acc[u * 5] + acc[u * 5 + 1]) + acc[u * 5 + 2]) + acc[u * 5 + 3]) + acc[u * 5 + 4]);
        } else {
            for (int j = 0; j < 10; ++j) asm volatile("" ::"v"(acc[j]));
        }
    }
    f32x4 s = tot[0] + tot[1];
    for (int j = 0; j < 10; ++j) s += acc[j];
    if (s.x == 12345.678f) out[threadIdx.x] = s.x + s.y + s.z + s.w;
}

template <int MODE>
void run(const char *name, float *out)
{
    const int iters = 200, blocks = 256 * 4 * 4 * 8;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, 10, 1.5f);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.5f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flops = (double)blocks * iters * 80 * 2048.0;
    printf("%-28s %.3f ms  %.1f TFLOP/s (%.0f %% of 157.3)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main()
{
    float *out; hipMalloc(&out, 4096);
    run<0>("mfma only", out);
    run<1>("+ mask multiply per chunk", out);
    run<2>("+ chain totals per tap", out);
    return 0;
}
