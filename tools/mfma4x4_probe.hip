// Tuning aid: v_mfma_f32_4x4x1_16b_f32 on gfx950 -- operand layout, numerics (is it a plain fmaf per element?) and issue rate /
// dependent latency next to v_mfma_f32_16x16x4_f32.  The column kernel's chain role uses it for 4-column groups.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma4x4_probe.hip -o tools/mfma4x4_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// assumed layout: lane l -> block b = l >> 2; A holds row i = l & 3 of block b, B holds column j = l & 3 of block b,
// D register r of lane l = D[b][row r][col l & 3]
__global__ void k_sem(const float *A, const float *B, const float *C, float *D)
{
    const int lane = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * 64;
    const float a = A[base + lane], b = B[base + lane];
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(base + lane) * 4 + r];
    f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(base + lane) * 4 + r] = d[r];
}

template <int MODE>
__global__ void k_rate(float *out, int iters, long long *cycles)
{
    // MODE 0: 8 independent 4x4x1 accumulators; 1: one dependent 4x4x1 chain; 2: 8 independent 16x16x4; 3: dependent 16x16x4
    // MODE 4: 4x4x1 independent x8 with one ds_read_b128 per 4 MFMAs feeding B
    __shared__ f32x4 sB[256];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 256) sB[threadIdx.x] = f32x4{1.0f, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    f32x4 acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + lane * 1e-3f, b = 1.0f - lane * 1e-3f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[q], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 0, 0, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
        } else if (MODE == 3) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
        } else {
            const f32x4 b0 = sB[(it * 2 + (lane & 3)) & 255], b1 = sB[(it * 2 + 1 + (lane & 3)) & 255];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b0[q], acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[4 + q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b1[q], acc[4 + q], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    f32x4 s = acc[0];
    for (int q = 1; q < 8; ++q) s = s + acc[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

static float rnd(unsigned &s, int spread)
{
    s = s * 1664525u + 1013904223u;
    float m = ((s >> 8) & 0xffffff) / 16777216.0f * 2.0f - 1.0f;
    s = s * 1664525u + 1013904223u;
    int e = (int)((s >> 10) % (2 * spread + 1)) - spread;
    return ldexpf(m, e);
}

int main()
{
    const int NB = 4096;
    std::vector<float> A(NB * 64), B(NB * 64), C(NB * 256), D(NB * 256);
    unsigned s = 777;
    for (auto &v : A) v = rnd(s, 6);
    for (auto &v : B) v = rnd(s, 6);
    for (auto &v : C) v = rnd(s, 8);
    // a few subnormal cases
    for (int q = 0; q < 256; ++q) { A[q] = ldexpf(1.0f, -100); B[q] = ldexpf(1.5f, -40); C[q * 4] = ldexpf(1.25f, -140); }
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sem, dim3(NB), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0, n = 0;
    for (int blk = 0; blk < NB; ++blk)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int b = l >> 2, j = l & 3;
                const float a = A[blk * 64 + b * 4 + r], bb = B[blk * 64 + b * 4 + j], c0 = C[(blk * 64 + l) * 4 + r];
                const float want = fmaf(a, bb, c0), got = D[(blk * 64 + l) * 4 + r];
                bad += memcmp(&want, &got, 4) != 0;
                ++n;
            }
    printf("4x4x1_16b layout+numerics: n=%ld mismatches vs fmaf (assumed layout) = %ld\n", n, bad);

    float *dO; long long *dCy;
    hipMalloc(&dO, 1024 * 1024 * 4); hipMalloc(&dCy, 8);
    const int iters = 20000;
    auto run = [&](auto kern, const char *name, int threads, double flop_per_mfma) {
        long long cy = 0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, dO, 100, dCy);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, dO, iters, dCy);
        hipMemcpy(&cy, dCy, 8, hipMemcpyDeviceToHost);
        printf("%-44s 1 WG x %4d thr: %.2f cycles per MFMA per wave\n", name, threads, (double)cy / (iters * 8.0));
        // whole chip
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, dO, iters, dCy);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s chip (1024 WG x 256): %.1f TFLOP/s\n", name, 1024.0 * 4 * iters * 8 * flop_per_mfma / (ms * 1e-3) / 1e12);
    };
    run(k_rate<0>, "4x4x1_16b independent x8", 64, 512.0);
    run(k_rate<0>, "4x4x1_16b independent x8 (4 waves)", 256, 512.0);
    run(k_rate<0>, "4x4x1_16b independent x8 (8 waves)", 512, 512.0);
    run(k_rate<1>, "4x4x1_16b dependent", 64, 512.0);
    run(k_rate<2>, "16x16x4 independent x8", 64, 2048.0);
    run(k_rate<3>, "16x16x4 dependent", 64, 2048.0);
    run(k_rate<4>, "4x4x1_16b x8 + ds_read_b128 per 4", 64, 512.0);
    run(k_rate<4>, "4x4x1_16b x8 + ds_read_b128 per 4 (4 waves)", 256, 512.0);
    return 0;
}
