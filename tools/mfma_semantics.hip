// Tuning aid: which scalar arithmetic reproduces v_mfma_f32_16x16x4_f32 bit for bit?
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_semantics.hip -o /tmp/mfma_sem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float *A, const float *B, const float *C, float *D)
{
    // A: [16 rows i][4 k], B: [4 k][16 cols j]; lane (kk = lane>>4, i = lane&15): a = A[i][kk], b = B[kk][i]
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    const float a = A[(blockIdx.x * 16 + i) * 4 + kk], b = B[(blockIdx.x * 4 + kk) * 16 + i];
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(blockIdx.x * 16 + 4 * kk + r) * 16 + i];
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(blockIdx.x * 16 + 4 * kk + r) * 16 + i] = d[r];
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
    unsigned s = 12345;
    for (auto &v : A) v = rnd(s, 6);
    for (auto &v : B) v = rnd(s, 6);
    for (auto &v : C) v = rnd(s, 8);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    long bad[6] = {0, 0, 0, 0, 0, 0}, n = 0;
    for (int blk = 0; blk < NB; ++blk)
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                float a[4], b[4];
                for (int q = 0; q < 4; ++q) { a[q] = A[(blk * 16 + r) * 4 + q]; b[q] = B[(blk * 4 + q) * 16 + c]; }
                const float c0 = C[(blk * 16 + r) * 16 + c], got = D[(blk * 16 + r) * 16 + c];
                float h[6];
                h[0] = fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], fmaf(a[0], b[0], c0))));       // fused, k ascending
                h[1] = fmaf(a[0], b[0], fmaf(a[1], b[1], fmaf(a[2], b[2], fmaf(a[3], b[3], c0))));       // fused, k descending
                { volatile float t = c0; for (int q = 0; q < 4; ++q) { volatile float p = a[q] * b[q]; t = t + p; } h[2] = t; }  // unfused ascending
                { double t = c0; for (int q = 0; q < 4; ++q) t += (double)a[q] * b[q]; h[3] = (float)t; }  // exact dot, one rounding
                { double t = 0; for (int q = 0; q < 4; ++q) t += (double)a[q] * b[q]; h[4] = c0 + (float)t; }  // dot rounded then added
                { float t = fmaf(a[1], b[1], a[0] * b[0]); float u = fmaf(a[3], b[3], a[2] * b[2]); h[5] = c0 + (t + u); }
                for (int q = 0; q < 6; ++q) bad[q] += memcmp(&h[q], &got, 4) != 0;
                ++n;
            }
    printf("n=%ld mismatches: fma_asc=%ld fma_desc=%ld unfused_asc=%ld exactdot=%ld dot_then_add=%ld tree=%ld\n", n, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
    return 0;
}
