// Tuning aid (round 5): is v_mfma_f32_32x32x2_f32 a chain of two fused multiply-adds in ascending k, and are TWO of them back to
// back the same bits as ONE v_mfma_f32_16x16x4_f32 (a chain of four, tools/mfma_semantics.hip) on the same four products?  If so a
// 32 x 32 tile can walk the canonical channel order of the engine (csrc/lmconv_device.h) and every bit-identity test stays valid.
// build + run (prints one line; also the issue rate of a dependent-free stream of each shape, one wave per SIMD):
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma32_semantics.hip -o /tmp/mfma32_sem && /tmp/mfma32_sem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: [32 rows m][4 k], B: [4 k][32 cols n], C/D: [32][32].  32x32x2: lane l supplies A[m = l % 32][k = l / 32], B[k = l / 32][n = l % 32];
// D register r of lane l is row 8 * (r / 4) + 4 * (l / 32) + r % 4, column l % 32.
__global__ void k32(const float *A, const float *B, const float *C, float *D)
{
    const int lane = threadIdx.x, m = lane & 31, kh = lane >> 5;
    const float *Ab = A + (size_t)blockIdx.x * 128, *Bb = B + (size_t)blockIdx.x * 128;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[(size_t)blockIdx.x * 1024 + (8 * (r / 4) + 4 * kh + r % 4) * 32 + m];
    // k = 0, 1 first, then k = 2, 3
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(Ab[m * 4 + kh], Bb[kh * 32 + m], c, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_32x32x2f32(Ab[m * 4 + 2 + kh], Bb[(2 + kh) * 32 + m], d, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(size_t)blockIdx.x * 1024 + (8 * (r / 4) + 4 * kh + r % 4) * 32 + m] = d[r];
}
// the same products through v_mfma_f32_16x16x4_f32: four 16 x 16 quadrants
__global__ void k16(const float *A, const float *B, const float *C, float *D)
{
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    const float *Ab = A + (size_t)blockIdx.x * 128, *Bb = B + (size_t)blockIdx.x * 128;
    for (int qm = 0; qm < 2; ++qm)
        for (int qn = 0; qn < 2; ++qn) {
            f32x4 c;
            for (int r = 0; r < 4; ++r) c[r] = C[(size_t)blockIdx.x * 1024 + (16 * qm + 4 * kk + r) * 32 + 16 * qn + i];
            const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(Ab[(16 * qm + i) * 4 + kk], Bb[kk * 32 + 16 * qn + i], c, 0, 0, 0);
            for (int r = 0; r < 4; ++r) D[(size_t)blockIdx.x * 1024 + (16 * qm + 4 * kk + r) * 32 + 16 * qn + i] = d[r];
        }
}
template <int SHAPE>
__global__ void k_rate(float *out, int iters, unsigned long long *cycles)
{
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    f32x4 b0 = {}, b1 = {}, b2 = {}, b3 = {};
    const float x = 1.0f + threadIdx.x * 1e-6f, y = 0.5f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 32) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        } else {
            b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b0, 0, 0, 0); b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b1, 0, 0, 0);
            b2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b2, 0, 0, 0); b3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + b0[0] + b1[1] + b2[2] + b3[3];
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
    const int NB = 1024;
    std::vector<float> A(NB * 128), B(NB * 128), C(NB * 1024), D32(NB * 1024), D16(NB * 1024);
    unsigned s = 4242;
    for (auto &v : A) v = rnd(s, 6);
    for (auto &v : B) v = rnd(s, 6);
    for (auto &v : C) v = rnd(s, 8);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D32.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k32, dim3(NB), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D32.data(), dD, D32.size() * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k16, dim3(NB), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D16.data(), dD, D16.size() * 4, hipMemcpyDeviceToHost);
    long bad_fma = 0, bad_16 = 0, bad_pairdot = 0, n = 0;
    for (int blk = 0; blk < NB; ++blk)
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                float a[4], b[4];
                for (int q = 0; q < 4; ++q) { a[q] = A[blk * 128 + r * 4 + q]; b[q] = B[blk * 128 + q * 32 + c]; }
                const float c0 = C[(size_t)blk * 1024 + r * 32 + c], got = D32[(size_t)blk * 1024 + r * 32 + c], g16 = D16[(size_t)blk * 1024 + r * 32 + c];
                const float h = fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], fmaf(a[0], b[0], c0))));
                double t = (double)a[0] * b[0] + (double)a[1] * b[1];
                float p = (float)((double)c0 + t);
                t = (double)a[2] * b[2] + (double)a[3] * b[3];
                p = (float)((double)p + t);
                bad_fma += memcmp(&h, &got, 4) != 0;
                bad_16 += memcmp(&g16, &got, 4) != 0;
                bad_pairdot += memcmp(&p, &got, 4) != 0;
                ++n;
            }
    unsigned long long *dcy, cy32 = 0, cy16 = 0;
    float *dout;
    hipMalloc(&dcy, 8); hipMalloc(&dout, 1024 * 64 * 4);
    const int iters = 4096;
    hipLaunchKernelGGL(k_rate<32>, dim3(1024), dim3(64), 0, 0, dout, iters, dcy);
    hipMemcpy(&cy32, dcy, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_rate<16>, dim3(1024), dim3(64), 0, 0, dout, iters, dcy);
    hipMemcpy(&cy16, dcy, 8, hipMemcpyDeviceToHost);
    printf("n=%ld  2 x v_mfma_f32_32x32x2_f32 vs: fmaf chain of four (k ascending) mismatches=%ld, v_mfma_f32_16x16x4_f32 mismatches=%ld, pairwise exact dot mismatches=%ld;  "
           "cycles per MFMA, 4 independent accumulators, one wave: 32x32x2 %.1f, 16x16x4 %.1f\n", n, bad_fma, bad_16, bad_pairdot,
           (double)cy32 / (4.0 * iters), (double)cy16 / (4.0 * iters));
    return 0;
}
