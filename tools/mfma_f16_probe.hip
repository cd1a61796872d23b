// Pins what a split-fp16 convolution needs to know about v_mfma_f32_32x32x16_f16 on gfx950:
//   (1) the A / B / D lane maps (A[m][k]: lane = m + 32 * (k / 8), element k % 8; B[k][n] likewise with n; D: col = lane & 31,
//       row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)), checked with an asymmetric integer product;
//   (2) whether fp16 SUBNORMAL inputs take part in the product (the low halves of split operands are subnormal for |x| < 0.1);
//   (3) that products are exact and the accumulation is fp32 (hi * hi of 11-bit mantissas = 22 bits).
// build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_probe.hip -o /tmp/mfma_f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16 *A, const _Float16 *B, float *D) {   // A (32, 16) row-major, B (16, 32) row-major, D (32, 32)
    const int lane = threadIdx.x, m = lane & 31, kb = (lane >> 5) * 8;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[m * 16 + kb + j]; b[j] = B[(kb + j) * 32 + m]; }
    f16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + m] = acc[r];
}

static int run(const std::vector<float> &Af, const std::vector<float> &Bf, const char *what, double tol) {
    std::vector<_Float16> A(512), B(512);
    for (int i = 0; i < 512; ++i) { A[i] = (_Float16)Af[i]; B[i] = (_Float16)Bf[i]; }
    _Float16 *dA, *dB; float *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    std::vector<float> D(1024);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    double worst = 0; int bad = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        double ref = 0;
        for (int kk = 0; kk < 16; ++kk) ref += (double)(float)A[m * 16 + kk] * (double)(float)B[kk * 32 + n];
        const double e = std::fabs(D[m * 32 + n] - ref) / (std::fabs(ref) + 1e-300);
        if (e > worst) worst = e;
        if (e > tol) ++bad;
    }
    printf("%-58s worst relative error %.3e, %d of 1024 beyond %.0e   D[0][0] = %.9g\n", what, worst, bad, tol, D[0]);
    hipFree(dA); hipFree(dB); hipFree(dD);
    return bad;
}

int main() {
    std::vector<float> A(512), B(512);
    for (int m = 0; m < 32; ++m) for (int kk = 0; kk < 16; ++kk) A[m * 16 + kk] = (float)((m * 7 + kk * 3) % 11 - 5);
    for (int kk = 0; kk < 16; ++kk) for (int n = 0; n < 32; ++n) B[kk * 32 + n] = (float)((kk * 5 + n * 13 + kk * n) % 9 - 4);
    int bad = run(A, B, "lane maps (asymmetric small integers)", 0);
    // subnormal B (2^-20 .. : below the smallest normal fp16 2^-14), normal A
    for (int i = 0; i < 512; ++i) B[i] = std::ldexp((float)((i * 37) % 15 + 1), -24);   // 1..15 * 2^-24: all subnormal
    bad += run(A, B, "subnormal B x integer A", 1e-6);
    for (int i = 0; i < 512; ++i) A[i] = std::ldexp((float)((i * 29) % 13 + 1), -24);
    for (int i = 0; i < 512; ++i) B[i] = (float)((i * 5) % 7 + 1) * 64.f;
    bad += run(A, B, "subnormal A x normal B", 1e-6);
    // full 11-bit mantissas: products need 22 bits, 16 of them summed in fp32
    for (int i = 0; i < 512; ++i) { A[i] = 1.f + ((i * 977) % 1024) / 1024.f; B[i] = 1.f + ((i * 613 + 7) % 1024) / 1024.f; }
    bad += run(A, B, "11-bit mantissas (exact products, fp32 sums)", 3e-7);
    printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
    return bad != 0;
}
