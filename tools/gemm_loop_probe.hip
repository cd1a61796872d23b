// Tuning aid (round 5): the MFMA loop of k_gemm_ws in isolation -- one (tap, chain) chunk after the other, A fragments from LDS a pair of
// tiles ahead, 10 (or 5) accumulators, the chain's value added into a running total -- with nothing around it: no staging, no barriers, no
// set-up.  What issue interval does the loop itself reach (the pipe's: 32 cycles per v_mfma_f32_16x16x4_f32), at 1, 2, 3 waves per SIMD?
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/gemm_loop_probe.hip -o tools/gemm_loop_probe.bin && tools/gemm_loop_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// VARIANT 0: as k_gemm_ws (pairs, prefetch one pair ahead, sched_barrier pins); 1: prefetch two pairs ahead; 2: no pins (compiler's order);
// 3: all A fragments of a half-chain read up front (registers permitting)
template <int NOT, int NGH, int VARIANT>
__attribute__((amdgpu_waves_per_eu(2, 2)))
__global__ __launch_bounds__(256) void k(int chunks, unsigned long long *out, float *sink)
{
    constexpr int CO = 16 * NOT, HALF = 4 * CO, CHUNK = NGH * HALF, NQ = NGH * NOT;
    __shared__ f32x4 sA[2 * CHUNK];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * CHUNK; i += 256) sA[i] = f32x4{1.0f + i * 1e-6f, 0.5f, 0.25f, 2.0f};
    __syncthreads();
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 tot[NOT], taptot[NOT], bv[NGH];
    for (int k2 = 0; k2 < NOT; ++k2) { tot[k2] = zero; taptot[k2] = zero; }
    for (int h = 0; h < NGH; ++h) bv[h] = f32x4{1.0f + lane, 2.0f, 3.0f, 4.0f};
    const unsigned long long c0 = clock64();
    int buf = 0;
    for (int n = 0; n < chunks; ++n) {
        const f32x4 *A = sA + buf * CHUNK + lane;
        f32x4 acc[NOT];
        if (VARIANT == 3) {
#pragma unroll
            for (int h = 0; h < NGH; ++h) {
                f32x4 w[NOT];
#pragma unroll
                for (int k2 = 0; k2 < NOT; ++k2) w[k2] = A[h * HALF + k2 * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int k2 = 0; k2 < NOT; ++k2)
                        acc[k2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k2][c], bv[h][c], h == 0 && c == 0 ? zero : acc[k2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            constexpr int D = VARIANT == 1 ? 2 : 1;          // pairs of look-ahead
            f32x4 w[2 * (D + 1)];
#pragma unroll
            for (int p = 0; p < 2 * D; ++p) w[p] = p < NQ ? A[(p / NOT) * HALF + (p % NOT) * 64] : zero;
#pragma unroll
            for (int q = 0; q < NQ; q += 2) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int qq = q + 2 * D + p;
                    if (qq < NQ) w[2 * D + p] = VARIANT == 4 ? w[p] : A[(qq / NOT) * HALF + (qq % NOT) * 64];
                }
                if (VARIANT != 2) __builtin_amdgcn_sched_barrier(0);
                const int h0 = q / NOT, k0 = q % NOT, h1 = (q + 1) / NOT, k1 = (q + 1) % NOT;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[k0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][c], bv[h0][c], h0 == 0 && c == 0 ? zero : acc[k0], 0, 0, 0);
                    if (q + 1 < NQ) acc[k1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1][c], bv[h1][c], h1 == 0 && c == 0 ? zero : acc[k1], 0, 0, 0);
                }
                if (VARIANT != 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 2 * D; ++p) w[p] = w[p + 2];
            }
        }
        const int j = n % 5;
        if (VARIANT == 5) {
#pragma unroll
            for (int k2 = 0; k2 < NOT; ++k2) taptot[k2] = acc[k2];
        } else
#pragma unroll
        for (int k2 = 0; k2 < NOT; ++k2) taptot[k2] = j == 0 ? acc[k2] : taptot[k2] + acc[k2];
        if (j == 4) {
#pragma unroll
            for (int k2 = 0; k2 < NOT; ++k2) tot[k2] = tot[k2] + taptot[k2];
        }
        buf ^= 1;
    }
    const unsigned long long c1 = clock64();
    float s = 0.0f;
    for (int k2 = 0; k2 < NOT; ++k2) s += tot[k2][0] + tot[k2][3];
    sink[blockIdx.x * 256 + tid] = s;
    if (lane == 0) out[blockIdx.x * 4 + (tid >> 6)] = c1 - c0;
}

template <int NOT, int NGH, int VARIANT>
static void run(int cus, int wps, int chunks)
{
    const int blocks = cus * wps;
    unsigned long long *dout; float *dsink;
    hipMalloc(&dout, (size_t)blocks * 4 * 8); hipMalloc(&dsink, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NOT, NGH, VARIANT>), dim3(blocks), dim3(256), 0, 0, chunks, dout, dsink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    const double per_wave = (double)chunks * NGH * NOT * 4;
    for (auto v : h) cyc.push_back((double)v / per_wave);
    std::sort(cyc.begin(), cyc.end());
    const double tf = (double)blocks * 4 * per_wave * 2048.0 / (ms * 1e-3) / 1e12;
    printf("tiles %2d groups %d variant %d, %d wave(s) per SIMD: cycles per MFMA and wave median %.1f (min %.1f, max %.1f) -> %.1f per SIMD; %.1f TFLOP/s = %.3f of nominal\n",
           NOT, NGH, VARIANT, wps, cyc[cyc.size() / 2], cyc.front(), cyc.back(), cyc[cyc.size() / 2] / wps, tf, tf / 157.3);
    hipFree(dout); hipFree(dsink);
}

int main()
{
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int ch = 4000;
    for (int wps = 1; wps <= 2; ++wps) {
        run<10, 2, 0>(cus, wps, ch); run<10, 2, 4>(cus, wps, ch); run<10, 2, 5>(cus, wps, ch); run<10, 2, 3>(cus, wps, ch);
    }
    return 0;
}
