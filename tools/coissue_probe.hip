// Tuning aid (round 5): do an MFMA wave and a helper wave on the SAME SIMD overlap, or do they take turns?  256-thread workgroups x 2 per
// CU would not pin the pairing, so: 512-thread workgroups (waves k and k + 4 share a SIMD, tools-side probe of HW_ID), waves 0-3 run a
// dependent-free v_mfma_f32_16x16x4_f32 stream, waves 4-7 one of: nothing, a dependent-free v_fma_f32 stream (VALU), ds_read_b128 +
// ds_write_b128 traffic (LDS), global loads of a 64 MB array (VMEM).  Times of each role alone and together (shader clock, per wave).
//   hipcc --offload-arch=gfx950 -O2 tools/coissue_probe.hip -o tools/coissue_probe.bin && tools/coissue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int HELPER, bool MFMA_ON>
__global__ __launch_bounds__(512) void k(int iters, const f32x4 *g, unsigned long long *out, float *sink)
{
    __shared__ f32x4 sL[4096];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 4096; i += 512) sL[i] = f32x4{1.0f, 2.0f, 3.0f, 4.0f};
    __syncthreads();
    const unsigned long long c0 = clock64();
    float res = 0.0f;
    if (wave < 4) {
        if (MFMA_ON) {
            f32x4 a[4] = {};
            const float x = 1.0f + lane * 1e-6f, y = 0.5f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[q], 0, 0, 0);
            }
            res = a[0][0] + a[1][1] + a[2][2] + a[3][3];
        }
    } else if (HELPER == 1) {           // VALU: 64 independent fmas per iteration (256 cycles of VALU per 8 MFMAs' 256 cycles of pipe)
        float v[16];
        for (int q = 0; q < 16; ++q) v[q] = 1.0f + lane + q;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = __builtin_fmaf(v[q], 1.0000001f, 0.5f);
        }
        for (int q = 0; q < 16; ++q) res += v[q];
    } else if (HELPER == 2) {           // LDS: 4 reads + 4 writes of 1 KB per iteration
        f32x4 t[4];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = sL[((it * 4 + q) * 64 + lane) & 4095];
#pragma unroll
            for (int q = 0; q < 4; ++q) sL[((it * 4 + q + 2048) * 64 + lane) & 4095] = t[q] + 1.0f;
        }
        res = t[0][0];
    } else if (HELPER == 3) {           // VMEM: 4 loads of 1 KB per iteration
        f32x4 acc = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += g[(((size_t)blockIdx.x * 977 + it * 4 + q) * 64 + lane) & ((1u << 22) - 1)];
        }
        res = acc[0];
    }
    const unsigned long long c1 = clock64();
    sink[blockIdx.x * 512 + tid] = res;
    if (lane == 0) out[blockIdx.x * 8 + wave] = c1 - c0;
}

template <int HELPER, bool MFMA_ON>
static void run(int cus, int iters, const f32x4 *g, const char *name)
{
    unsigned long long *dout; float *dsink;
    hipMalloc(&dout, (size_t)cus * 8 * 8); hipMalloc(&dsink, (size_t)cus * 512 * 4);
    hipLaunchKernelGGL((k<HELPER, MFMA_ON>), dim3(cus), dim3(512), 0, 0, iters, g, dout, dsink);
    std::vector<unsigned long long> h((size_t)cus * 8);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> m, hp;
    for (int b = 0; b < cus; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : hp).push_back((double)h[b * 8 + w] / iters);
    std::sort(m.begin(), m.end()); std::sort(hp.begin(), hp.end());
    printf("%-34s MFMA waves: %7.1f cycles per iteration of 8 MFMAs (= %.1f per MFMA);  helper waves: %7.1f cycles per iteration\n", name,
           m[m.size() / 2], m[m.size() / 2] / 8.0, hp[hp.size() / 2]);
    hipFree(dout); hipFree(dsink);
}

int main()
{
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    f32x4 *g; hipMalloc(&g, (size_t)(1u << 22) * 16); hipMemset(g, 0, (size_t)(1u << 22) * 16);
    const int it = 20000;
    run<0, true>(cus, it, g, "MFMA alone");
    run<1, false>(cus, it, g, "VALU helper alone");
    run<1, true>(cus, it, g, "MFMA + VALU helper");
    run<2, false>(cus, it, g, "LDS helper alone");
    run<2, true>(cus, it, g, "MFMA + LDS helper");
    run<3, false>(cus, it, g, "VMEM helper alone");
    run<3, true>(cus, it, g, "MFMA + VMEM helper");
    return 0;
}
