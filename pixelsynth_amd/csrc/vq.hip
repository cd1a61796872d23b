// vq.hip -- vector quantisation around the AR loop for gfx950 (MI355X).
//
// Replaces, behind the C ABI of include/pixelsynth_hip.h, the two pieces of the VQ-VAE-2 top level that sit
// directly on the novel-view path (models/vqvae2/vqvae.py):
//   Quantize.forward, inference part   :41-51   nearest codebook entry of every latent vector  -> ps_vq_nearest_f32
//   Quantize.embed_code + permute      :77-78, :306-307   codes -> (B,D,H,W) latent grid        -> ps_vq_embed_f32
// so that the codes the PixelCNN consumes / produces never leave the device (the reference goes through
// a one-hot (N,512) matrix, a (N,512) distance matrix and a host-visible argmax).
//
// Definition of the distance (the reference leaves the summation order to the BLAS it runs on):
//   dist[n][k] = (zz[n] - 2 * dot[n][k]) + ee[k],  zz / ee / dot accumulated in ascending d with fused multiply-adds,
//   idx[n] = the smallest k among the minimisers of dist[n][.]   (torch.max returns the first maximum of -dist)
#include "ps_common.h"

namespace {

constexpr int VQ_ROWS = 16;     // latent vectors per workgroup
constexpr int VQ_THREADS = 256;
constexpr int VQ_MAX_D = 64;

// z: layout 0 = (N, D) row-major; layout 1 = (B, D, HW), row n = b * HW + p  (NCHW, no permute needed)
__global__ __launch_bounds__(VQ_THREADS) void k_vq_nearest(const float *z, int layout, const float *embed, int N, int D, int K,
                                                           int HW, int32_t *idx, float *mindist)
{
    __shared__ float sZ[VQ_ROWS][VQ_MAX_D + 1];
    __shared__ float sZZ[VQ_ROWS];
    __shared__ float sBestD[VQ_ROWS][VQ_THREADS / 64];
    __shared__ int sBestK[VQ_ROWS][VQ_THREADS / 64];
    const int t = threadIdx.x, n0 = blockIdx.x * VQ_ROWS;
    for (int e = t; e < VQ_ROWS * D; e += VQ_THREADS) {
        const int r = e / D, d = e - r * D, n = n0 + r;
        float v = 0.0f;
        if (n < N) v = layout == 0 ? z[(size_t)n * D + d] : z[((size_t)(n / HW) * D + d) * HW + (n % HW)];
        sZ[r][d] = v;
    }
    __syncthreads();
    if (t < VQ_ROWS) {
        float s = 0.0f;
        for (int d = 0; d < D; ++d) s = __builtin_fmaf(sZ[t][d], sZ[t][d], s);
        sZZ[t] = s;
    }
    __syncthreads();
    float bestD[VQ_ROWS];
    int bestK[VQ_ROWS];
#pragma unroll
    for (int r = 0; r < VQ_ROWS; ++r) { bestD[r] = INFINITY; bestK[r] = 0x7fffffff; }
    for (int k = t; k < K; k += VQ_THREADS) {   // codes of this thread, ascending: strict < keeps the smallest k
        float dot[VQ_ROWS];
#pragma unroll
        for (int r = 0; r < VQ_ROWS; ++r) dot[r] = 0.0f;
        float ee = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float e = embed[(size_t)d * K + k];   // coalesced over k
            ee = __builtin_fmaf(e, e, ee);
#pragma unroll
            for (int r = 0; r < VQ_ROWS; ++r) dot[r] = __builtin_fmaf(sZ[r][d], e, dot[r]);
        }
#pragma unroll
        for (int r = 0; r < VQ_ROWS; ++r) {
            const float dist = (sZZ[r] - 2.0f * dot[r]) + ee;
            if (dist < bestD[r]) { bestD[r] = dist; bestK[r] = k; }
        }
    }
    // (dist, k) lexicographic minimum over the 256 threads of the workgroup, row by row
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int r = 0; r < VQ_ROWS; ++r) {
        float bd = bestD[r];
        int bk = bestK[r];
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(bd, off, 64);
            const int ok = __shfl_xor(bk, off, 64);
            if (od < bd || (od == bd && ok < bk)) { bd = od; bk = ok; }
        }
        if (lane == 0) { sBestD[r][wave] = bd; sBestK[r][wave] = bk; }
    }
    __syncthreads();
    if (t < VQ_ROWS && n0 + t < N) {
        float bd = sBestD[t][0];
        int bk = sBestK[t][0];
        for (int w = 1; w < VQ_THREADS / 64; ++w) {
            const float od = sBestD[t][w];
            const int ok = sBestK[t][w];
            if (od < bd || (od == bd && ok < bk)) { bd = od; bk = ok; }
        }
        idx[n0 + t] = bk;
        if (mindist) mindist[n0 + t] = bd;
    }
}

// out (B, D, HW) <- embed (D, K) columns selected by idx (B, HW); invalid codes give zeros
__global__ void k_vq_embed(const int32_t *idx, const float *embed, int B, int HW, int D, int K, float *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)B * D * HW;
    if (i >= total) return;
    const int p = (int)(i % HW), d = (int)((i / HW) % D), b = (int)(i / ((size_t)HW * D));
    const int k = idx[(size_t)b * HW + p];
    out[i] = (k >= 0 && k < K) ? embed[(size_t)d * K + k] : 0.0f;
}

}  // namespace

extern "C" {

int ps_vq_nearest_f32(const float *z, int layout, const float *embed, int N, int D, int K, int HW, int32_t *idx,
                      float *mindist, void *stream)
{
    PS_REQUIRE(z && embed && idx, "vq_nearest: null pointer");
    PS_REQUIRE(N > 0 && D > 0 && D <= VQ_MAX_D && K > 0, "vq_nearest: bad sizes (N=%d, D=%d <= %d, K=%d)", N, D, VQ_MAX_D, K);
    PS_REQUIRE(layout == 0 || (layout == 1 && HW > 0 && N % HW == 0), "vq_nearest: layout 1 needs N = B * HW");
    hipLaunchKernelGGL(k_vq_nearest, dim3((N + VQ_ROWS - 1) / VQ_ROWS), dim3(VQ_THREADS), 0, (hipStream_t)stream, z, layout, embed,
                       N, D, K, HW, idx, mindist);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_vq_embed_f32(const int32_t *idx, const float *embed, int B, int HW, int D, int K, float *out, void *stream)
{
    PS_REQUIRE(idx && embed && out, "vq_embed: null pointer");
    PS_REQUIRE(B > 0 && HW > 0 && D > 0 && K > 0, "vq_embed: bad sizes");
    const size_t total = (size_t)B * D * HW;
    hipLaunchKernelGGL(k_vq_embed, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, embed, B, HW, D,
                       K, out);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // extern "C"
