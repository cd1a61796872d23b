// vq_ends.hip -- the two 3-channel ends of the VQ-VAE (models/vqvae2/vqvae.py:100-161) for gfx950 (MI355X): the first
// convolution of enc_b (Conv2d(3, 64, 4, stride 2, padding 1), :107) and the last transposed convolution of dec
// (ConvTranspose2d(64, 3, 4, stride 2, padding 1), :150), fp32 in / out on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32
// products).  MIOpen ran the first as im2col + one GEMM PER IMAGE (2.2 ms per 128 views for 0.1 GFLOP each) and the second as a
// backward-data kernel (1.4-2.6 ms); both are a few hundred MB of traffic and a few GFLOP.
//
//   ps_vq_stem_s2d_f32   image (B, 3, H, W) NCHW -> the layer's output WITH ITS BIAS, already in the space-to-depth form the next layer
//                        (the 4 x 4 stride-2 convolution, run as a 3 x 3 one over 2 x 2 blocks: vqvae.py `s2d_weight`) reads:
//                        (B, H / 4, W / 4, 4 * 64), channel (sy, sx, co) = output pixel (2 y + sy, 2 x + sx).  K = (ci, ky, kx) = 48 = 12 MFMA
//                        steps; a wave takes four blocks = 16 output pixels: 12 loads per lane, 48 MFMAs, 64 bytes out per lane.
//   ps_vq_head_f32       h (B, Hh, Wh, 64) NHWC -> image (B, 3, 2 Hh, 2 Wh) NCHW = conv_transpose(relu(h)) + bias: at the input's resolution the
//                        four output parities x 3 channels are 12 rows of ONE 16-row MFMA tile, K = 9 taps x 64 channels with the taps a parity
//                        does not touch as zero weights (`convt_weight`); a wave takes 16 positions of a row.
// Sum order: fixed (taps / channels in the order below, fp32 accumulation); compared against fp64 layers in tests/test_vqvae_gpu.py.
#include "ps_common.h"

#include <algorithm>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int VE_WAVES = 4, VE_THREADS = 64 * VE_WAVES;

// ---------------------------------------------------------------------------------------------------------------------------------
struct StemArgs {
    const float *x, *w, *bias;   // (B, 3, H, W); (64, 3, 4, 4); (64)
    float *y;                    // (B, H / 4, W / 4, 256)
    int B, H, W;
};

__global__ __launch_bounds__(VE_THREADS) void k_vq_stem(StemArgs a)
{
    __shared__ float sW[4 * 12 * 64];        // [output tile][step j][lane (co i, kk)] = w[co][4 j + kk]  (k = ci * 16 + ky * 4 + kx)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kk = lane >> 4;
    for (int e = tid; e < 4 * 12 * 64; e += VE_THREADS) {
        const int l = e & 63, j = (e >> 6) % 12, t = (e >> 6) / 12;
        sW[e] = a.w[(size_t)(t * 16 + (l & 15)) * 48 + 4 * j + (l >> 4)];
    }
    __syncthreads();
    const int H = a.H, W = a.W, Hb = H / 4, Wb = W / 4, tiles_row = Wb / 4;
    const size_t ntiles = (size_t)a.B * Hb * tiles_row;
    const int sx = i & 1, sy = (i >> 1) & 1, xb = i >> 2;
    for (size_t tile = (size_t)blockIdx.x * VE_WAVES + wave; tile < ntiles; tile += (size_t)gridDim.x * VE_WAVES) {
        const int xg = (int)(tile % tiles_row);
        const size_t rest = tile / tiles_row;
        const int yb = (int)(rest % Hb), b = (int)(rest / Hb);
        const int oy = 2 * yb + sy, ox = 2 * (4 * xg + xb) + sx;        // this lane's output pixel of the layer (at H / 2 x W / 2)
        const int ix = 2 * ox - 1 + kk;
        const float *img = a.x + (size_t)b * 3 * H * W;
        float v[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const int ci = j >> 2, iy = 2 * oy - 1 + (j & 3);
            v[j] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? img[((size_t)ci * H + iy) * W + ix] : 0.0f;
        }
        float *dst = a.y + (((size_t)b * Hb + yb) * Wb + 4 * xg + xb) * 256 + (sy * 2 + sx) * 64 + 4 * kk;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = *(const f32x4 *)(a.bias + 16 * t + 4 * kk);
#pragma unroll
            for (int j = 0; j < 12; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sW[(t * 12 + j) * 64 + lane], v[j], acc, 0, 0, 0);
            *(f32x4 *)(dst + 16 * t) = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
struct HeadArgs {
    const float *h, *wt, *bias;   // (B, Hh, Wh, 64); (64, 3, 4, 4); (3)
    float *y;                     // (B, 3, 2 Hh, 2 Wh)
    int B, Hh, Wh;
};

__global__ __launch_bounds__(VE_THREADS) void k_vq_head(HeadArgs a)
{
    __shared__ f32x4 sW[9 * 4 * 64];        // [tap (dy, dx)][16-channel chunk c][lane (row m, kk)] = W'[m][tap][16 c + 4 kk ..]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kk = lane >> 4;
    for (int e = tid; e < 9 * 4 * 64; e += VE_THREADS) {
        const int l = e & 63, c = (e >> 6) & 3, tap = e >> 8, m = l & 15, k4 = l >> 4;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1, py = m / 6, px = (m / 3) & 1, co = m % 3;
        const int ky = py + 1 - 2 * dy, kx = px + 1 - 2 * dx;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (m < 12 && ky >= 0 && ky <= 3 && kx >= 0 && kx <= 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = a.wt[(((size_t)(16 * c + 4 * k4 + q) * 3 + co) * 4 + ky) * 4 + kx];
        }
        sW[e] = v;
    }
    __syncthreads();
    const int Hh = a.Hh, Wh = a.Wh, tiles_row = Wh / 16;
    const size_t ntiles = (size_t)a.B * Hh * tiles_row;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (size_t tile = (size_t)blockIdx.x * VE_WAVES + wave; tile < ntiles; tile += (size_t)gridDim.x * VE_WAVES) {
        const int xg = (int)(tile % tiles_row);
        const size_t rest = tile / tiles_row;
        const int y = (int)(rest % Hh), b = (int)(rest / Hh), x = 16 * xg + i;
        const float *frame = a.h + (size_t)b * Hh * Wh * 64 + 4 * kk;
        f32x4 acc = zero;
        // A row of the 3 x 3 neighbourhood is read ONCE: lane (i, kk) fetches its own position and -- lanes 0 / 15 -- the position left /
        // right of the tile; the dx = -1 / +1 operands are the neighbouring lanes' registers (DPP row shifts: a row of 16 lanes is the 16
        // positions of one kk), the edge value standing in where the shift runs out of the row.  3.4 instead of 9 fetches per position.
        const int xe = i == 0 ? x - 1 : i == 15 ? x + 1 : x;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            const bool rin = yy >= 0 && yy < Hh, ein = rin && xe >= 0 && xe < Wh;
            const float *rc = frame + ((size_t)(rin ? yy : y) * Wh + x) * 64, *re = frame + ((size_t)(rin ? yy : y) * Wh + (ein ? xe : x)) * 64;
            f32x4 ctr[4], edge[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ctr[c] = *(const f32x4 *)(rc + 16 * c);
                edge[c] = *(const f32x4 *)(re + 16 * c);
                ctr[c] = rin ? __builtin_elementwise_max(ctr[c], zero) : zero;       // the ReLU in front of the layer; zero padding
                edge[c] = ein ? __builtin_elementwise_max(edge[c], zero) : zero;
            }
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int tap = (dy + 1) * 3 + dx + 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    f32x4 v = ctr[c];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {   // row_shr:1 / row_shl:1; the lane the shift leaves without a source keeps `old` = its edge value
                        if (dx < 0) v[q] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge[c][q]), __float_as_int(ctr[c][q]), 0x111, 0xF, 0xF, false));
                        if (dx > 0) v[q] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge[c][q]), __float_as_int(ctr[c][q]), 0x101, 0xF, 0xF, false));
                    }
                    const f32x4 w4 = sW[(tap * 4 + c) * 64 + lane];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[q], v[q], acc, 0, 0, 0);
                }
            }
        }
        // lane (position i, kk) holds rows m = 4 kk + r of the tile: (parity, channel) = (m / 3, m % 3)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 4 * kk + r;
            if (m >= 12) continue;
            const int py = m / 6, px = (m / 3) & 1, co = m % 3;
            a.y[(((size_t)b * 3 + co) * (2 * Hh) + 2 * y + py) * (2 * Wh) + 2 * x + px] = acc[r] + a.bias[co];
        }
    }
}

int grid_waves(size_t ntiles, int per_cu)
{
    static int cus_of[64] = {};   // per device, looked up once
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (!cus_of[dev] && hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus_of[dev] = 256;
        cus = cus_of[dev];
    }
    const size_t wgs = (ntiles + VE_WAVES - 1) / VE_WAVES;
    return (int)std::min<size_t>(wgs, (size_t)cus * per_cu);
}

}  // namespace

extern "C" {

int ps_vq_stem_s2d_f32(const float *x, const float *w, const float *bias, int B, int H, int W, float *y, void *stream)
{
    PS_REQUIRE(x && w && bias && y, "vq_stem: null pointer");
    PS_REQUIRE(B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 16 == 0, "vq_stem: H a multiple of 4 and W of 16 required (H = %d, W = %d)", H, W);
    PS_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0, "vq_stem: 16-byte aligned y / bias required");
    StemArgs a{x, w, bias, y, B, H, W};
    const size_t ntiles = (size_t)B * (H / 4) * (W / 16);
    hipLaunchKernelGGL(k_vq_stem, dim3(grid_waves(ntiles, 8)), dim3(VE_THREADS), 0, (hipStream_t)stream, a);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_vq_head_f32(const float *h, const float *wt, const float *bias, int B, int Hh, int Wh, float *y, void *stream)
{
    PS_REQUIRE(h && wt && bias && y, "vq_head: null pointer");
    PS_REQUIRE(B > 0 && Hh > 0 && Wh > 0 && Wh % 16 == 0, "vq_head: Wh a multiple of 16 required (Wh = %d)", Wh);
    PS_REQUIRE(((uintptr_t)h & 15) == 0, "vq_head: 16-byte aligned h required");
    HeadArgs a{h, wt, bias, y, B, Hh, Wh};
    const size_t ntiles = (size_t)B * Hh * (Wh / 16);
    hipLaunchKernelGGL(k_vq_head, dim3(grid_waves(ntiles, 4)), dim3(VE_THREADS), 0, (hipStream_t)stream, a);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // extern "C"
