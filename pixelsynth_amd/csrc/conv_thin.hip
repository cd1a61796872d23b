// conv_thin.hip -- the two THIN 3 x 3 convolutions at either end of the refinement decoder (SURVEY 8f row 2;
// models/networks/architectures.py:126-167, models/layers/blocks.py:34-73) for gfx950 (MI355X), fp32 throughout:
//   ps_conv3x3_thin_in_nhwc_f32    4 input channels -> Co (block 0: the reprojected RGB + mask into 64 channels)
//   ps_conv3x3_thin_out_nhwc_f32   Ci -> at most 4 output channels (block 7: 128 channels into RGB)
// Neither is a matrix product worth the name (K = 36; N = 3): through MIOpen's implicit GEMM they take 0.5 and 0.9 ms per 16 views,
// as long as the decoder's 128 -> 128 layers do on the fp16 pipe (csrc/conv_f16x3.hip), while their arithmetic is 2 % of those.
// Here they are what they are -- one pass over the wide side of the layer (the 64-channel output / the 128-channel input) at memory
// speed, fp32 FMAs on the vector ALU (thin_in: two neighbouring pixels per thread, weights by broadcast LDS reads; thin_out: the patch
// in LDS, weights through the scalar cache -- two pixels per thread with the weights in LDS too was SLOWER, 394 against 338 us: a
// broadcast ds_read_b128 costs what any other does, and every weight read feeds only 6 CO FMAs) -- with the block's norm + ReLU,
// y = max(x * scale[b][c] - shift[b][c], 0) (models/layers/normalization.py:21-47), applied on the way in.  No bias: the caller
// folds it into the next pass, as for every convolution of the decoder.
#include "ps_common.h"

#include <algorithm>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TW = 64, TH = 8;                   // output pixels per 256-thread workgroup: a thread owns two neighbours in a row
constexpr int OTW = 32, OTH = 8;                 // thin_out: one pixel per thread
constexpr int OPW = OTW + 2, OPP = (OTH + 2) * OPW;   // its patch: 10 x 34 = 340 pixels

// ---- 4 -> Co.  A thread owns two neighbouring output pixels: their 3 x 4 x 4 inputs in registers, four output channels at a time;
// the weights in LDS ([tap][ci][Co], read as broadcast float4s: one read feeds eight FMAs of every lane).
template <bool FUSE> __global__ __launch_bounds__(256) void k_thin_in(const f32x4 *__restrict__ x, const f32x4 *__restrict__ scale,
                                                                       const f32x4 *__restrict__ shift, const float *__restrict__ w,
                                                                       float *__restrict__ y, int H, int W, int Co, int tiles_x,
                                                                       int tiles_per_frame)
{
    extern __shared__ f32x4 wl[];   // 36 * Co floats
    for (int i = threadIdx.x; i < 9 * Co; i += 256) wl[i] = ((const f32x4 *)w)[i];
    const int b = blockIdx.x / tiles_per_frame, tf = blockIdx.x - b * tiles_per_frame;
    const int oy = (tf / tiles_x) * TH + (threadIdx.x >> 5), ox = (tf % tiles_x) * TW + 2 * (threadIdx.x & 31);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 sc = zero, sh = zero;
    if (FUSE) { sc = scale[b]; sh = shift[b]; }
    f32x4 in[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int iy = oy + r - 1, ix = ox + c - 1;
            f32x4 v = zero;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                v = x[((size_t)b * H + iy) * W + ix];
                if (FUSE) v = __builtin_elementwise_max(v * sc - sh, zero);
            }
            in[r][c] = v;
        }
    __syncthreads();
    float *yp = y + (((size_t)b * H + oy) * W + ox) * Co;
    const int Co4 = Co >> 2;
    for (int c = 0; c < Co4; ++c) {
        f32x2 a0l = {0.f, 0.f}, a0h = a0l, a1l = a0l, a1h = a0l;   // pairs: v_pk_fma_f32, two FMAs per lane and issue slot
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 wv = wl[(t * 4 + j) * Co4 + c];
                const f32x2 wlo = {wv[0], wv[1]}, whi = {wv[2], wv[3]};
                const float u0 = in[t / 3][t % 3][j], u1 = in[t / 3][t % 3 + 1][j];
                const f32x2 uu0 = {u0, u0}, uu1 = {u1, u1};
                a0l = __builtin_elementwise_fma(uu0, wlo, a0l);
                a0h = __builtin_elementwise_fma(uu0, whi, a0h);
                a1l = __builtin_elementwise_fma(uu1, wlo, a1l);
                a1h = __builtin_elementwise_fma(uu1, whi, a1h);
            }
        *(f32x4 *)(yp + 4 * c) = (f32x4){a0l[0], a0l[1], a0h[0], a0h[1]};
        *(f32x4 *)(yp + Co + 4 * c) = (f32x4){a1l[0], a1l[1], a1h[0], a1h[1]};
    }
}

// ---- Ci -> CO <= 4.  The workgroup's 10 x 34 patch of 32 input channels in LDS as [c / 4][pixel] float4s (a wave's reads are
// consecutive), normalised on the way in; a thread owns one output pixel and CO accumulators.  w: [tap 9][Ci][CO]
template <int CO, bool FUSE> __global__ __launch_bounds__(256) void k_thin_out(const float *__restrict__ x, const float *__restrict__ scale,
                                                                               const float *__restrict__ shift, const float *__restrict__ w,
                                                                               float *__restrict__ y, int H, int W, int Ci, int tiles_x,
                                                                               int tiles_per_frame)
{
    __shared__ f32x4 patch[8][OPP];
    const int b = blockIdx.x / tiles_per_frame, tf = blockIdx.x - b * tiles_per_frame;
    const int ty0 = (tf / tiles_x) * OTH, tx0 = (tf % tiles_x) * OTW, ty = threadIdx.x >> 5, tx = threadIdx.x & 31;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float *xb = x + (size_t)b * H * W * Ci;
    float acc[CO];
#pragma unroll
    for (int k = 0; k < CO; ++k) acc[k] = 0.f;
    for (int c0 = 0; c0 < Ci; c0 += 32) {
        for (int i = threadIdx.x; i < OPP * 8; i += 256) {
            const int p = i >> 3, c4 = i & 7, pr = p / OPW, pc = p - pr * OPW, iy = ty0 + pr - 1, ix = tx0 + pc - 1;
            f32x4 v = zero;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                v = *(const f32x4 *)(xb + ((size_t)iy * W + ix) * Ci + c0 + 4 * c4);
                if (FUSE) {
                    const f32x4 sc = *(const f32x4 *)(scale + (size_t)b * Ci + c0 + 4 * c4), sh = *(const f32x4 *)(shift + (size_t)b * Ci + c0 + 4 * c4);
                    v = __builtin_elementwise_max(v * sc - sh, zero);
                }
            }
            patch[c4][p] = v;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int pp = (ty + t / 3) * OPW + tx + t % 3;
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) {
                const f32x4 v = patch[c4][pp];
                const float *wv = w + ((size_t)t * Ci + c0 + 4 * c4) * CO;   // (uniform: scalar loads)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < CO; ++k) acc[k] = __builtin_fmaf(v[j], wv[j * CO + k], acc[k]);
            }
        }
        __syncthreads();
    }
    float *yp = y + (((size_t)b * H + ty0 + ty) * W + tx0 + tx) * CO;
#pragma unroll
    for (int k = 0; k < CO; ++k) yp[k] = acc[k];
}

// ---- 4 -> 64 on the fp16 matrix pipe (split operands as in csrc/conv_f16x3.hip: v = hi + lo, three MFMAs per product, fp32 sums).
// K = 9 taps x 4 channels = 36, padded to two steps of v_mfma_f32_16x16x32_f16; a wave takes 16 consecutive pixels of a row per
// tile: lane (m, kb) = (lane & 15, lane >> 4) holds A[m][8 kb .. 8 kb + 7] = the four channels of taps 2 kb and 2 kb + 1 at pixel m
// (second step: tap 8 in kb = 0, zeros elsewhere), and, for each of the four 16-channel output tiles, B[8 kb ..][n] = the weights of
// those (tap, channel) pairs for output channel n -- loaded and split once per wave, 64 registers.  D: column = lane & 15 = channel,
// row = 4 (lane >> 4) + register = pixel.  w: [tap 9][ci 4][64].
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <bool FUSE> __global__ __launch_bounds__(256) void k_thin_in_mfma(const f32x4 *__restrict__ x, const f32x4 *__restrict__ scale,
                                                                            const f32x4 *__restrict__ shift, const float *__restrict__ w,
                                                                            float *__restrict__ y, int H, int W, int tiles_per_row, int ntiles,
                                                                            int *__restrict__ overflow)
{
    constexpr int CO = 64;
    const int lane = threadIdx.x & 63, m = lane & 15, kb = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int over = 0;
    auto split = [&](const float (&v)[8], h8 &hi, h8 &lo) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            over |= !(__builtin_fabsf(v[j]) <= 65000.f);
            hi[j] = (_Float16)v[j];
            lo[j] = (_Float16)(v[j] - (float)hi[j]);
        }
    };
    // the weights of this lane's (tap, channel) pairs, every output tile, both steps
    h8 bh[4][2], bl[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = ks * 32 + kb * 8 + j;
                v[j] = k < 36 ? w[(size_t)k * CO + nt * 16 + m] : 0.f;
            }
            split(v, bh[nt][ks], bl[nt][ks]);
        }
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int x0 = (tile % tiles_per_row) * 16, rowid = tile / tiles_per_row, b = rowid / H, oy = rowid - b * H, ox = x0 + m;
        f32x4 sc = zero, sh = zero;
        if (FUSE) { sc = scale[b]; sh = shift[b]; }
        auto pixel = [&](int t) {   // the (normalised) input of tap t at this lane's pixel; zeros outside the image and beyond tap 8
            f32x4 v = zero;
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            if (t < 9 && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                v = x[((size_t)b * H + iy) * W + ix];
                if (FUSE) v = __builtin_elementwise_max(v * sc - sh, zero);
            }
            return v;
        };
        const f32x4 p0 = pixel(2 * kb), p1 = pixel(2 * kb + 1), p2 = pixel(kb == 0 ? 8 : 9);
        const float v0[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
        const float v1[8] = {p2[0], p2[1], p2[2], p2[3], 0.f, 0.f, 0.f, 0.f};
        h8 a0h, a0l, a1h, a1l;
        split(v0, a0h, a0l);
        split(v1, a1h, a1l);
        float *yp = y + (((size_t)b * H + oy) * W + x0 + 4 * kb) * CO + m;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 acc = zero;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0l, bh[nt][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, bl[nt][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l, bh[nt][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, bl[nt][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, bh[nt][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, bh[nt][0], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) yp[(size_t)r * CO + nt * 16] = acc[r];
        }
    }
    if (over) *overflow = 1;
}

}  // namespace

extern "C" {

int ps_conv3x3_thin_in_f16x3_nhwc(const float *x, const float *scale, const float *shift, const float *w, int B, int H, int W, int Co,
                                  float *y, int *overflow, void *stream)
{
    PS_REQUIRE(x && w && y && overflow, "conv3x3_thin_in_f16x3: null pointer");
    PS_REQUIRE((scale == nullptr) == (shift == nullptr), "conv3x3_thin_in_f16x3: scale and shift come together");
    PS_REQUIRE(B > 0 && H > 0 && W > 0 && W % 16 == 0 && Co == 64, "conv3x3_thin_in_f16x3: W a multiple of 16 and Co = 64 required (W = %d, Co = %d)", W, Co);
    const size_t nt = (size_t)B * H * (W / 16);
    PS_REQUIRE(nt < ((size_t)1 << 31), "conv3x3_thin_in_f16x3: too many tiles");
    const int grid = (int)std::min<size_t>((nt + 3) / 4, 256 * 8);
    if (scale)
        hipLaunchKernelGGL(k_thin_in_mfma<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x, (const f32x4 *)scale,
                           (const f32x4 *)shift, w, y, H, W, W / 16, (int)nt, overflow);
    else
        hipLaunchKernelGGL(k_thin_in_mfma<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x, (const f32x4 *)nullptr,
                           (const f32x4 *)nullptr, w, y, H, W, W / 16, (int)nt, overflow);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_conv3x3_thin_in_nhwc_f32(const float *x, const float *scale, const float *shift, const float *w, int B, int H, int W, int Co,
                                float *y, void *stream)
{
    PS_REQUIRE(x && w && y, "conv3x3_thin_in: null pointer");
    PS_REQUIRE((scale == nullptr) == (shift == nullptr), "conv3x3_thin_in: scale and shift come together");
    PS_REQUIRE(B > 0 && H > 0 && W > 0 && H % TH == 0 && W % TW == 0 && Co > 0 && Co % 4 == 0 && Co <= 256,
               "conv3x3_thin_in: H a multiple of 8, W of 64, Co of 4 (at most 256) required (H = %d, W = %d, Co = %d)", H, W, Co);
    const int tiles_x = W / TW, tpf = (H / TH) * tiles_x;
    PS_REQUIRE((size_t)B * tpf < ((size_t)1 << 31), "conv3x3_thin_in: too many tiles");
    if (scale)
        hipLaunchKernelGGL(k_thin_in<true>, dim3(B * tpf), dim3(256), 36 * Co * sizeof(float), (hipStream_t)stream, (const f32x4 *)x, (const f32x4 *)scale,
                           (const f32x4 *)shift, w, y, H, W, Co, tiles_x, tpf);
    else
        hipLaunchKernelGGL(k_thin_in<false>, dim3(B * tpf), dim3(256), 36 * Co * sizeof(float), (hipStream_t)stream, (const f32x4 *)x, (const f32x4 *)nullptr,
                           (const f32x4 *)nullptr, w, y, H, W, Co, tiles_x, tpf);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_conv3x3_thin_out_nhwc_f32(const float *x, const float *scale, const float *shift, const float *w, int B, int H, int W, int Ci,
                                 int Co, float *y, void *stream)
{
    PS_REQUIRE(x && w && y, "conv3x3_thin_out: null pointer");
    PS_REQUIRE((scale == nullptr) == (shift == nullptr), "conv3x3_thin_out: scale and shift come together");
    PS_REQUIRE(B > 0 && H > 0 && W > 0 && H % OTH == 0 && W % OTW == 0 && Ci > 0 && Ci % 32 == 0 && Co >= 1 && Co <= 4,
               "conv3x3_thin_out: H a multiple of 8, W of 32, Ci of 32 and 1 <= Co <= 4 required (H = %d, W = %d, Ci = %d, Co = %d)", H, W, Ci, Co);
    const int tiles_x = W / OTW, tpf = (H / OTH) * tiles_x;
    PS_REQUIRE((size_t)B * tpf < ((size_t)1 << 31), "conv3x3_thin_out: too many tiles");
#define PS_THIN_OUT(CO)                                                                                                                  \
    do {                                                                                                                                 \
        if (scale)                                                                                                                       \
            hipLaunchKernelGGL((k_thin_out<CO, true>), dim3(B * tpf), dim3(256), 0, (hipStream_t)stream, x, scale, shift, w, y, H, W, Ci, \
                               tiles_x, tpf);                                                                                            \
        else                                                                                                                             \
            hipLaunchKernelGGL((k_thin_out<CO, false>), dim3(B * tpf), dim3(256), 0, (hipStream_t)stream, x, scale, shift, w, y, H, W, Ci, \
                               tiles_x, tpf);                                                                                            \
    } while (0)
    switch (Co) {
    case 1: PS_THIN_OUT(1); break;
    case 2: PS_THIN_OUT(2); break;
    case 3: PS_THIN_OUT(3); break;
    default: PS_THIN_OUT(4); break;
    }
#undef PS_THIN_OUT
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // extern "C"
