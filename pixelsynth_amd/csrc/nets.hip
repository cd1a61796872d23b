// nets.hip -- the elementwise passes of the refinement decoder's blocks for gfx950 (MI355X).
//
// A ResNet_Block of the reference (models/layers/blocks.py:34-73) is, between its convolutions,
//   LinearNoiseLayer + stored-statistics batch norm + ReLU   (models/layers/normalization.py:21-47, :117-184)
//   avg_pool 3x3 / 2  or  bilinear x2 of BOTH branches, then their sum                 (blocks.py:61-73)
// which torch runs as 2 + 3 full passes over (B, C, H, W) activations per block -- about a third of the decoder's time
// next to MIOpen's convolutions.  Here each is ONE pass over channels-last (NHWC) memory, 16 bytes per lane:
//   ps_affine_relu_nhwc_f32      y = max(x * scale[b][c] - shift[b][c], 0)
//   ps_pool_add_nhwc_f32         out = avg_pool2d(a, 3, 2, 1) + avg_pool2d(b, 3, 2, 1)       (count_include_pad, as torch's default)
//   ps_upsample_add_nhwc_f32     out = bilinear_x2(a) + bilinear_x2(b)                       (align_corners = False)
//   ps_add_bias_nhwc_f32         out = a + b + bias[c]
//   ps_cat_mask_nhwc_f32         the decoder's input, cat((rgb, 1 - background_mask), 1), NCHW -> NHWC in one pass
//   ps_noise_affine_f32          the (B, C) scale / shift of a LinearNoiseLayer from its noise draw: one launch instead of eight
// (b may be NULL: the resampled branch alone.)  The bias of a convolution is a pass of its own in torch; here the convolutions
// run without it and the per-channel constant rides along in the pass that consumes their output: folded into `shift` of the
// next norm (host), or the `bias` argument of the resampling / residual kernels.  The convolutions themselves are csrc/conv_f16x3.hip,
// conv_thin.hip and conv1x1.hip (rounds 5 and 6).
#include "ps_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// x, y: (B, HW, C) channels-last; scale, shift: (B, C)
__global__ __launch_bounds__(256) void k_affine_relu(const f32x4 *__restrict__ x, const f32x4 *__restrict__ scale,
                                                     const f32x4 *__restrict__ shift, size_t per_frame4, int C4, size_t total4,
                                                     f32x4 *__restrict__ y)
{
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const size_t b = i / per_frame4;
        const int c4 = (int)(i % C4);
        const f32x4 v = x[i] * scale[b * C4 + c4] - shift[b * C4 + c4];
        y[i] = __builtin_elementwise_max(v, zero);
    }
}

// 3x3 window sums of a (and b) around input pixel (2 yo, 2 xo), zero padding, divisor 9
// `bias` (per channel, may be null) is a constant that belongs to every INPUT pixel of the pooled tensors (the bias of the
// convolutions that produced them, not yet added): it reaches the output with the share of the window inside the image
// `post` (may be null): a tensor of the OUTPUT's shape added after the pooling -- the other branch of a down-sampling block when its
// 1 x 1 convolution was run on the pooled input instead (pooling and a 1 x 1 convolution commute; the convolution then costs a quarter)
// (Round 6, measured and not kept: one workgroup per output row with 32-bit index arithmetic, 725 us per launch against this form's 685;
// the row's input columns summed over their three rows in LDS first -- every input element fetched once instead of 2.25 times --: 1 266 us.)
__global__ __launch_bounds__(256) void k_pool_add(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b,
                                                  const f32x4 *__restrict__ bias, const f32x4 *__restrict__ post, int H, int W, int C4,
                                                  size_t total4, f32x4 *__restrict__ out)
{
    const int Ho = H / 2, Wo = W / 2;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int xo = (int)(p % Wo);
        p /= Wo;
        const int yo = (int)(p % Ho);
        const size_t f = p / Ho;
        f32x4 sa = zero, sb = zero;
        int inside = 0;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = 2 * yo + dy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = 2 * xo + dx;
                if (xx < 0 || xx >= W) continue;
                const size_t at = ((f * H + yy) * W + xx) * C4 + c4;
                sa += a[at];
                if (b) sb += b[at];
                ++inside;
            }
        }
        const float ninth = 1.0f / 9.0f;
        f32x4 v = b ? sa * ninth + sb * ninth : sa * ninth;
        if (bias) v += bias[c4] * ((float)inside * ninth);
        if (post) v += post[i];
        out[i] = v;
    }
}

// bilinear x2, align_corners = False: source coordinate (d + 0.5) / 2 - 0.5, clamped at 0; the two source samples and the
// weight of the second one
__device__ __forceinline__ void up_src(int d, int n, int &i0, int &i1, float &w1)
{
    const float s = fmaxf((d + 0.5f) * 0.5f - 0.5f, 0.0f);
    i0 = (int)s;
    i1 = min(i0 + 1, n - 1);
    w1 = s - (float)i0;
}

// One workgroup per INPUT row (frame, y), grid-stride: a thread owns (x, channel quad) and writes the 2 x 2 output block it grows into
// from the 3 x 3 input neighbourhood -- 9 fetches per 4 outputs where a thread per output fetched 16 (round 6: the pass was bound by those
// fetches, 2.7 TB/s of 5 the chip streams).  Every output is the SAME expression of the same four samples and weights as before (up_src's
// indices and weights, spelled out: output 2 x reads (x - 1, x; 0.75) -- (0, 1; 0) at the border --, output 2 x + 1 reads (x, x + 1; 0.25)).
__global__ __launch_bounds__(256) void k_upsample_add(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b,
                                                      const f32x4 *__restrict__ bias, int H, int W, int C4, unsigned rows,
                                                      f32x4 *__restrict__ out)
{
    const int per_row = W * C4, out_row = 2 * per_row;
    for (unsigned row = blockIdx.x; row < rows; row += gridDim.x) {
        const unsigned f = row / (unsigned)H;
        const int y = (int)(row - f * (unsigned)H);
        const int yt = max(y - 1, 0), yb = min(y + 1, H - 1);
        const size_t fb = (size_t)f * H * W * C4;
        // output row 2 y: (y0, y1, wy) = (y - 1, y, 0.75), at the border (0, 1, 0); output row 2 y + 1: (y, y + 1 clamped, 0.25)
        const float wy0 = y > 0 ? 0.75f : 0.0f, wy1 = 0.25f;
        f32x4 *o0 = out + ((size_t)f * 2 * H + 2 * y) * out_row, *o1 = o0 + out_row;
        for (int e = threadIdx.x; e < per_row; e += 256) {
            const int x = e / C4, c4 = e - x * C4;
            const int xl = max(x - 1, 0), xr = min(x + 1, W - 1);
            const float wx0 = x > 0 ? 0.75f : 0.0f, wx1 = 0.25f;      // output column 2 x: (x - 1, x; 0.75) or (0, 1; 0); 2 x + 1: (x, xr; 0.25)
            f32x4 res[2][2];
#pragma unroll
            for (int src = 0; src < 2; ++src) {
                const f32x4 *t = src == 0 ? a : b;
                if (!t) continue;
                t += fb + c4;
                const f32x4 *T = t + (size_t)yt * per_row, *C = t + (size_t)y * per_row, *B = t + (size_t)yb * per_row;
                const f32x4 tl = T[xl * C4], tm = T[x * C4], tr = T[xr * C4], cl = C[xl * C4], cm = C[x * C4], cr = C[xr * C4],
                            bl = B[xl * C4], bm = B[x * C4], br = B[xr * C4];
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    // the two source rows of this output row, as (left, middle, right) triples
                    const bool top_pair = py == 0 && y > 0;          // rows (y - 1, y)
                    const f32x4 u0l = py == 0 ? (top_pair ? tl : cl) : cl, u0m = py == 0 ? (top_pair ? tm : cm) : cm, u0r = py == 0 ? (top_pair ? tr : cr) : cr;
                    const f32x4 u1l = py == 0 ? (top_pair ? cl : bl) : bl, u1m = py == 0 ? (top_pair ? cm : bm) : bm, u1r = py == 0 ? (top_pair ? cr : br) : br;
                    const float wy = py == 0 ? wy0 : wy1;
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        const bool left_pair = px == 0 && x > 0;     // columns (x - 1, x)
                        const f32x4 v00 = px == 0 ? (left_pair ? u0l : u0m) : u0m, v01 = px == 0 ? (left_pair ? u0m : u0r) : u0r;
                        const f32x4 v10 = px == 0 ? (left_pair ? u1l : u1m) : u1m, v11 = px == 0 ? (left_pair ? u1m : u1r) : u1r;
                        const float wx = px == 0 ? wx0 : wx1;
                        const float w00 = (1.0f - wy) * (1.0f - wx), w01 = (1.0f - wy) * wx, w10 = wy * (1.0f - wx), w11 = wy * wx;
                        const f32x4 v = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
                        res[py][px] = src == 0 ? v : res[py][px] + v;
                    }
                }
            }
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    f32x4 v = res[py][px];
                    if (bias) v += bias[c4];   // (the interpolation weights add up to one)
                    (py == 0 ? o0 : o1)[(2 * x + px) * C4 + c4] = v;
                }
        }
    }
}

// out = a + b + bias[c]
__global__ __launch_bounds__(256) void k_add_bias(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, const f32x4 *__restrict__ bias,
                                                  int C4, size_t total4, f32x4 *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        f32x4 v = a[i] + b[i];
        if (bias) v += bias[i % C4];
        out[i] = v;
    }
}

// LinearNoiseLayer + stored-statistics batch norm as ONE (sample, channel) affine (normalization.py:21-47, :170-184):
//   scale[b][c] = rsqrt(var[c] + eps) * (1 + <noise[b], Wg[c]>),   shift[b][c] = mean[c] * scale[b][c] - <noise[b], Wb[c]> - pend[c] * scale[b][c]
// noise (B, K), Wg / Wb (C, K); pend: the bias of the convolution in front, not yet added to the normalised tensor (may be null).
__global__ __launch_bounds__(256) void k_noise_affine(const float *__restrict__ noise, const float *__restrict__ wg, const float *__restrict__ wb,
                                                      const float *__restrict__ mean, const float *__restrict__ var, const float *__restrict__ pend,
                                                      float eps, int B, int C, int K, float *__restrict__ scale, float *__restrict__ shift)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    float g = 0.f, h = 0.f;
    for (int k = 0; k < K; ++k) {
        const float n = noise[b * K + k];
        g = __builtin_fmaf(n, wg[c * K + k], g);
        h = __builtin_fmaf(n, wb[c * K + k], h);
    }
    const float sc = (1.0f / __builtin_sqrtf(var[c] + eps)) * (1.0f + g);
    float sh = mean[c] * sc - h;
    if (pend) sh -= pend[c] * sc;
    scale[i] = sc;
    shift[i] = sh;
}

// The decoder's input: cat((x, 1 - background_mask), 1) of an NCHW RGB image and a (B, H, W) bool mask, straight into the NHWC
// (B, H, W, 4) tensor the first block reads (torch: a cat and a strided layout copy of a 4-channel tensor, 0.28 ms per 16 views).
__global__ __launch_bounds__(256) void k_cat_mask_nhwc(const float *__restrict__ x, const unsigned char *__restrict__ bg, size_t HW, size_t total,
                                                       f32x4 *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / HW, p = i - b * HW;
        const float *xb = x + b * 3 * HW + p;
        const f32x4 v = {xb[0], xb[HW], xb[2 * HW], bg[i] ? 0.0f : 1.0f};
        out[i] = v;
    }
}

unsigned grid_for(size_t total4) { return (unsigned)std::min<size_t>((total4 + 255) / 256, 256 * 32); }

}  // namespace

extern "C" {

int ps_affine_relu_nhwc_f32(const float *x, const float *scale, const float *shift, int B, int HW, int C, float *y, void *stream)
{
    PS_REQUIRE(x && scale && shift && y, "affine_relu: null pointer");
    PS_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0, "affine_relu: B, HW > 0 and C a positive multiple of 4 required (C = %d)", C);
    const size_t per_frame4 = (size_t)HW * (C / 4), total4 = per_frame4 * B;
    hipLaunchKernelGGL(k_affine_relu, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)x,
                       (const f32x4 *)scale, (const f32x4 *)shift, per_frame4, C / 4, total4, (f32x4 *)y);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_pool_add_post_nhwc_f32(const float *a, const float *b, const float *bias, const float *post, int B, int H, int W, int C, float *out,
                              void *stream)
{
    PS_REQUIRE(a && out, "pool_add: null pointer");
    PS_REQUIRE(B > 0 && H > 1 && W > 1 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0,
               "pool_add: even H, W and C a multiple of 4 required (H = %d, W = %d, C = %d)", H, W, C);
    const size_t total4 = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(k_pool_add, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)a, (const f32x4 *)b,
                       (const f32x4 *)bias, (const f32x4 *)post, H, W, C / 4, total4, (f32x4 *)out);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_pool_add_nhwc_f32(const float *a, const float *b, const float *bias, int B, int H, int W, int C, float *out, void *stream)
{
    return ps_pool_add_post_nhwc_f32(a, b, bias, nullptr, B, H, W, C, out, stream);
}

int ps_upsample_add_nhwc_f32(const float *a, const float *b, const float *bias, int B, int H, int W, int C, float *out, void *stream)
{
    PS_REQUIRE(a && out, "upsample_add: null pointer");
    PS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample_add: C a positive multiple of 4 required (C = %d)", C);
    PS_REQUIRE((size_t)B * H < ((size_t)1 << 31) && (size_t)2 * W * (C / 4) < ((size_t)1 << 31), "upsample_add: frame too large");
    const unsigned rows = (unsigned)B * (unsigned)H;
    hipLaunchKernelGGL(k_upsample_add, dim3(std::min(rows, 256u * 64u)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)a, (const f32x4 *)b,
                       (const f32x4 *)bias, H, W, C / 4, rows, (f32x4 *)out);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_add_bias_nhwc_f32(const float *a, const float *b, const float *bias, int B, int HW, int C, float *out, void *stream)
{
    PS_REQUIRE(a && b && out, "add_bias: null pointer");
    PS_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0, "add_bias: C a positive multiple of 4 required (C = %d)", C);
    const size_t total4 = (size_t)B * HW * (C / 4);
    hipLaunchKernelGGL(k_add_bias, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)a, (const f32x4 *)b,
                       (const f32x4 *)bias, C / 4, total4, (f32x4 *)out);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_cat_mask_nhwc_f32(const float *x, const unsigned char *background_mask, int B, int H, int W, float *out, void *stream)
{
    PS_REQUIRE(x && background_mask && out, "cat_mask: null pointer");
    PS_REQUIRE(B > 0 && H > 0 && W > 0, "cat_mask: B, H, W > 0 required");
    const size_t HW = (size_t)H * W, total = HW * B;
    hipLaunchKernelGGL(k_cat_mask_nhwc, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, background_mask, HW, total, (f32x4 *)out);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_noise_affine_f32(const float *noise, const float *wg, const float *wb, const float *mean, const float *var, const float *pend,
                        float eps, int B, int C, int K, float *scale, float *shift, void *stream)
{
    PS_REQUIRE(noise && wg && wb && mean && var && scale && shift, "noise_affine: null pointer");
    PS_REQUIRE(B > 0 && C > 0 && K > 0, "noise_affine: B, C, K > 0 required");
    hipLaunchKernelGGL(k_noise_affine, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, noise, wg, wb, mean, var, pend, eps, B, C, K,
                       scale, shift);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // extern "C"
