// lmconv.hip -- locally masked convolution, the PixelSynth PixelCNN and its autoregressive loop
// for gfx950 (MI355X).
//
// Replaces, behind the C ABI of include/pixelsynth_hip.h:
//   _locally_masked_conv2d.forward      models/lmconv/locally_masked_convolution.py:11-50
//   nin / gated_resnet / PONO            models/lmconv/layers.py:20-38, 136-163, 231-243
//   concat_elu                           models/lmconv/utils.py:31-35
//   OurPixelCNN.forward                  models/lmconv/model.py:110-155
//   sample() hot loop                    models/lmconv/sample.py:54-66
//
// Design (DESIGN.md "AR path"):
//   * Activations live channels-last in per-location caches  R[node] (raw u, 80 ch),
//     E[node] = concat_elu(u) (160 ch), X[g] = concat_elu(x) inside gated resnet g (160 ch).
//   * Every masked conv / 1x1 is ONE kernel, k_gemm: out[item][o] = sum_tap sum_c W_tap[o][c] *
//     mask_tap[item] * in[neighbour_tap(item)][c].  An "item" is a (frame, location) pair.  The 9 taps
//     are split-K slots (one wave = 16 output channels x 1 tap x 16 items per MFMA tile), the weights
//     are pre-packed [tap][c/4][o][4] so both MFMA operands are 16-byte loads, the products run on
//     v_mfma_f32_16x16x4_f32 (exact fp32, fma-chain numerics).  Masked taps cost nothing but a zero store.
//   * The same kernel serves the whole-grid forward (items = F*L, the reference-faithful mode and the
//     cache build) and the incremental AR column step (items = F, location = order[f][step]).
//   * k_post_* kernels reduce the tap slots in a fixed order (deterministic), add bias and apply
//     PONO / concat-ELU / gate / residual, writing the caches the next stage reads.
//   * The AR loop replays one captured hipGraph per order position; the step index lives in device memory.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ps_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NF = 80;        // nr_filters          (models/z_buffermodel.py:63)
constexpr int NCLS = 512;     // input_channels / classes
constexpr int NNODE = 19;     // u0..u8 (up pass) + d0..d9 (down pass)
constexpr int NGATED = 14;
constexpr int MAX_TAPS = 10;  // 9 conv taps + 1 nin_skip slot

struct GemmTap {
    const float *in;   // channels-last input [F][L][ld]
    const float *w;    // packed weights of this tap [Cin/4][Co_pad][4]
    int dr, dc;        // neighbour offset (already times dilation)
    int mask_row;      // row of the (F,9,L) mask, -1 = unmasked
    int ld;            // channels per location in `in`
};

// ==========================================================================================
// whole-grid mode: items = (frame, location) pairs of the full grid
// ==========================================================================================
struct GemmArgs {
    GemmTap tap[MAX_TAPS];
    int ntaps, Cin, Co_pad, H, W, L, nitems, tiles_per_block;
    const float *mask;
    size_t mask_fstride;
    float *partial;  // [ntaps][nitems][Co_pad]
};

// 5 channel groups (80 input channels) of one tap: all ten 16-byte operand loads are issued before the
// 20 MFMAs; even groups accumulate into acc0, odd groups into acc1 (two independent chains).  Both the
// whole-grid and the column kernels go through this function, so their summation order is identical.
__device__ __forceinline__ void mfma_chunk5(const f32x4 (&av)[5], const f32x4 (&bv)[5], f32x4 &acc0, f32x4 &acc1)
{
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        f32x4 &acc = (j & 1) ? acc1 : acc0;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc, 0, 0, 0);
    }
}

// grid (Co_pad/16, ntaps, item blocks), one wave per block
__global__ __launch_bounds__(64) void k_gemm(GemmArgs a)
{
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    const int o0 = blockIdx.x * 16;
    const GemmTap tp = a.tap[blockIdx.y];
    const int ngroups = a.Cin >> 4;
    const float *wbase = tp.w + ((size_t)kk * a.Co_pad + o0 + i) * 4;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int tt = 0; tt < a.tiles_per_block; ++tt) {
        const int tile = blockIdx.z * a.tiles_per_block + tt;
        if (tile * 16 >= a.nitems) break;
        const int item = tile * 16 + i;
        const bool valid = item < a.nitems;
        const float *src = nullptr;
        float mv = 0.0f;
        if (valid) {
            const int f = item / a.L, q = item - f * a.L;
            const int r = q / a.W, c = q - r * a.W;
            const int rr = r + tp.dr, cc = c + tp.dc;
            if (rr >= 0 && rr < a.H && cc >= 0 && cc < a.W) {
                mv = tp.mask_row >= 0 ? a.mask[(size_t)f * a.mask_fstride + (size_t)tp.mask_row * a.L + q] : 1.0f;
                src = tp.in + ((size_t)f * a.L + rr * a.W + cc) * tp.ld + 4 * kk;
            }
        }
        const bool live = valid && mv != 0.0f;
        f32x4 acc0 = zero, acc1 = zero;
        if (__any(live)) {
            int g = 0;
            for (; g + 5 <= ngroups; g += 5) {
                f32x4 av[5], bv[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    av[j] = *(const f32x4 *)(wbase + (size_t)(g + j) * 16 * a.Co_pad);
                    bv[j] = live ? *(const f32x4 *)(src + 16 * (g + j)) * mv : zero;
                }
                mfma_chunk5(av, bv, acc0, acc1);
            }
            for (; g < ngroups; ++g) {  // ragged channel counts of the generic lmconv entry point only
                const f32x4 av = *(const f32x4 *)(wbase + (size_t)g * 16 * a.Co_pad);
                const f32x4 bv = live ? *(const f32x4 *)(src + 16 * g) * mv : zero;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc0, 0, 0, 0);
            }
        }
        // D: row (output channel) = kk*4 + reg, col (item) = i
        if (valid)
            *(f32x4 *)(a.partial + ((size_t)blockIdx.y * a.nitems + item) * a.Co_pad + o0 + kk * 4) = acc0 + acc1;
    }
}

// ------------------------------------------------------------------------------------------
// per-item post ops, shared by the whole-grid kernels and the column-step prologue.  An item is
// handled by 16 lanes; lane `sub` owns channels sub + 16*k, k = 0..4 (NF = 80).  Every reduction uses
// the same association order in both modes, so column steps and whole-grid passes agree bit for bit.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : (expf(x) - 1.0f); }

__device__ __forceinline__ float sum16(float v)
{
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// PONO over the NF channels of one item (models/lmconv/layers.py:231-236), unbiased variance, eps 1e-5
__device__ __forceinline__ void pono16x5(float (&v)[5])
{
    const float mean = sum16((((v[0] + v[1]) + v[2]) + v[3]) + v[4]) / (float)NF;
    float d[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) d[k] = v[k] - mean;
    const float ss = sum16((((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3]) + d[4] * d[4]);
    const float sd = sqrtf(ss / (float)(NF - 1) + 1e-5f);
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = d[k] / sd;
}

enum { POST_CONVIN = 0, POST_GATE = 1, POST_DIL = 2 };

// KIND = POST_CONVIN: out = PONO(sum conv_input + b) [+ nin_skip + b2]            (layers.py:153-156)
//        POST_GATE:   out = rin + PONO(p) * sigmoid(g), (p,g) = sum conv_out + b   (layers.py:159-163)
//        POST_DIL:    out = PONO(sum dilated conv + b)                             (model.py:138-140,148-150)
template <int KIND>
__device__ __forceinline__ void post_item(const float *__restrict__ P, int nitems, int item, int Co_pad,
                                          const float *__restrict__ bias, const float *__restrict__ bias2,
                                          bool has_skip, const float *__restrict__ rin, int sub, float (&out)[5])
{
    float v[5], g[5];
    float pv[9][5], pg[9][5];
    float sk[5];
#pragma unroll
    for (int s = 0; s < 9; ++s) {  // every load is issued before the first add (one L2 round trip, not nine)
        const float *row = P + ((size_t)s * nitems + item) * Co_pad;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            pv[s][k] = row[sub + 16 * k];
            if (KIND == POST_GATE) pg[s][k] = row[sub + 16 * k + NF];
        }
    }
    if (KIND == POST_CONVIN && has_skip) {
#pragma unroll
        for (int k = 0; k < 5; ++k) sk[k] = P[((size_t)9 * nitems + item) * Co_pad + sub + 16 * k] + bias2[sub + 16 * k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = sub + 16 * k;
        v[k] = bias[c];
        if (KIND == POST_GATE) g[k] = bias[c + NF];
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            v[k] += pv[s][k];
            if (KIND == POST_GATE) g[k] += pg[s][k];
        }
    }
    pono16x5(v);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = sub + 16 * k;
        if (KIND == POST_CONVIN) {
            if (has_skip) v[k] += sk[k];
            out[k] = v[k];
        } else if (KIND == POST_GATE) {
            out[k] = rin[c] + v[k] * (1.0f / (1.0f + expf(-g[k])));
        } else {
            out[k] = v[k];
        }
    }
}

// u_init on one-hot input as a gather, type-A mask (model.py:132):
//   y[o] = b[o] + sum_t m_t * (W[t][512][o] + W[t][code(nbr_t)][o]) ; then norm_init (PONO)
__device__ __forceinline__ void uinit_item(const int32_t *__restrict__ codes_f, const float *mA /*9 values*/,
                                           const float *__restrict__ w, const float *__restrict__ bias, int q, int H,
                                           int W, int sub, float (&out)[5])
{
    const int r = q / W, c0 = q - r * W;
    float v[5];
    int code[9];
    float mv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int rr = r + t / 3 - 1, cc = c0 + t % 3 - 1;
        const bool in = rr >= 0 && rr < H && cc >= 0 && cc < W;
        mv[t] = in ? mA[t] : 0.0f;
        code[t] = (in && mv[t] != 0.0f) ? codes_f[rr * W + cc] : -1;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = bias[sub + 16 * k];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (mv[t] == 0.0f) continue;
        const float *w1 = w + ((size_t)t * (NCLS + 1) + NCLS) * NF;
        const float *wc = w + ((size_t)t * (NCLS + 1) + (code[t] >= 0 ? code[t] : 0)) * NF;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float x = w1[sub + 16 * k];
            if (code[t] >= 0) x += wc[sub + 16 * k];
            v[k] += mv[t] * x;
        }
    }
    pono16x5(v);
#pragma unroll
    for (int k = 0; k < 5; ++k) out[k] = v[k];
}

__device__ __forceinline__ void store_raw_celu(float *R, float *E, size_t loc, int sub, const float (&u)[5])
{
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = sub + 16 * k;
        R[loc * NF + c] = u[k];
        E[loc * (2 * NF) + c] = elu1(u[k]);
        E[loc * (2 * NF) + NF + c] = elu1(-u[k]);
    }
}

struct PostArgs {
    const float *partial;
    int nitems, Co_pad, L, has_skip;
    const float *bias, *bias2;
    const float *Rin;
    float *Rout, *Eout, *Xout;
};

// whole-grid post op: 16 items per 256-thread block
template <int KIND>
__global__ __launch_bounds__(256) void k_post_grid(PostArgs a)
{
    const int item = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (item >= a.nitems) return;  // whole 16-lane groups leave together
    const size_t loc = item;       // item = f*L + q
    float out[5];
    post_item<KIND>(a.partial, a.nitems, item, a.Co_pad, a.bias, a.bias2, a.has_skip != 0,
                    KIND == POST_GATE ? a.Rin + loc * NF : nullptr, sub, out);
    if (KIND == POST_CONVIN) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int c = sub + 16 * k;
            a.Xout[loc * (2 * NF) + c] = elu1(out[k]);
            a.Xout[loc * (2 * NF) + NF + c] = elu1(-out[k]);
        }
    } else {
        store_raw_celu(a.Rout, a.Eout, loc, sub, out);
    }
}

struct UinitArgs {
    const int32_t *codes;  // (F,L), -1 = all-zero input
    const float *mask;     // mask_init (F,9,L)
    const float *w;        // [9][513][NF]
    const float *bias;
    float *Rout, *Eout;
    int H, W, L, nitems;
};

__global__ __launch_bounds__(256) void k_uinit_grid(UinitArgs a)
{
    const int item = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (item >= a.nitems) return;
    const int f = item / a.L, q = item - f * a.L;
    float mA[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) mA[t] = a.mask[((size_t)f * 9 + t) * a.L + q];
    float u[5];
    uinit_item(a.codes + (size_t)f * a.L, mA, a.w, a.bias, q, a.H, a.W, sub, u);
    store_raw_celu(a.Rout, a.Eout, (size_t)item, sub, u);
}

// logits = nin_out partial + bias; nchw: (F,512,H,W) like the reference, else (nitems,512)
__global__ __launch_bounds__(256) void k_logits_grid(const float *partial, const float *bias, int nitems, int L,
                                                     int nchw, float *logits)
{
    const int item = blockIdx.x;
    const int f = item / L, q = item - f * L;
    for (int o = threadIdx.x; o < NCLS; o += 256) {
        const float v = bias[o] + partial[(size_t)item * NCLS + o];
        if (nchw) logits[((size_t)f * NCLS + o) * L + q] = v;
        else logits[(size_t)item * NCLS + o] = v;
    }
}

// ==========================================================================================
// column mode: one location per frame per order position (the incremental AR step)
// ==========================================================================================
struct StepCtx {
    int step, q;
    float m[3][9];  // mask values of location q: [0] type A dil 1, [1] type B dil 1, [2] type B dil 2
};

struct CtxArgs {
    StepCtx *ctx;
    const int32_t *order;
    const float *mask[3];
    int F, L;
};

__device__ __forceinline__ void ctx_fill(const CtxArgs &a, int f, int step, int t /*thread 0..31*/)
{
    if (step >= a.L) {
        if (t == 0) a.ctx[f].step = step;
        return;
    }
    const int q = a.order[(size_t)f * a.L + step];
    if (t < 27) a.ctx[f].m[t / 9][t % 9] = a.mask[t / 9][((size_t)f * 9 + t % 9) * a.L + q];
    if (t == 27) { a.ctx[f].step = step; a.ctx[f].q = q; }
}

__global__ __launch_bounds__(32) void k_ctx_init(CtxArgs a, int step) { ctx_fill(a, blockIdx.x, step, threadIdx.x); }

enum { PRO_UINIT = 0, PRO_CONVIN = 1, PRO_GATE = 2, PRO_DIL = 3 };
enum { IN_CELU = 0, IN_RAW = 1, IN_ELU = 2 };

struct ColArgs {
    GemmTap tap[MAX_TAPS];
    int ntaps, Cin, Co_pad, H, W, L, F, center_tap, mask_kind;
    float *partial;            // this stage: [ntaps][F][Co_pad]
    const StepCtx *ctx;
    // prologue: the post op of the PREVIOUS stage, evaluated by the centre-tap blocks
    const float *prev_partial;
    int prev_Co_pad, prev_has_skip;
    const float *prev_bias, *prev_bias2;
    const float *Rin;
    float *Rout, *Eout, *Xout;  // caches written by the (blockIdx.x == 0) centre-tap block
    const int32_t *codes;       // PRO_UINIT
    const float *uinit_w, *uinit_b;
};

constexpr int SIN_LD = 2 * NF + 4;

// grid (ceil(Co_pad/64), ntaps, ceil(F/16)); 4 waves = 4 output-channel tiles sharing one tap.
// Non-centre taps read finished columns of earlier order positions from the caches; the centre tap is
// the current location, whose input is produced here from the previous stage's tap slots.
template <int PRO, int IN>
__global__ __launch_bounds__(256) void k_col(ColArgs a)
{
    constexpr int NG = (IN == IN_CELU) ? 10 : 5;  // 16-channel groups of the input: 160 or 80 channels
    __shared__ __attribute__((aligned(16))) float sIn[16][SIN_LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tapi = blockIdx.y;
    const GemmTap tp = a.tap[tapi];
    const int f0 = blockIdx.z * 16;
    const bool center = tapi == a.center_tap;
    const int cot = blockIdx.x * 4 + wave;
    const bool has_tile = cot * 16 < a.Co_pad;
    const int o0 = cot * 16, i = lane & 15, kk = lane >> 4;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};

    // (1) the weight operands do not depend on anything computed this step: get them in flight first
    f32x4 av[NG];
    if (has_tile) {
        const float *wbase = tp.w + ((size_t)kk * a.Co_pad + o0 + i) * 4;
#pragma unroll
        for (int g = 0; g < NG; ++g) av[g] = *(const f32x4 *)(wbase + (size_t)g * 16 * a.Co_pad);
    }

    // (2) centre tap: the current location's input = post op of the previous stage (16 lanes per item)
    if (center) {
        const int jl = tid >> 4, sub = tid & 15;
        const int f = f0 + jl;
        if (f < a.F) {
            const int q = a.ctx[f].q;
            const size_t loc = (size_t)f * a.L + q;
            float u[5];
            if (PRO == PRO_UINIT) {
                float mA[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) mA[t] = a.ctx[f].m[0][t];
                uinit_item(a.codes + (size_t)f * a.L, mA, a.uinit_w, a.uinit_b, q, a.H, a.W, sub, u);
            } else if (PRO == PRO_CONVIN) {
                post_item<POST_CONVIN>(a.prev_partial, a.F, f, a.prev_Co_pad, a.prev_bias, a.prev_bias2,
                                       a.prev_has_skip != 0, nullptr, sub, u);
            } else if (PRO == PRO_GATE) {
                post_item<POST_GATE>(a.prev_partial, a.F, f, a.prev_Co_pad, a.prev_bias, nullptr, false,
                                     a.Rin + loc * NF, sub, u);
            } else {
                post_item<POST_DIL>(a.prev_partial, a.F, f, a.prev_Co_pad, a.prev_bias, nullptr, false, nullptr, sub, u);
            }
            const bool writer = blockIdx.x == 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int c = sub + 16 * k;
                const float ep = elu1(u[k]), en = elu1(-u[k]);
                if (IN == IN_CELU) { sIn[jl][c] = ep; sIn[jl][NF + c] = en; }
                else if (IN == IN_RAW) sIn[jl][c] = u[k];
                else sIn[jl][c] = ep;
                if (writer) {
                    if (PRO == PRO_CONVIN) {
                        a.Xout[loc * (2 * NF) + c] = ep;
                        a.Xout[loc * (2 * NF) + NF + c] = en;
                    } else {
                        a.Rout[loc * NF + c] = u[k];
                        a.Eout[loc * (2 * NF) + c] = ep;
                        a.Eout[loc * (2 * NF) + NF + c] = en;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (!has_tile) return;

    // (3) the activation operands: LDS for the centre tap, finished columns in the caches otherwise
    const int f = f0 + i;
    const bool valid = f < a.F;
    const float *src = nullptr;
    float mv = 0.0f;
    if (valid) {
        const int q = a.ctx[f].q;
        const int r = q / a.W, c = q - r * a.W;
        const int rr = r + tp.dr, cc = c + tp.dc;
        if (rr >= 0 && rr < a.H && cc >= 0 && cc < a.W) {
            mv = tp.mask_row >= 0 ? a.ctx[f].m[a.mask_kind][tp.mask_row] : 1.0f;
            src = tp.in + ((size_t)f * a.L + rr * a.W + cc) * tp.ld + 4 * kk;
        }
    }
    const bool live = valid && mv != 0.0f;
    f32x4 acc0 = zero, acc1 = zero;
    if (__any(live)) {
        f32x4 bv[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            bv[g] = zero;
            if (live) {
                if (center) bv[g] = *(const f32x4 *)(&sIn[i][16 * g + 4 * kk]) * mv;
                else bv[g] = *(const f32x4 *)(src + 16 * g) * mv;
            }
        }
        // (4) two independent accumulation chains, same order as the whole-grid kernel
#pragma unroll
        for (int g0 = 0; g0 < NG; g0 += 5) {
            const f32x4 (&a5)[5] = *reinterpret_cast<const f32x4 (*)[5]>(&av[g0]);
            const f32x4 (&b5)[5] = *reinterpret_cast<const f32x4 (*)[5]>(&bv[g0]);
            mfma_chunk5(a5, b5, acc0, acc1);
        }
    }
    if (valid) *(f32x4 *)(a.partial + ((size_t)tapi * a.F + f) * a.Co_pad + o0 + kk * 4) = acc0 + acc1;
}

// ------------------------------------------------------------------------------------------
// end of an order position: logits = nin_out + bias, categorical draw (models/lmconv/sample.py:60-66:
// softmax(logits/T), one draw, one-hot write), then the context of the next position.
// ------------------------------------------------------------------------------------------
struct FinishArgs {
    const float *partial;     // nin_out slot [F][512]
    const float *bias;
    CtxArgs cx;
    int32_t *codes;           // (F,L) or null (logits only)
    const uint8_t *region;    // (F,L) by location
    const int32_t *forced;    // (F,L) by location or null
    const float *uniforms;    // (F,L) by location or null
    float *out_logits;        // (F,L,512) by location or null
    float *step_logits;       // (F,512) or null
    float temperature;
    int advance;              // 1: write the context of step+1
};

__global__ __launch_bounds__(512) void k_finish(FinishArgs a)
{
    __shared__ float sh[NCLS];
    __shared__ float red[8];
    const int f = blockIdx.x, o = threadIdx.x, L = a.cx.L;
    const int step = a.cx.ctx[f].step, q = a.cx.ctx[f].q;
    const size_t loc = (size_t)f * L + q;
    const float lg = a.partial[(size_t)f * NCLS + o] + a.bias[o];
    if (a.out_logits) a.out_logits[loc * NCLS + o] = lg;
    if (a.step_logits) a.step_logits[(size_t)f * NCLS + o] = lg;
    const bool draw = a.codes && a.region[loc];
    if (draw && a.forced) {
        if (o == 0) a.codes[loc] = a.forced[loc];
    } else if (draw) {
        const float x = lg / a.temperature;
        float m = x;
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
        if ((o & 63) == 0) red[o >> 6] = m;
        __syncthreads();
        m = red[0];
        for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
        sh[o] = expf(x - m);
        __syncthreads();
        for (int off = 1; off < NCLS; off <<= 1) {  // inclusive scan
            const float v = o >= off ? sh[o - off] : 0.0f;
            __syncthreads();
            sh[o] += v;
            __syncthreads();
        }
        const float target = a.uniforms[loc] * sh[NCLS - 1];
        const int cnt = __syncthreads_count(sh[o] <= target);  // classes whose cdf is <= target
        if (o == 0) a.codes[loc] = min(cnt, NCLS - 1);
    }
    __syncthreads();  // every thread has read ctx[f] before it is advanced
    if (a.advance && o < 32) ctx_fill(a.cx, f, step + 1, o);
}

__global__ void k_mask_codes(int32_t *codes, const uint8_t *region, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && region[i]) codes[i] = -1;
}

// ------------------------------------------------------------------------------------------
// generic NCHW lmconv helpers
// ------------------------------------------------------------------------------------------
__global__ void k_nchw_to_cl(const float *x, int B, int C, int Cpad, int L, float *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * L * Cpad) return;
    const int c = i % Cpad;
    const size_t bl = i / Cpad;
    const int l = bl % L;
    const int b = bl / L;
    out[i] = c < C ? x[((size_t)b * C + c) * L + l] : 0.0f;
}

// (Co,Ci,3,3) -> [9][Cpad/4][Co_pad][4]
__global__ void k_pack_conv(const float *w, int Co, int Ci, int Co_pad, int Cpad, float *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_tap = (size_t)Cpad * Co_pad;
    if (i >= 9 * per_tap) return;
    const int t = i / per_tap;
    const size_t r = i % per_tap;
    const int c4 = r / ((size_t)Co_pad * 4);
    const int o = (r / 4) % Co_pad;
    const int c = c4 * 4 + (r & 3);
    out[i] = (o < Co && c < Ci) ? w[((size_t)o * Ci + c) * 9 + t] : 0.0f;
}

__global__ void k_reduce_nchw(const float *partial, const float *bias, int B, int Co, int Co_pad, int L, float *y)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * Co * L) return;
    const int l = i % L;
    const int o = (i / L) % Co;
    const int b = i / ((size_t)L * Co);
    const size_t nitems = (size_t)B * L, item = (size_t)b * L + l;
    float v = bias ? bias[o] : 0.0f;
    for (int s = 0; s < 9; ++s) v += partial[(s * nitems + item) * Co_pad + o];
    y[i] = v;
}

// ------------------------------------------------------------------------------------------
// host: weight packing
// ------------------------------------------------------------------------------------------
inline int pad16(int v) { return (v + 15) / 16 * 16; }

// (Co,Ci,3,3) host -> [9][Ci/4][Co_pad][4]
std::vector<float> pack_conv_host(const float *w, int Co, int Ci)
{
    const int Cp = pad16(Ci), Cop = pad16(Co);
    std::vector<float> out((size_t)9 * Cp * Cop, 0.0f);
    for (int t = 0; t < 9; ++t)
        for (int c = 0; c < Ci; ++c)
            for (int o = 0; o < Co; ++o)
                out[(((size_t)t * (Cp / 4) + c / 4) * Cop + o) * 4 + (c & 3)] = w[((size_t)o * Ci + c) * 9 + t];
    return out;
}

// weight-normed Linear (Co,Ci): W = v * (g / ||v||_row)  (torch._weight_norm, layers.py:23-24) -> [Ci/4][Co_pad][4]
std::vector<float> pack_nin_host(const float *v, const float *g, int Co, int Ci)
{
    const int Cp = pad16(Ci), Cop = pad16(Co);
    std::vector<float> out((size_t)Cp * Cop, 0.0f);
    for (int o = 0; o < Co; ++o) {
        float ss = 0.0f;
        for (int c = 0; c < Ci; ++c) ss += v[(size_t)o * Ci + c] * v[(size_t)o * Ci + c];
        const float scale = g[o] / sqrtf(ss);
        for (int c = 0; c < Ci; ++c)
            out[(((size_t)(c / 4)) * Cop + o) * 4 + (c & 3)] = v[(size_t)o * Ci + c] * scale;
    }
    return out;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// the handle
// ------------------------------------------------------------------------------------------
struct ps_pixelcnn {
    int H = 0, W = 0, L = 0, maxF = 0;
    std::vector<void *> allocs;
    struct Gated {
        float *w_in, *b_in, *w_out, *b_out, *w_skip, *b_skip;
        int node_in, node_skip, node_out;
    } gated[NGATED];
    struct Dil { float *w, *b; int node_in, node_out; } dil[4];
    float *uinit_w = nullptr, *uinit_b = nullptr, *out_w = nullptr, *out_b = nullptr;
    float *R[NNODE], *E[NNODE], *X[NGATED];
    float *partial = nullptr;                 // whole-grid tap slots [10][maxF*L][160]
    float *col_partial[2] = {nullptr, nullptr};  // column-mode tap slots, ping-pong between stages
    float *col_logits = nullptr;
    StepCtx *ctx = nullptr;
    hipStream_t stream = nullptr;   // internal stream for graph capture/replay
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipGraph_t graph = nullptr;          // step graph of the last ar_run (kept alive until replaced)
    hipGraphExec_t graph_exec = nullptr;
    bool use_graph = true;
    // bench.py profiling aid (ps_pixelcnn_time_column_step): event pair around every launch, by kernel tag
    struct ProfRec { int tag; hipEvent_t e0, e1; };
    std::vector<ProfRec> *prof = nullptr;
    double prof_gemm_flops = 0.0, prof_gemm_wbytes = 0.0;
};

namespace {

template <typename T>
int dev_alloc(ps_pixelcnn *h, T **p, size_t count)
{
    void *d = nullptr;
    PS_HIP_CHECK(hipMalloc(&d, count * sizeof(T)));
    h->allocs.push_back(d);
    *p = (T *)d;
    return PS_OK;
}

int upload(ps_pixelcnn *h, float **p, const float *src, size_t count)
{
    if (int rc = dev_alloc(h, p, count)) return rc;
    PS_HIP_CHECK(hipMemcpy(*p, src, count * sizeof(float), hipMemcpyHostToDevice));
    return PS_OK;
}

// the executable graph must outlive its launches: drain the internal stream before dropping it
void release_graph(ps_pixelcnn *h)
{
    if (!h->graph_exec && !h->graph) return;
    (void)hipStreamSynchronize(h->stream);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->graph_exec = nullptr;
    h->graph = nullptr;
}

struct Masks { const float *init, *und, *dil; };

enum { TAG_GEMM = 0, TAG_POST = 1, TAG_UINIT = 2, TAG_LOGITS = 3, TAG_FINISH = 4 };

template <typename Fn>
void timed(ps_pixelcnn *h, hipStream_t st, int tag, Fn &&launch)
{
    if (!h->prof) { launch(); return; }
    ps_pixelcnn::ProfRec r{tag, nullptr, nullptr};
    (void)hipEventCreate(&r.e0);
    (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, st);
    launch();
    (void)hipEventRecord(r.e1, st);
    h->prof->push_back(r);
}

template <typename Args>
void conv_taps(Args &a, const float *in, int ld, const float *wp, int Cin, int Co_pad, int dil)
{
    a.ntaps = 9;
    a.Cin = Cin;
    a.Co_pad = Co_pad;
    const size_t per_tap = (size_t)Cin * Co_pad;
    for (int t = 0; t < 9; ++t)
        a.tap[t] = GemmTap{in, wp + t * per_tap, (t / 3 - 1) * dil, (t % 3 - 1) * dil, t, ld};
}

// ------------------------------------------------------------------------------------------
// whole-grid evaluation (reference-faithful forward; cache build before the column steps)
// logits: null (caches only), (F,512,H,W) when nchw, else (F*L,512) by location
// ------------------------------------------------------------------------------------------
void run_grid(ps_pixelcnn *h, int F, const int32_t *codes, const Masks &m, float *logits, bool nchw, hipStream_t st)
{
    const int nitems = F * h->L;
    const int pblocks = (nitems + 15) / 16;
    auto gemm = [&](GemmArgs &a, const float *mask) {
        a.H = h->H; a.W = h->W; a.L = h->L; a.nitems = nitems;
        a.mask = mask; a.mask_fstride = (size_t)9 * h->L; a.partial = h->partial; a.tiles_per_block = 8;
        const int tiles = (nitems + 15) / 16;
        hipLaunchKernelGGL(k_gemm, dim3(a.Co_pad / 16, a.ntaps, (tiles + 7) / 8), dim3(64), 0, st, a);
    };
    {   // u_init + norm_init  (model.py:132)
        UinitArgs u{codes, m.init, h->uinit_w, h->uinit_b, h->R[0], h->E[0], h->H, h->W, h->L, nitems};
        hipLaunchKernelGGL(k_uinit_grid, dim3(pblocks), dim3(256), 0, st, u);
    }
    auto gated = [&](int g) {
        const ps_pixelcnn::Gated &G = h->gated[g];
        GemmArgs a{};
        conv_taps(a, h->E[G.node_in], 2 * NF, G.w_in, 2 * NF, NF, 1);                 // conv_input (layers.py:153)
        if (G.node_skip >= 0) {                                                         // nin_skip   (layers.py:155-156)
            a.tap[9] = GemmTap{h->E[G.node_skip], G.w_skip, 0, 0, -1, 2 * NF};
            a.ntaps = 10;
        }
        gemm(a, m.und);
        PostArgs p{h->partial, nitems, NF, h->L, G.node_skip >= 0, G.b_in, G.b_skip, nullptr, nullptr, nullptr, h->X[g]};
        hipLaunchKernelGGL(k_post_grid<POST_CONVIN>, dim3(pblocks), dim3(256), 0, st, p);
        GemmArgs b{};
        conv_taps(b, h->X[g], 2 * NF, G.w_out, 2 * NF, 2 * NF, 1);                     // conv_out   (layers.py:159)
        gemm(b, m.und);
        PostArgs q{h->partial, nitems, 2 * NF, h->L, 0, G.b_out, nullptr, h->R[G.node_in], h->R[G.node_out],
                   h->E[G.node_out], nullptr};
        hipLaunchKernelGGL(k_post_grid<POST_GATE>, dim3(pblocks), dim3(256), 0, st, q);   // gate + residual (:160-163)
    };
    auto dilated = [&](int d) {
        const ps_pixelcnn::Dil &D = h->dil[d];
        GemmArgs a{};
        conv_taps(a, h->R[D.node_in], NF, D.w, NF, NF, 2);                              // model.py:138,148
        gemm(a, m.dil);
        PostArgs p{h->partial, nitems, NF, h->L, 0, D.b, nullptr, nullptr, h->R[D.node_out], h->E[D.node_out], nullptr};
        hipLaunchKernelGGL(k_post_grid<POST_DIL>, dim3(pblocks), dim3(256), 0, st, p);
    };
    gated(0); gated(1); dilated(0); gated(2); gated(3); dilated(1); gated(4); gated(5);     // up pass
    gated(6); gated(7); dilated(2); gated(8); gated(9); gated(10); dilated(3);              // down pass
    gated(11); gated(12); gated(13);
    if (!logits) return;
    GemmArgs a{};                                                                         // nin_out(elu(u)) model.py:153
    a.ntaps = 1; a.Cin = NF; a.Co_pad = NCLS;
    a.tap[0] = GemmTap{h->E[NNODE - 1], h->out_w, 0, 0, -1, 2 * NF};
    gemm(a, nullptr);
    hipLaunchKernelGGL(k_logits_grid, dim3(nitems), dim3(256), 0, st, h->partial, h->out_b, nitems, h->L, nchw ? 1 : 0,
                       logits);
}

// ------------------------------------------------------------------------------------------
// one column step: 33 k_col launches (each evaluates the previous stage's post op in its centre-tap
// blocks) + k_finish.  h->ctx must describe the current order position.
// ------------------------------------------------------------------------------------------
struct Prev {
    int pro = PRO_UINIT;
    const float *partial = nullptr, *bias = nullptr, *bias2 = nullptr, *Rin = nullptr;
    int Co_pad = 0, has_skip = 0;
    float *Rout = nullptr, *Eout = nullptr, *Xout = nullptr;
};

template <int IN>
void launch_col(ps_pixelcnn *h, hipStream_t st, int pro, const ColArgs &a, dim3 grid)
{
    timed(h, st, TAG_GEMM, [&]() {
        switch (pro) {
        case PRO_UINIT: hipLaunchKernelGGL((k_col<PRO_UINIT, IN>), grid, dim3(256), 0, st, a); break;
        case PRO_CONVIN: hipLaunchKernelGGL((k_col<PRO_CONVIN, IN>), grid, dim3(256), 0, st, a); break;
        case PRO_GATE: hipLaunchKernelGGL((k_col<PRO_GATE, IN>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((k_col<PRO_DIL, IN>), grid, dim3(256), 0, st, a); break;
        }
    });
}

void run_column(ps_pixelcnn *h, int F, const int32_t *codes, FinishArgs fin, hipStream_t st)
{
    int stage = 0;
    Prev prev;  // stage 0's prologue is u_init itself
    prev.Rout = h->R[0];
    prev.Eout = h->E[0];
    auto stage_launch = [&](ColArgs &a, int in_form, int mask_kind, int center) {
        a.H = h->H; a.W = h->W; a.L = h->L; a.F = F;
        a.center_tap = center; a.mask_kind = mask_kind;
        a.partial = h->col_partial[stage & 1];
        a.ctx = h->ctx;
        a.prev_partial = prev.partial; a.prev_Co_pad = prev.Co_pad; a.prev_has_skip = prev.has_skip;
        a.prev_bias = prev.bias; a.prev_bias2 = prev.bias2; a.Rin = prev.Rin;
        a.Rout = prev.Rout; a.Eout = prev.Eout; a.Xout = prev.Xout;
        a.codes = codes; a.uinit_w = h->uinit_w; a.uinit_b = h->uinit_b;
        const dim3 grid((a.Co_pad + 63) / 64, a.ntaps, (F + 15) / 16);
        if (h->prof)
            for (int t = 0; t < a.ntaps; ++t) {
                h->prof_gemm_flops += 2.0 * a.Co_pad * a.Cin * F;
                h->prof_gemm_wbytes += 4.0 * a.Co_pad * a.Cin;
            }
        if (in_form == IN_CELU) launch_col<IN_CELU>(h, st, prev.pro, a, grid);
        else if (in_form == IN_RAW) launch_col<IN_RAW>(h, st, prev.pro, a, grid);
        else launch_col<IN_ELU>(h, st, prev.pro, a, grid);
        prev = Prev();
        prev.partial = a.partial;
        prev.Co_pad = a.Co_pad;
        ++stage;
    };
    auto gated = [&](int g) {
        const ps_pixelcnn::Gated &G = h->gated[g];
        ColArgs a{};
        conv_taps(a, h->E[G.node_in], 2 * NF, G.w_in, 2 * NF, NF, 1);
        if (G.node_skip >= 0) {
            a.tap[9] = GemmTap{h->E[G.node_skip], G.w_skip, 0, 0, -1, 2 * NF};
            a.ntaps = 10;
        }
        stage_launch(a, IN_CELU, 1, 4);
        prev.pro = PRO_CONVIN; prev.bias = G.b_in; prev.bias2 = G.b_skip; prev.has_skip = G.node_skip >= 0;
        prev.Xout = h->X[g];
        ColArgs b{};
        conv_taps(b, h->X[g], 2 * NF, G.w_out, 2 * NF, 2 * NF, 1);
        stage_launch(b, IN_CELU, 1, 4);
        prev.pro = PRO_GATE; prev.bias = G.b_out; prev.Rin = h->R[G.node_in];
        prev.Rout = h->R[G.node_out]; prev.Eout = h->E[G.node_out];
    };
    auto dilated = [&](int d) {
        const ps_pixelcnn::Dil &D = h->dil[d];
        ColArgs a{};
        conv_taps(a, h->R[D.node_in], NF, D.w, NF, NF, 2);
        stage_launch(a, IN_RAW, 2, 4);
        prev.pro = PRO_DIL; prev.bias = D.b; prev.Rout = h->R[D.node_out]; prev.Eout = h->E[D.node_out];
    };
    gated(0); gated(1); dilated(0); gated(2); gated(3); dilated(1); gated(4); gated(5);
    gated(6); gated(7); dilated(2); gated(8); gated(9); gated(10); dilated(3);
    gated(11); gated(12); gated(13);
    ColArgs a{};  // nin_out(elu(u)): one unmasked centre tap; its prologue is the last gate
    a.ntaps = 1; a.Cin = NF; a.Co_pad = NCLS;
    a.tap[0] = GemmTap{h->E[NNODE - 1], h->out_w, 0, 0, -1, 2 * NF};
    stage_launch(a, IN_ELU, 1, 0);
    fin.partial = prev.partial;
    fin.bias = h->out_b;
    timed(h, st, TAG_FINISH, [&]() { hipLaunchKernelGGL(k_finish, dim3(F), dim3(512), 0, st, fin); });
}

CtxArgs make_ctx_args(ps_pixelcnn *h, const int32_t *order, const Masks &m, int F)
{
    CtxArgs cx{};
    cx.ctx = h->ctx; cx.order = order;
    cx.mask[0] = m.init; cx.mask[1] = m.und; cx.mask[2] = m.dil;
    cx.F = F; cx.L = h->L;
    return cx;
}

int check_handle(ps_pixelcnn *h, int F)
{
    PS_REQUIRE(h, "pixelcnn: null handle");
    PS_REQUIRE(F > 0 && F <= h->maxF, "pixelcnn: F=%d outside [1, max_frames=%d]", F, h->maxF);
    return PS_OK;
}

}  // namespace

extern "C" {

int ps_pixelcnn_create(const float *const *params, int n_params, int H, int W, int max_frames, ps_pixelcnn **out)
{
    PS_REQUIRE(params && out, "pixelcnn_create: null pointer");
    PS_REQUIRE(n_params == PS_PIXELCNN_NUM_PARAMS, "pixelcnn_create: expected %d tensors, got %d",
               PS_PIXELCNN_NUM_PARAMS, n_params);
    PS_REQUIRE(H > 0 && W > 0 && max_frames > 0, "pixelcnn_create: bad sizes");
    for (int i = 0; i < n_params; ++i) PS_REQUIRE(params[i], "pixelcnn_create: tensor %d is null", i);
    ps_pixelcnn *h = new ps_pixelcnn();
    h->H = H; h->W = W; h->L = H * W; h->maxF = max_frames;
    const char *env = getenv("PS_AR_GRAPH");
    h->use_graph = !(env && env[0] == '0');
    int rc = PS_OK;
    auto fail_out = [&](int code) { ps_pixelcnn_destroy(h); return code; };

    // ---- schedule: node numbering u0..u8 = 0..8, d0..d9 = 9..18 (model.py:132-151)
    const int g_in[NGATED] = {0, 1, 3, 4, 6, 7, 8, 9, 11, 12, 13, 15, 16, 17};
    const int g_out[NGATED] = {1, 2, 4, 5, 7, 8, 9, 10, 12, 13, 14, 16, 17, 18};
    const int g_skip[NGATED] = {-1, -1, -1, -1, -1, -1, 7, 6, 5, 4, 3, 2, 1, 0};
    // parameter indices in reference state_dict order
    auto down_base = [](int k) { return k * 7; };            // 8 down blocks x 7 tensors
    auto up_base = [](int k) { return 56 + k * 4; };         // 6 up blocks x 4 tensors
    for (int g = 0; g < NGATED; ++g) {
        ps_pixelcnn::Gated &G = h->gated[g];
        G.node_in = g_in[g]; G.node_out = g_out[g]; G.node_skip = g_skip[g];
        G.w_skip = G.b_skip = nullptr;
        const float *w_in, *b_in, *w_out, *b_out;
        if (g < 6) {
            const int b = up_base(g);
            w_in = params[b]; b_in = params[b + 1]; w_out = params[b + 2]; b_out = params[b + 3];
        } else {
            const int b = down_base(g - 6);
            w_in = params[b]; b_in = params[b + 1]; w_out = params[b + 5]; b_out = params[b + 6];
            std::vector<float> ws = pack_nin_host(params[b + 4], params[b + 3], NF, 2 * NF);
            if ((rc = upload(h, &G.w_skip, ws.data(), ws.size()))) return fail_out(rc);
            if ((rc = upload(h, &G.b_skip, params[b + 2], NF))) return fail_out(rc);
        }
        std::vector<float> pi = pack_conv_host(w_in, NF, 2 * NF), po = pack_conv_host(w_out, 2 * NF, 2 * NF);
        if ((rc = upload(h, &G.w_in, pi.data(), pi.size()))) return fail_out(rc);
        if ((rc = upload(h, &G.b_in, b_in, NF))) return fail_out(rc);
        if ((rc = upload(h, &G.w_out, po.data(), po.size()))) return fail_out(rc);
        if ((rc = upload(h, &G.b_out, b_out, 2 * NF))) return fail_out(rc);
    }
    const int d_in[4] = {2, 5, 10, 14}, d_out[4] = {3, 6, 11, 15};
    for (int d = 0; d < 4; ++d) {
        const int b = 82 + d * 2;  // downsize_u_stream.{0,1}, upsize_u_stream.{0,1}
        h->dil[d].node_in = d_in[d]; h->dil[d].node_out = d_out[d];
        std::vector<float> pw = pack_conv_host(params[b], NF, NF);
        if ((rc = upload(h, &h->dil[d].w, pw.data(), pw.size()))) return fail_out(rc);
        if ((rc = upload(h, &h->dil[d].b, params[b + 1], NF))) return fail_out(rc);
    }
    {   // u_init (80,513,3,3) -> [9][513][80]
        std::vector<float> wu((size_t)9 * (NCLS + 1) * NF);
        const float *w = params[80];
        for (int t = 0; t < 9; ++t)
            for (int c = 0; c <= NCLS; ++c)
                for (int o = 0; o < NF; ++o) wu[((size_t)t * (NCLS + 1) + c) * NF + o] = w[((size_t)o * (NCLS + 1) + c) * 9 + t];
        if ((rc = upload(h, &h->uinit_w, wu.data(), wu.size()))) return fail_out(rc);
        if ((rc = upload(h, &h->uinit_b, params[81], NF))) return fail_out(rc);
        std::vector<float> wo = pack_nin_host(params[92], params[91], NCLS, NF);
        if ((rc = upload(h, &h->out_w, wo.data(), wo.size()))) return fail_out(rc);
        if ((rc = upload(h, &h->out_b, params[90], NCLS))) return fail_out(rc);
    }
    const size_t locs = (size_t)max_frames * h->L;
    for (int n = 0; n < NNODE; ++n) {
        if ((rc = dev_alloc(h, &h->R[n], locs * NF))) return fail_out(rc);
        if ((rc = dev_alloc(h, &h->E[n], locs * 2 * NF))) return fail_out(rc);
    }
    for (int g = 0; g < NGATED; ++g)
        if ((rc = dev_alloc(h, &h->X[g], locs * 2 * NF))) return fail_out(rc);
    size_t pfloats = (size_t)MAX_TAPS * locs * 2 * NF;
    if (locs * NCLS > pfloats) pfloats = locs * NCLS;
    if ((rc = dev_alloc(h, &h->partial, pfloats))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->col_logits, (size_t)max_frames * NCLS))) return fail_out(rc);
    for (int k = 0; k < 2; ++k)
        if ((rc = dev_alloc(h, &h->col_partial[k], (size_t)MAX_TAPS * max_frames * NCLS))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->ctx, (size_t)max_frames))) return fail_out(rc);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) != hipSuccess) {
        ps::fail(PS_ERR_HIP, "pixelcnn_create: stream/event creation failed");
        return fail_out(PS_ERR_HIP);
    }
    *out = h;
    return PS_OK;
}

void ps_pixelcnn_destroy(ps_pixelcnn *h)
{
    if (!h) return;
    if (h->stream) release_graph(h);
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->ev_out) (void)hipEventDestroy(h->ev_out);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int ps_pixelcnn_forward_f32(ps_pixelcnn *h, const int32_t *codes, const float *mask_init, const float *mask_undilated,
                            const float *mask_dilated, int F, float *logits, void *stream)
{
    if (int rc = check_handle(h, F)) return rc;
    PS_REQUIRE(codes && mask_init && mask_undilated && mask_dilated && logits, "pixelcnn_forward: null pointer");
    run_grid(h, F, codes, Masks{mask_init, mask_undilated, mask_dilated}, logits, true, (hipStream_t)stream);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_pixelcnn_ar_step(ps_pixelcnn *h, const int32_t *codes, const int32_t *order, const float *mask_init,
                        const float *mask_undilated, const float *mask_dilated, int F, int step, int first_step,
                        float *logits, void *stream)
{
    if (int rc = check_handle(h, F)) return rc;
    PS_REQUIRE(codes && order && mask_init && mask_undilated && mask_dilated && logits, "pixelcnn_ar_step: null pointer");
    PS_REQUIRE(step >= 0 && step < h->L && first_step >= 0 && first_step <= step, "pixelcnn_ar_step: bad step");
    hipStream_t st = (hipStream_t)stream;
    const Masks m{mask_init, mask_undilated, mask_dilated};
    if (step == first_step) run_grid(h, F, codes, m, nullptr, false, st);
    FinishArgs fin{};
    fin.cx = make_ctx_args(h, order, m, F);
    hipLaunchKernelGGL(k_ctx_init, dim3(F), dim3(32), 0, st, fin.cx, step);
    fin.step_logits = logits;
    fin.temperature = 1.0f;
    run_column(h, F, codes, fin, st);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_pixelcnn_ar_run(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region,
                       const float *mask_init, const float *mask_undilated, const float *mask_dilated,
                       const int32_t *forced, const float *uniforms, float temperature, int F, int first_step,
                       float *out_logits, void *stream)
{
    if (int rc = check_handle(h, F)) return rc;
    PS_REQUIRE(codes && order && sample_region && mask_init && mask_undilated && mask_dilated, "pixelcnn_ar_run: null pointer");
    PS_REQUIRE((forced != nullptr) != (uniforms != nullptr), "pixelcnn_ar_run: give exactly one of forced / uniforms");
    PS_REQUIRE(first_step >= 0 && first_step <= h->L, "pixelcnn_ar_run: first_step out of range");
    PS_REQUIRE(temperature > 0.0f, "pixelcnn_ar_run: temperature must be > 0");
    hipStream_t caller = (hipStream_t)stream;
    hipStream_t st = h->use_graph ? h->stream : caller;
    if (h->use_graph) {  // hand over from the caller's stream to the internal (capturable) one
        release_graph(h);
        PS_HIP_CHECK(hipEventRecord(h->ev_in, caller));
        PS_HIP_CHECK(hipStreamWaitEvent(st, h->ev_in, 0));
    }
    const Masks m{mask_init, mask_undilated, mask_dilated};
    const size_t n = (size_t)F * h->L;
    hipLaunchKernelGGL(k_mask_codes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, codes, sample_region, n);
    // whole-grid pass: exact for every location that precedes the first sampled one; with out_logits it also
    // yields their logits, by location (the walked positions are overwritten by the column steps)
    run_grid(h, F, codes, m, out_logits, false, st);
    FinishArgs fin{};
    fin.cx = make_ctx_args(h, order, m, F);
    fin.codes = codes; fin.region = sample_region; fin.forced = forced; fin.uniforms = uniforms;
    fin.out_logits = out_logits; fin.temperature = temperature; fin.advance = 1;
    hipLaunchKernelGGL(k_ctx_init, dim3(F), dim3(32), 0, st, fin.cx, first_step);
    PS_LAUNCH_CHECK();
    const int nsteps = h->L - first_step;
    if (nsteps > 0) {
        if (h->use_graph) {
            PS_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            run_column(h, F, codes, fin, st);
            PS_HIP_CHECK(hipStreamEndCapture(st, &h->graph));
            PS_HIP_CHECK(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
            for (int sidx = 0; sidx < nsteps; ++sidx) PS_HIP_CHECK(hipGraphLaunch(h->graph_exec, st));
        } else {
            for (int sidx = 0; sidx < nsteps; ++sidx) run_column(h, F, codes, fin, st);
        }
    }
    PS_LAUNCH_CHECK();
    if (h->use_graph) {
        PS_HIP_CHECK(hipEventRecord(h->ev_out, st));
        PS_HIP_CHECK(hipStreamWaitEvent(caller, h->ev_out, 0));
    }
    return PS_OK;
}

int ps_pixelcnn_time_column_step(ps_pixelcnn *h, const int32_t *codes, const int32_t *order, const float *mask_init,
                                 const float *mask_undilated, const float *mask_dilated, int F, int step, int reps,
                                 int *launches, float *total_ms, double *gemm_flops_per_step,
                                 double *gemm_weight_bytes_per_step, void *stream)
{
    if (int rc = check_handle(h, F)) return rc;
    PS_REQUIRE(codes && order && mask_init && mask_undilated && mask_dilated && launches && total_ms,
               "pixelcnn_time_column_step: null pointer");
    PS_REQUIRE(step >= 0 && step < h->L && reps > 0, "pixelcnn_time_column_step: bad step / reps");
    hipStream_t st = (hipStream_t)stream;
    std::vector<ps_pixelcnn::ProfRec> recs;
    FinishArgs fin{};
    fin.cx = make_ctx_args(h, order, Masks{mask_init, mask_undilated, mask_dilated}, F);
    fin.step_logits = h->col_logits;
    fin.temperature = 1.0f;
    hipLaunchKernelGGL(k_ctx_init, dim3(F), dim3(32), 0, st, fin.cx, step);
    run_column(h, F, codes, fin, st);  // untimed warm-up
    h->prof = &recs;
    h->prof_gemm_flops = h->prof_gemm_wbytes = 0.0;
    for (int r = 0; r < reps; ++r) run_column(h, F, codes, fin, st);
    h->prof = nullptr;
    PS_HIP_CHECK(hipStreamSynchronize(st));
    for (int t = 0; t < PS_PROF_NTAGS; ++t) { launches[t] = 0; total_ms[t] = 0.0f; }
    for (auto &r : recs) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        launches[r.tag] += 1;
        total_ms[r.tag] += ms;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    if (gemm_flops_per_step) *gemm_flops_per_step = h->prof_gemm_flops / reps;
    if (gemm_weight_bytes_per_step) *gemm_weight_bytes_per_step = h->prof_gemm_wbytes / reps;
    PS_LAUNCH_CHECK();
    return PS_OK;
}

size_t ps_lmconv_workspace_bytes(int B, int Ci, int Co, int H, int W)
{
    if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return 0;
    const size_t L = (size_t)H * W, Cp = pad16(Ci), Cop = pad16(Co);
    size_t o = 0;
    o = ps::align_up(o + (size_t)B * L * Cp * 4, 256);
    o = ps::align_up(o + 9 * Cp * Cop * 4, 256);
    o = ps::align_up(o + 9 * (size_t)B * L * Cop * 4, 256);
    return o;
}

int ps_lmconv_forward_f32(const float *x, const float *mask, size_t mask_batch_stride, const float *weight,
                          const float *bias, int B, int Ci, int Co, int H, int W, int dilation, float *y,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    PS_REQUIRE(x && mask && weight && y && workspace, "lmconv_forward: null pointer");
    PS_REQUIRE(B > 0 && Ci > 0 && Co > 0 && H > 0 && W > 0 && dilation > 0, "lmconv_forward: bad sizes");
    const size_t need = ps_lmconv_workspace_bytes(B, Ci, Co, H, W);
    if (workspace_bytes < need)
        return ps::fail(PS_ERR_WORKSPACE, "lmconv_forward: workspace %zu < required %zu bytes", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const int L = H * W, Cp = pad16(Ci), Cop = pad16(Co);
    char *ws = (char *)workspace;
    float *xcl = (float *)ws;
    size_t o = ps::align_up((size_t)B * L * Cp * 4, 256);
    float *wp = (float *)(ws + o);
    o = ps::align_up(o + (size_t)9 * Cp * Cop * 4, 256);
    float *partial = (float *)(ws + o);
    const size_t n1 = (size_t)B * L * Cp, n2 = (size_t)9 * Cp * Cop, n3 = (size_t)B * Co * L;
    hipLaunchKernelGGL(k_nchw_to_cl, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, x, B, Ci, Cp, L, xcl);
    hipLaunchKernelGGL(k_pack_conv, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, weight, Co, Ci, Cop, Cp, wp);
    GemmArgs a{};
    conv_taps(a, xcl, Cp, wp, Cp, Cop, dilation);
    a.H = H; a.W = W; a.L = L; a.nitems = B * L; a.mask = mask; a.mask_fstride = mask_batch_stride;
    a.partial = partial; a.tiles_per_block = 8;
    const int tiles = (a.nitems + 15) / 16;
    hipLaunchKernelGGL(k_gemm, dim3(Cop / 16, 9, (tiles + 7) / 8), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_reduce_nchw, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, partial, bias, B, Co, Cop, L, y);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // extern "C"
