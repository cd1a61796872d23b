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
//   * Every masked conv / 1x1 is the same product: out[item][o] = sum_tap sum_c W_tap[o][c] * mask_tap[item] *
//     in[neighbour_tap(item)][c], an "item" being a (frame, location) pair.  16 items x 16 output channels
//     form one v_mfma_f32_16x16x4_f32 tile (exact fp32, fma-chain numerics); the weights are pre-packed
//     [tap][c/4][o][4] so both MFMA operands are 16-byte loads; masked taps are skipped.
//   * The taps are grouped in split-K slots NA (taps 0..3), C (the location itself), NB (taps 5..8) and SKIP
//     (nin_skip); every consumer adds them as ((bias + NA) + C) + NB, and every kernel walks taps and
//     80-channel chunks in the same order -- so the two evaluation modes below agree bit for bit.
//   * Whole-grid mode (k_gemm + k_post_grid, items = F*L): the reference-faithful OurPixelCNN.forward and the
//     cache build for the prefix of observed locations.
//   * Column mode (the incremental AR step, items = F): two launches per order position.  k_nbr computes the
//     NA/NB slots of all 32 convs at once (they only read finished columns of earlier positions); k_chain
//     walks the 33 dependent stages inside one workgroup per 16 frames (centre-tap products on MFMA,
//     post ops one wave per frame, LDS hand-off), draws the code and writes the next position's context.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ps_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Pointers that reach a kernel through a descriptor in memory (StageDesc) are "generic" to the compiler,
// which then emits FLAT loads/stores.  FLAT ops also count on lgkmcnt, so an LDS-only barrier
// (s_waitcnt lgkmcnt(0)) would drain every weight / slot prefetch in flight.  All descriptor pointers
// are device-global memory: say so, and get global_load / global_store.
#define PS_G(T, p) ((__attribute__((address_space(1))) T *)(p))
#define PS_GC(T, p) ((const __attribute__((address_space(1))) T *)(p))

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

// Split-K slots of a masked 3x3 conv.  Every consumer adds them in this order:
//   y = ((bias + NA) + C) + NB          (+ SKIP after the norm, layers.py:155-156)
enum { SLOT_NA = 0 /* taps 0..3 */, SLOT_C = 1 /* tap 4, the location itself */, SLOT_NB = 2 /* taps 5..8 */,
       SLOT_SKIP = 3 /* nin_skip 1x1 */ };

// 5 channel groups (80 input channels) of one tap: all ten 16-byte operand loads are issued before the
// 20 MFMAs; even groups accumulate into acc0, odd groups into acc1 (two independent chains).  Every
// kernel goes through this function and walks taps / chunks in the same order, so the whole-grid
// pass and the column steps produce identical bits.
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

// ==========================================================================================
// whole-grid mode: items = (frame, location) pairs of the full grid
// ==========================================================================================
struct GemmArgs {
    GemmTap tap[MAX_TAPS];
    int slot_first[5];  // slot s covers taps [slot_first[s], slot_first[s+1])
    int nslots, Cin, Co_pad, H, W, L, nitems, tiles_per_block;
    const float *mask;
    size_t mask_fstride;
    float *partial;  // [nslots][nitems][Co_pad]
};

// grid (Co_pad/16, nslots, item blocks), one wave per block
__global__ __launch_bounds__(64) void k_gemm(GemmArgs a)
{
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    const int o0 = blockIdx.x * 16, slot = blockIdx.y;
    const int ngroups = a.Cin >> 4;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int tt = 0; tt < a.tiles_per_block; ++tt) {
        const int tile = blockIdx.z * a.tiles_per_block + tt;
        if (tile * 16 >= a.nitems) break;
        const int item = tile * 16 + i;
        const bool valid = item < a.nitems;
        int f = 0, r = 0, c = 0, q = 0;
        if (valid) {
            f = item / a.L;
            q = item - f * a.L;
            r = q / a.W;
            c = q - r * a.W;
        }
        f32x4 acc0 = zero, acc1 = zero;
        for (int t = a.slot_first[slot]; t < a.slot_first[slot + 1]; ++t) {
            const GemmTap tp = a.tap[t];
            const float *src = nullptr;
            float mv = 0.0f;
            const int rr = r + tp.dr, cc = c + tp.dc;
            if (valid && rr >= 0 && rr < a.H && cc >= 0 && cc < a.W) {
                mv = tp.mask_row >= 0 ? a.mask[(size_t)f * a.mask_fstride + (size_t)tp.mask_row * a.L + q] : 1.0f;
                src = tp.in + ((size_t)f * a.L + rr * a.W + cc) * tp.ld + 4 * kk;
            }
            const bool live = mv != 0.0f;
            if (!__any(live)) continue;  // a masked tap adds exact zeros: skipping it does not change the bits
            const float *wbase = tp.w + ((size_t)kk * a.Co_pad + o0 + i) * 4;
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
            *(f32x4 *)(a.partial + ((size_t)slot * a.nitems + item) * a.Co_pad + o0 + kk * 4) = acc0 + acc1;
    }
}

// ------------------------------------------------------------------------------------------
// per-item post ops, shared by the whole-grid kernels and the column chain.  An item (one location of one
// frame, NF = 80 channels) is handled by ONE WAVE: lane l owns channel l and, for l < 16, channel 64 + l
// (NCH = 2 slots per lane, the second one mostly empty).  Keeping the per-lane work this small is what
// bounds the serial prologue of every stage of k_chain.  Every reduction uses the same association
// order in both modes, so column steps and whole-grid passes agree bit for bit.
// ------------------------------------------------------------------------------------------
constexpr int NCH = 2;
__device__ __forceinline__ int chan(int lane, int k) { return lane + 64 * k; }
__device__ __forceinline__ bool owns(int lane, int k) { return k == 0 || lane < NF - 64; }

// Elementwise math of the post ops.  These sit on the sequential critical path of every AR order position
// (k_chain), so they use the hardware transcendental units directly (v_exp_f32 / v_rcp_f32 / v_rsq_f32,
// ~1 ulp) instead of the libm-exact sequences; the result stays ~1e-7 relative to the exact value,
// far inside the 1e-4 logit tolerance, and both evaluation modes share these functions bit for bit.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
// concat_elu of one value: (elu(x), elu(-x)) with a single exponential (utils.py:31-35)
__device__ __forceinline__ void celu_pair(float x, float &ep, float &en)
{
    const float e = fast_exp(-fabsf(x)) - 1.0f;
    ep = x > 0.0f ? x : e;
    en = x > 0.0f ? e : -x;
}
__device__ __forceinline__ float sigmoid1(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }

// all-reduce over the 64 lanes of a wave: DPP row rotations inside each 16-lane row (rotation by 8, 4, 2, 1
// pairs each lane with the same partners as an xor butterfly and a+b == b+a bitwise, so every lane of a
// row ends with identical bits), then the four row sums are read with v_readlane and added in a fixed order.
template <int N>
__device__ __forceinline__ float row_ror(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return ((r0 + r1) + r2) + r3;
}

// PONO over the NF channels of one item (models/lmconv/layers.py:231-236), unbiased variance, eps 1e-5.
// Slots a lane does not own must hold 0 on entry and hold 0 on exit.
__device__ __forceinline__ void pono_wave(float (&v)[NCH], int lane)
{
    const float mean = wave_sum(v[0] + v[1]) * (1.0f / (float)NF);
    const float d0 = v[0] - mean, d1 = owns(lane, 1) ? v[1] - mean : 0.0f;
    const float ss = wave_sum(d0 * d0 + d1 * d1);
    const float inv = __builtin_amdgcn_rsqf(ss * (1.0f / (float)(NF - 1)) + 1e-5f);
    v[0] = d0 * inv;
    v[1] = d1 * inv;
}

__device__ __forceinline__ float slot_sum(float bias, float na, float c, float nb) { return ((bias + na) + c) + nb; }

enum { POST_CONVIN = 0, POST_GATE = 1, POST_DIL = 2 };

// v (and g for the gate): conv output INCLUDING bias, already slot-summed by the caller (0 in unowned slots).
// KIND = POST_CONVIN: out = PONO(v) [+ skip]                         (layers.py:153-156)
//        POST_GATE:   out = rin + PONO(v) * sigmoid(g)                (layers.py:159-163)
//        POST_DIL:    out = PONO(v)                                   (model.py:138-140,148-150)
template <int KIND>
__device__ __forceinline__ void post_math(float (&v)[NCH], const float (&g)[NCH], const float (&skip)[NCH], bool has_skip,
                                          const float (&rin)[NCH], int lane, float (&out)[NCH])
{
    pono_wave(v, lane);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (KIND == POST_CONVIN) out[k] = has_skip ? v[k] + skip[k] : v[k];
        else if (KIND == POST_GATE) out[k] = rin[k] + v[k] * sigmoid1(g[k]);
        else out[k] = v[k];
        if (!owns(lane, k)) out[k] = 0.0f;
    }
}

// u_init on one-hot input as a gather, type-A mask (model.py:132), BEFORE norm_init:
//   y[o] = b[o] + sum_t m_t * (W[t][512][o] + W[t][code(nbr_t)][o])
// Only earlier order positions contribute (the centre of a type-A mask is 0), so in column mode this
// belongs to the neighbour kernel, not to the chain.
__device__ __forceinline__ void uinit_gather(const int32_t *__restrict__ codes_f, const float *mA /*9 values*/,
                                             const float *__restrict__ w, const float *__restrict__ bias, int q, int H,
                                             int W, int lane, float (&out)[NCH])
{
    const int r = q / W, c0 = q - r * W;
    float v[NCH];
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
    for (int k = 0; k < NCH; ++k) v[k] = owns(lane, k) ? bias[chan(lane, k)] : 0.0f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (mv[t] == 0.0f) continue;
        const float *w1 = w + ((size_t)t * (NCLS + 1) + NCLS) * NF;
        const float *wc = w + ((size_t)t * (NCLS + 1) + (code[t] >= 0 ? code[t] : 0)) * NF;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (!owns(lane, k)) continue;
            float x = w1[chan(lane, k)];
            if (code[t] >= 0) x += wc[chan(lane, k)];
            v[k] += mv[t] * x;
        }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) out[k] = v[k];
}

__device__ __forceinline__ void store_raw_celu(float *R, float *E, size_t loc, int lane, const float (&u)[NCH])
{
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (!owns(lane, k)) continue;
        const int c = chan(lane, k);
        float ep, en;
        celu_pair(u[k], ep, en);
        R[loc * NF + c] = u[k];
        E[loc * (2 * NF) + c] = ep;
        E[loc * (2 * NF) + NF + c] = en;
    }
}

struct PostArgs {
    const float *partial;  // [slots][nitems][Co_pad]
    int nitems, Co_pad, L, has_skip;
    const float *bias, *bias2;
    const float *Rin;
    float *Rout, *Eout, *Xout;
};

// whole-grid post op: one wave per item, 4 items per 256-thread block
template <int KIND>
__global__ __launch_bounds__(256) void k_post_grid(PostArgs a)
{
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= a.nitems) return;  // whole waves leave together
    const size_t loc = item;       // item = f*L + q
    const size_t ss = (size_t)a.nitems * a.Co_pad;
    const float *P = a.partial + (size_t)item * a.Co_pad;
    float v[NCH], g[NCH], skip[NCH], rin[NCH], out[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        v[k] = g[k] = skip[k] = rin[k] = 0.0f;
        if (!owns(lane, k)) continue;
        const int c = chan(lane, k);
        v[k] = slot_sum(a.bias[c], P[SLOT_NA * ss + c], P[SLOT_C * ss + c], P[SLOT_NB * ss + c]);
        if (KIND == POST_GATE) {
            g[k] = slot_sum(a.bias[c + NF], P[SLOT_NA * ss + c + NF], P[SLOT_C * ss + c + NF], P[SLOT_NB * ss + c + NF]);
            rin[k] = a.Rin[loc * NF + c];
        }
        if (KIND == POST_CONVIN && a.has_skip) skip[k] = P[SLOT_SKIP * ss + c] + a.bias2[c];
    }
    post_math<KIND>(v, g, skip, a.has_skip != 0, rin, lane, out);
    if (KIND == POST_CONVIN) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (!owns(lane, k)) continue;
            const int c = chan(lane, k);
            float ep, en;
            celu_pair(out[k], ep, en);
            a.Xout[loc * (2 * NF) + c] = ep;
            a.Xout[loc * (2 * NF) + NF + c] = en;
        }
    } else {
        store_raw_celu(a.Rout, a.Eout, loc, lane, out);
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
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= a.nitems) return;
    const int f = item / a.L, q = item - f * a.L;
    float mA[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) mA[t] = a.mask[((size_t)f * 9 + t) * a.L + q];
    float u[NCH];
    uinit_gather(a.codes + (size_t)f * a.L, mA, a.w, a.bias, q, a.H, a.W, lane, u);
    pono_wave(u, lane);  // norm_init
    store_raw_celu(a.Rout, a.Eout, (size_t)item, lane, u);
}

// logits = nin_out partial + bias; nchw: (F,512,H,W) like the reference, else (nitems,512)
__global__ __launch_bounds__(256) void k_logits_grid(const float *partial, const float *bias, int nitems, int L,
                                                     int nchw, float *logits)
{
    const int item = blockIdx.x;
    const int f = item / L, q = item - f * L;
    for (int o = threadIdx.x; o < NCLS; o += 256) {
        const float v = partial[(size_t)item * NCLS + o] + bias[o];
        if (nchw) logits[((size_t)f * NCLS + o) * L + q] = v;
        else logits[(size_t)item * NCLS + o] = v;
    }
}

// ==========================================================================================
// column mode: one location per frame per order position (the incremental AR step).
// Two launches per order position:
//   k_nbr    every NEIGHBOUR-tap partial sum (slots NA, NB) of all 32 masked convs at once.  They only
//            read finished columns of earlier order positions, so they do not depend on this
//            position's chain and run fully parallel (one wave = one stage x slot x 16 channels).
//   k_chain  one workgroup per 16 frames walks the 33 stages in order.  Only the centre taps
//            (1x1 products on the fresh activation) and the post ops are sequential; activations go
//            stage to stage through LDS, weights stream from L2.  Ends with the categorical draw and
//            the context of the next order position.
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
constexpr int NST = 33;       // 14 x (conv_input, conv_out) + 4 dilated convs + nin_out
constexpr int NBR_LD = 2 * NF;
constexpr int SIN_LD = 2 * NF + 4;
constexpr int SL_LD = NCLS + 4;
constexpr int CHAIN_WAVES = 16;  // one wave per frame of the 16-frame tile in the post ops; waves 0..9 own the MFMA tiles

struct __attribute__((aligned(16))) StageDesc {
    // control words first, 16-byte aligned: k_chain fetches them with two ds_read_b128 per stage
    int pro, in_form, save_slot /* keep this u in LDS, -1 */, p_has_skip;
    int NG, Co_pad, center_tap, skip_slot /* saved u_k feeding w_skip, -1 */;
    const float *w;       // packed weights [taps][NG*4][Co_pad][4]
    const float *w_skip;  // packed nin_skip [40][80][4] or null
    const float *in;      // cache the neighbour taps gather from (E / X / R at earlier order positions)
    int in_ld, dil, mask_kind, has_nbr;
    // prologue of this stage = post op of the previous stage
    const float *pbias, *pbias2;
    float *outR, *outE, *outX;  // caches the prologue writes at the current location
};

struct NbrWork { int stage, half, cog; };

struct NbrArgs {
    const StageDesc *stages;
    const NbrWork *work;
    const StepCtx *ctx;
    float *nbr;   // [NST][2][F][NBR_LD]
    float *upre;  // [F][NF] u_init before norm_init
    const int32_t *codes;
    const float *uinit_w, *uinit_b;
    int nwork, H, W, L, F;
};

template <int NG>
__device__ __forceinline__ void nbr_taps(const StageDesc &sd, const NbrArgs &a, int half, int o0, int f, bool valid,
                                         int i, int kk, f32x4 &acc0, f32x4 &acc1)
{
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    int q = 0, r = 0, c = 0;
    if (valid) {
        q = a.ctx[f].q;
        r = q / a.W;
        c = q - r * a.W;
    }
    const size_t per_tap = (size_t)NG * 16 * sd.Co_pad;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const int t = half * 5 + tt;  // taps 0..3 or 5..8
        const int rr = r + (t / 3 - 1) * sd.dil, cc = c + (t % 3 - 1) * sd.dil;
        float mv = 0.0f;
        const float *src = nullptr;
        if (valid && rr >= 0 && rr < a.H && cc >= 0 && cc < a.W) {
            mv = a.ctx[f].m[sd.mask_kind][t];
            src = sd.in + ((size_t)f * a.L + rr * a.W + cc) * sd.in_ld + 4 * kk;
        }
        const bool live = mv != 0.0f;
        if (!__any(live)) continue;
        const float *wbase = sd.w + t * per_tap + ((size_t)kk * sd.Co_pad + o0 + i) * 4;
#pragma unroll
        for (int g0 = 0; g0 < NG; g0 += 5) {
            f32x4 av[5], bv[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                av[j] = *PS_GC(f32x4, wbase + (size_t)(g0 + j) * 16 * sd.Co_pad);
                bv[j] = live ? *PS_GC(f32x4, src + 16 * (g0 + j)) * mv : zero;
            }
            mfma_chunk5(av, bv, acc0, acc1);
        }
    }
}

// grid (work items, ceil(F/16)); 4 waves = 4 output-channel tiles of one (stage, slot)
__global__ __launch_bounds__(256) void k_nbr(NbrArgs a)
{
    if ((int)blockIdx.x >= a.nwork) {  // last 4 work items: the u_init gather, one wave per frame
        const int f = blockIdx.y * 16 + ((int)blockIdx.x - a.nwork) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (f >= a.F) return;
        float mA[9], v[NCH];
#pragma unroll
        for (int t = 0; t < 9; ++t) mA[t] = a.ctx[f].m[0][t];
        uinit_gather(a.codes + (size_t)f * a.L, mA, a.uinit_w, a.uinit_b, a.ctx[f].q, a.H, a.W, lane, v);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (owns(lane, k)) a.upre[(size_t)f * NF + chan(lane, k)] = v[k];
        return;
    }
    const NbrWork wk = a.work[blockIdx.x];
    const StageDesc sd = a.stages[wk.stage];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    const int cot = wk.cog * 4 + wave;
    if (cot * 16 >= sd.Co_pad) return;
    const int o0 = cot * 16;
    const int f = blockIdx.y * 16 + i;
    const bool valid = f < a.F;
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
    if (sd.NG == 10) nbr_taps<10>(sd, a, wk.half, o0, f, valid, i, kk, acc0, acc1);
    else nbr_taps<5>(sd, a, wk.half, o0, f, valid, i, kk, acc0, acc1);
    if (valid) *(f32x4 *)(a.nbr + (((size_t)wk.stage * 2 + wk.half) * a.F + f) * NBR_LD + o0 + kk * 4) = acc0 + acc1;
}

struct ChainArgs {
    const StageDesc *stages;
    const float *nbr;
    const float *upre;        // [F][NF] from k_nbr
    CtxArgs cx;
    const float *out_b;
    int H, W, L, F;
    // end of the order position
    int32_t *codes;           // (F,L) written for sampled locations, or null (logits only)
    const uint8_t *region;    // (F,L) by location
    const int32_t *forced;    // (F,L) by location or null
    const float *uniforms;    // (F,L) by location or null
    float *out_logits;        // (F,L,512) by location or null
    float *step_logits;       // (F,512) or null
    float temperature;
    int advance;              // 1: write the context of step+1
    unsigned long long *trace; // optional [NST][10] shader-clock stamps of workgroup 0 (tuning aid)
    int ablate;                // tuning aid (PS_CHAIN_ABLATE): 1 no cache stores, 2 no slot prefetch, 4 no MFMA, 8 no post math
};

// Workgroup barrier that only drains LDS traffic.  __syncthreads() also waits for every outstanding
// global access (vmcnt(0)), which would serialise the weight / neighbour-slot prefetches of k_chain
// against its two barriers per stage; the data exchanged between the waves here lives in LDS only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one centre-tap product: out[16 items][16 channels] = W (registers, loaded early) x LDS input rows
template <int NG>
__device__ __forceinline__ f32x4 center_tile(const f32x4 *av, const float (*sIn)[SIN_LD], int i, int kk)
{
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 acc0 = zero, acc1 = zero;
#pragma unroll
    for (int g0 = 0; g0 < NG; g0 += 5) {
        f32x4 a5[5], b5[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            a5[j] = av[g0 + j];
            b5[j] = *(const f32x4 *)(&sIn[i][16 * (g0 + j) + 4 * kk]);
        }
        mfma_chunk5(a5, b5, acc0, acc1);
    }
    return acc0 + acc1;
}

template <int NG>
__device__ __forceinline__ void load_tile_weights(const float *__restrict__ w, int Co_pad, int o0, int i, int kk, f32x4 *av)
{
    const float *wbase = w + ((size_t)kk * Co_pad + o0 + i) * 4;
#pragma unroll
    for (int g = 0; g < NG; ++g) av[g] = *PS_GC(f32x4, wbase + (size_t)g * 16 * Co_pad);
}

// operands of a prologue (post op of the previous stage), fetched one stage ahead as RAW values: no
// arithmetic at fetch time, so nothing waits for the loads until the next stage consumes them
struct PreOps { float pb[NCH], na[NCH], nb[NCH], pbg[NCH], nag[NCH], nbg[NCH], skb[NCH]; };

template <int PRO>
__device__ __forceinline__ void chain_prefetch(PreOps &p, const float *pbias, const float *pbias2, bool has_skip,
                                               const float *nA, const float *nB, int lane)
{
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (!owns(lane, k)) continue;
        const int c = chan(lane, k);
        p.pb[k] = PS_GC(float, pbias)[c];
        p.na[k] = PS_GC(float, nA)[c];
        p.nb[k] = PS_GC(float, nB)[c];
        if (PRO == PRO_GATE) {
            p.pbg[k] = PS_GC(float, pbias)[c + NF];
            p.nag[k] = PS_GC(float, nA)[c + NF];
            p.nbg[k] = PS_GC(float, nB)[c + NF];
        }
        if (PRO == PRO_CONVIN && has_skip) p.skb[k] = PS_GC(float, pbias2)[c];
    }
}

// post op of the previous stage for one item: slots summed as ((bias + NA) + C) + NB like the whole-grid pass
template <int PRO>
__device__ __forceinline__ void chain_post(const PreOps &p, bool has_skip, const float *sCrow, const float *sSrow,
                                           int lane, const float (&ucur)[NCH], float (&out)[NCH])
{
    float v[NCH], g[NCH], skip[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        v[k] = g[k] = skip[k] = 0.0f;
        if (!owns(lane, k)) continue;
        const int c = chan(lane, k);
        if (PRO == PRO_UINIT) {
            v[k] = p.pb[k];  // u_init before norm_init, from k_nbr
        } else {
            v[k] = ((p.pb[k] + p.na[k]) + sCrow[c]) + p.nb[k];
            if (PRO == PRO_GATE) g[k] = ((p.pbg[k] + p.nag[k]) + sCrow[c + NF]) + p.nbg[k];
            if (PRO == PRO_CONVIN && has_skip) skip[k] = sSrow[c] + p.skb[k];
        }
    }
    if (PRO == PRO_CONVIN) post_math<POST_CONVIN>(v, g, skip, has_skip, ucur, lane, out);
    else if (PRO == PRO_GATE) post_math<POST_GATE>(v, g, skip, false, ucur, lane, out);
    else post_math<POST_DIL>(v, g, skip, false, ucur, lane, out);  // PRO_UINIT: norm_init is the same PONO
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__global__ __launch_bounds__(CHAIN_WAVES * 64) void k_chain(ChainArgs a)
{
    __shared__ __attribute__((aligned(16))) float sIn[16][SIN_LD];    // input of the centre tap
    __shared__ __attribute__((aligned(16))) float sSkip[16][SIN_LD];  // concat_elu(u_k) feeding nin_skip
    __shared__ __attribute__((aligned(16))) float sC[16][SIN_LD];     // centre-tap results
    __shared__ __attribute__((aligned(16))) float sS[16][NF + 4];     // nin_skip results
    __shared__ float sU[8][16][NF];                                   // u0..u7 of this location (skip connections)
    __shared__ __attribute__((aligned(16))) float sL[16][SL_LD];      // nin_out results
    __shared__ StageDesc sSt[NST];                                    // the chain description, read every stage
    const int tid = threadIdx.x, wave = uni(tid >> 6), lane = tid & 63, i = lane & 15, kk = lane >> 4;
    const int f0 = blockIdx.x * 16;
    {
        const int *src = (const int *)a.stages;
        int *dst = (int *)sSt;
        for (int k = tid; k < (int)(NST * sizeof(StageDesc) / 4); k += CHAIN_WAVES * 64) dst[k] = src[k];
    }
    // post-op role: wave w handles frame f0 + w of the tile
    const int pf = f0 + wave;
    const bool pact = pf < a.F;
    int pq = 0;
    if (pact) pq = uni(a.cx.ctx[pf].q);
    const size_t ploc = (size_t)pf * a.L + pq;
    const size_t off80 = ploc * NF, off160 = ploc * (2 * NF);                  // this location in the R / E,X caches
    const float *nbr_item = a.nbr + (size_t)pf * NBR_LD;                       // this frame's rows of the k_nbr slots
    const size_t nbr_half = (size_t)a.F * NBR_LD, nbr_stage = 2 * nbr_half;
    float ucur[NCH] = {0.0f, 0.0f};
    PreOps pre;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        pre.pb[k] = pre.na[k] = pre.nb[k] = pre.pbg[k] = pre.nag[k] = pre.nbg[k] = pre.skb[k] = 0.0f;
        if (pact && owns(lane, k)) pre.pb[k] = a.upre[(size_t)pf * NF + chan(lane, k)];  // stage 0: u_init before norm_init
    }
    if (!pact) {  // rows of absent frames feed zeros into the MFMA tiles
        for (int c = lane; c < 2 * NF; c += 64) { sIn[wave][c] = 0.0f; sSkip[wave][c] = 0.0f; }
    }
    __syncthreads();

#define PS_TRACE(slot) do { if (a.trace && blockIdx.x == 0 && tid == 0) a.trace[s * 10 + (slot)] = clock64(); } while (0)
    for (int s = 0; s < NST - 1; ++s) {
        // stage control words: wave-uniform scalars, read once
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const i32x4 c0 = *(const i32x4 *)&sSt[s].pro, c1 = *(const i32x4 *)&sSt[s].NG;
        const int pro = uni(c0.x), in_form = uni(c0.y), save_slot = uni(c0.z), p_has_skip = uni(c0.w), NG = uni(c1.x),
                  Co_pad = uni(c1.y), center_tap = uni(c1.z);
        const float *w = sSt[s].w, *w_skip = sSt[s].w_skip;
        PS_TRACE(0);
        // ---- (1) this stage's centre-tap weights: independent of everything computed here, issue first.
        //      Waves 0..9 own the tiles of the main product, waves 10..14 those of the nin_skip 1x1.
        const int ntile = Co_pad >> 4;
        const bool main_w = wave < ntile, skip_w = w_skip != nullptr && wave >= 10 && wave < 15;
        f32x4 av[10];
        if (main_w) {
            const float *wc = w + (size_t)center_tap * NG * 16 * Co_pad;
            if (NG == 10) load_tile_weights<10>(wc, Co_pad, wave * 16, i, kk, av);
            else load_tile_weights<5>(wc, Co_pad, wave * 16, i, kk, av);
        } else if (skip_w) {
            load_tile_weights<10>(w_skip, NF, (wave - 10) * 16, i, kk, av);
        }
        PS_TRACE(5);
        // ---- (2) post op of stage s-1 on this wave's frame
        if (pact) {
            // (2a) consume the operands fetched during the previous stage ...
            const PreOps cur = pre;
            // (2b) ... and put the NEXT prologue's operands (k_nbr slots of THIS stage + biases) in flight before
            //      the post-op math, whose dependent chain (LDS read, two wave reductions, exp/rcp) hides them
            {
                const int npro = uni(sSt[s + 1].pro), nskip = uni(sSt[s + 1].p_has_skip);
                const float *pbias = sSt[s + 1].pbias, *pbias2 = sSt[s + 1].pbias2;
                const float *nA = nbr_item + (size_t)s * nbr_stage, *nB = nA + nbr_half;
                if (a.ablate & 2) {
                } else if (npro == PRO_CONVIN) chain_prefetch<PRO_CONVIN>(pre, pbias, pbias2, nskip != 0, nA, nB, lane);
                else if (npro == PRO_GATE) chain_prefetch<PRO_GATE>(pre, pbias, pbias2, false, nA, nB, lane);
                else chain_prefetch<PRO_DIL>(pre, pbias, pbias2, false, nA, nB, lane);
            }
            PS_TRACE(8);
            float out[NCH];
            if (a.ablate & 8) {
#pragma unroll
                for (int k = 0; k < NCH; ++k) out[k] = owns(lane, k) ? cur.pb[k] + sC[wave][chan(lane, k)] : 0.0f;
            } else if (pro == PRO_CONVIN) chain_post<PRO_CONVIN>(cur, p_has_skip != 0, sC[wave], sS[wave], lane, ucur, out);
            else if (pro == PRO_GATE) chain_post<PRO_GATE>(cur, false, sC[wave], sS[wave], lane, ucur, out);
            else if (pro == PRO_DIL) chain_post<PRO_DIL>(cur, false, sC[wave], sS[wave], lane, ucur, out);
            else chain_post<PRO_UINIT>(cur, false, sC[wave], sS[wave], lane, ucur, out);
            PS_TRACE(6);
            float ep[NCH], en[NCH];
#pragma unroll
            for (int k = 0; k < NCH; ++k) celu_pair(out[k], ep[k], en[k]);
            const bool is_u = pro != PRO_CONVIN;
            float *o1 = is_u ? sSt[s].outR : sSt[s].outX, *o2 = sSt[s].outE;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (!owns(lane, k)) continue;
                const int c = chan(lane, k);
                if (in_form == IN_CELU) { sIn[wave][c] = ep[k]; sIn[wave][NF + c] = en[k]; }
                else sIn[wave][c] = out[k];  // IN_RAW: dilated convs read the raw u
                if (is_u) {
                    ucur[k] = out[k];
                    if (save_slot >= 0) sU[save_slot][wave][c] = out[k];
                }
                if (!(a.ablate & 1)) {
                    if (is_u) {
                        PS_G(float, o1)[off80 + c] = out[k];
                        PS_G(float, o2)[off160 + c] = ep[k];
                        PS_G(float, o2)[off160 + NF + c] = en[k];
                    } else {
                        PS_G(float, o1)[off160 + c] = ep[k];
                        PS_G(float, o1)[off160 + NF + c] = en[k];
                    }
                }
            }
            PS_TRACE(7);
            const int next_skip = uni(sSt[s + 1].skip_slot);
            if (next_skip >= 0) {  // stage the NEXT stage's nin_skip input (u_k of this location, from the up pass)
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    if (!owns(lane, k)) continue;
                    const int c = chan(lane, k);
                    float sp, sn;
                    celu_pair(sU[next_skip][wave][c], sp, sn);
                    sSkip[wave][c] = sp;
                    sSkip[wave][NF + c] = sn;
                }
            }
        }
        PS_TRACE(1);
        lds_barrier();
        PS_TRACE(2);
        // ---- (4) centre-tap products and the nin_skip 1x1 (its input was staged one stage earlier)
        if (!(a.ablate & 4)) {
            if (main_w) {
                const f32x4 r = NG == 10 ? center_tile<10>(av, sIn, i, kk) : center_tile<5>(av, sIn, i, kk);
                *(f32x4 *)(&sC[i][wave * 16 + kk * 4]) = r;
            } else if (skip_w) {
                const f32x4 r = center_tile<10>(av, sSkip, i, kk);
                *(f32x4 *)(&sS[i][(wave - 10) * 16 + kk * 4]) = r;
            }
        }
        PS_TRACE(3);
        lds_barrier();
        PS_TRACE(4);
    }
#undef PS_TRACE

    {   // ---- nin_out(elu(u)) (model.py:153): its prologue is the last gate; 32 tiles, two per wave
        const StageDesc &sd = sSt[NST - 1];
        const int ntile = sd.Co_pad >> 4;
        f32x4 avL[2][5];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ct = wave + r * CHAIN_WAVES;
            if (ct < ntile) load_tile_weights<5>(sd.w, sd.Co_pad, ct * 16, i, kk, avL[r]);
        }
        if (pact) {
            float out[NCH];
            chain_post<PRO_GATE>(pre, false, sC[wave], sS[wave], lane, ucur, out);
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                if (!owns(lane, k)) continue;
                const int c = chan(lane, k);
                float ep, en;
                celu_pair(out[k], ep, en);
                sIn[wave][c] = ep;
                PS_G(float, sd.outR)[off80 + c] = out[k];
                PS_G(float, sd.outE)[off160 + c] = ep;
                PS_G(float, sd.outE)[off160 + NF + c] = en;
            }
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ct = wave + r * CHAIN_WAVES;
            if (ct < ntile) *(f32x4 *)(&sL[i][ct * 16 + kk * 4]) = center_tile<5>(avL[r], sIn, i, kk);
        }
        lds_barrier();
    }

    // ---- end of the order position: logits, categorical draw (sample.py:60-66), next context; wave = frame
    if (pact) {
        const int f = pf, j = wave;
        const size_t loc = ploc;
        float lg[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) lg[k] = sL[j][lane * 8 + k] + a.out_b[lane * 8 + k];
        if (a.out_logits) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a.out_logits[loc * NCLS + lane * 8 + k] = lg[k];
        }
        if (a.step_logits) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a.step_logits[(size_t)f * NCLS + lane * 8 + k] = lg[k];
        }
        if (a.codes && a.region[loc]) {
            if (a.forced) {
                if (lane == 0) a.codes[loc] = a.forced[loc];
            } else {
                float x[8], m = -INFINITY;
#pragma unroll
                for (int k = 0; k < 8; ++k) { x[k] = lg[k] / a.temperature; m = fmaxf(m, x[k]); }
                for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
                float e[8], ls = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k) { e[k] = expf(x[k] - m); ls += e[k]; }
                float incl = ls;  // inclusive scan of the per-lane sums (classes are lane-major)
                for (int off = 1; off < 64; off <<= 1) {
                    const float t = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += t;
                }
                const float total = __shfl(incl, 63, 64);
                const float target = a.uniforms[loc] * total;
                float run = incl - ls;
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) { run += e[k]; cnt += run <= target ? 1 : 0; }  // classes whose cdf <= target
                for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
                if (lane == 0) a.codes[loc] = min(cnt, NCLS - 1);
            }
        }
        if (a.advance) {  // this wave is the only reader of ctx[f] in this launch
            const int step = a.cx.ctx[f].step;
            if (lane < 32) ctx_fill(a.cx, f, step + 1, lane);
        }
    }
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
    const size_t ss = nitems * Co_pad, at = item * Co_pad + o;
    y[i] = slot_sum(bias ? bias[o] : 0.0f, partial[SLOT_NA * ss + at], partial[SLOT_C * ss + at], partial[SLOT_NB * ss + at]);
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
    float *partial = nullptr;       // whole-grid slots [4][maxF*L][160]
    float *nbr = nullptr;           // column mode: neighbour slots [NST][2][maxF][160]
    float *upre = nullptr;          // column mode: u_init before norm_init [maxF][80]
    float *col_logits = nullptr;
    StepCtx *ctx = nullptr;
    StageDesc *stages = nullptr;    // device copy of the 33-stage chain description
    NbrWork *work = nullptr;
    int nwork = 0;
    hipStream_t stream = nullptr;   // internal stream for graph capture/replay
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipGraph_t graph = nullptr;          // step graph of the last ar_run (kept alive until replaced)
    hipGraphExec_t graph_exec = nullptr;
    bool use_graph = true;
    // bench.py profiling aid (ps_pixelcnn_time_column_step): event pair around every launch, by kernel tag
    struct ProfRec { int tag; hipEvent_t e0, e1; };
    std::vector<ProfRec> *prof = nullptr;
    double flops_nbr = 0.0, flops_chain = 0.0, wbytes_nbr = 0.0, wbytes_chain = 0.0;  // dense work of one step, per frame
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

enum { TAG_NBR = 0, TAG_CHAIN = 1 };

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

// 3x3 taps in slot order: NA = taps 0..3, C = tap 4, NB = taps 5..8 (+ optional SKIP appended by the caller)
void conv_taps(GemmArgs &a, const float *in, int ld, const float *wp, int Cin, int Co_pad, int dil)
{
    a.Cin = Cin;
    a.Co_pad = Co_pad;
    const size_t per_tap = (size_t)Cin * Co_pad;
    for (int t = 0; t < 9; ++t)
        a.tap[t] = GemmTap{in, wp + t * per_tap, (t / 3 - 1) * dil, (t % 3 - 1) * dil, t, ld};
    a.nslots = 3;
    a.slot_first[0] = 0; a.slot_first[1] = 4; a.slot_first[2] = 5; a.slot_first[3] = 9; a.slot_first[4] = 9;
}

// ------------------------------------------------------------------------------------------
// whole-grid evaluation (reference-faithful forward; cache build before the column steps)
// logits: null (caches only), (F,512,H,W) when nchw, else (F*L,512) by location
// ------------------------------------------------------------------------------------------
void run_grid(ps_pixelcnn *h, int F, const int32_t *codes, const Masks &m, float *logits, bool nchw, hipStream_t st)
{
    const int nitems = F * h->L;
    const int pblocks = (nitems + 3) / 4;
    auto gemm = [&](GemmArgs &a, const float *mask) {
        a.H = h->H; a.W = h->W; a.L = h->L; a.nitems = nitems;
        a.mask = mask; a.mask_fstride = (size_t)9 * h->L; a.partial = h->partial; a.tiles_per_block = 8;
        const int tiles = (nitems + 15) / 16;
        hipLaunchKernelGGL(k_gemm, dim3(a.Co_pad / 16, a.nslots, (tiles + 7) / 8), dim3(64), 0, st, a);
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
            a.slot_first[4] = 10;
            a.nslots = 4;
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
    a.Cin = NF; a.Co_pad = NCLS; a.nslots = 1;
    a.slot_first[0] = 0; a.slot_first[1] = 1;
    a.tap[0] = GemmTap{h->E[NNODE - 1], h->out_w, 0, 0, -1, 2 * NF};
    gemm(a, nullptr);
    hipLaunchKernelGGL(k_logits_grid, dim3(nitems), dim3(256), 0, st, h->partial, h->out_b, nitems, h->L, nchw ? 1 : 0,
                       logits);
}

// ------------------------------------------------------------------------------------------
// the 33-stage chain description consumed by k_nbr / k_chain (built once per handle)
// ------------------------------------------------------------------------------------------
int build_stage_table(ps_pixelcnn *h)
{
    std::vector<StageDesc> st;
    std::vector<NbrWork> work;
    struct Prev { int pro; const float *bias, *bias2; int has_skip; float *R, *E, *X; int save; } prev;
    prev = Prev{PRO_UINIT, nullptr, nullptr, 0, h->R[0], h->E[0], nullptr, 0};  // u0 is saved in LDS slot 0
    auto push = [&](const float *w, const float *w_skip, const float *in, int in_ld, int NG, int Co, int dil,
                    int mask_kind, int center, int has_nbr, int in_form, int skip_slot) {
        StageDesc d{};
        d.w = w; d.w_skip = w_skip; d.in = in; d.in_ld = in_ld; d.NG = NG; d.Co_pad = Co; d.dil = dil;
        d.mask_kind = mask_kind; d.center_tap = center; d.has_nbr = has_nbr;
        d.pro = prev.pro; d.in_form = in_form; d.skip_slot = skip_slot; d.save_slot = prev.save;
        d.pbias = prev.bias; d.pbias2 = prev.bias2; d.p_has_skip = prev.has_skip;
        d.outR = prev.R; d.outE = prev.E; d.outX = prev.X;
        const int s = (int)st.size();
        if (has_nbr)
            for (int half = 0; half < 2; ++half)
                for (int cog = 0; cog < (Co / 16 + 3) / 4; ++cog) work.push_back(NbrWork{s, half, cog});
        // dense algorithmic work per frame of this stage (taps x 2*Co*Cin flops, fp32 weights once)
        const double taps_nbr = has_nbr ? 8.0 : 0.0, cin = NG * 16.0;
        h->flops_nbr += taps_nbr * 2.0 * Co * cin;
        h->wbytes_nbr += taps_nbr * 4.0 * Co * cin;
        h->flops_chain += 2.0 * Co * cin + (w_skip ? 2.0 * NF * 2 * NF : 0.0);
        h->wbytes_chain += 4.0 * Co * cin + (w_skip ? 4.0 * NF * 2 * NF : 0.0);
        st.push_back(d);
    };
    auto gated = [&](int g) {
        const ps_pixelcnn::Gated &G = h->gated[g];
        push(G.w_in, G.w_skip, h->E[G.node_in], 2 * NF, 10, NF, 1, 1, 4, 1, IN_CELU, G.node_skip >= 0 ? G.node_skip : -1);
        prev = Prev{PRO_CONVIN, G.b_in, G.b_skip, G.node_skip >= 0, nullptr, nullptr, h->X[g], -1};
        push(G.w_out, nullptr, h->X[g], 2 * NF, 10, 2 * NF, 1, 1, 4, 1, IN_CELU, -1);
        prev = Prev{PRO_GATE, G.b_out, nullptr, 0, h->R[G.node_out], h->E[G.node_out], nullptr, -1};
        prev.save = (G.node_out >= 1 && G.node_out <= 7) ? G.node_out : -1;  // LDS slot k holds u_k
    };
    auto dilated = [&](int d) {
        const ps_pixelcnn::Dil &D = h->dil[d];
        push(D.w, nullptr, h->R[D.node_in], NF, 5, NF, 2, 2, 4, 1, IN_RAW, -1);
        prev = Prev{PRO_DIL, D.b, nullptr, 0, h->R[D.node_out], h->E[D.node_out], nullptr, -1};
        prev.save = (D.node_out >= 1 && D.node_out <= 7) ? D.node_out : -1;
    };
    gated(0); gated(1); dilated(0); gated(2); gated(3); dilated(1); gated(4); gated(5);
    gated(6); gated(7); dilated(2); gated(8); gated(9); gated(10); dilated(3);
    gated(11); gated(12); gated(13);
    push(h->out_w, nullptr, nullptr, 0, 5, NCLS, 1, 1, 0, 0, IN_ELU, -1);  // nin_out(elu(u)), prologue = last gate
    if ((int)st.size() != NST) return ps::fail(PS_ERR_STATE, "stage table has %d entries, expected %d", (int)st.size(), NST);
    if (int rc = dev_alloc(h, &h->stages, st.size())) return rc;
    PS_HIP_CHECK(hipMemcpy(h->stages, st.data(), st.size() * sizeof(StageDesc), hipMemcpyHostToDevice));
    if (int rc = dev_alloc(h, &h->work, work.size())) return rc;
    PS_HIP_CHECK(hipMemcpy(h->work, work.data(), work.size() * sizeof(NbrWork), hipMemcpyHostToDevice));
    h->nwork = (int)work.size();
    return PS_OK;
}

// one order position: neighbour taps of every conv, then the centre-tap chain + draw.  h->ctx must
// describe the current position.
void run_column(ps_pixelcnn *h, int F, const int32_t *codes, ChainArgs ca, hipStream_t st)
{
    NbrArgs na{h->stages, h->work, h->ctx, h->nbr, h->upre, codes, h->uinit_w, h->uinit_b, h->nwork, h->H, h->W, h->L, F};
    const int tiles = (F + 15) / 16;
    timed(h, st, TAG_NBR, [&]() { hipLaunchKernelGGL(k_nbr, dim3(h->nwork + 4, tiles), dim3(256), 0, st, na); });
    ca.stages = h->stages; ca.nbr = h->nbr; ca.upre = h->upre;
    ca.out_b = h->out_b;
    ca.H = h->H; ca.W = h->W; ca.L = h->L; ca.F = F;
    timed(h, st, TAG_CHAIN, [&]() { hipLaunchKernelGGL(k_chain, dim3(tiles), dim3(CHAIN_WAVES * 64), 0, st, ca); });
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
    // Two launches per order position keep the host far ahead of the GPU, so the loop is launched eagerly on the
    // caller's stream by default; PS_AR_GRAPH=1 replays it as a hipGraph on a stream owned by the handle instead.
    const char *env = getenv("PS_AR_GRAPH");
    h->use_graph = env && env[0] == '1';
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
    size_t pfloats = (size_t)4 * locs * 2 * NF;
    if (locs * NCLS > pfloats) pfloats = locs * NCLS;
    if ((rc = dev_alloc(h, &h->partial, pfloats))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->col_logits, (size_t)max_frames * NCLS))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->nbr, (size_t)NST * 2 * max_frames * NBR_LD))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->upre, (size_t)max_frames * NF))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->ctx, (size_t)max_frames))) return fail_out(rc);
    if ((rc = build_stage_table(h))) return fail_out(rc);
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
    ChainArgs ca{};
    ca.cx = make_ctx_args(h, order, m, F);
    hipLaunchKernelGGL(k_ctx_init, dim3(F), dim3(32), 0, st, ca.cx, step);
    ca.step_logits = logits;
    ca.temperature = 1.0f;
    run_column(h, F, codes, ca, st);
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
    ChainArgs ca{};
    ca.cx = make_ctx_args(h, order, m, F);
    ca.codes = codes; ca.region = sample_region; ca.forced = forced; ca.uniforms = uniforms;
    ca.out_logits = out_logits; ca.temperature = temperature; ca.advance = 1;
    hipLaunchKernelGGL(k_ctx_init, dim3(F), dim3(32), 0, st, ca.cx, first_step);
    PS_LAUNCH_CHECK();
    const int nsteps = h->L - first_step;
    if (nsteps > 0) {
        if (h->use_graph) {
            PS_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            run_column(h, F, codes, ca, st);
            PS_HIP_CHECK(hipStreamEndCapture(st, &h->graph));
            PS_HIP_CHECK(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
            for (int sidx = 0; sidx < nsteps; ++sidx) PS_HIP_CHECK(hipGraphLaunch(h->graph_exec, st));
        } else {
            for (int sidx = 0; sidx < nsteps; ++sidx) run_column(h, F, codes, ca, st);
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
                                 int *launches, float *total_ms, double *flops_per_launch, double *weight_bytes_per_launch,
                                 void *stream)
{
    if (int rc = check_handle(h, F)) return rc;
    PS_REQUIRE(codes && order && mask_init && mask_undilated && mask_dilated && launches && total_ms,
               "pixelcnn_time_column_step: null pointer");
    PS_REQUIRE(step >= 0 && step < h->L && reps > 0, "pixelcnn_time_column_step: bad step / reps");
    hipStream_t st = (hipStream_t)stream;
    std::vector<ps_pixelcnn::ProfRec> recs;
    ChainArgs ca{};
    ca.cx = make_ctx_args(h, order, Masks{mask_init, mask_undilated, mask_dilated}, F);
    ca.step_logits = h->col_logits;
    ca.temperature = 1.0f;
    if (const char *ab = getenv("PS_CHAIN_ABLATE")) ca.ablate = atoi(ab);
    hipLaunchKernelGGL(k_ctx_init, dim3(F), dim3(32), 0, st, ca.cx, step);
    run_column(h, F, codes, ca, st);  // untimed warm-up
    if (const char *tp = getenv("PS_CHAIN_TRACE")) {  // tuning aid: per-stage shader-clock stamps of workgroup 0
        unsigned long long *d = nullptr;
        if (hipMalloc(&d, NST * 10 * 8) == hipSuccess) {
            (void)hipMemsetAsync(d, 0, NST * 10 * 8, st);
            ChainArgs ct = ca;
            ct.trace = d;
            run_column(h, F, codes, ct, st);
            std::vector<unsigned long long> hst(NST * 10);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(hst.data(), d, NST * 10 * 8, hipMemcpyDeviceToHost);
            if (FILE *fp = fopen(tp, "w")) {
                for (int s2 = 0; s2 < NST - 1; ++s2)
                    fprintf(fp, "%d %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", s2, hst[s2 * 10], hst[s2 * 10 + 5],
                            hst[s2 * 10 + 6], hst[s2 * 10 + 7], hst[s2 * 10 + 8], hst[s2 * 10 + 1], hst[s2 * 10 + 2],
                            hst[s2 * 10 + 3], hst[s2 * 10 + 4]);
                fclose(fp);
            }
            (void)hipFree(d);
        }
    }
    h->prof = &recs;
    for (int r = 0; r < reps; ++r) run_column(h, F, codes, ca, st);
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
    if (flops_per_launch) { flops_per_launch[TAG_NBR] = h->flops_nbr * F; flops_per_launch[TAG_CHAIN] = h->flops_chain * F; }
    if (weight_bytes_per_launch) { weight_bytes_per_launch[TAG_NBR] = h->wbytes_nbr; weight_bytes_per_launch[TAG_CHAIN] = h->wbytes_chain; }
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
    o = ps::align_up(o + 3 * (size_t)B * L * Cop * 4, 256);
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
    hipLaunchKernelGGL(k_gemm, dim3(Cop / 16, a.nslots, (tiles + 7) / 8), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_reduce_nchw, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, partial, bias, B, Co, Cop, L, y);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // extern "C"
