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

struct GemmArgs {
    GemmTap tap[MAX_TAPS];
    int ntaps, Cin, Co_pad, H, W, L, nitems, tiles_per_block;
    const float *mask;
    size_t mask_fstride;
    const int32_t *order;     // COLUMN mode: (F,L) location visited at each order position
    const int32_t *step_ptr;  // COLUMN mode: current order position (device memory, graph-replayable)
    float *partial;           // [ntaps][nitems][Co_pad]
};

template <bool COLUMN>
__device__ __forceinline__ void item_map(int item, int L, const int32_t *order, const int32_t *step_ptr, int &f, int &q)
{
    if (COLUMN) {
        f = item;
        q = order[(size_t)f * L + *step_ptr];
    } else {
        f = item / L;
        q = item - f * L;
    }
}

// grid (Co_pad/16, ntaps, item blocks), one wave per block
template <bool COLUMN>
__global__ __launch_bounds__(64) void k_gemm(GemmArgs a)
{
    const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
    const int o0 = blockIdx.x * 16;
    const GemmTap tp = a.tap[blockIdx.y];
    const int ngroups = a.Cin >> 4;
    const float *wbase = tp.w + ((size_t)kk * a.Co_pad + o0 + i) * 4;
    for (int tt = 0; tt < a.tiles_per_block; ++tt) {
        const int tile = blockIdx.z * a.tiles_per_block + tt;
        if (tile * 16 >= a.nitems) break;
        const int item = tile * 16 + i;
        const bool valid = item < a.nitems;
        const float *src = nullptr;
        float mv = 0.0f;
        if (valid) {
            int f, q;
            item_map<COLUMN>(item, a.L, a.order, a.step_ptr, f, q);
            const int r = q / a.W, c = q - r * a.W;
            const int rr = r + tp.dr, cc = c + tp.dc;
            if (rr >= 0 && rr < a.H && cc >= 0 && cc < a.W) {
                mv = tp.mask_row >= 0 ? a.mask[(size_t)f * a.mask_fstride + (size_t)tp.mask_row * a.L + q] : 1.0f;
                src = tp.in + ((size_t)f * a.L + rr * a.W + cc) * tp.ld + 4 * kk;
            }
        }
        const bool live = valid && mv != 0.0f;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if (__any(live)) {
#pragma unroll 2
            for (int g = 0; g < ngroups; ++g) {
                const f32x4 av = *(const f32x4 *)(wbase + (size_t)g * 16 * a.Co_pad);
                f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
                if (live) bv = *(const f32x4 *)(src + 16 * g) * mv;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
            }
        }
        // D: row (output channel) = kk*4 + reg, col (item) = i
        if (valid)
            *(f32x4 *)(a.partial + ((size_t)blockIdx.y * a.nitems + item) * a.Co_pad + o0 + kk * 4) = acc;
    }
}

// ------------------------------------------------------------------------------------------
// per-item post ops
// ------------------------------------------------------------------------------------------
struct PostArgs {
    const float *partial;
    int nslots, nitems, Co_pad, L;
    const float *bias, *bias2;
    const float *Rin;
    float *Rout, *Eout, *Xout;
    float *logits;      // POST_LOGITS output
    int logits_nchw;    // 1: (F,512,H,W)   0: (nitems,512)
    const int32_t *order, *step_ptr;
};

__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : (expf(x) - 1.0f); }

// block-wide sum over `n` (<= 128) values, one per thread (threads >= n pass 0)
__device__ __forceinline__ float block_sum128(float v, float *sh)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float s = sh[0] + sh[1];
    __syncthreads();
    return s;
}

// PONO over NF channels held one per thread (threads < NF) (models/lmconv/layers.py:231-236)
__device__ __forceinline__ float pono80(float x, bool act, float *sh)
{
    const float mean = block_sum128(act ? x : 0.0f, sh) / (float)NF;
    const float d = act ? x - mean : 0.0f;
    const float var = block_sum128(d * d, sh) / (float)(NF - 1);  // unbiased
    return d / sqrtf(var + 1e-5f);
}

enum { POST_CONVIN = 0, POST_GATE = 1, POST_DIL = 2 };

template <int KIND, bool COLUMN>
__global__ __launch_bounds__(128) void k_post(PostArgs a)
{
    __shared__ float sh[2];
    const int item = blockIdx.x, o = threadIdx.x;
    const bool act = o < NF;
    int f, q;
    item_map<COLUMN>(item, a.L, a.order, a.step_ptr, f, q);
    const size_t loc = (size_t)f * a.L + q;
    const int nmain = 9;
    float v = 0.0f, g = 0.0f;
    if (act) {
        v = a.bias[o];
        for (int s = 0; s < nmain; ++s) v += a.partial[((size_t)s * a.nitems + item) * a.Co_pad + o];
        if (KIND == POST_GATE) {
            g = a.bias[o + NF];
            for (int s = 0; s < nmain; ++s) g += a.partial[((size_t)s * a.nitems + item) * a.Co_pad + o + NF];
        }
    }
    float n = pono80(v, act, sh);
    if (!act) return;
    if (KIND == POST_CONVIN) {
        if (a.nslots > nmain) n += a.partial[((size_t)nmain * a.nitems + item) * a.Co_pad + o] + a.bias2[o];
        a.Xout[loc * (2 * NF) + o] = elu1(n);
        a.Xout[loc * (2 * NF) + NF + o] = elu1(-n);
    } else {
        float u = n;
        if (KIND == POST_GATE) u = a.Rin[loc * NF + o] + n * (1.0f / (1.0f + expf(-g)));
        a.Rout[loc * NF + o] = u;
        a.Eout[loc * (2 * NF) + o] = elu1(u);
        a.Eout[loc * (2 * NF) + NF + o] = elu1(-u);
    }
}

template <bool COLUMN>
__global__ __launch_bounds__(256) void k_post_logits(PostArgs a)
{
    const int item = blockIdx.x;
    int f, q;
    item_map<COLUMN>(item, a.L, a.order, a.step_ptr, f, q);
    for (int o = threadIdx.x; o < NCLS; o += 256) {
        const float v = a.bias[o] + a.partial[(size_t)item * a.Co_pad + o];
        if (a.logits_nchw) a.logits[((size_t)f * NCLS + o) * a.L + q] = v;
        else a.logits[(size_t)item * NCLS + o] = v;
    }
}

// u_init on one-hot input as a gather (type-A mask): y[o] = b[o] + sum_t m_t (W[t][code(nbr_t)][o] + W[t][512][o])
struct UinitArgs {
    const int32_t *codes;  // (F,L), -1 = all-zero input
    const float *mask;     // mask_init (F,9,L)
    size_t mask_fstride;
    const float *w;        // [9][513][NF]
    const float *bias;
    float *Rout, *Eout;
    int H, W, L;
    const int32_t *order, *step_ptr;
};

template <bool COLUMN>
__global__ __launch_bounds__(128) void k_uinit(UinitArgs a)
{
    __shared__ float sh[2];
    const int item = blockIdx.x, o = threadIdx.x;
    const bool act = o < NF;
    int f, q;
    item_map<COLUMN>(item, a.L, a.order, a.step_ptr, f, q);
    const int r = q / a.W, c = q - r * a.W;
    float v = 0.0f;
    if (act) {
        v = a.bias[o];
        for (int t = 0; t < 9; ++t) {
            const int rr = r + t / 3 - 1, cc = c + t % 3 - 1;
            if (rr < 0 || rr >= a.H || cc < 0 || cc >= a.W) continue;
            const float mv = a.mask[(size_t)f * a.mask_fstride + (size_t)t * a.L + q];
            if (mv == 0.0f) continue;
            const int code = a.codes[(size_t)f * a.L + rr * a.W + cc];
            float w = a.w[((size_t)t * (NCLS + 1) + NCLS) * NF + o];
            if (code >= 0) w += a.w[((size_t)t * (NCLS + 1) + code) * NF + o];
            v += mv * w;
        }
    }
    const float u = pono80(v, act, sh);
    if (!act) return;
    const size_t loc = (size_t)f * a.L + q;
    a.Rout[loc * NF + o] = u;
    a.Eout[loc * (2 * NF) + o] = elu1(u);
    a.Eout[loc * (2 * NF) + NF + o] = elu1(-u);
}

// ------------------------------------------------------------------------------------------
// sampling (models/lmconv/sample.py:60-66): softmax(logits/T), one categorical draw, one-hot write
// ------------------------------------------------------------------------------------------
struct SampleArgs {
    const float *logits;      // (F,512) of this step
    int32_t *codes;           // (F,L)
    const int32_t *order;
    const uint8_t *region;    // (F,L) by location
    const int32_t *forced;    // (F,L) by location or null
    const float *uniforms;    // (F,L) by location or null
    float *out_logits;        // (F,L,512) by location or null
    const int32_t *step_ptr;
    float temperature;
    int L;
};

__global__ __launch_bounds__(512) void k_sample(SampleArgs a)
{
    __shared__ float sh[NCLS];
    __shared__ float red[8];
    const int f = blockIdx.x, o = threadIdx.x;
    const int q = a.order[(size_t)f * a.L + *a.step_ptr];
    const size_t loc = (size_t)f * a.L + q;
    const float lg = a.logits[(size_t)f * NCLS + o];
    if (a.out_logits) a.out_logits[loc * NCLS + o] = lg;
    if (!a.region[loc]) return;
    if (a.forced) {
        if (o == 0) a.codes[loc] = a.forced[loc];
        return;
    }
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
    const bool below = sh[o] <= target;  // chosen = number of classes whose cdf is <= target
    const int cnt = __syncthreads_count(below);
    if (o == 0) a.codes[loc] = min(cnt, NCLS - 1);
}

__global__ void k_mask_codes(int32_t *codes, const uint8_t *region, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && region[i]) codes[i] = -1;
}

__global__ void k_set_step(int32_t *step, int v) { *step = v; }
__global__ void k_inc_step(int32_t *step) { *step += 1; }

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
    float *partial = nullptr, *col_logits = nullptr;
    int32_t *step = nullptr;
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

struct Ctx {
    ps_pixelcnn *h;
    bool column;
    int F, nitems;
    const int32_t *codes, *order;
    Masks m;
    hipStream_t st;
    bool logits_nchw = false;  // grid mode: (F,512,H,W) like the reference, else (nitems,512)
};

enum { TAG_GEMM = 0, TAG_POST = 1, TAG_UINIT = 2, TAG_POST_LOGITS = 3, TAG_SAMPLE = 4 };

template <typename Fn>
void timed(const Ctx &c, int tag, Fn &&launch)
{
    if (!c.h->prof) { launch(); return; }
    ps_pixelcnn::ProfRec r{tag, nullptr, nullptr};
    (void)hipEventCreate(&r.e0);
    (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, c.st);
    launch();
    (void)hipEventRecord(r.e1, c.st);
    c.h->prof->push_back(r);
}

void launch_gemm(const Ctx &c, GemmArgs &a)
{
    a.H = c.h->H; a.W = c.h->W; a.L = c.h->L;
    a.nitems = c.nitems;
    a.mask_fstride = (size_t)9 * c.h->L;
    a.order = c.order;
    a.step_ptr = c.h->step;
    a.partial = c.h->partial;
    const int tiles = (c.nitems + 15) / 16;
    a.tiles_per_block = c.column ? 1 : 8;
    const dim3 grid(a.Co_pad / 16, a.ntaps, (tiles + a.tiles_per_block - 1) / a.tiles_per_block);
    if (c.h->prof) {  // algorithmic work of this launch: dense 2*Co*Cin per tap and item; weights streamed once
        for (int t = 0; t < a.ntaps; ++t) {
            const int co = a.Co_pad;  // 80, 160, 512: no padding in the PixelSynth configuration
            c.h->prof_gemm_flops += 2.0 * co * a.Cin * c.nitems;
            c.h->prof_gemm_wbytes += 4.0 * co * a.Cin;
        }
    }
    timed(c, TAG_GEMM, [&]() {
        if (c.column) hipLaunchKernelGGL(k_gemm<true>, grid, dim3(64), 0, c.st, a);
        else hipLaunchKernelGGL(k_gemm<false>, grid, dim3(64), 0, c.st, a);
    });
}

void conv_taps(GemmArgs &a, const float *in, int ld, const float *wp, int Cin, int Co_pad, int dil, const float *mask)
{
    a.ntaps = 9;
    a.Cin = Cin;
    a.Co_pad = Co_pad;
    a.mask = mask;
    const size_t per_tap = (size_t)Cin * Co_pad;
    for (int t = 0; t < 9; ++t)
        a.tap[t] = GemmTap{in, wp + t * per_tap, (t / 3 - 1) * dil, (t % 3 - 1) * dil, t, ld};
}

template <int KIND>
void launch_post(const Ctx &c, PostArgs &p)
{
    p.partial = c.h->partial;
    p.nitems = c.nitems;
    p.L = c.h->L;
    p.order = c.order;
    p.step_ptr = c.h->step;
    timed(c, TAG_POST, [&]() {
        if (c.column) hipLaunchKernelGGL((k_post<KIND, true>), dim3(c.nitems), dim3(128), 0, c.st, p);
        else hipLaunchKernelGGL((k_post<KIND, false>), dim3(c.nitems), dim3(128), 0, c.st, p);
    });
}

// One evaluation of the network over the context's items (whole grid or one column per frame).
// logits: (F,512,H,W) in grid mode, (F,512) in column mode; may be null in grid mode (cache build only).
void run_network(const Ctx &c, float *logits)
{
    ps_pixelcnn *h = c.h;
    {   // u_init + norm_init  (model.py:132)
        UinitArgs u{c.codes, c.m.init, (size_t)9 * h->L, h->uinit_w, h->uinit_b, h->R[0], h->E[0], h->H, h->W, h->L,
                    c.order, h->step};
        timed(c, TAG_UINIT, [&]() {
            if (c.column) hipLaunchKernelGGL(k_uinit<true>, dim3(c.nitems), dim3(128), 0, c.st, u);
            else hipLaunchKernelGGL(k_uinit<false>, dim3(c.nitems), dim3(128), 0, c.st, u);
        });
    }
    auto gated = [&](int g) {
        const ps_pixelcnn::Gated &G = h->gated[g];
        GemmArgs a{};
        conv_taps(a, h->E[G.node_in], 2 * NF, G.w_in, 2 * NF, NF, 1, c.m.und);      // conv_input (layers.py:153)
        if (G.node_skip >= 0) {                                                         // nin_skip   (layers.py:155-156)
            a.tap[9] = GemmTap{h->E[G.node_skip], G.w_skip, 0, 0, -1, 2 * NF};
            a.ntaps = 10;
        }
        launch_gemm(c, a);
        PostArgs p{};
        p.nslots = a.ntaps; p.Co_pad = NF; p.bias = G.b_in; p.bias2 = G.b_skip; p.Xout = h->X[g];
        launch_post<POST_CONVIN>(c, p);
        GemmArgs b{};
        conv_taps(b, h->X[g], 2 * NF, G.w_out, 2 * NF, 2 * NF, 1, c.m.und);          // conv_out   (layers.py:159)
        launch_gemm(c, b);
        PostArgs q{};
        q.nslots = 9; q.Co_pad = 2 * NF; q.bias = G.b_out; q.Rin = h->R[G.node_in];
        q.Rout = h->R[G.node_out]; q.Eout = h->E[G.node_out];
        launch_post<POST_GATE>(c, q);                                                   // gate + residual (:160-163)
    };
    auto dilated = [&](int d) {
        const ps_pixelcnn::Dil &D = h->dil[d];
        GemmArgs a{};
        conv_taps(a, h->R[D.node_in], NF, D.w, NF, NF, 2, c.m.dil);                   // model.py:138,148
        launch_gemm(c, a);
        PostArgs p{};
        p.nslots = 9; p.Co_pad = NF; p.bias = D.b; p.Rout = h->R[D.node_out]; p.Eout = h->E[D.node_out];
        launch_post<POST_DIL>(c, p);
    };
    gated(0); gated(1); dilated(0); gated(2); gated(3); dilated(1); gated(4); gated(5);     // up pass
    gated(6); gated(7); dilated(2); gated(8); gated(9); gated(10); dilated(3);              // down pass
    gated(11); gated(12); gated(13);
    if (!logits) return;
    GemmArgs a{};                                                                         // nin_out(elu(u)) model.py:153
    a.ntaps = 1; a.Cin = NF; a.Co_pad = NCLS; a.mask = nullptr;
    a.tap[0] = GemmTap{h->E[NNODE - 1], h->out_w, 0, 0, -1, 2 * NF};
    launch_gemm(c, a);
    PostArgs p{};
    p.partial = h->partial; p.nitems = c.nitems; p.Co_pad = NCLS; p.L = h->L; p.bias = h->out_b;
    p.logits = logits; p.logits_nchw = c.logits_nchw ? 1 : 0; p.order = c.order; p.step_ptr = h->step;
    timed(c, TAG_POST_LOGITS, [&]() {
        if (c.column) hipLaunchKernelGGL(k_post_logits<true>, dim3(c.nitems), dim3(256), 0, c.st, p);
        else hipLaunchKernelGGL(k_post_logits<false>, dim3(c.nitems), dim3(256), 0, c.st, p);
    });
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
    if ((rc = dev_alloc(h, &h->step, 4))) return fail_out(rc);
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
    Ctx c{h, false, F, F * h->L, codes, nullptr, Masks{mask_init, mask_undilated, mask_dilated}, (hipStream_t)stream, true};
    run_network(c, logits);
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
    if (step == first_step) {
        Ctx g{h, false, F, F * h->L, codes, nullptr, m, st};
        run_network(g, nullptr);
    }
    hipLaunchKernelGGL(k_set_step, dim3(1), dim3(1), 0, st, h->step, step);
    Ctx c{h, true, F, F, codes, order, m, st};
    run_network(c, logits);
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
        PS_HIP_CHECK(hipEventRecord(h->ev_in, caller));
        PS_HIP_CHECK(hipStreamWaitEvent(st, h->ev_in, 0));
    }
    const Masks m{mask_init, mask_undilated, mask_dilated};
    const size_t n = (size_t)F * h->L;
    hipLaunchKernelGGL(k_mask_codes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, codes, sample_region, n);
    Ctx g{h, false, F, F * h->L, codes, nullptr, m, st};
    // whole-grid pass: exact for every location that precedes the first sampled one; with out_logits it also
    // yields their logits, by location (the walked positions are overwritten by the column steps)
    run_network(g, out_logits);
    hipLaunchKernelGGL(k_set_step, dim3(1), dim3(1), 0, st, h->step, first_step);
    PS_LAUNCH_CHECK();
    const int nsteps = h->L - first_step;
    if (nsteps > 0) {
        Ctx c{h, true, F, F, codes, order, m, st};
        SampleArgs s{h->col_logits, codes, order, sample_region, forced, uniforms, out_logits, h->step, temperature, h->L};
        auto body = [&]() {
            run_network(c, h->col_logits);
            hipLaunchKernelGGL(k_sample, dim3(F), dim3(512), 0, st, s);
            hipLaunchKernelGGL(k_inc_step, dim3(1), dim3(1), 0, st, h->step);
        };
        if (h->use_graph) {
            release_graph(h);
            PS_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            body();
            PS_HIP_CHECK(hipStreamEndCapture(st, &h->graph));
            PS_HIP_CHECK(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
            for (int sidx = 0; sidx < nsteps; ++sidx) PS_HIP_CHECK(hipGraphLaunch(h->graph_exec, st));
        } else {
            for (int sidx = 0; sidx < nsteps; ++sidx) body();
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
    hipLaunchKernelGGL(k_set_step, dim3(1), dim3(1), 0, st, h->step, step);
    Ctx c{h, true, F, F, codes, order, Masks{mask_init, mask_undilated, mask_dilated}, st};
    run_network(c, h->col_logits);  // untimed warm-up
    h->prof = &recs;
    h->prof_gemm_flops = h->prof_gemm_wbytes = 0.0;
    for (int r = 0; r < reps; ++r) run_network(c, h->col_logits);
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
    conv_taps(a, xcl, Cp, wp, Cp, Cop, dilation, mask);
    a.H = H; a.W = W; a.L = L; a.nitems = B * L; a.mask_fstride = mask_batch_stride;
    a.order = nullptr; a.step_ptr = nullptr; a.partial = partial; a.tiles_per_block = 8;
    const int tiles = (a.nitems + 15) / 16;
    hipLaunchKernelGGL(k_gemm<false>, dim3(Cop / 16, 9, (tiles + 7) / 8), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_reduce_nchw, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, partial, bias, B, Co, Cop, L, y);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // extern "C"
