// conv1x1.hip -- the 1 x 1 convolutions either side of the hot path for gfx950 (MI355X), fp32 in / fp32 out on
// the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation -- no split operands, no overflow guard).
//
// Replaces torch.nn.Conv2d(Ci, Co, 1) -- the projection branch of a ResNet_Block (models/layers/blocks.py:46-57: 4 -> 64, 64 -> 128,
// 128 -> 256, 256 -> 128, 128 -> 128, 128 -> 3 in the refinement decoder) and the VQ-VAE's ResBlock / quantize_conv_t projections
// (models/vqvae2/vqvae.py:81-97, :262) -- which MIOpen ran as implicit-GEMM kernels, the batch cut at 2 GiB and concatenated again:
// the decoder's six layers 4.8 -> 3.3 ms per 128 views (tools/conv1x1_time.py), bound by their bytes (3-5 TB/s) at the wide ends.
//
// On channels-last memory a 1 x 1 convolution is y (P, Co) = x (P, Ci) . w^T (Co, Ci), P = B H W pixels: a skinny GEMM whose K is
// the whole weight.  So:
//   * the weights live in LDS for the life of a workgroup, in MFMA A-fragment order (Co x Ci x 4 B <= 128 KB: every layer above);
//   * a wave takes 32 pixels at a time (two B tiles): lane (pixel i, kk) reads the 16 bytes x[p_i][16 c + 4 kk ..] of every 16-channel
//     chunk c -- straight from memory into the B registers, nobody else needs them -- and walks the output tiles: one ds_read_b128 of
//     weights feeds 8 MFMAs, the 16 x 16 results of a tile leave as 16 bytes per lane (y[p_i][16 t + 4 kk ..]);
//   * persistent workgroups of eight waves, the grid a multiple of the compute units.
// Memory-side the layers move 0.05-0.5 ms of bytes per 128 views each; the matrix pipe is busy about as long.
// Sum order: channels 16 c + 4 kk + j, j innermost, accumulated over c in fp32 (deterministic; not torch's order -- compared against an
// fp64 convolution in tests/test_networks_gpu.py).
#include "ps_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C1_WAVES = 8, C1_THREADS = 64 * C1_WAVES, C1_PT = 2;   // waves per workgroup; pixel tiles (of 16) per wave and trip
constexpr int C1_MAX_LDS_FLOATS = 32768;                             // 128 KB of weights

struct C1Args {
    const float *x, *w;
    float *y;
    size_t npix;
    int Ci, Co, cot;      // cot = output tiles of 16 (Co rounded up)
    int ldx;              // floats between the rows of x (>= Ci: the first Ci channels of a wider activation)
    const float *bias;    // (Co) or null
    const float *res;     // (npix, Co) or null: added on the way out
    int relu_in, relu_res;   // max(x, 0) on the way in; max(res, 0) instead of res
};

// a tile's 4 results of one lane -- y[p][co .. co + 3] -- with the bias and the other branch added, stored
__device__ __forceinline__ void c1_store(const C1Args &a, size_t p, int co, f32x4 acc, bool vec_out)
{
    const int Co = a.Co;
    float *dst = a.y + p * (size_t)Co + co;
    if (vec_out) {
        if (co >= Co) return;
        if (a.bias) acc = acc + *(const f32x4 *)(a.bias + co);
        if (a.res) {
            f32x4 r = *(const f32x4 *)(a.res + p * (size_t)Co + co);
            if (a.relu_res) r = __builtin_elementwise_max(r, (f32x4){0.0f, 0.0f, 0.0f, 0.0f});
            acc = acc + r;
        }
        *(f32x4 *)dst = acc;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (co + r >= Co) continue;
            float v = acc[r];
            if (a.bias) v = v + a.bias[co + r];
            if (a.res) {
                const float q = a.res[p * (size_t)Co + co + r];
                v = v + (a.relu_res ? fmaxf(q, 0.0f) : q);
            }
            dst[r] = v;
        }
    }
}

// NCH > 0: Ci = 16 NCH.  NCH = 0: Ci = 4 (one MFMA per tile: lane (i, kk) carries channel kk)
template <int NCH>
__global__ __launch_bounds__(C1_THREADS) void k_conv1x1(C1Args a)
{
    extern __shared__ f32x4 sW[];            // NCH > 0: [tile][chunk][lane] f32x4;  NCH = 0: floats [tile][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kk = lane >> 4;
    const int Ci = a.Ci, Co = a.Co;
    (void)Ci;
    if (NCH > 0) {
        for (int e = tid; e < a.cot * NCH * 64; e += C1_THREADS) {
            const int l = e & 63, c = (e >> 6) % NCH, t = (e >> 6) / NCH, co = t * 16 + (l & 15);
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (co < Co) v = *(const f32x4 *)(a.w + (size_t)co * Ci + 16 * c + 4 * (l >> 4));
            sW[e] = v;
        }
    } else {
        float *sWf = (float *)sW;
        for (int e = tid; e < a.cot * 64; e += C1_THREADS) {
            const int l = e & 63, t = e >> 6, co = t * 16 + (l & 15);
            sWf[e] = co < Co ? a.w[(size_t)co * 4 + (l >> 4)] : 0.0f;
        }
    }
    __syncthreads();
    const bool vec_out = (Co & 3) == 0;
    const size_t ntrips = (a.npix + 16 * C1_PT * C1_WAVES - 1) / (16 * C1_PT * C1_WAVES);
    for (size_t trip = blockIdx.x; trip < ntrips; trip += gridDim.x) {
        const size_t p0 = (trip * C1_WAVES + wave) * (16 * C1_PT);
        if (p0 >= a.npix) continue;
        size_t p[C1_PT];
#pragma unroll
        for (int t = 0; t < C1_PT; ++t) p[t] = p0 + 16 * t + i;
        if (NCH > 0) {
            f32x4 b[C1_PT][NCH];
#pragma unroll
            for (int t = 0; t < C1_PT; ++t) {
                const float *row = a.x + (p[t] < a.npix ? p[t] : a.npix - 1) * (size_t)a.ldx + 4 * kk;   // (a pixel past the end: loaded, never stored)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    b[t][c] = *(const f32x4 *)(row + 16 * c);
                    if (a.relu_in) b[t][c] = __builtin_elementwise_max(b[t][c], (f32x4){0.0f, 0.0f, 0.0f, 0.0f});
                }
            }
            for (int ot = 0; ot < a.cot; ++ot) {
                const f32x4 *A = sW + (size_t)ot * NCH * 64 + lane;
                f32x4 acc[C1_PT];
#pragma unroll
                for (int t = 0; t < C1_PT; ++t) acc[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const f32x4 w4 = A[c * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int t = 0; t < C1_PT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[j], b[t][c][j], acc[t], 0, 0, 0);
                }
                const int co = ot * 16 + 4 * kk;
#pragma unroll
                for (int t = 0; t < C1_PT; ++t)
                    if (p[t] < a.npix) c1_store(a, p[t], co, acc[t], vec_out);
            }
        } else {
            const float *sWf = (const float *)sW;
            float b[C1_PT];
#pragma unroll
            for (int t = 0; t < C1_PT; ++t) {
                b[t] = a.x[(p[t] < a.npix ? p[t] : a.npix - 1) * (size_t)a.ldx + kk];
                if (a.relu_in) b[t] = fmaxf(b[t], 0.0f);
            }
            for (int ot = 0; ot < a.cot; ++ot) {
                const float w1 = sWf[ot * 64 + lane];
                const int co = ot * 16 + 4 * kk;
#pragma unroll
                for (int t = 0; t < C1_PT; ++t) {
                    const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, b[t], (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
                    if (p[t] < a.npix) c1_store(a, p[t], co, acc, vec_out);
                }
            }
        }
    }
}

template <int NCH>
int launch_conv1x1(const C1Args &a, hipStream_t st)
{
    const size_t lds = NCH > 0 ? (size_t)a.cot * NCH * 64 * sizeof(f32x4) : (size_t)a.cot * 64 * sizeof(float);
    // per instantiation AND per device: the dynamic-LDS limit is a property of the function on a device (a process may drive several GPUs)
    constexpr int MAX_DEV = 64;
    static bool attr_set[MAX_DEV] = {};
    static int cus_of[MAX_DEV] = {};
    int dev = 0;
    PS_HIP_CHECK(hipGetDevice(&dev));
    PS_REQUIRE(dev >= 0 && dev < MAX_DEV, "conv1x1: device %d", dev);
    if (!attr_set[dev]) {
        PS_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv1x1<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, C1_MAX_LDS_FLOATS * 4));
        PS_HIP_CHECK(hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev));
        attr_set[dev] = true;
    }
    const int cus = cus_of[dev] > 0 ? cus_of[dev] : 256;
    const size_t ntrips = (a.npix + 16 * C1_PT * C1_WAVES - 1) / (16 * C1_PT * C1_WAVES);
    // workgroups per compute unit the weights leave room for (160 KB of LDS; two waves per SIMD each)
    const int per_cu = lds > 80 * 1024 ? 1 : lds > 40 * 1024 ? 2 : 4;
    const int grid = (int)std::min<size_t>(ntrips, (size_t)cus * per_cu);
    hipLaunchKernelGGL(k_conv1x1<NCH>, dim3(grid), dim3(C1_THREADS), lds, st, a);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // namespace

extern "C" {

int ps_conv1x1_takes(int Ci, int Co)
{
    const int cot = (Co + 15) / 16;
    const bool k = Ci == 4 || Ci == 32 || Ci == 64 || Ci == 128 || Ci == 256;
    return k && Co >= 1 && (size_t)cot * 16 * Ci <= (size_t)C1_MAX_LDS_FLOATS;
}

int ps_conv1x1_ex_nhwc_f32(const float *x, int ldx, const float *w, const float *bias, const float *res, int flags, size_t npix, int Ci,
                           int Co, float *y, void *stream)
{
    PS_REQUIRE(x && w && y, "conv1x1: null pointer");
    PS_REQUIRE(npix > 0, "conv1x1: no pixels");
    PS_REQUIRE(ps_conv1x1_takes(Ci, Co), "conv1x1: Ci in {4, 32, 64, 128, 256} and ceil16(Co) * Ci <= 32768 required (Ci = %d, Co = %d)", Ci, Co);
    PS_REQUIRE(ldx >= Ci && ldx % 4 == 0, "conv1x1: ldx >= Ci and a multiple of 4 required (ldx = %d)", ldx);
    PS_REQUIRE((flags & ~3) == 0, "conv1x1: unknown flags %d", flags);
    PS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0 &&
               ((uintptr_t)res & 15) == 0, "conv1x1: 16-byte aligned buffers required");
    C1Args a{x, w, y, npix, Ci, Co, (Co + 15) / 16, ldx, bias, res, flags & 1, (flags >> 1) & 1};
    hipStream_t st = (hipStream_t)stream;
    switch (Ci) {
    case 4: return launch_conv1x1<0>(a, st);
    case 32: return launch_conv1x1<2>(a, st);
    case 64: return launch_conv1x1<4>(a, st);
    case 128: return launch_conv1x1<8>(a, st);
    default: return launch_conv1x1<16>(a, st);
    }
}

int ps_conv1x1_nhwc_f32(const float *x, const float *w, size_t npix, int Ci, int Co, float *y, void *stream)
{
    return ps_conv1x1_ex_nhwc_f32(x, Ci, w, nullptr, nullptr, 0, npix, Ci, Co, y, stream);
}

}  // extern "C"
