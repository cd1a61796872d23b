// conv_f16x3.hip -- the 3 x 3 convolutions of the refinement decoder (models/networks/architectures.py:126-167,
// models/layers/blocks.py:34-73) as an implicit GEMM on the fp16 matrix pipe of gfx950 (MI355X), fp32 in, fp32 out.
//
// Why not fp32 MFMA: v_mfma_f32_*_f32 runs at the vector rate (157 TFLOP/s); MIOpen's fp32 implicit GEMM already sits at two thirds of
// that, and the decoder is four fifths of an end-to-end frame.  v_mfma_f32_32x32x16_f16 is 16 x faster, its products of two fp16
// values are exact and its sums are fp32 (tools/mfma_f16_probe.hip, which also pins the lane maps used below and that SUBNORMAL
// fp16 inputs take part in the product).  So every operand is split   v = hi + lo,  hi = fp16(v),  lo = fp16(v - hi)
// (22 of fp32's 24 mantissa bits; below 2^-3 the low half is subnormal and the split is exact to 2^-25 absolute) and a product is
// three MFMAs:  a * b ~= a.hi * b.hi + a.hi * b.lo + a.lo * b.hi;  the dropped a.lo * b.lo is below 2^-22 relative.  The result
// differs from an fp32 convolution's by about as much as two fp32 summation orders differ from each other (tests/test_nets_gpu.py
// holds both against an fp64 reference).  fp16 overflows at 65504: an activation beyond 65000 raises `overflow` (the layers in
// front of these convolutions are normalisations; the caller checks the flag and falls back).
//
// One workgroup = 8 waves = a 16 x 16 tile of output pixels x 128 output channels; a wave owns 4 rows x 16 pixels x 64 channels as
// 2 x 2 MFMA tiles of 32 x 32 (64 accumulator registers).  K runs over (32 input channels) x (9 taps) x ...:
//   * the tile's 18 x 18 input patch of the next 32 channels is fetched into registers while the current 32 are multiplied, then
//     (optionally through the block's norm + ReLU: y = max(x * scale[b][c] - shift[b][c], 0) -- the pass in front of every one of these
//     convolutions, models/layers/normalization.py:21-47 -- fused here instead of a round trip through memory) split and written to LDS
//     as planes [hi | lo][k / 8][pixel][8 fp16]: an A fragment of any tap is one conflict-free ds_read_b128 at a shifted pixel;
//   * the weights come packed in fragment order (ps_conv3x3_f16x3_pack) and reach LDS by LDS-DMA (global_load_lds_dwordx4), 16 KB
//     per tap, two taps ahead in a ring of three buffers, waited for with counted s_waitcnt vmcnt; one raw barrier per tap.
#include "ps_common.h"

namespace psconv {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TH = 16, TW = 16;            // output pixels per workgroup
constexpr int PW = TW + 2, PP = (TH + 2) * PW;   // the patch: 18 x 18 = 324 pixels
constexpr int CK = 32;                     // input channels per chunk (two MFMA steps of 16)
constexpr int COT = 128;                   // output channels per workgroup
constexpr int NT = 512;                    // threads
constexpr int A_PLANE = PP * 16;           // bytes: [pixel][8 fp16]
constexpr int A_BUF = 8 * A_PLANE;         // planes [part 2][ks 2][kb 2]
constexpr int B_TAP = CK * COT * 4;        // 16 KB: [ks 2][ct 4][part 2][lane 64][8 fp16]
constexpr int LDS_B = 2 * A_BUF;
constexpr int LDS_TOTAL = LDS_B + 3 * B_TAP;   // 132 096 bytes
constexpr int NA = (PP + 63) / 64;         // patch pixels per thread: 6 (8 threads = 32 channels per pixel, 64 pixels per sweep)

#define PS_GCP(p) ((const __attribute__((address_space(1))) void *)(p))

// LDS-DMA by hand (the builtin makes the compiler wait vmcnt(0) in front of the next read of the array, which is the ring being consumed)
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_base)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(PS_GCP(gsrc)), "s"(lds_base) : "memory", "m0");
}
__device__ __forceinline__ f32x4 load16_async(const void *g)   // the result is there after a vmcnt wait the CALLER places
{
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(PS_GCP(g)) : "memory");
    return r;
}
template <int N> __device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct ConvArgs {
    const float *x;        // (B, H, W, Ci)
    const float *scale;    // (B, Ci) or null
    const float *shift;
    const char *wp;        // packed weights
    float *y;              // (B, H, W, Co)
    int *overflow;
    int H, W, Ci, Co, tiles_x, tiles_per_frame, ncb, nblocks;
};

// FUSE: norm + ReLU on the way in
template <bool FUSE> __global__ __launch_bounds__(NT) void k_conv3x3_f16x3(ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (tile, output-channel block): block b runs on XCD b % 8; an XCD takes a contiguous run of (tile, cb) pairs, cb fastest,
    // so that neighbouring patches and the two channel blocks of one patch meet in one L2
    int L = blockIdx.x;
    if ((a.nblocks & 7) == 0) L = (L & 7) * (a.nblocks >> 3) + (L >> 3);
    const int cb = L % a.ncb, tile = L / a.ncb;
    const int b = tile / a.tiles_per_frame, tf = tile - b * a.tiles_per_frame;
    const int ty0 = (tf / a.tiles_x) * TH, tx0 = (tf % a.tiles_x) * TW;
    const int H = a.H, W = a.W, Ci = a.Ci, nchunk = Ci / CK, G = nchunk * 9;
    const float *xb = a.x + (size_t)b * H * W * Ci;

    // ---- staging role of this thread: channels 4 c4 .. 4 c4 + 3 of the chunk, patch pixels s, s + 64, ...
    const int c4 = tid & 7, s = tid >> 3;
    unsigned xoff[NA];
    unsigned valid = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int p = s + 64 * i, pr = p / PW, pc = p - pr * PW, iy = ty0 + pr - 1, ix = tx0 + pc - 1;
        const bool ok = p < PP && iy >= 0 && iy < H && ix >= 0 && ix < W;
        xoff[i] = ok ? (unsigned)((iy * W + ix) * Ci + 4 * c4) : 0u;
        valid |= (unsigned)ok << i;
    }
    const unsigned a_wr = (unsigned)(((c4 >> 1) & 1) + 2 * (c4 >> 2)) * A_PLANE + (c4 & 1) * 8 + s * 16;   // + 64 * 16 per sweep; lo: + 4 planes
    f32x4 ra[NA], rsc, rsh;
    int over = 0;
    auto fetch = [&](int c) {   // chunk c of the patch into registers (asynchronous)
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = load16_async(xb + xoff[i] + c * CK);
        if (FUSE) {
            rsc = load16_async(a.scale + (size_t)b * Ci + c * CK + 4 * c4);
            rsh = load16_async(a.shift + (size_t)b * Ci + c * CK + 4 * c4);
        }
    };
    constexpr int NFETCH = NA + (FUSE ? 2 : 0);
    auto pin = [&]() {   // the fetched registers are read only below this point (which the caller has put behind a vmcnt wait)
        static_assert(NA == 6, "pin() names the registers one by one");
        asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]));
        if (FUSE) asm volatile("" : "+v"(rsc), "+v"(rsh));
    };
    auto stash = [&](int buf) {   // registers -> split -> LDS
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (s + 64 * i >= PP) continue;
            f32x4 v = ra[i];
            if (FUSE) v = __builtin_elementwise_max(v * rsc - rsh, (f32x4){0.f, 0.f, 0.f, 0.f});
            if (!((valid >> i) & 1)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            h4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                over |= !(__builtin_fabsf(v[j]) <= 65000.f);
                hi[j] = (_Float16)v[j];
                lo[j] = (_Float16)(v[j] - (float)hi[j]);
            }
            char *dst = lds + buf * A_BUF + a_wr + i * 64 * 16;
            *(h4 *)dst = hi;
            *(h4 *)(dst + 4 * A_PLANE) = lo;
        }
    };
    // ---- weight ring: tap g of the (chunk, tap) sequence -> buffer g % 3; this wave copies pieces `wave` and `wave + 8` of its 16 KB
    const char *wsrc = a.wp + (size_t)cb * G * B_TAP + (size_t)wave * 1024 + lane * 16;
    auto dma_tap = [&](int g, int buf) {
        const unsigned dst = (unsigned)(LDS_B + buf * B_TAP + wave * 1024);
        dma16(wsrc + (size_t)g * B_TAP, dst);
        dma16(wsrc + (size_t)g * B_TAP + 8 * 1024, dst + 8 * 1024);
    };
    // ---- MFMA role: rows 4 pg .. 4 pg + 3 of the tile (two M tiles of 2 rows x 16 pixels), channel tiles 2 chh, 2 chh + 1
    const int pg = wave & 3, chh = wave >> 2, m = lane & 31, kb = lane >> 5;
    const unsigned a_rd = (unsigned)(((4 * pg + (m >> 4)) * PW + (m & 15)) * 16 + kb * A_PLANE);
    const unsigned b_rd = (unsigned)(LDS_B + lane * 16 + chh * 2 * 2048);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    fetch(0);
    dma_tap(0, 0);
    if (G > 1) dma_tap(1, 1);
    wait_vm<4>();   // (G >= 9: two taps of two copies each are newer than the patch)
    pin();
    stash(0);

    for (int c = 0; c < nchunk; ++c) {
        const bool more = c + 1 < nchunk;
        const char *Ab = lds + (c & 1) * A_BUF + a_rd;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int g = c * 9 + tap;
            // tap g's weights have landed: everything older than the requests that may still be in flight
            // (tap g + 1's two copies; the next chunk's patch, requested at tap 0 behind tap 2's copies)
            if (g + 1 >= G) wait_vm<0>();
            else if ((tap == 1 || tap == 2) && more) wait_vm<2 + NFETCH>();
            else wait_vm<2>();
            lds_barrier();
            if (g + 2 < G) dma_tap(g + 2, (tap + 2) % 3);
            if (tap == 0 && more) fetch(c + 1);
            const char *Bb = lds + b_rd + (tap % 3) * B_TAP;
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const char *p = Ab + ks * 2 * A_PLANE + ((2 * mt + ky) * PW + kx) * 16;
                    ah[mt] = *(const h8 *)p;
                    al[mt] = *(const h8 *)(p + 4 * A_PLANE);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const char *p = Bb + ks * 8192 + nt * 2048;
                    bh[nt] = *(const h8 *)p;
                    bl[nt] = *(const h8 *)(p + 1024);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    }
            }
            if (tap == 8 && more) {   // (the patch was waited for at tap 3: tap 3's weights were requested behind it)
                pin();
                stash((c + 1) & 1);
            }
        }
    }
    // ---- out: D[row][col]: col = lane & 31 = channel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = pixel of the M tile
    float *yb = a.y + ((size_t)b * H * W) * a.Co + cb * COT + chh * 64 + m;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
            const int oy = ty0 + 4 * pg + 2 * mt + (row >> 4), ox = tx0 + (row & 15);
            float *yp = yb + ((size_t)oy * W + ox) * a.Co;
            yp[0] = acc[mt][0][r];
            yp[32] = acc[mt][1][r];
        }
    if (over) *a.overflow = 1;
}

// (Co, 3, 3, Ci) fp32 -> [cb][chunk][tap][ks][ct][part][lane][8] fp16: B[k][n] of the MFMA: n = lane & 31 = channel 32 ct + n of block cb,
// k = 8 (lane >> 5) + j = input channel 32 chunk + 16 ks + k
__global__ __launch_bounds__(256) void k_pack(const float *__restrict__ w, int Co, int Ci, h8 *__restrict__ out, int total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int lane = i & 63;
    int q = i >> 6;
    const int ct = q & 3; q >>= 2;
    const int ks = q & 1; q >>= 1;
    const int tap = q % 9; q /= 9;
    const int nchunk = Ci / CK, chunk = q % nchunk, cb = q / nchunk;
    const int co = cb * COT + ct * 32 + (lane & 31), ci = chunk * CK + ks * 16 + (lane >> 5) * 8;
    const float *src = w + ((size_t)co * 9 + tap) * Ci + ci;
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = src[j];
        hi[j] = (_Float16)v;
        lo[j] = (_Float16)(v - (float)hi[j]);
    }
    h8 *dst = out + ((size_t)(i >> 6) * 2) * 64 + lane;
    dst[0] = hi;
    dst[64] = lo;
}

}  // namespace psconv

extern "C" {

size_t ps_conv3x3_f16x3_packed_bytes(int Co, int Ci) { return (size_t)9 * Co * Ci * 4; }

int ps_conv3x3_f16x3_pack(const float *w, int Co, int Ci, void *packed, void *stream)
{
    PS_REQUIRE(w && packed, "conv3x3_f16x3_pack: null pointer");
    PS_REQUIRE(Co > 0 && Co % psconv::COT == 0 && Ci > 0 && Ci % psconv::CK == 0,
               "conv3x3_f16x3_pack: Co a multiple of 128 and Ci a multiple of 32 required (Co = %d, Ci = %d)", Co, Ci);
    const int total = (Co / psconv::COT) * (Ci / psconv::CK) * 9 * 2 * 4 * 64;
    hipLaunchKernelGGL(psconv::k_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, Co, Ci, (psconv::h8 *)packed, total);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_conv3x3_f16x3_nhwc(const float *x, const float *scale, const float *shift, const void *packed, int B, int H, int W, int Ci, int Co,
                          float *y, int *overflow, void *stream)
{
    using namespace psconv;
    PS_REQUIRE(x && packed && y && overflow, "conv3x3_f16x3: null pointer");
    PS_REQUIRE((scale == nullptr) == (shift == nullptr), "conv3x3_f16x3: scale and shift come together");
    PS_REQUIRE(B > 0 && H > 0 && W > 0 && H % TH == 0 && W % TW == 0, "conv3x3_f16x3: H and W multiples of 16 required (H = %d, W = %d)", H, W);
    PS_REQUIRE(Co > 0 && Co % COT == 0 && Ci > 0 && Ci % CK == 0,
               "conv3x3_f16x3: Co a multiple of 128 and Ci a multiple of 32 required (Co = %d, Ci = %d)", Co, Ci);
    PS_REQUIRE((size_t)H * W * Ci < ((size_t)1 << 31), "conv3x3_f16x3: a frame of %d x %d x %d does not fit 32-bit offsets", H, W, Ci);
    ConvArgs a;
    a.x = x; a.scale = scale; a.shift = shift; a.wp = (const char *)packed; a.y = y; a.overflow = overflow;
    a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
    a.tiles_x = W / TW; a.tiles_per_frame = (H / TH) * a.tiles_x; a.ncb = Co / COT;
    const size_t nb = (size_t)B * a.tiles_per_frame * a.ncb;
    PS_REQUIRE(nb < ((size_t)1 << 31), "conv3x3_f16x3: too many tiles");
    a.nblocks = (int)nb;
    static bool attr_set[2] = {false, false};   // (per process; the attribute is per function)
    if (scale) {
        if (!attr_set[1]) { PS_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv3x3_f16x3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL)); attr_set[1] = true; }
        hipLaunchKernelGGL(k_conv3x3_f16x3<true>, dim3(a.nblocks), dim3(NT), LDS_TOTAL, (hipStream_t)stream, a);
    } else {
        if (!attr_set[0]) { PS_HIP_CHECK(hipFuncSetAttribute((const void *)k_conv3x3_f16x3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL)); attr_set[0] = true; }
        hipLaunchKernelGGL(k_conv3x3_f16x3<false>, dim3(a.nblocks), dim3(NT), LDS_TOTAL, (hipStream_t)stream, a);
    }
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // extern "C"
