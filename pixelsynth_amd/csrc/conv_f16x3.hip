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
//     per tap, three taps ahead in a ring of four buffers, waited for with counted s_waitcnt vmcnt; one raw barrier per tap.
// Tuning builds (never the product library): -DPS_CONV_EXP=<bits> -- 1: no MFMAs, 2: fragments read once, 4: no barriers, 8: no weight
// copies (1-8: results invalid, for timing what the rest costs), 16: every workgroup's shader cycles and 100 MHz ticks summed
// (ps_conv_debug_clock; tools/conv_f16x3_time.py --clock); -DPS_CONV_NO_SCHED: no sched_group_barrier pattern.  PS_CONV_WGS (environment):
// workgroups per launch.
#include "ps_common.h"

#include <algorithm>
#include <cstdlib>

namespace psconv {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TH = 16, TW = 16;            // output pixels per workgroup.  (An MFMA tile = two rows of 16 pixels, 288 bytes apart in LDS: under
                                           // ds_read_b128's lane groups every A read is a 2-way bank conflict, SQ_LDS_BANK_CONFLICT = 35 % of the
                                           // LDS cycles.  8 x 32 tiles -- one image row per MFMA tile, conflict-free -- measured equal to 2 %
                                           // slower: the larger halo costs what the conflicts did; the kernel is not bound by LDS.)
constexpr int PW = TW + 2, PP = (TH + 2) * PW;   // the patch: 18 x 18 = 324 pixels
constexpr int CK = 32;                     // input channels per chunk (two MFMA steps of 16)
constexpr int COT = 128;                   // output channels per workgroup
constexpr int NT = 512;                    // threads
constexpr int A_PLANE = PP * 16;           // bytes: [pixel][8 fp16]
constexpr int A_BUF = 8 * A_PLANE;         // planes [part 2][ks 2][kb 2]
constexpr int B_TAP = CK * COT * 4;        // 16 KB: [ks 2][ct 4][part 2][lane 64][8 fp16]
constexpr int LDS_B = 2 * A_BUF;
constexpr int LDS_TOTAL = LDS_B + 4 * B_TAP;   // 148 480 bytes
constexpr int NA = (PP + 63) / 64;         // patch pixels per thread: 6 (8 threads = 32 channels per pixel, 64 pixels per sweep)

#define PS_GCP(p) ((const __attribute__((address_space(1))) void *)(p))

// LDS-DMA by hand (the builtin makes the compiler wait vmcnt(0) in front of the next read of the array, which is the ring being consumed)
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_base)
{
#if defined(PS_CONV_EXP) && (PS_CONV_EXP & 8)   // tuning build: no weight traffic (results invalid)
    return;
#endif
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
#if defined(PS_CONV_EXP) && (PS_CONV_EXP & 8)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
#endif
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
__device__ __forceinline__ void lds_barrier()
{
#if defined(PS_CONV_EXP) && (PS_CONV_EXP & 4)   // tuning build: no barrier (results invalid)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

struct ConvArgs {
    const float *x;        // (B, H, W, Ci)
    const float *scale;    // (B, Ci) or null
    const float *shift;
    const char *wp;        // packed weights
    float *y;              // (B, H, W, Co)
    const float *bias;     // (Co) or null: added to the results
    const float *res;      // (B, H, W, Co) or null: added to the results (the block's other branch)
    int *overflow;
    int H, W, Ci, Co, tiles_x, tiles_per_frame, ncb, nblocks;
    int s2d_C;             // > 0: x is (B, 2 H, 2 W, s2d_C), read as its space-to-depth form (B, H, W, Ci = 4 s2d_C), channel (sy, sx, c)
    int co_live;           // output channels [co_live, Co) have all-zero weights (padding): 64-channel halves of a block that lie there are not multiplied
    int d2s_C;             // > 0: y is (B, 2 H, 2 W, d2s_C), written from the depth-to-space of (B, H, W, Co = 4 d2s_C), channel (py, px, c)
};

#if defined(PS_CONV_EXP) && (PS_CONV_EXP & 16)   // tuning build: shader cycles and 100 MHz ticks of every workgroup's first wave, summed
__device__ unsigned long long g_conv_clk[2];
#endif

// One (tile, output-channel block) of the launch: which patch, which weights, where the results go.
struct Item {
    const float *xb;      // the frame
    const char *w;        // the channel block's packed weights (+ this wave's piece)
    int b, cb, ty0, tx0;
};

// FUSE: norm + ReLU on the way in.  PERSISTENT workgroups: one per compute unit (148 KB of LDS), each walking its share of the
// launch's (tile, channel block) items with the chunk pipeline running ACROSS items -- the first patch of the next item is fetched
// and stashed during the last chunk of this one, the weight ring never drains, and an item's results are stored behind its last
// MFMA while the next item's first taps multiply.  Worth 1-2 % over one workgroup per item (PS_CONV_WGS=0), more on the two-chunk
// layers: the kernel is bound by its MFMA stream (with the fragment reads switched off it takes as long), and that stream by the
// chip's power -- every SIMD issuing v_mfma_f32_32x32x16_f16 on random operands and nothing else holds 1.72 GHz = 1.77 PFLOP/s,
// 0.71 of the nominal 2.5 (tools/mfma_f16_clock_probe.hip; 2.39 GHz on constant operands); this kernel reaches 1.1-1.3.
// PERM: the space-to-depth read / depth-to-space store of ps_conv3x3_f16x3_ex_nhwc compiled in (an instantiation of its own: the runtime
// branches cost the plain form 1 % of the decoder's time).
template <bool FUSE, bool PERM = false> __global__ __launch_bounds__(NT) void k_conv3x3_f16x3(ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, Ci = a.Ci, nchunk = Ci / CK, G = nchunk * 9;
    const int s2dC = PERM ? a.s2d_C : 0, d2sC = PERM ? a.d2s_C : 0;
    // block -> items: block b runs on XCD b % 8; an XCD takes a contiguous run of (tile, cb) pairs, cb fastest, so that neighbouring
    // patches and the two channel blocks of one patch meet in one L2; the XCD's workgroups deal that run among themselves
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, J = gridDim.x >> 3;
    const int lo_x = (int)(((long long)a.nblocks * xcd) >> 3), hi_x = (int)(((long long)a.nblocks * (xcd + 1)) >> 3);
    const int nitems = lo_x + j0 < hi_x ? (hi_x - lo_x - j0 + J - 1) / J : 0;
    if (nitems == 0) return;
    auto item_at = [&](int i) {
        const int L = lo_x + j0 + i * J;
        Item it;
        it.cb = L % a.ncb;
        const int tile = L / a.ncb;
        it.b = tile / a.tiles_per_frame;
        const int tf = tile - it.b * a.tiles_per_frame;
        it.ty0 = (tf / a.tiles_x) * TH;
        it.tx0 = (tf % a.tiles_x) * TW;
        it.xb = a.x + (size_t)it.b * H * W * Ci;
        it.w = a.wp + (size_t)it.cb * G * B_TAP + (size_t)wave * 1024 + lane * 16;
        return it;
    };

    // ---- staging role of this thread: channels 4 c4 .. 4 c4 + 3 of the chunk, patch pixels s, s + 64, ...
    const int c4 = tid & 7, s = tid >> 3;
    unsigned xoff[NA];       // of the item whose patch is fetched next
    unsigned valid = 0;
    auto aim = [&](const Item &it) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int p = s + 64 * i, pr = p / PW, pc = p - pr * PW, iy = it.ty0 + pr - 1, ix = it.tx0 + pc - 1;
            const bool ok = p < PP && iy >= 0 && iy < H && ix >= 0 && ix < W;
            xoff[i] = ok ? (unsigned)((s2dC ? iy * W * Ci + ix * (Ci >> 1) : (iy * W + ix) * Ci) + 4 * c4) : 0u;
            valid = (valid & ~(1u << i)) | ((unsigned)ok << i);
        }
    };
    const unsigned a_wr = (unsigned)(((c4 >> 1) & 1) + 2 * (c4 >> 2)) * A_PLANE + (c4 & 1) * 8 + s * 16;   // + 64 * 16 per sweep; lo: + 4 planes
    f32x4 ra[NA], rsc, rsh;
    int over = 0;
    auto fetch = [&](const Item &it, int c) {   // chunk c of the item's patch into registers (asynchronous)
        int coff = c * CK;
        if (s2dC) {   // chunk c of the space-to-depth channels = sub-position q of the 2 x 2 block, channels cc .. cc + 31 of the real tensor
            const int q = coff / s2dC, cc = coff - q * s2dC;
            coff = ((q >> 1) * 2 * W + (q & 1)) * s2dC + cc;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = load16_async(it.xb + xoff[i] + coff);
        if (FUSE) {
            rsc = load16_async(a.scale + (size_t)it.b * Ci + c * CK + 4 * c4);
            rsh = load16_async(a.shift + (size_t)it.b * Ci + c * CK + 4 * c4);
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
    // ---- weight ring: the workgroup's taps in the order it multiplies them, tap number gg -> buffer gg & 3; this wave copies pieces
    // `wave` and `wave + 8` of a tap's 16 KB
    auto dma_tap = [&](const Item &it, int g, int gg) {   // tap g = 9 chunk + tap of the item
        const unsigned dst = (unsigned)(LDS_B + (gg & 3) * B_TAP + wave * 1024);
        dma16(it.w + (size_t)g * B_TAP, dst);
        dma16(it.w + (size_t)g * B_TAP + 8 * 1024, dst + 8 * 1024);
    };
    // ---- MFMA role: rows 4 pg .. 4 pg + 3 of the tile (two M tiles of 2 rows x 16 pixels), channel tiles 2 chh, 2 chh + 1
    const int pg = wave & 3, chh = wave >> 2, m = lane & 31, kb = lane >> 5;
    const unsigned a_rd = (unsigned)(((4 * pg + (m >> 4)) * PW + (m & 15)) * 16 + kb * A_PLANE);
    const unsigned b_rd = (unsigned)(LDS_B + lane * 16 + chh * 2 * 2048);
    f32x16 acc[2][2];
    auto clear = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    // D[row][col]: col = lane & 31 = channel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = pixel of the M tile
    auto store = [&](const Item &it) {
        const int co = it.cb * COT + chh * 64 + m;
        // depth-to-space on the way out: channel co = parity * d2s_C + c goes to pixel (2 oy + py, 2 ox + px), channel c (co and
        // co + 32 share their parity: d2s_C is a multiple of 64)
        const int par = d2sC ? co / d2sC : 0, py = par >> 1, px = par & 1;
        const size_t base = ((size_t)it.b * H * W) * a.Co + (d2sC ? co - par * d2sC : co);
        float *yb = a.y + base;
        const bool ok0 = co < a.Co, ok1 = co + 32 < a.Co;   // (Co = 64: the block's upper half is padding)
        float b0 = 0.f, b1 = 0.f;
        if (a.bias) {
            if (ok0) b0 = a.bias[co];
            if (ok1) b1 = a.bias[co + 32];
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            // all of an M tile's residual values first, then its stores: load / store pairs one by one would each wait for the store in
            // front of them (one counter, in order) -- 20 us per item instead of 2
            float rv[16][2];
            if (a.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
                    const int oy = it.ty0 + 4 * pg + 2 * mt + (row >> 4), ox = it.tx0 + (row & 15);
                    const float *rp = a.res + base + ((size_t)oy * W + ox) * a.Co;
                    rv[r][0] = ok0 ? rp[0] : 0.f;
                    rv[r][1] = ok1 ? rp[32] : 0.f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r][0] = rv[r][1] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
                const int oy = it.ty0 + 4 * pg + 2 * mt + (row >> 4), ox = it.tx0 + (row & 15);
                const size_t off = d2sC ? ((size_t)(2 * oy + py) * (2 * W) + 2 * ox + px) * d2sC : ((size_t)oy * W + ox) * a.Co;
                if (ok0) yb[off] = acc[mt][0][r] + b0 + rv[r][0];
                if (ok1) yb[off + 32] = acc[mt][1][r] + b1 + rv[r][1];
            }
        }
    };

    // The MFMA steps (tap, 16-channel half) run as a software pipeline: while a step's twelve MFMAs execute, the wave reads the NEXT
    // step's eight fragments into the other register set.  So tap t + 1's weights are resident when tap t starts (ring of four,
    // requested three taps ahead) and the next chunk's patch is stashed during tap 6.
    struct Frag { h8 ah[2], al[2], bh[2], bl[2]; };
    Frag F[2];
    bool fresh = false;
    [[maybe_unused]] bool first_load = true;
    auto load_frags = [&](Frag &f, const char *Ab, const char *Bb, int tap, int ks) {
#if defined(PS_CONV_EXP) && (PS_CONV_EXP & 2)   // tuning build: fragments read once
        if (!first_load) return;
#endif
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const char *p = Ab + ks * 2 * A_PLANE + ((2 * mt + ky) * PW + kx) * 16;
            f.ah[mt] = *(const h8 *)p;
            f.al[mt] = *(const h8 *)(p + 4 * A_PLANE);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const char *p = Bb + ks * 8192 + nt * 2048;
            f.bh[nt] = *(const h8 *)p;
            f.bl[nt] = *(const h8 *)(p + 1024);
        }
    };
    // Does this wave's 64-channel half of the block carry weights at all (wave-uniform, per item)?  64 output channels leave the upper half of
    // the 128-channel block as padding.  A wave's MFMAs on zero weights cost the SIMD it shares with a live wave as much as real ones -- the
    // kernel is bound by its MFMA stream -- so a dead wave keeps the barriers, the copies and the staging and skips fragments and MFMAs.
    // (Tried and dropped: also skipping the upper 32-channel tile INSIDE a live wave -- the branch inside the step costs the scheduler its
    // interleaving: the decoder 53.9 -> 64.2 ms.)
    bool lv = true;
#ifdef PS_CONV_NO_SKIP   // tuning build: every wave multiplies, padding or not
    auto live_of = [&](const Item &) { return true; };
#else
    auto live_of = [&](const Item &it) { return it.cb * COT + chh * 64 < a.co_live; };
#endif
    auto mfma_step = [&](const Frag &f) {
#if defined(PS_CONV_EXP) && (PS_CONV_EXP & 1)   // tuning build: no MFMAs (the fragments still have to arrive)
        asm volatile("" : : "v"(f.ah[0]), "v"(f.ah[1]), "v"(f.al[0]), "v"(f.al[1]), "v"(f.bh[0]), "v"(f.bh[1]), "v"(f.bl[0]), "v"(f.bl[1]));
        return;
#endif
        // term by term over the four accumulators: an MFMA never waits for the result of the one in front of it
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? f.al[mt] : f.ah[mt], term == 1 ? f.bl[nt] : f.bh[nt],
                                                                         acc[mt][nt], 0, 0, 0);
#ifndef PS_CONV_NO_SCHED
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // one fragment read behind each of the first eight MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#endif
    };

#if defined(PS_CONV_EXP) && (PS_CONV_EXP & 16)
    const unsigned long long clk0 = clock64(), wall0 = wall_clock64();
#endif
    Item cur = item_at(0), nxt = cur;
    lv = live_of(cur);
    const int Q = nitems * nchunk, GG = Q * 9;   // chunk instances and taps of this workgroup
    aim(cur);
    fetch(cur, 0);
    dma_tap(cur, 0, 0);
    dma_tap(cur, 1, 1);
    dma_tap(cur, 2, 2);
    wait_vm<6>();   // (three taps of two copies each are newer than the patch)
    pin();
    stash(0);
    clear();
    bool stored = false;   // the previous item's results are on their way out

    int c = 0, item = 0;   // chunk of the current item; q = chunk instance
    for (int q = 0; q < Q; ++q) {
        const bool last = c + 1 == nchunk, more = q + 1 < Q;
        if (last && more) nxt = item_at(item + 1);
        const Item &nx = last ? nxt : cur;       // the item of the next chunk instance
        const int cn = last ? 0 : c + 1;          // and its chunk
        const char *Ab = lds + (q & 1) * A_BUF + a_rd, *An = lds + ((q + 1) & 1) * A_BUF + a_rd;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int gg = q * 9 + tap;
            // tap gg + 1's weights have landed: everything older than the requests that may still be in flight (tap gg + 2's two
            // copies; the next chunk's patch, requested at tap 0 behind tap 3's copies)
            if (gg + 2 >= GG) wait_vm<0>();
            else if (stored && tap < 2) ;                      // (landed before the stores went out, see below)
            else if ((tap == 1 || tap == 2) && more) wait_vm<2 + NFETCH>();
            else wait_vm<2>();
            if (tap == 1) stored = false;
            lds_barrier();
            if (gg + 3 < GG) {
                if (tap < 6) dma_tap(cur, c * 9 + tap + 3, gg + 3);
                else dma_tap(nx, cn * 9 + tap - 6, gg + 3);
            }
            if (tap == 0 && more) {
                if (last) aim(nx);
                fetch(nx, cn);
            }
            const char *Bb = lds + b_rd + (gg & 3) * B_TAP, *Bn = lds + b_rd + ((gg + 1) & 3) * B_TAP;
            if (lv) {
                if (gg == 0 || fresh) load_frags(F[0], Ab, Bb, tap, 0);
                fresh = false;
                load_frags(F[1], Ab, Bb, tap, 1);
                first_load = false;
                mfma_step(F[0]);
                if (tap < 8) load_frags(F[0], Ab, Bn, tap + 1, 0);
                else if (more) load_frags(F[0], An, Bn, 0, 0);
                mfma_step(F[1]);
            } else {
                fresh = true;      // (a wave that comes back to life at a later item reads its first fragments itself)
            }
            if (tap == 6 && more) {   // (the patch was waited for at tap 3, whose wait covers tap 1's copies, requested behind it)
                pin();                // (the two waves of a SIMD stashing at different taps, 5 and 6: 1-3 % slower; the stash spread a
                                      // piece per step over taps 3 .. 5 with its conversions scheduled behind the MFMAs: 2 % slower)
                stash((q + 1) & 1);
            }
        }
        if (last) {
            // Stores and loads retire in no fixed order, so a counted wait behind 64 stores would wait for all of them.  Instead:
            // everything requested so far (the next item's taps 0 .. 2) is waited for HERE, the next two taps then need no wait, and
            // tap 2's counts only the loads behind it (a store still in flight makes that wait longer, never shorter).
            if (more) wait_vm<0>();
            store(cur);
            clear();
            stored = more;
            cur = nxt;
            lv = live_of(cur);
            ++item;
            c = 0;
        } else {
            ++c;
        }
    }
    if (over) *a.overflow = 1;
#if defined(PS_CONV_EXP) && (PS_CONV_EXP & 16)
    if (tid == 0) {
        atomicAdd(&g_conv_clk[0], clock64() - clk0);
        atomicAdd(&g_conv_clk[1], wall_clock64() - wall0);
    }
#endif
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
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = co < Co ? w[((size_t)co * 9 + tap) * Ci + ci + j] : 0.f;   // (Co = 64: the block's upper half is zeros)
        hi[j] = (_Float16)v;
        lo[j] = (_Float16)(v - (float)hi[j]);
    }
    h8 *dst = out + ((size_t)(i >> 6) * 2) * 64 + lane;
    dst[0] = hi;
    dst[64] = lo;
}

}  // namespace psconv

extern "C" {

#if defined(PS_CONV_EXP) && (PS_CONV_EXP & 16)
// tuning build: (shader cycles, 100 MHz ticks) summed over the workgroups since the last call; resets
int ps_conv_debug_clock(unsigned long long *out)
{
    unsigned long long zero[2] = {0, 0};
    PS_HIP_CHECK(hipDeviceSynchronize());
    PS_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(psconv::g_conv_clk), 16));
    PS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(psconv::g_conv_clk), zero, 16));
    return PS_OK;
}
#endif

size_t ps_conv3x3_f16x3_packed_bytes(int Co, int Ci) { return (size_t)9 * ((Co + psconv::COT - 1) / psconv::COT * psconv::COT) * Ci * 4; }

int ps_conv3x3_f16x3_pack(const float *w, int Co, int Ci, void *packed, void *stream)
{
    PS_REQUIRE(w && packed, "conv3x3_f16x3_pack: null pointer");
    PS_REQUIRE(Co > 0 && Co % 64 == 0 && Ci > 0 && Ci % psconv::CK == 0,
               "conv3x3_f16x3_pack: Co a multiple of 64 and Ci a multiple of 32 required (Co = %d, Ci = %d)", Co, Ci);
    const int total = ((Co + psconv::COT - 1) / psconv::COT) * (Ci / psconv::CK) * 9 * 2 * 4 * 64;
    hipLaunchKernelGGL(psconv::k_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, Co, Ci, (psconv::h8 *)packed, total);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_conv3x3_f16x3_nhwc(const float *x, const float *scale, const float *shift, const void *packed, const float *bias, const float *res,
                          int B, int H, int W, int Ci, int Co, float *y, int *overflow, void *stream)
{
    return ps_conv3x3_f16x3_ex_nhwc(x, scale, shift, packed, bias, res, B, H, W, Ci, Co, 0, 0, 0, y, overflow, stream);
}

int ps_conv3x3_f16x3_ex_nhwc(const float *x, const float *scale, const float *shift, const void *packed, const float *bias, const float *res,
                             int B, int H, int W, int Ci, int Co, int co_live, int in_s2d, int out_d2s, float *y, int *overflow, void *stream)
{
    using namespace psconv;
    PS_REQUIRE(x && packed && y && overflow, "conv3x3_f16x3: null pointer");
    PS_REQUIRE(co_live >= 0 && co_live <= Co, "conv3x3_f16x3: co_live in 0 .. Co required (%d)", co_live);
    PS_REQUIRE(!in_s2d || (Ci % 4 == 0 && (Ci / 4) % CK == 0), "conv3x3_f16x3: space-to-depth input needs Ci / 4 a multiple of 32 (Ci = %d)", Ci);
    PS_REQUIRE(!out_d2s || (Co % 4 == 0 && (Co / 4) % 64 == 0 && !res),
               "conv3x3_f16x3: depth-to-space output needs Co / 4 a multiple of 64 and no res (Co = %d)", Co);
    PS_REQUIRE((scale == nullptr) == (shift == nullptr), "conv3x3_f16x3: scale and shift come together");
    PS_REQUIRE(B > 0 && H > 0 && W > 0 && H % TH == 0 && W % TW == 0, "conv3x3_f16x3: H and W multiples of 16 required (H = %d, W = %d)", H, W);
    PS_REQUIRE(Co > 0 && Co % 64 == 0 && Ci > 0 && Ci % CK == 0,
               "conv3x3_f16x3: Co a multiple of 64 and Ci a multiple of 32 required (Co = %d, Ci = %d)", Co, Ci);
    PS_REQUIRE((size_t)H * W * Ci < ((size_t)1 << 31), "conv3x3_f16x3: a frame of %d x %d x %d does not fit 32-bit offsets", H, W, Ci);
    ConvArgs a;
    a.x = x; a.scale = scale; a.shift = shift; a.wp = (const char *)packed; a.y = y; a.bias = bias; a.res = res; a.overflow = overflow;
    a.H = H; a.W = W; a.Ci = Ci; a.Co = Co;
    a.s2d_C = in_s2d ? Ci / 4 : 0; a.d2s_C = out_d2s ? Co / 4 : 0;
    a.co_live = co_live > 0 && co_live < Co ? co_live : Co;
    a.tiles_x = W / TW; a.tiles_per_frame = (H / TH) * a.tiles_x; a.ncb = (Co + COT - 1) / COT;
    const size_t nb = (size_t)B * a.tiles_per_frame * a.ncb;
    PS_REQUIRE(nb < ((size_t)1 << 30), "conv3x3_f16x3: too many tiles");
    a.nblocks = (int)nb;
    // one persistent workgroup per compute unit (148 KB of LDS each), a multiple of the eight XCDs; per DEVICE, as is the functions'
    // dynamic-LDS limit (a process may drive several GPUs, one per thread)
    constexpr int MAX_DEV = 64;
    static int cus_of[MAX_DEV] = {};
    static bool attr_set[MAX_DEV][4] = {};
    int dev = 0;
    PS_HIP_CHECK(hipGetDevice(&dev));
    PS_REQUIRE(dev >= 0 && dev < MAX_DEV, "conv3x3_f16x3: device %d", dev);
    if (!cus_of[dev]) {
        hipDeviceProp_t prop;
        PS_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        cus_of[dev] = prop.multiProcessorCount >= 8 ? prop.multiProcessorCount / 8 * 8 : 8;
    }
    const int cus = cus_of[dev];
    static int wgs = -2;     // PS_CONV_WGS (tuning): workgroups per launch; 0 = one per item (no persistence); default: one per compute unit
    if (wgs == -2) {
        const char *e = getenv("PS_CONV_WGS");
        wgs = e ? atoi(e) : -1;
    }
    const int want = wgs < 0 ? cus : wgs == 0 ? (int)((nb + 7) / 8 * 8) : (wgs + 7) / 8 * 8;
    const int grid = (int)std::min<size_t>((size_t)want, (nb + 7) / 8 * 8);
    auto launch = [&](auto kernel, int slot) -> int {
        if (!attr_set[dev][slot]) {
            PS_HIP_CHECK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
            attr_set[dev][slot] = true;
        }
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(NT), LDS_TOTAL, (hipStream_t)stream, a);
        return PS_OK;
    };
    const bool perm = a.s2d_C || a.d2s_C;
    int rc;
    if (scale) rc = perm ? launch(k_conv3x3_f16x3<true, true>, 3) : launch(k_conv3x3_f16x3<true, false>, 1);
    else rc = perm ? launch(k_conv3x3_f16x3<false, true>, 2) : launch(k_conv3x3_f16x3<false, false>, 0);
    if (rc != PS_OK) return rc;
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // extern "C"
