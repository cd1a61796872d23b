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
//   * fp32 MFMA is a chain of fused multiply-adds in ascending k, so MFMA tiles and v_fma_f32 loops that walk one
//     canonical order (five accumulation chains per tap, mfma_chunk5) produce identical bits.
//   * Whole-grid mode (k_gemm + k_post_grid, items = F*L or the observed prefix of every order): the
//     reference-faithful OurPixelCNN.forward and the cache build an AR run starts from.
//   * Column mode (the incremental AR evaluation; a column = one order position of one frame): k_column, ONE launch per
//     WAVEFRONT of columns that do not depend on each other (the walk position by position is the special case of one
//     column per frame), with two workgroup roles that start together -- nbr_role computes the NA/NB slots of all
//     32 convs (MFMA; they only read finished columns), chain_role walks the 33 dependent stages of one column on one
//     CU (centre taps as per-thread FMA chains on weights held in registers, post op by a dedicated wave, LDS
//     hand-off) and draws the code.  Completion counters per (stage, column tile) carry the neighbour slots across;
//     every wait is bounded.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "ps_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Pointers that reach a kernel through a descriptor in memory (NbrWork, control records) are "generic" to the compiler,
// which then emits FLAT loads/stores.  FLAT ops also count on lgkmcnt, so an LDS-only barrier
// (s_waitcnt lgkmcnt(0)) would drain every weight / slot prefetch in flight.  All descriptor pointers
// are device-global memory: say so, and get global_load / global_store.
#define PS_G(T, p) ((__attribute__((address_space(1))) T *)(p))
#define PS_GC(T, p) ((const __attribute__((address_space(1))) T *)(p))

constexpr int NF = 80;        // nr_filters          (models/z_buffermodel.py:63)
constexpr int NCLS = 512;     // input_channels / classes
constexpr int NNODE = 19;     // u0..u8 (up pass) + d0..d9 (down pass)
constexpr int R_LD = 96;      // row stride of the raw-u caches R[node]: 80 channels padded to three 128-byte lines, so that a
                              // cache line never spans two locations (E / X rows are 160 floats = five lines)
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
// 20 MFMAs; group j of the chunk accumulates into acc[j] (five independent chains, so consecutive MFMAs
// never wait on each other).  v_mfma_f32_16x16x4_f32 is a chain of four fused multiply-adds in ascending k
// (tools/mfma_semantics.hip: 0 mismatches in 2^20), so chain j of output o is, in order,
//     for group g in (j, 5 + j, ...): for c in 0..3: for kk in 0..3: acc = fma(W[o][16g + 4kk + c], x[16g + 4kk + c], acc)
// and the tap's value is chunk_total(acc).  Every kernel -- MFMA or VALU -- walks taps, chunks and chains in this
// order, so the whole-grid pass and the column steps produce identical bits.
struct Acc5 { f32x4 v[5]; };
__device__ __forceinline__ Acc5 acc5_zero()
{
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    return Acc5{{z, z, z, z, z}};
}
__device__ __forceinline__ void mfma_chunk5(const f32x4 (&av)[5], const f32x4 (&bv)[5], Acc5 &acc)
{
#pragma unroll
    for (int j = 0; j < 5; ++j) acc.v[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc.v[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 5; ++j) acc.v[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc.v[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 5; ++j) acc.v[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc.v[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 5; ++j) acc.v[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc.v[j], 0, 0, 0);
}
// value of one tap from its five chains
template <typename T>
__device__ __forceinline__ T chain_total(const T &a0, const T &a1, const T &a2, const T &a3, const T &a4) { return (((a0 + a1) + a2) + a3) + a4; }
__device__ __forceinline__ f32x4 chunk_total(const Acc5 &a) { return chain_total(a.v[0], a.v[1], a.v[2], a.v[3], a.v[4]); }

// ==========================================================================================
// whole-grid mode: items = (frame, location) pairs of the full grid
// ==========================================================================================
// Items of a whole-grid pass: every (frame, location) pair, or -- with a generation order -- only the first
// `npre` locations of each frame in that order (the observed prefix an AR run starts from; later locations
// are produced by the column steps, and no earlier location ever reads them).
struct ItemMap {
    const int32_t *order;  // (F, L) location by rank, or null = all L locations in raster order
    int npre;              // locations per frame
    const int32_t *start;  // (F) or null: ranks below start[f] are NOT evaluated at this stage -- nothing reads them
                           // (k_prefix_starts); only with an order
    int f0;                // first frame of the pass (a pass over frames [f0, f0 + n): item 0 is rank 0 of frame f0)
};
__device__ __forceinline__ void item_loc(const ItemMap &m, int item, int L, int &f, int &q)
{
    const int fl = item / m.npre;
    const int r = item - fl * m.npre;
    f = m.f0 + fl;
    q = m.order ? m.order[(size_t)f * L + r] : r;
}
// is the item evaluated at this stage?
__device__ __forceinline__ bool item_wanted(const ItemMap &m, int item)
{
    if (!m.start) return true;
    const int fl = item / m.npre;
    return item - fl * m.npre >= m.start[m.f0 + fl];
}

constexpr int N_XCD = 8;  // gfx950: 8 XCDs, workgroup ids are dealt round-robin over them

struct GemmArgs {
    GemmTap tap[MAX_TAPS];
    ItemMap items;
    int slot_first[5];  // slot s covers taps [slot_first[s], slot_first[s+1])
    int nslots, Cin, Co_pad, H, W, L, nitems, tiles_per_block;
    int nx, ny, tpx;    // launch geometry (launch_gemm): channel blocks, item blocks, item blocks per XCD
    int zgrid;          // slots along the grid (nslots), or 1 = every wave walks all slots
    int wg_reverse;     // k_gemm_wg: workgroups walk the items from the end (tuning)
    const float *sum_bias;  // zgrid == 1 only: the wave adds its slots up itself, y = ((bias + NA) + C) + NB, and stores y in
                            // place of slot NA (a third of the partial traffic); null = raw slots
    const float *mask;
    size_t mask_fstride;
    float *partial;  // [nslots][nitems][Co_pad]
};

// a row of zeros: the input row of a lane whose tap is closed, when every mask value of the wave is 0 or 1 (the
// reference's masks always are): the closed lanes then LOAD their zeros and the chunk loop carries no mask arithmetic
// (40 vector instructions per chunk that compete with the MFMAs for issue: tools/mfma_rate_probe.hip, 95 % -> 84 %)
constexpr int ZERO_ROW = 4096;   // floats: as many input channels as a closed lane may walk through it
__device__ float g_zero_row[ZERO_ROW];

// grid z -> slot, long slots first: (NA, NB, C, SKIP)
__device__ __forceinline__ int gemm_slot_of(const GemmArgs &a, int z) { return z == 0 ? SLOT_NA : z == 1 && a.nslots > 2 ? SLOT_NB : z == 2 ? SLOT_C : z; }

// One wave = 16 items x (T x 16) output channels of one slot: the gathered input rows (B operand) are loaded once
// per 80-channel chunk and reused by the T output tiles, so the kernel is bound by the MFMA pipe rather than by
// the per-CU L1 fill rate (at T = 1 every 40 MFMAs needed 20 KB of operands).
constexpr int SY_LD = 168;   // floats per item of the fused kernel's LDS tile (160 channels + pad: 16-byte rows, 8 banks apart)
template <int T>
__device__ __forceinline__ void gemm_tiles(const GemmArgs &a, int o0, int z0, int z1, int first_tile, float *sY = nullptr,
                                           float *sS = nullptr)
{

    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    const int ngroups = a.Cin >> 4;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int tt = 0; tt < a.tiles_per_block; ++tt) {
        const int tile = first_tile + tt;
        if (tile * 16 >= a.nitems) break;
        const int item = tile * 16 + i;
        const bool valid = item < a.nitems && item_wanted(a.items, item);
        if (!__any(valid) && !sY) continue;   // a tile nobody reads at this stage (the fused kernel still needs its barrier)
        int f = 0, r = 0, c = 0, q = 0;
        if (valid) {
            item_loc(a.items, item, a.L, f, q);
            r = q / a.W;
            c = q - r * a.W;
        }
        const bool summing = a.sum_bias != nullptr;   // (then z0 = 0, z1 = nslots, slots in the order of the sum: NA, C, NB, SKIP)
        f32x4 ysum[T];
#pragma unroll
        for (int u = 0; u < T; ++u) ysum[u] = zero;
        for (int z = z0; z < z1; ++z) {
        const int slot = summing ? z : gemm_slot_of(a, z);
        // slot value = taps of the slot added in order, each tap from fresh accumulators: P_t = chunk_total(acc)
        f32x4 tot[T];
#pragma unroll
        for (int u = 0; u < T; ++u) tot[u] = zero;
        // the mask values of all (at most four) taps of the slot are requested together, before the first tap needs one:
        // fetched inside the tap loop each is a dependent round trip in front of the tap's operand loads
        const int t0 = a.slot_first[slot], nt = a.slot_first[slot + 1] - t0;
        float mvs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mvs[k] = 0.0f;
            if (k < nt) {
                const GemmTap &tq = a.tap[t0 + k];
                const int rr = r + tq.dr, cc = c + tq.dc;
                if (valid && rr >= 0 && rr < a.H && cc >= 0 && cc < a.W)
                    mvs[k] = tq.mask_row >= 0 ? a.mask[(size_t)f * a.mask_fstride + (size_t)tq.mask_row * a.L + q] : 1.0f;
            }
        }
        for (int t = t0; t < t0 + nt; ++t) {
            const GemmTap tp = a.tap[t];
            // Lanes without a live input row still LOAD (row 0 of the cache, a valid address) and discard: a load under a
            // lane condition compiles to branch / load / s_waitcnt vmcnt(0) per load, i.e. the five input loads of a
            // chunk one round trip after the other (k_gemm: 48.8 -> 43.6 us per launch).
            const int k = t - t0;
            const float mv = k == 0 ? mvs[0] : k == 1 ? mvs[1] : k == 2 ? mvs[2] : mvs[3];
            const int rr = r + tp.dr, cc = c + tp.dc;
            const bool live = mv != 0.0f;
            if (!__any(live)) continue;  // a masked tap is an exact zero: skipping it does not change the bits
            // (a masked row is not fetched either)
            const bool unit = a.Cin <= ZERO_ROW && __all(mv == 0.0f || mv == 1.0f);   // wave-uniform: 0/1 masks need no multiply
            const float *src = live ? tp.in + ((size_t)f * a.L + rr * a.W + cc) * tp.ld + 4 * kk
                                    : (unit ? g_zero_row : tp.in) + 4 * kk;
#ifdef PS_GEMM_EXP_HOTB   // timing experiment only (wrong results): every input row is row 0 -> the B operand always hits L1
            src = tp.in + 4 * kk;
#endif
            Acc5 acc[T];
#pragma unroll
            for (int u = 0; u < T; ++u) acc[u] = acc5_zero();
            // weights through a buffer descriptor: uniform base + uniform offset in SGPRs, ONE 32-bit lane offset -- the ten
            // weight loads of a chunk need no per-load 64-bit address registers (19 spilled VGPRs otherwise)
            const uint32_t woff = (uint32_t)((kk * a.Co_pad + o0 + i) * 16);
            const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)tp.w, 0, 0x7fffffff, 0x00020000);
            auto wload = [&](int grp, int u) {
                return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, woff, (grp * 16 * a.Co_pad + 64 * u) * 4, 0));
            };
            int g = 0;
            for (; g + 5 <= ngroups; g += 5) {
                f32x4 bv[5];
#pragma unroll
#ifdef PS_GEMM_EXP_NOLOAD
                for (int j = 0; j < 5; ++j) bv[j] = f32x4{mv, (float)g, (float)j, 1.0f};
#else
                for (int j = 0; j < 5; ++j) bv[j] = *(const f32x4 *)(src + 16 * (g + j));
#endif
                // (the loads must stay unconditional: left to itself the compiler sinks the last one under `live` and waits
                // for it with vmcnt(0) -- the B round trip and the A round trip of the chunk then run one after the other)
#pragma unroll
                for (int j = 0; j < 5; ++j) asm volatile("" : "+v"(bv[j]));
                if (!unit) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) bv[j] = live ? bv[j] * mv : zero;
                }
#pragma unroll
                for (int u = 0; u < T; ++u) {
                    f32x4 av[5];
#pragma unroll
#ifdef PS_GEMM_EXP_NOLOAD   // timing experiment only (wrong results): operands made up in registers, no loads at all
                    for (int j = 0; j < 5; ++j) { av[j] = bv[j] + (float)(g + u); asm volatile("" : "+v"(av[j])); }
#elif defined(PS_GEMM_EXP_HOTA)   // timing experiment only (wrong results): five fixed weight vectors -> the A operand always hits L1
                    for (int j = 0; j < 5; ++j) av[j] = wload(j, 0);
#else
                    for (int j = 0; j < 5; ++j) av[j] = wload(g + j, u);
#endif
                    mfma_chunk5(av, bv, acc[u]);
                }
            }
            for (; g < ngroups; ++g) {  // ragged channel counts of the generic lmconv entry point only
                const f32x4 raw = *(const f32x4 *)(src + 16 * g);
                const f32x4 bv = unit ? raw : live ? raw * mv : zero;
#pragma unroll
                for (int u = 0; u < T; ++u) {
                    const f32x4 av = wload(g, u);
                    f32x4 &a0 = acc[u].v[0];
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, a0, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, a0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < T; ++u) tot[u] = tot[u] + chunk_total(acc[u]);
        }
        // D: row (output channel) = kk*4 + reg, col (item) = i
        if (summing && slot != SLOT_SKIP) {
#pragma unroll
            for (int u = 0; u < T; ++u)
                ysum[u] = (slot == SLOT_NA ? *(const f32x4 *)(a.sum_bias + o0 + 16 * u + kk * 4) : ysum[u]) + tot[u];
            if (slot != SLOT_NB) continue;
#pragma unroll
            for (int u = 0; u < T; ++u) tot[u] = ysum[u];
        }
        if (sY) {   // fused stage kernel: y (and the nin_skip slot) stay in the workgroup's LDS tile
            float *dst = (slot == SLOT_SKIP ? sS : sY) + i * SY_LD + o0 + kk * 4;
#pragma unroll
            for (int u = 0; u < T; ++u) *(f32x4 *)(dst + 16 * u) = tot[u];
        } else if (valid) {
            const int at = summing && slot == SLOT_NB ? SLOT_NA : slot;
#pragma unroll
            for (int u = 0; u < T; ++u)
                *(f32x4 *)(a.partial + ((size_t)at * a.nitems + item) * a.Co_pad + o0 + 16 * u + kk * 4) = tot[u];
        }
        }
    }
}

// grid (ceil(Co_pad / 64), item blocks, nslots), one wave per block.  Blocks are dispatched x fastest, z slowest, and a
// wave of the four-tap slots NA / NB lives four times as long as one of the single-tap slots C / SKIP: the slot is the
// slowest dimension, long slots first, so that the kernel's tail is made of short waves.
// 16-channel output tiles per wave: 2 (with four waves per SIMD) measured best -- 4: 55.7 us, 2: 51.1, 2 at four waves per
// SIMD: 49.3, 1: 60.2 us per launch at 16 frames
#ifndef PS_GEMM_T
#define PS_GEMM_T 2
#endif
constexpr int GEMM_T = PS_GEMM_T;
#if PS_GEMM_T <= 2
#define PS_GEMM_WAVES 4
#else
#define PS_GEMM_WAVES 3
#endif
__attribute__((amdgpu_waves_per_eu(PS_GEMM_WAVES, PS_GEMM_WAVES)))
__global__ __launch_bounds__(64) void k_gemm(GemmArgs a)
{
    // Workgroup ids go round-robin over the 8 XCDs, each with its own L2.  Every XCD gets a contiguous range of item
    // blocks with ALL their channel blocks and slots (the waves that gather the same input rows, and the rows of
    // neighbouring items, meet in one L2) instead of five channel blocks of one tile on five XCDs.
    const int xcd = blockIdx.x & (N_XCD - 1), j = blockIdx.x >> 3;
    const int x = j % a.nx, t = (j / a.nx) % a.tpx, z = j / (a.nx * a.tpx);
    const int y = xcd * a.tpx + t;
    if (y >= a.ny) return;
    // zgrid = 1: one wave walks ALL slots of its (tile, channel block) -- the wave's start-up (kernel arguments, order and
    // mask look-ups: two or three dependent round trips) is paid once per nine or ten taps instead of once per slot, and a
    // single-tap C / SKIP wave was mostly start-up
    const int z0 = a.zgrid == 1 ? 0 : z, z1 = a.zgrid == 1 ? a.nslots : z + 1;
    const int o0 = x * 16 * GEMM_T, first_tile = y * a.tiles_per_block;
    const int T = min(GEMM_T, (a.Co_pad - o0) >> 4);
    if (GEMM_T >= 4 && T == 4) gemm_tiles<4>(a, o0, z0, z1, first_tile);
    else if (GEMM_T >= 3 && T == 3) gemm_tiles<3>(a, o0, z0, z1, first_tile);
    else if (GEMM_T >= 2 && T == 2) gemm_tiles<2>(a, o0, z0, z1, first_tile);
    else gemm_tiles<1>(a, o0, z0, z1, first_tile);
}

// ------------------------------------------------------------------------------------------
// per-item post ops, shared by the whole-grid kernels and the column chain.
// One wave per item, TWO adjacent channels per lane: lane l < 40 owns channels 2l and 2l + 1 (8-byte accesses, packed
// fp32 add / mul / fma for everything but the transcendentals); lanes 40..63 carry zeros.
// The statistics of PONO are reduced in ONE association order everywhere (pono_total): s_l = y[2l] + y[2l+1], a
// butterfly over the lanes of each row of 16 (DPP), then R2 + (R1 + R0) -- so column steps and whole-grid passes
// agree bit for bit.
// ------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int PONO_LANES = NF / 2;  // 40 lanes carry data

// Elementwise math of the post ops.  These sit on the sequential critical path of every AR order position
// (the chain role), so they use the hardware transcendental units directly (v_exp_f32 / v_rcp_f32 / v_rsq_f32,
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

// PONO statistics (models/lmconv/layers.py:231-236: unbiased variance, eps 1e-5)
__device__ __forceinline__ float pono_mean(float total) { return total * (1.0f / (float)NF); }
__device__ __forceinline__ float pono_inv(float ss_total) { return __builtin_amdgcn_rsqf(ss_total * (1.0f / (float)(NF - 1)) + 1e-5f); }

// y = ((bias + NA) + C) + NB
__device__ __forceinline__ float slot_sum(float bias, float na, float c, float nb) { return ((bias + na) + c) + nb; }

enum { POST_CONVIN = 0, POST_GATE = 1, POST_DIL = 2 };

// n = PONO-normalised value.  KIND = POST_CONVIN: out = n [+ skip]              (layers.py:153-156)
//                                   POST_GATE:   out = rin + n * sigmoid(g)      (layers.py:159-163)
//                                   POST_DIL:    out = n                         (model.py:138-140,148-150)
__device__ __forceinline__ f32x2 sigmoid2(const f32x2 &x) { return f32x2{sigmoid1(x.x), sigmoid1(x.y)}; }
__device__ __forceinline__ void celu_pair2(const f32x2 &x, f32x2 &ep, f32x2 &en)
{
    float p0, p1, n0, n1;
    celu_pair(x.x, p0, n0);
    celu_pair(x.y, p1, n1);
    ep = f32x2{p0, p1};
    en = f32x2{n0, n1};
}
// y = ((bias + NA) + C) + NB on a channel pair
__device__ __forceinline__ f32x2 slot_sum2(const f32x2 &bias, const f32x2 &na, const f32x2 &c, const f32x2 &nb) { return ((bias + na) + c) + nb; }

template <int KIND>
__device__ __forceinline__ f32x2 post_finish(const f32x2 &n, const f32x2 &g, const f32x2 &skip, bool has_skip, const f32x2 &rin)
{
    if (KIND == POST_CONVIN) return has_skip ? n + skip : n;
    if (KIND == POST_GATE) return rin + n * sigmoid2(g);
    return n;
}

template <int CTRL>
__device__ __forceinline__ float dpp_xadd(float x)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false);
    return x + __int_as_float(moved);
}
__device__ __forceinline__ float lane_value(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
// sum over the 80 channels of an item, two per lane; `own` = this lane is one of the 40 data lanes (others count 0).
// The result is wave-uniform.  s_l = v.x + v.y; butterfly over the 16 lanes of every row (pairs, quads, octets, row);
// then T = R2 + (R1 + R0) with R_k the sum of row k (row 2 = lanes 32..39 + zeros).
__device__ __forceinline__ float pono_total(const f32x2 &v, bool own)
{
    float x = own ? v.x + v.y : 0.0f;
    x = dpp_xadd<0xB1>(x);    // quad_perm [1,0,3,2]: pairs
    x = dpp_xadd<0x4E>(x);    // quad_perm [2,3,0,1]: quads
    x = dpp_xadd<0x141>(x);   // row_half_mirror: octets
    x = dpp_xadd<0x140>(x);   // row_mirror: every lane of row k now holds R_k
    // row_bcast:15 into rows 1 (and 3): R1 + R0;  row_bcast:31 into rows 2 (and 3): R2 + (R1 + R0)
    x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xa, 0xf, false));
    x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x143, 0xc, 0xf, false));
    return lane_value(x, 47);
}

// u_init on one-hot input as a gather, type-A mask (model.py:132), BEFORE norm_init:
//   y[o] = b[o] + sum_t m_t * (W[t][512][o] + W[t][code(nbr_t)][o])
// Only earlier order positions contribute (the centre of a type-A mask is 0), so in column mode this
// belongs to the neighbour kernel, not to the chain.  V = float (channel c) or f32x4 (channels c .. c+3).
// `code[t]`: the neighbour's class, -1 = all-zero input (not sampled yet), UINIT_CLOSED = tap closed / outside the grid
constexpr int UINIT_CLOSED = -2;
template <typename V>
__device__ __forceinline__ V uinit_from_codes(const int *code /*9*/, const float *mA /*9 values*/, const float *__restrict__ w,
                                              const float *__restrict__ bias, int c)
{
    V v = *(const V *)(bias + c);
    // all eighteen rows are requested before any is used (closed taps re-read the ones row and drop it): fetched under
    // the tap's condition they come one round trip after the other, up to nine of them at the start of every launch
    V ones[9], rows[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float *wt = w + (size_t)t * (NCLS + 1) * NF + c;
        ones[t] = *(const V *)(wt + (size_t)NCLS * NF);
        rows[t] = *(const V *)(wt + (size_t)(code[t] >= 0 ? code[t] : NCLS) * NF);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (code[t] == UINIT_CLOSED) continue;
        V x = ones[t];
        if (code[t] >= 0) x = x + rows[t];
        v = v + x * mA[t];
    }
    return v;
}
template <typename V>
__device__ __forceinline__ V uinit_gather(const int32_t *__restrict__ codes_f, const float *mA /*9 values*/,
                                          const float *__restrict__ w, const float *__restrict__ bias, int q, int H, int W, int c)
{
    const int r = q / W, c0 = q - r * W;
    int code[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int rr = r + t / 3 - 1, cc = c0 + t % 3 - 1;
        const bool in = rr >= 0 && rr < H && cc >= 0 && cc < W;
        const int raw = codes_f[in ? rr * W + cc : q];  // (loaded unconditionally: see uinit_from_codes)
        code[t] = (in && mA[t] != 0.0f) ? raw : UINIT_CLOSED;
    }
    return uinit_from_codes<V>(code, mA, w, bias, c);
}

__device__ __forceinline__ void store_raw_celu2(float *R, float *E, size_t loc, int c, const f32x2 &u)
{
    f32x2 ep, en;
    celu_pair2(u, ep, en);
    *(f32x2 *)(R + loc * R_LD + c) = u;
    *(f32x2 *)(E + loc * (2 * NF) + c) = ep;
    *(f32x2 *)(E + loc * (2 * NF) + NF + c) = en;
}

struct PostArgs {
    ItemMap items;
    const float *partial;  // [slots][nitems][Co_pad]
    int nitems, Co_pad, L, has_skip;
    int summed;            // slot NA of `partial` already holds y = ((bias + NA) + C) + NB (k_gemm with sum_bias)
    const float *bias, *bias2;
    const float *Rin;
    float *Rout, *Eout, *Xout;
};

// post op of one item by one wave.  `P` points at channel pair c of the item's y (raw slots `ss` floats apart unless
// a.summed), `S` at the same pair of its nin_skip slot.
template <int KIND>
__device__ __forceinline__ void post_item(const PostArgs &a, int item, int lane, const float *Pbase, size_t ss, const float *Sbase)
{
    int f, q;
    item_loc(a.items, item, a.L, f, q);
    const size_t loc = (size_t)f * a.L + q;
    const bool own = lane < PONO_LANES;
    const int c = own ? 2 * lane : 0;
    const float *P = Pbase + c;
    const f32x2 zero = {0.0f, 0.0f};
    auto ld = [](const float *p) { return *(const f32x2 *)p; };
    f32x2 g = zero, skip = zero, rin = zero;
    const f32x2 y = a.summed ? ld(P + SLOT_NA * ss)
                             : slot_sum2(ld(a.bias + c), ld(P + SLOT_NA * ss), ld(P + SLOT_C * ss), ld(P + SLOT_NB * ss));
    if (KIND == POST_GATE) {
        g = a.summed ? ld(P + SLOT_NA * ss + NF)
                     : slot_sum2(ld(a.bias + NF + c), ld(P + SLOT_NA * ss + NF), ld(P + SLOT_C * ss + NF), ld(P + SLOT_NB * ss + NF));
        rin = ld(a.Rin + loc * R_LD + c);
    }
    if (KIND == POST_CONVIN && a.has_skip) skip = ld(Sbase + c) + ld(a.bias2 + c);
    const float mean = pono_mean(pono_total(y, own));
    const f32x2 d = y - mean;
    const float inv = pono_inv(pono_total(d * d, own));
    if (!own) return;
    const f32x2 out = post_finish<KIND>(d * inv, g, skip, a.has_skip != 0, rin);
    if (KIND == POST_CONVIN) {
        f32x2 ep, en;
        celu_pair2(out, ep, en);
        *(f32x2 *)(a.Xout + loc * (2 * NF) + c) = ep;
        *(f32x2 *)(a.Xout + loc * (2 * NF) + NF + c) = en;
    } else {
        store_raw_celu2(a.Rout, a.Eout, loc, c, out);
    }
}

// whole-grid post op: one wave per item, 4 items per 256-thread block
template <int KIND>
__global__ __launch_bounds__(256) void k_post_grid(PostArgs a)
{
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= a.nitems || !item_wanted(a.items, item)) return;  // whole waves leave together
    const size_t ss = (size_t)a.nitems * a.Co_pad;
    const float *P = a.partial + (size_t)item * a.Co_pad;
    post_item<KIND>(a, item, lane, P, ss, P + SLOT_SKIP * ss);
}

// ------------------------------------------------------------------------------------------
// k_gemm_wg: the whole-grid products with the receptive-field window shared through LDS (round 3).
//
// k_gemm gives every wave its own 16 items x 32 output channels and lets it gather its input rows and fetch its weights from
// L1 / L2 by itself.  Here a WORKGROUP of four waves -- one per SIMD: five-wave workgroups (one wave per 16 of 80 channels)
// were measured first and never got more than two of them resident on a CU, 3 + 3 + 2 + 2 waves, the doubly loaded SIMDs
// setting the pace -- owns a tile of 16 * TI items and ALL output channels of the conv, 20 MFMA tiles, five per wave:
//   conv_out (160 channels, TI = 2)   wave w: output tiles 2w, 2w + 1 for both item tiles, + output tile 8 + w / 2 for item tile w & 1
//   conv_input / dilated (80, TI = 4) wave w: output tile w for the four item tiles,        + output tile 4 for item tile w
//   * the gathered input rows of a tap (operand B: 16 * TI items x Cin channels, mask applied, closed or absent rows as zeros)
//     are staged in LDS ONCE per workgroup, in the lane order of the MFMA fragment, and read from there by all four waves
//     (conflict-free ds_read_b128); the rows of the NEXT open tap are requested before the MFMAs of this one and parked after
//     them (two buffers and one barrier per tap; conv_input, whose 64 x 160 rows take 40 KB, has one buffer and two barriers);
//   * a wave's weights (operand A) come straight from L2 into registers, one accumulation chain ahead of their use, and are
//     used for up to four item tiles (k_gemm: one);
//   * the four waves walk the SAME items, so the barriers cost no skew; a tap that is closed for the whole tile of items is
//     skipped by all of them (an exact zero); a tap that is open for some of them is computed for all, on zeros where it
//     is closed -- tot + 0 is tot, so the bits do not change -- which keeps the tap body free of branches.
// Arithmetic and order are k_gemm's summing form to the bit: per tap five accumulation chains (chain j = channel groups j,
// j + 5 in MFMA order), tap value (((a0 + a1) + a2) + a3) + a4, taps added in order into the slot, y = ((bias + NA) + C) + NB
// stored in place of slot NA, the nin_skip slot raw.  Taken for launches of at least PS_GEMM_WG_MIN item tiles.
// ------------------------------------------------------------------------------------------
constexpr int GW_WAVES = 4, GW_THREADS = 64 * GW_WAVES;
enum { GW_CONVOUT = 0, GW_CONVIN = 1, GW_DIL = 2 };
#ifdef PS_WG_TRACE_BUILD   // tuning builds: shader-clock stamps of wave 0 of the first 32 workgroups, per kernel variant
__device__ unsigned long long g_wg_trace[3][32][16];
__device__ unsigned long long g_wg_span[3][4096][2];   // wall clock (100 MHz) at the start and the end of every workgroup, + hw id
#define WG_STAMP(k) do { if (y < 32 && tid == 0 && (k) < 16) g_wg_trace[KIND][y][(k)] = clock64(); } while (0)
#else
#define WG_STAMP(k) do { } while (0)
#endif
// waves per SIMD the register budget is cut for: the full-size forms (five tiles per wave) take two, the others three
constexpr int gw_occ(int kind, int ti) { return (kind == GW_CONVOUT && ti == 2) || (kind == GW_CONVIN && ti == 4) ? 2 : 3; }
template <int KIND, int TI>
__attribute__((amdgpu_waves_per_eu(gw_occ(KIND, TI), gw_occ(KIND, TI))))
__global__ __launch_bounds__(GW_THREADS) void k_gemm_wg(GemmArgs a, PostArgs pa, int fuse_post)
{
    constexpr int NGH = KIND == GW_DIL ? 1 : 2, NG = 5 * NGH, MI = 16 * TI;
    constexpr int POSTK = KIND == GW_CONVOUT ? POST_GATE : KIND == GW_CONVIN ? POST_CONVIN : POST_DIL;
    constexpr int YLD = KIND == GW_DIL ? 84 : 168;      // floats per item of the post op's LDS tile: y (+ gate half / nin_skip slot) + pad
    constexpr int NAU = KIND == GW_CONVOUT ? 3 : 2;     // distinct output tiles (A operands) of a wave
    constexpr int NBU = TI + 1;                         // B operands of a wave: the TI item tiles + the fifth tile's own copy
    constexpr bool DB = MI * NG * 16 <= 6144;           // two B buffers while they take no more than 48 KB
    constexpr int BUF = TI * NG * 64;                   // f32x4 per B buffer: [item tile][channel group][lane]
    constexpr int SU = (TI * NG + GW_WAVES - 1) / GW_WAVES;   // 1 KB staging units per wave and tap (the last one may be absent)
    constexpr bool SU_EVEN = TI * NG % GW_WAVES == 0;
    static_assert(MI <= 64, "one lane per item in the set-up");
    constexpr int NB4 = (DB ? 2 : 1) * BUF > MI * YLD / 4 ? (DB ? 2 : 1) * BUF : MI * YLD / 4;   // (the post op's tile reuses the B buffers)
    __shared__ f32x4 sB[NB4];
    __shared__ int sRow[MAX_TAPS * MI];      // input row of (tap, item), -1 = closed (mask 0, outside the grid, item not evaluated)
    __shared__ float sMv[MAX_TAPS * MI];     // its mask value
    __shared__ int sItem[MI];                // item index, -1 = not evaluated here
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15, kk = lane >> 4;
    const int xcd = blockIdx.x & (N_XCD - 1), tb = blockIdx.x >> 3;
    const int yy = xcd * a.tpx + tb;         // contiguous item ranges per XCD, as in k_gemm
    if (tb >= a.tpx || yy >= a.ny) return;
    const int y = a.wg_reverse ? a.ny - 1 - yy : yy;   // (tuning: PS_WG_REVERSE)
    const int item0 = y * MI;
    const int ntaps = a.slot_first[a.nslots];
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    WG_STAMP(0);
#ifdef PS_WG_TRACE_BUILD
    if (tid == 0 && y < 4096) g_wg_span[KIND][y][0] = wall_clock64();
#endif
    // ---- set-up: rows and mask values of every (tap, item) of the tile; wave w does taps w, w + 4, w + 8
    {
        const int m = lane & (MI - 1);
        const int item = item0 + m;
        const bool valid = item < a.nitems && item_wanted(a.items, item);
        int f = 0, q = 0, r = 0, c = 0;
        if (valid) {
            item_loc(a.items, item, a.L, f, q);
            r = q / a.W;
            c = q - r * a.W;
        }
        if (wave == 0 && lane < MI) sItem[m] = valid ? item : -1;
        for (int t = wave; t < ntaps; t += GW_WAVES) {
            const GemmTap tp = a.tap[t];
            const int rr = r + tp.dr, cc = c + tp.dc;
            float mv = 0.0f;
            if (valid && rr >= 0 && rr < a.H && cc >= 0 && cc < a.W)
                mv = tp.mask_row >= 0 ? a.mask[(size_t)f * a.mask_fstride + (size_t)tp.mask_row * a.L + q] : 1.0f;
            if (lane < MI) {
                sRow[t * MI + m] = mv != 0.0f ? (f * a.L + rr * a.W + cc) : -1;
                sMv[t * MI + m] = mv;
            }
        }
    }
    __syncthreads();
    WG_STAMP(1);
    // live bits [4 t, 4 t + TI): item tile ti has an open lane at tap t (wave-uniform; the same in every wave)
    unsigned long long live = 0;
    for (int t = 0; t < ntaps; ++t) {
        const unsigned long long b = __ballot(sRow[t * MI + (lane & (MI - 1))] >= 0);
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
            if ((b >> (16 * ti)) & 0xFFFFull) live |= 1ull << (4 * t + ti);
    }
    if (live == 0 && __ballot(sItem[lane & (MI - 1)] >= 0) == 0ull) return;   // nothing of this tile is evaluated here
    auto tiles_of = [&](int t) { return (unsigned)((live >> (4 * t)) & 0xFull); };
    auto next_live = [&](int t) {   // first tap after t with an open item tile, or ntaps
        int n = t + 1;
        while (n < ntaps && tiles_of(n) == 0) ++n;
        return n;
    };
    // ---- staging: unit u of this wave = (item tile, channel group) (wave + 4 u); lane (kk, i) carries channels 16 g + 4 kk .. + 3 of
    // item i; rows and mask values are looked up once per item tile
    f32x4 sv[SU];
    auto stage_load = [&](int t) {
        const GemmTap tp = a.tap[t];
        int row[TI];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) row[ti] = sRow[t * MI + ti * 16 + i];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int unit = wave + GW_WAVES * u, ti = unit / NG, g = unit - ti * NG;
            if (!SU_EVEN && u == SU - 1 && unit >= TI * NG) continue;   // (wave-uniform)
            int r = row[0];
#pragma unroll
            for (int k = 1; k < TI; ++k) r = ti == k ? row[k] : r;
            sv[u] = *(const f32x4 *)(tp.in + (size_t)(r >= 0 ? r : 0) * tp.ld + 16 * g + 4 * kk);   // (unconditional: see gemm_tiles)
        }
    };
    auto stage_store = [&](int t, int buf) {
        int row[TI];
        float mvv[TI];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) { row[ti] = sRow[t * MI + ti * 16 + i]; mvv[ti] = sMv[t * MI + ti * 16 + i]; }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int unit = wave + GW_WAVES * u, ti = unit / NG, g = unit - ti * NG;
            if (!SU_EVEN && u == SU - 1 && unit >= TI * NG) continue;
            int r = row[0];
            float mv = mvv[0];
#pragma unroll
            for (int k = 1; k < TI; ++k) { r = ti == k ? row[k] : r; mv = ti == k ? mvv[k] : mv; }
            sB[buf * BUF + (ti * NG + g) * 64 + lane] = r >= 0 ? sv[u] * mv : zero;   // (x * 1.0f is x: 0 / 1 masks cost nothing)
        }
    };
    // ---- the wave's five tiles: A operand (output tile) and B operand (item tile) of each
    //   conv_out:          (a0,b0) (a0,b1) (a1,b0) (a1,b1) (a2,bx)     a0 = 2w, a1 = 2w + 1, a2 = 8 + w / 2, bx = item tile w & 1
    //   conv_input / dil:  (a0,b0) (a0,b1) (a0,b2) (a0,b3) (a1,bx)     a0 = w, a1 = 4, bx = item tile w
    // (with fewer item tiles than the full-size forms -- conv_out TI = 1, conv_input / dilated TI = 2 -- a wave has the tiles of
    // its first NT4 = 2 (conv_out: its two output tiles) or TI combinations, and the fifth tile exists for the waves whose item tile
    // it would be: 3 + 3 + 2 + 2 or 3 + 2 + 3 + 2 tiles; three such workgroups fit a CU and even each other's SIMDs out)
    constexpr int NT4 = KIND == GW_CONVOUT ? 2 * TI : TI;       // tiles ahead of the "fifth" one
    constexpr int NTL = NT4 + 1;
    auto a_of = [](int k) constexpr { return KIND == GW_CONVOUT ? (k < NT4 ? k / TI : 2) : (k < NT4 ? 0 : 1); };
    auto b_of = [](int k) constexpr { return k < NT4 ? (KIND == GW_CONVOUT ? k % TI : k) : TI; };
    const int tixr = KIND == GW_CONVOUT ? (wave & 1) : wave;    // item tile of the fifth tile ...
    const bool has5 = tixr < TI;                                // ... if the workgroup has that item tile
    const int tix = has5 ? tixr : 0;
    int ot[NAU];                                                // output tile of A operand n
    if (KIND == GW_CONVOUT) { ot[0] = 2 * wave; ot[1] = 2 * wave + 1; ot[NAU - 1] = 8 + (wave >> 1); }
    else { ot[0] = wave; ot[1] = 4; }
    static_assert(KIND != GW_CONVOUT || TI <= 2, "conv_out: the fifth tile's item tile is w & 1");
    auto o_of = [&](int k) { return 16 * ot[a_of(k)]; };                       // first output channel of tile k
    auto m_of = [&](int k) { return (k < NT4 ? b_of(k) : tix) * 16 + i; };     // this lane's item of tile k (column i of the tile)
    f32x4 tot[NTL], ysum[NTL];
#pragma unroll
    for (int k = 0; k < NTL; ++k) { tot[k] = zero; ysum[k] = zero; }
    uint32_t woff[NAU];
#pragma unroll
    for (int n = 0; n < NAU; ++n) woff[n] = (uint32_t)((kk * a.Co_pad + 16 * ot[n] + i) * 16);
    auto wload = [&](const __amdgpu_buffer_rsrc_t &wrs, int grp, int n) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, woff[n], grp * 16 * a.Co_pad * 4, 0));
    };
    f32x4 av[NGH][NAU];           // chain 0's weights of the CURRENT tap: requested before the previous tap's barrier (a0_load), so that a
    auto a0_load = [&](int t) {   // tap does not open with a memory round trip that every wave of the workgroup sits through together
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)a.tap[t].w, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int h = 0; h < NGH; ++h)
#pragma unroll
            for (int n = 0; n < NAU; ++n) av[h][n] = wload(wrs, 5 * h, n);
    };
    // products of tap t from buffer `buf`, added to tot[].  ALL: every item tile has an open lane -- straight-line code; else the
    // tiles of closed item tiles are left out (an exact zero) behind wave-uniform branches, one per tile and chain.
    constexpr bool BPRE = gw_occ(KIND, TI) == 2;   // B operands read a chain ahead too, where the register budget is the large one
    auto tap_products = [&](int t, int nxt, int buf, auto ALLc) {
        constexpr bool ALL = decltype(ALLc)::value;     // every tile of this wave is computed
        const unsigned tl = tiles_of(t);
        bool lv[NTL];
#pragma unroll
        for (int k = 0; k < NTL; ++k) lv[k] = ALL || (k < NT4 ? ((tl >> b_of(k)) & 1u) != 0 : has5 && ((tl >> tix) & 1u));
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)a.tap[t].w, 0, 0x7fffffff, 0x00020000);
        // (the last chain requests chain 0 of the NEXT open tap -- of this tap again when there is none, a valid address: the
        // request count stays the same on every path -- so that a tap does not open with a memory round trip)
        const __amdgpu_buffer_rsrc_t wnx = __builtin_amdgcn_make_buffer_rsrc((void *)a.tap[nxt < ntaps ? nxt : t].w, 0, 0x7fffffff, 0x00020000);
        f32x4 taptot[NTL], an[NGH][NAU], bn[NGH][NBU];
        auto bload = [&](int j, f32x4 (&dst)[NGH][NBU]) {
#pragma unroll
            for (int h = 0; h < NGH; ++h) {
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) dst[h][ti] = sB[buf * BUF + (ti * NG + j + 5 * h) * 64 + lane];
                dst[h][TI] = sB[buf * BUF + (tix * NG + j + 5 * h) * 64 + lane];
            }
        };
        if (BPRE) bload(0, bn);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            // the next chain's weights are requested under this chain's MFMAs ...
#pragma unroll
            for (int h = 0; h < NGH; ++h)
#pragma unroll
                for (int n = 0; n < NAU; ++n) an[h][n] = j < 4 ? wload(wrs, j + 1 + 5 * h, n) : wload(wnx, 5 * h, n);
            f32x4 bv[NGH][NBU], acc[NTL];
            if (BPRE) {
#pragma unroll
                for (int h = 0; h < NGH; ++h)
#pragma unroll
                    for (int q = 0; q < NBU; ++q) bv[h][q] = bn[h][q];
                if (j < 4) bload(j + 1, bn);
            } else {
                bload(j, bv);
            }
            __builtin_amdgcn_sched_barrier(0);   // ... and the scheduler may not pull their consumers up to them
            // chain j of the five tiles: group j (c = 0..3), then group j + 5 -- the tiles are independent accumulators
            if (ALL) {
#pragma unroll
                for (int h = 0; h < NGH; ++h)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int k = 0; k < NTL; ++k)
                            acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][a_of(k)][c], bv[h][b_of(k)][c], h == 0 && c == 0 ? zero : acc[k], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NTL; ++k) taptot[k] = j == 0 ? acc[k] : taptot[k] + acc[k];
            } else {
#pragma unroll
                for (int k = 0; k < NTL; ++k) {
                    if (!lv[k]) continue;
#pragma unroll
                    for (int h = 0; h < NGH; ++h)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][a_of(k)][c], bv[h][b_of(k)][c], h == 0 && c == 0 ? zero : acc[k], 0, 0, 0);
                    taptot[k] = j == 0 ? acc[k] : taptot[k] + acc[k];
                }
            }
#pragma unroll
            for (int h = 0; h < NGH; ++h)
#pragma unroll
                for (int n = 0; n < NAU; ++n) av[h][n] = an[h][n];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < NTL; ++k)
            if (lv[k]) tot[k] = tot[k] + taptot[k];
    };
    auto tap_dispatch = [&](int t, int nxt, int buf) {
        if (has5 && tiles_of(t) == (1u << TI) - 1u) tap_products(t, nxt, buf, std::integral_constant<bool, true>{});
        else tap_products(t, nxt, buf, std::integral_constant<bool, false>{});
    };
    // ---- the taps in slot order NA, C, NB (, SKIP); the open ones staged through sB
    int cur = next_live(-1), buf = 0;
    if (cur < ntaps) {
        stage_load(cur);
        a0_load(cur);
        stage_store(cur, 0);
    }
    __syncthreads();
    WG_STAMP(2);
    int nstamp = 3;
    (void)nstamp;
    for (int slot = 0; slot < a.nslots; ++slot) {
        for (int t = a.slot_first[slot]; t < a.slot_first[slot + 1]; ++t) {
            if (t != cur) continue;              // no open lane in the whole tile: an exact zero, skipped by every wave
            const int nxt = next_live(t);
            if (nxt < ntaps) stage_load(nxt);    // in flight under the MFMAs
            tap_dispatch(t, nxt, buf);
            if (DB) {
                if (nxt < ntaps) stage_store(nxt, buf ^ 1);
                __syncthreads();                 // next tap's rows visible; everybody is done with this tap's
                buf ^= 1;
            } else {
                __syncthreads();                 // everybody is done with this tap's rows
                if (nxt < ntaps) stage_store(nxt, 0);
                __syncthreads();
            }
            cur = nxt;
            WG_STAMP(nstamp);
            ++nstamp;
        }
        if (slot == SLOT_SKIP) {
            if (fuse_post) {   // (all taps are done: the B buffers are free -- the last tap ended with a barrier)
                float *sY = (float *)sB;
#pragma unroll
                for (int k = 0; k < NTL; ++k)
                    if (k < NT4 || has5) *(f32x4 *)(sY + m_of(k) * YLD + NF + o_of(k) + kk * 4) = tot[k];
            } else {
#pragma unroll
                for (int k = 0; k < NTL; ++k) {
                    const int item = (k < NT4 || has5) ? sItem[m_of(k)] : -1;
                    if (item >= 0) *(f32x4 *)(a.partial + ((size_t)SLOT_SKIP * a.nitems + item) * a.Co_pad + o_of(k) + kk * 4) = tot[k];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NTL; ++k)
                ysum[k] = (slot == SLOT_NA ? *(const f32x4 *)(a.sum_bias + o_of(k) + kk * 4) : ysum[k]) + tot[k];
            if (slot == SLOT_NB && !fuse_post) {
#pragma unroll
                for (int k = 0; k < NTL; ++k) {
                    const int item = (k < NT4 || has5) ? sItem[m_of(k)] : -1;
                    if (item >= 0) *(f32x4 *)(a.partial + ((size_t)SLOT_NA * a.nitems + item) * a.Co_pad + o_of(k) + kk * 4) = ysum[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NTL; ++k) tot[k] = zero;
    }
    // ---- the post op of the stage, in the same launch: y (and the gate half / the nin_skip slot) of the tile's items go through
    // LDS -- a row per item, where post_item (the code of k_post_grid) finds them -- and the four waves share out the items.
    // No partial sums in HBM, no second launch; the other workgroups of the CU keep the matrix pipes busy meanwhile.
    if (fuse_post) {
        float *sY = (float *)sB;
        // (conv_input with nin_skip: the skip slot was parked above, after the barrier of the last tap; here the taps are done too)
#pragma unroll
        for (int k = 0; k < NTL; ++k)
            if (k < NT4 || has5) *(f32x4 *)(sY + m_of(k) * YLD + o_of(k) + kk * 4) = ysum[k];
        __syncthreads();
        for (int m = wave; m < MI; m += GW_WAVES) {
            const int item = sItem[m];
            if (item < 0) continue;   // (wave-uniform)
            post_item<POSTK>(pa, item, lane, sY + m * YLD, 0, sY + m * YLD + NF);
        }
    }
    WG_STAMP(15);
#ifdef PS_WG_TRACE_BUILD
    if (tid == 0 && y < 4096) {
        unsigned hw = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_wg_span[KIND][y][1] = (wall_clock64() << 16) | (hw & 0xffffu);
    }
#endif
}

// One stage of the whole-grid pass in ONE kernel: a workgroup = one tile of 16 items, one wave per block of 32 output
// channels (3 waves for the 80-channel convs, 5 for conv_out); every wave walks all slots of its block (gemm_tiles,
// summing form), parks y -- and the nin_skip slot -- in LDS, and after one barrier the waves share out the 16 items for
// the post op.  No partial sums in HBM, no second launch.  Same arithmetic in the same order as k_gemm + k_post_grid.
template <int KIND>
__attribute__((amdgpu_waves_per_eu(PS_GEMM_WAVES, PS_GEMM_WAVES)))
__global__ __launch_bounds__(320) void k_stage_fused(GemmArgs a, PostArgs p)
{
    __shared__ __attribute__((aligned(16))) float sY[16 * SY_LD];
    __shared__ __attribute__((aligned(16))) float sS[KIND == POST_CONVIN ? 16 * SY_LD : 4];
    const int xcd = blockIdx.x & (N_XCD - 1), t = blockIdx.x >> 3;
    const int y = xcd * a.tpx + t;   // tile
    if (t >= a.tpx || y >= a.ny) return;
    const int x = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int o0 = x * 32;
    if (a.Co_pad - o0 >= 32) gemm_tiles<2>(a, o0, 0, a.nslots, y, sY, sS);
    else gemm_tiles<1>(a, o0, 0, a.nslots, y, sY, sS);
    __syncthreads();
    for (int it = x; it < 16; it += nw) {
        const int item = y * 16 + it;
        if (item >= a.nitems) break;
        if (!item_wanted(p.items, item)) continue;
        post_item<KIND>(p, item, lane, sY + it * SY_LD, 0, sS + it * SY_LD);
    }
}

// ------------------------------------------------------------------------------------------
// Which prefix items does anybody read?  The whole-grid pass over the observed prefix of an AR run exists for ONE reason:
// the column steps read the finished activations of earlier neighbours.  A column reads, per stage, the open taps of its
// location -- so from the prefix only a band along the frontier; those items read their own open taps one stage
// earlier, and so on backwards through the 32 stages: a dependency cone, not the whole prefix at every stage (63-83 %
// of the work for PixelSynth's orders).  Because the generation order sweeps towards the frontier, the cone
// of a stage is -- up to a few items -- a SUFFIX of the prefix in rank order, so it is kept as one number per (stage,
// frame): the smallest rank anyone reads; items of lower rank are skipped at that stage (their cache rows keep whatever
// they held; nothing reads them).  The taps come from the kernel masks themselves, exactly what the kernels follow.
// One workgroup per frame; starts[(stage id) * F + f] with stage ids: 0 u_init, 1 + g conv_input / nin_skip of gated
// block g, 15 + g its conv_out, 29 + d dilated conv d.
// ------------------------------------------------------------------------------------------
constexpr int N_EVAL = 1 + 2 * NGATED + 4;   // 33
struct StartsArgs {
    const int32_t *order;   // (F, L)
    const float *mask_und, *mask_dil;   // (F, 9, L): type B dilation 1 / dilation 2
    int H, W, L, npre, F;
    int g_in[NGATED], g_out[NGATED], g_skip[NGATED], d_in[4], d_out[4];
    int32_t *starts;        // (N_EVAL, F)
    int f0;                 // frames [f0, f0 + gridDim.x) of the F
};
constexpr int STARTS_MAXL = 4096;
__global__ __launch_bounds__(1024) void k_prefix_starts(StartsArgs a)
{
    __shared__ int rank[STARTS_MAXL];   // by location
    __shared__ int s1[STARTS_MAXL];     // by rank < npre: min rank among the open dilation-1 taps of ranks >= r (suffix minimum)
    __shared__ int s2[STARTS_MAXL];     //                 the same, dilation-2 taps of the dilated mask
    __shared__ int cmin[2];             // min rank the COLUMNS (ranks >= npre) read through dilation-1 / dilation-2 taps
    const int f = a.f0 + blockIdx.x, t = threadIdx.x, L = a.L, npre = a.npre;
    const int32_t *ord = a.order + (size_t)f * L;
    for (int r = t; r < L; r += 1024) rank[ord[r]] = r;
    if (t < 2) cmin[t] = npre;
    __syncthreads();
    for (int r = t; r < L; r += 1024) {
        const int q = ord[r], y = q / a.W, x = q - y * a.W;
        int m1 = npre, m2 = npre;
        for (int tap = 0; tap < 9; ++tap) {
            if (tap == 4) continue;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if (a.mask_und[((size_t)f * 9 + tap) * L + q] != 0.0f) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) m1 = min(m1, rank[yy * a.W + xx]);
            }
            if (a.mask_dil[((size_t)f * 9 + tap) * L + q] != 0.0f) {
                const int yy = y + 2 * dy, xx = x + 2 * dx;
                if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) m2 = min(m2, rank[yy * a.W + xx]);
            }
        }
        if (r < npre) { s1[r] = m1; s2[r] = m2; }
        else { atomicMin(&cmin[0], m1); atomicMin(&cmin[1], m2); }
    }
    __syncthreads();
    for (int off = 1; off < npre; off <<= 1) {   // suffix minima by doubling
        int v1[STARTS_MAXL / 1024], v2[STARTS_MAXL / 1024];
#pragma unroll
        for (int k = 0; k < STARTS_MAXL / 1024; ++k) {
            const int r = t + 1024 * k;
            if (r < npre) {
                v1[k] = r + off < npre ? min(s1[r], s1[r + off]) : s1[r];
                v2[k] = r + off < npre ? min(s2[r], s2[r + off]) : s2[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < STARTS_MAXL / 1024; ++k) {
            const int r = t + 1024 * k;
            if (r < npre) { s1[r] = v1[k]; s2[r] = v2[k]; }
        }
        __syncthreads();
    }
    if (t != 0) return;
    auto suf = [&](const int *s, int r0) { return r0 >= npre ? npre : min(r0, s[r0]); };   // ranks [r0, npre) and all they read
    int need[NNODE], needX[NGATED];
    for (int n = 0; n < NNODE; ++n) need[n] = npre;
    for (int g = 0; g < NGATED; ++g) { needX[g] = cmin[0]; need[a.g_in[g]] = min(need[a.g_in[g]], cmin[0]); }
    for (int d = 0; d < 4; ++d) need[a.d_in[d]] = min(need[a.d_in[d]], cmin[1]);
    // backwards through the stages in execution order (run_grid): G = gated block, D = dilated conv
    const int kind[18] = {0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    const int idx[18] = {0, 1, 0, 2, 3, 1, 4, 5, 6, 7, 2, 8, 9, 10, 3, 11, 12, 13};
    int32_t *out = a.starts + f;
    for (int e = 17; e >= 0; --e) {
        if (kind[e] == 0) {
            const int g = idx[e];
            const int so = need[a.g_out[g]];                     // conv_out + gate evaluated from rank so on
            out[(size_t)(15 + g) * a.F] = so;
            needX[g] = min(needX[g], suf(s1, so));               //   reads conv_input's output at its open taps
            need[a.g_in[g]] = min(need[a.g_in[g]], so);          //   and the residual input at the same location
            const int si = needX[g];                             // conv_input (+ nin_skip) evaluated from rank si on
            out[(size_t)(1 + g) * a.F] = si;
            need[a.g_in[g]] = min(need[a.g_in[g]], suf(s1, si));
            if (a.g_skip[g] >= 0) need[a.g_skip[g]] = min(need[a.g_skip[g]], si);
        } else {
            const int d = idx[e];
            const int sd = need[a.d_out[d]];
            out[(size_t)(29 + d) * a.F] = sd;
            need[a.d_in[d]] = min(need[a.d_in[d]], suf(s2, sd));
        }
    }
    out[0] = need[0];   // u_init + norm_init
}

struct UinitArgs {
    ItemMap items;
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
    if (item >= a.nitems || !item_wanted(a.items, item)) return;
    int f, q;
    item_loc(a.items, item, a.L, f, q);
    const size_t loc = (size_t)f * a.L + q;
    const bool own = lane < PONO_LANES;
    const int c = own ? 2 * lane : 0;
    float mA[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) mA[t] = a.mask[((size_t)f * 9 + t) * a.L + q];
    const f32x2 y = uinit_gather<f32x2>(a.codes + (size_t)f * a.L, mA, a.w, a.bias, q, a.H, a.W, c);
    const float mean = pono_mean(pono_total(y, own));   // norm_init
    const f32x2 d = y - mean;
    const float inv = pono_inv(pono_total(d * d, own));
    if (own) store_raw_celu2(a.Rout, a.Eout, loc, c, d * inv);
}

// logits = nin_out partial + bias; nchw: (F,512,H,W) like the reference, else (nitems,512)
__global__ __launch_bounds__(256) void k_logits_grid(ItemMap items, const float *partial, const float *bias, int nitems, int L,
                                                     int nchw, float *logits)
{
    const int item = blockIdx.x;
    int f, q;
    item_loc(items, item, L, f, q);
    for (int o = threadIdx.x; o < NCLS; o += 256) {
        const float v = partial[(size_t)item * NCLS + o] + bias[o];
        if (nchw) logits[((size_t)f * NCLS + o) * L + q] = v;
        else logits[((size_t)f * L + q) * NCLS + o] = v;
    }
}

// ==========================================================================================
// column mode: one location per frame per order position (the incremental AR step) -- k_column below.
//   neighbour role   every NEIGHBOUR-tap partial sum (slots NA, NB) of all 32 masked convs at once.  They only
//                    read finished columns of earlier order positions, so they do not depend on this
//                    position's chain and run fully parallel (one wave = one tap of one stage x slot x 16 channels).
//   chain role       one workgroup per frame walks the 33 stages in order.  Only the centre taps (1x1 products on
//                    the fresh activation) and the post ops are sequential; activations go stage to stage through
//                    LDS, weights stream from L2 into registers.  Ends with the categorical draw and the context of
//                    the next order position.
// ==========================================================================================
// A COLUMN = one order position of one frame.  Everything a launch needs about a column that does not depend on the
// run so far sits in one 160-byte record, so that it costs one memory round trip: frame, location, the mask values of
// the location and where the u_init gather finds the codes of its (earlier) neighbours.  The records of a whole run
// are written once (k_ctx_build), in schedule order: a launch works on a contiguous slice of them.
//
// Wavefronts.  Column (f, i) reads the finished columns of the locations that are BOTH a tap neighbour (3x3, dilation 1
// or 2) of its location and earlier in the frame's order -- nothing else; in particular not the column of position
// i - 1 unless that one happens to be such a neighbour.  So the columns of a frame form a DAG whose depth (60-110 for
// PixelSynth's orders over 400-700 walked positions: the order sweeps a frontier, and along a frontier only every
// other cell or so depends on the previous one) is the number of dependent launches, not the number of positions:
// all columns of one DAG level (a "wavefront", host: ps_ar_wavefronts) go into ONE launch, each with its own chain
// workgroup.  Every column is computed exactly as in the position-by-position walk (which is the special case of one
// column per frame and launch), so the results are bit-identical.
struct StepCtx {
    int q;            // location
    int f;            // frame
    float m[3][9];    // mask values of location q: [0] type A dil 1, [1] type B dil 1, [2] type B dil 2
    int nloc[9];      // location of the type-A neighbour of every tap (u_init gather), -1 where the tap is closed
    int pad[2];
};
static_assert(sizeof(StepCtx) == 160, "one record = 160 bytes");

// cache rows (frame * L + location) of the eight neighbour taps of a column for the two mask kinds the convs use
// (type B dilation 1, type B dilation 2), -1 = tap closed or outside the grid; taps 0..3 = slot NA, 4..7 = slot NB
struct ColTaps { int row[2][8]; };
static_assert(sizeof(ColTaps) == 64, "one record = 64 bytes");

struct CtxArgs {
    StepCtx *ctx;     // [columns of the run]
    ColTaps *taps;    // [columns of the run] neighbour rows for the throughput form (k_column_tp)
    const int32_t *order;
    const float *mask[3];
    int F, L;
};

// neighbour code of type-A tap t of location q (-1: closed tap or outside the grid)
__device__ __forceinline__ int ctx_nbr_loc(int q, int t, int H, int W)
{
    const int r = q / W, c = q - r * W, rr = r + t / 3 - 1, cc = c + t % 3 - 1;
    return (rr >= 0 && rr < H && cc >= 0 && cc < W) ? rr * W + cc : -1;
}

// records of `ncols` columns: cols = (frame, order position) pairs in schedule order, or null for the plain walk
// (column k = frame k % F at position first + k / F)
__global__ __launch_bounds__(32) void k_ctx_build(CtxArgs a, const int32_t *cols, int ncols, int first, int H, int W, int *err)
{
    const int k = blockIdx.x, t = threadIdx.x;
    if (k >= ncols) return;
    int f = cols ? cols[2 * k] : k % a.F, i = cols ? cols[2 * k + 1] : first + k / a.F;
    if (f < 0 || f >= a.F || i < first || i >= a.L) {  // a schedule that does not belong to this run: flag it, stay in bounds
        if (t == 0) *err = 2;
        f = 0;
        i = first;
    }
    StepCtx *c = a.ctx + k;
    const int q = a.order[(size_t)f * a.L + i];
    if (t < 27) c->m[t / 9][t % 9] = a.mask[t / 9][((size_t)f * 9 + t % 9) * a.L + q];
    if (t == 27) { c->q = q; c->f = f; }
    if (t < 9) {
        const int loc = ctx_nbr_loc(q, t, H, W);
        const float mA = a.mask[0][((size_t)f * 9 + t) * a.L + q];
        c->nloc[t] = (loc >= 0 && mA != 0.0f) ? loc : -1;
    }
    if (t < 16) {  // neighbour rows for k_column_tp: kind 0 = type B dilation 1, kind 1 = type B dilation 2; the masks are 0 / 1
        const int kind = t >> 3, tq = t & 7, tap = tq < 4 ? tq : tq + 1, dil = kind + 1;
        const int r = q / W, cc = q - r * W, rr = r + (tap / 3 - 1) * dil, c2 = cc + (tap % 3 - 1) * dil;
        const bool in = rr >= 0 && rr < H && c2 >= 0 && c2 < W;
        const float mv = a.mask[1 + kind][((size_t)f * 9 + tap) * a.L + q];
        a.taps[k].row[kind][tq] = (in && mv != 0.0f) ? f * a.L + rr * W + c2 : -1;
    }
}

enum { PRO_UINIT = 0, PRO_CONVIN = 1, PRO_GATE = 2, PRO_DIL = 3 };
enum { IN_CELU = 0, IN_RAW = 1, IN_ELU = 2 };
constexpr int NST = 33;       // 14 x (conv_input, conv_out) + 4 dilated convs + nin_out
constexpr int NBR_LD = 2 * NF;

// Host-side description of one of the 33 stages (build_stage_table); the kernels read the tables derived from it:
// NbrWork records (neighbour role) and the control records of the chain role.
struct StageDesc {
    int pro, in_form, save_slot /* keep this u in LDS, -1 */, p_has_skip;
    int NG, Co_pad, center_tap, skip_slot /* saved u_k feeding w_skip, -1 */;
    const float *w;       // packed weights [taps][NG*4][Co_pad][4]
    const float *w_skip;  // packed nin_skip [40][80][4] or null
    const float *in;      // cache the neighbour taps gather from (E / X / R at earlier order positions)
    int in_ld, dil, mask_kind, has_nbr;
    // prologue of this stage = post op of the previous stage
    const float *pbias, *pbias2;
    float *outR, *outE, *outX;  // caches the prologue writes at the current location
    // the centre-tap (+ nin_skip) weights again, laid out for the chain role: [nstep][nchain][4]
    const float *wv;
    int nchain, nstep;
};

// a work item of the neighbour role = (stage, slot NA|NB, 16 output channels), with everything it needs of the stage inline:
// one dependent fetch instead of work item -> stage description -> data
struct __attribute__((aligned(16))) NbrWork {
    const float *w;   // packed weights of the conv [taps][NG*4][Co_pad][4]
    const float *in;  // cache the taps gather from
    int stage, half, cog, NG;
    int Co_pad, in_ld, dil, mask_kind;
};

// Completion counters of the neighbour role: one per (stage, 16-column tile of the launch), each on its own 128-byte
// line -- several thousand items finish per launch, and atomics on one line are served one after the other by the
// memory side (counters packed in two lines made the neighbour role atomics-bound and every chain's polls queue behind
// them: 128 columns 84 -> see DESIGN).  A chain only watches the counters of its own tile.
constexpr int COL_CAP = 128;  // columns per launch: 4 chain XCDs x 32 CUs (larger wavefronts are split)
constexpr int MAX_TILES = COL_CAP / 16, CNT_PAD = 32 /* dwords */;
__device__ __host__ __forceinline__ size_t cnt_index(int stage, int tile) { return ((size_t)stage * MAX_TILES + tile) * CNT_PAD; }

struct NbrArgs {
    const NbrWork *work;
    const StepCtx *ctx;   // records of this launch's columns
    float *nbr;           // [NST][2][col_stride][NBR_LD]
    int nwork, H, W, L;
    int ncols;            // columns of this launch
    int col_stride;       // column capacity of the nbr buffer
    int tiles;            // 16-column tiles = ceil(ncols / 16)
    int chain_xcds;       // the chain workgroups are the blocks on XCDs 0 .. chain_xcds-1 (see k_column)
    unsigned *cnt;        // [NST][MAX_TILES] padded completion counters of this handle: work items done, ever (cnt_index)
    int nbr_wgs, groups;  // neighbour-role workgroups of the launch; work items each of them runs at a time (2 or 4)
    int debug;            // tuning only
    int *err;             // set to 1 if a bounded wait ran out (ps_pixelcnn_status)
    // look-ahead (as in the throughput form, nbr_role_tp): work entries [w_from, nwork) for this launch's columns, then entries
    // [0, w_upto) -- the stages below the split -- for the NEXT launch's columns, into the other half of the double-buffered
    // slots / counters, each item once the chain workgroups of this launch have published the input of its stage (`done`)
    int w_from, w_upto;
    const StepCtx *ctx_next;
    int ncols_next, tiles_next;
    float *nbr_next;
    unsigned *cnt_next;
    const unsigned *done;
    unsigned done_target;
    int split;            // the stages below it are the look-ahead's
};

// one neighbour tap of one conv for 16 columns x 16 output channels, from fresh accumulators
// AHEAD: an item of the NEXT launch's columns.  Some of its rows were stored (write-through) by chain workgroups of THIS launch
// on other XCDs; a stale copy can only be in this CU's L1 / this XCD's L2 if the line was read earlier in the launch: rows of
// 80 floats share lines with their neighbours' (device-scope loads for those), and the dummy reads of closed lanes -- which is why
// a closed lane reads a row another lane gathers anyway, in every launch (see nbr_item_tp).
template <int NG, bool EAGER, bool AHEAD = false>
__device__ __forceinline__ f32x4 nbr_tap(const NbrWork &sd, const NbrArgs &a, const StepCtx *recs, int t, int o0, int col, bool valid,
                                         int i, int kk)
{
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    float mv = 0.0f;
    int row = -1;
    if (valid) {
        const StepCtx &cx = recs[col];
        const int q = cx.q, f = cx.f;
        const int r = q / a.W, c = q - r * a.W;
        const int rr = r + (t / 3 - 1) * sd.dil, cc = c + (t % 3 - 1) * sd.dil;
        if (rr >= 0 && rr < a.H && cc >= 0 && cc < a.W) {
            mv = cx.m[sd.mask_kind][t];
            row = f * a.L + rr * a.W + cc;
        }
    }
    const bool live = mv != 0.0f;
    const unsigned long long open = __builtin_amdgcn_ballot_w64(live);
    if (open == 0ull) return zero;
    const int safe = __shfl(row, __builtin_ctzll(open), 64);   // a masked row is not fetched: the lane reads one that is being read anyway
    const int rowq = live ? row : safe;
    const float *src = sd.in + (size_t)rowq * sd.in_ld + 4 * kk;
    Acc5 acc = acc5_zero();
    const float *wbase = sd.w + (size_t)t * NG * 16 * sd.Co_pad + ((size_t)kk * sd.Co_pad + o0 + i) * 4;
    f32x4 av[NG], bv[NG];
    // How many input-row loads a wave keeps in flight matters beyond this role: with all ten at once (EAGER) the neighbour
    // role alone is 13 % faster, but a large launch as a whole 6 % slower -- the chains on the other XCDs wait longer for
    // their own operands.  So only the FIRST round of a launch is eager (its items are the stages the chains are already
    // waiting for); later rounds fetch one 80-channel chunk of rows at a time, multiply it, then fetch the next.  (Loading
    // under the lane condition `live ? *p : 0` -- one round trip per load, see k_gemm -- was within 1 % of that.)
#pragma unroll
    for (int g = 0; g < NG; ++g) av[g] = *PS_GC(f32x4, wbase + (size_t)g * 16 * sd.Co_pad);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void *)sd.in, 0, 0x7fffffff, 0x00020000);
    [[maybe_unused]] const int voff = (rowq * sd.in_ld + 4 * kk) * 4;
    auto brow = [&](int g) {
        if (AHEAD && NG == 5) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, voff + 64 * g, 0, 16 /* sc1 */));
        return *PS_GC(f32x4, src + 16 * g);
    };
    if (EAGER) {
#pragma unroll
        for (int g = 0; g < NG; ++g) bv[g] = brow(g);
    }
#pragma unroll
    for (int g0 = 0; g0 < NG; g0 += 5) {
        if (!EAGER) {
#pragma unroll
            for (int g = g0; g < g0 + 5; ++g) bv[g] = brow(g);
        }
#ifndef PS_NBR_NO_PIN
        // (the loads stay unconditional: otherwise the compiler sinks one of them under `live` and waits for it with
        // vmcnt(0), which also drains the weight loads in flight -- see gemm_tiles)
#pragma unroll
        for (int g = g0; g < g0 + 5; ++g) asm volatile("" : "+v"(bv[g]));
#endif
#pragma unroll
        for (int g = g0; g < g0 + 5; ++g) bv[g] = live ? bv[g] * mv : zero;
        const f32x4 (&a5)[5] = *reinterpret_cast<const f32x4 (*)[5]>(&av[g0]);
        const f32x4 (&b5)[5] = *reinterpret_cast<const f32x4 (*)[5]>(&bv[g0]);
        mfma_chunk5(a5, b5, acc);
        if (!EAGER) asm volatile("" ::: "memory");  // keeps the next chunk's loads behind this chunk's MFMAs
    }
    return chunk_total(acc);
}

// Results that another workgroup of the SAME launch consumes (k_column: neighbour slots -> chain) leave with
// write-through stores (sc1: past this XCD's L2, which is not coherent with the consumer's); the consumer reads them
// with device-scope loads after it has seen the completion counter.
// (hipcc pads no hazard wait states around an asm statement: a store of more than 64 bits still reads its data registers
// when the next instruction issues, and the compiler is free to overwrite them there -- two of these back to back, the
// second address computed into the first one's data registers, stored address bits for a quarter of the lanes.  The
// s_nop covers the VMEM-store-data hazard.)
__device__ __forceinline__ void store_through(float *p, const f32x4 &v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" : : "v"(PS_G(f32x4, p)), "v"(v) : "memory");
}
__device__ __forceinline__ void store_through2(float *p, const f32x2 &v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" : : "v"(PS_G(f32x2, p)), "v"(v) : "memory");
}
__device__ __forceinline__ void signal_done(unsigned *counter, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left
    if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Neighbour-tap role of k_column.  A work item = (stage, slot NA|NB, 16 output channels) for a tile of 16 columns: its
// 4 waves take the 4 taps of the slot and the partials are added in tap order through LDS (the order k_gemm uses).
// A workgroup runs `groups` items at a time, four waves each (2 for small launches: with all sixteen waves at work every
// SIMD interleaves four MFMA chains, 4 x 40 x 32 cycles = 2.2 us before the first result; 4 when there are more items
// than CUs x 2), and walks the item list round by round: item (round * workgroups + nb) * groups + group -- stage-major
// over the tiles, so the first stages of every tile come first.  An item's completion is published (its stage's
// counter) once its write-through stores have left; that wait is folded into the NEXT round's wait for its operands
// (vmcnt is in order), only the last round drains on its own.  The workgroups of a launch are all resident (at most
// one per CU), so nothing here ever waits for another workgroup.
// (Tried and dropped, each slower because the 128-register budget of a 1024-thread workgroup spills: fetching the next
// round's records a round ahead; one wave per item with its four taps in sequence and no barrier; items of two column
// tiles that keep the tap's weights in registers.)
// Bound of the in-launch waits on the neighbour role's completion counters: a hang guard, not a schedule.  A wait is normally
// over before it starts; it lasts when workgroups of the launch are not resident yet because kernels of ANOTHER stream hold their
// CUs (bench.py / driver.py run the next batch's splat under this batch's AR run: a stream of 64-thread workgroups can keep a
// 512- or 1024-thread workgroup that needs most of a CU's LDS waiting for as long as that kernel lasts, milliseconds).  Round 2's
// bounds (20 000 / 40 000 polls of >= 128 clocks: a few ms) were inside that range and expired now and then (one bench run in
// six); 2^24 polls are seconds -- still finite, so a lost workgroup ends as an error from ps_pixelcnn_status, not as a hung GPU.
constexpr int WAIT_SPINS = 1 << 24;
constexpr int NBR_MAX_GROUPS = 4;
constexpr int NWORK_MAX = 512;  // work-table entries the neighbour role can stage (this network: 460)

__device__ __forceinline__ void nbr_role(const NbrArgs &a, int nb)
{
    __shared__ __attribute__((aligned(16))) float sNP[2][NBR_MAX_GROUPS][4][16][20];
    // the launch's column records and the work table, staged once: a round then starts with two LDS reads instead of
    // two dependent trips to memory (work record -> column record) before its operands can even be requested
    __shared__ __attribute__((aligned(16))) StepCtx sCtx[COL_CAP];
    __shared__ __attribute__((aligned(16))) StepCtx sCtxN[COL_CAP];   // the NEXT launch's records (look-ahead)
    __shared__ __attribute__((aligned(16))) NbrWork sWork[NWORK_MAX];
    __shared__ unsigned sArr[2][NBR_MAX_GROUPS], sRd[NBR_MAX_GROUPS], sGo[NBR_MAX_GROUPS];   // sGo: look-ahead stages wave 0 has seen published, + 1
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    const int grp4 = wave >> 2, w4 = wave & 3;
    if (nb >= a.nbr_wgs) return;
    {
        if (threadIdx.x < NBR_MAX_GROUPS) { sArr[0][threadIdx.x] = 0; sArr[1][threadIdx.x] = 0; sRd[threadIdx.x] = 0; sGo[threadIdx.x] = 0; }
        const int nc = a.ncols * (int)(sizeof(StepCtx) / 16), nw = a.nwork * (int)(sizeof(NbrWork) / 16);
        const int nx = a.w_upto > 0 ? a.ncols_next * (int)(sizeof(StepCtx) / 16) : 0;
        for (int k = threadIdx.x; k < nc + nw + nx; k += (int)blockDim.x) {
            if (k < nc) ((uint4 *)sCtx)[k] = ((const uint4 *)a.ctx)[k];
            else if (k < nc + nw) ((uint4 *)sWork)[k - nc] = ((const uint4 *)a.work)[k - nc];
            else ((uint4 *)sCtxN)[k - nc - nw] = ((const uint4 *)a.ctx_next)[k - nc - nw];
        }
        __syncthreads();
    }
    if (grp4 >= a.groups) return;
    // From here on the four waves of a group only synchronise with each other, through two monotone LDS counters (no
    // workgroup barrier: the groups drift apart, so one group's MFMAs run under another group's operand fetches instead
    // of all sixteen waves fetching, multiplying and exchanging in lock-step):
    //   sArr[r & 1][g]  partials the tap waves 1..3 have written in rounds of that parity (3 per round; per parity,
    //            because a tap wave may be one round ahead of wave 0); wave 0 adds up round r once it reads 3 (r / 2 + 1);
    //   sRd[g]   rounds wave 0 has consumed; a tap wave reuses exchange buffer r & 1 once rounds <= r - 2 are consumed.
    // Every wait is bounded (a lost wave sets the handle's error flag instead of hanging the GPU).
    const int n_own = (a.nwork - a.w_from) * a.tiles;
    const int nitems = n_own + a.w_upto * a.tiles_next, per_round = a.nbr_wgs * a.groups;
    unsigned *pending = nullptr;  // counter of the item this group finished in the previous round, not yet published
    auto spin_until = [&](const unsigned *flag, unsigned want) {
        int spins = 0;
        while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - want) < 0) {
            if (++spins > (1 << 22)) { if (lane == 0) *a.err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    unsigned r = 0;  // rounds this group has worked on
    int ready_upto = -1;   // look-ahead: stages whose input the chain workgroups of this launch are known to have published
    for (int base = 0; base < nitems; base += per_round) {
        const int item = base + nb * a.groups + grp4;
        if (item >= nitems) break;  // (the four waves of a group agree)
        const bool ahead = item >= n_own;
        int witem, ctile;
        if (!ahead) {
            const int q = item / a.tiles;
            witem = a.w_from + q; ctile = item - q * a.tiles;
        } else {
            const int j = item - n_own;
            witem = j / a.tiles_next; ctile = j - witem * a.tiles_next;
        }
        const int par = r & 1;
        const NbrWork wk = sWork[witem];
        const int col = ctile * 16 + i;
        const bool valid = col < (ahead ? a.ncols_next : a.ncols);
        const int t = wk.half * 5 + w4;  // taps 0..3 (NA) or 5..8 (NB)
        f32x4 part;
        if (ahead && wk.stage > ready_upto) {
            // The chain workgroups of this launch must have stored the input of the item's stage (`done`).  Only the group's wave 0
            // looks at the counters in memory -- the other three take its word through LDS -- and it polls slowly: with every wave
            // polling every 0.2 us the counters' lines were hammered from 2752 waves, and the chains' own device-scope traffic (and
            // their publishing atomics, on the same lines) slowed down by 1.4 us per look-ahead stage.  One look at the LAST
            // look-ahead stage's counter settles it for the rest of the launch when the chains are that far already; otherwise
            // wait for this stage's (bounded) -- but not with this group's previous item unpublished behind the wait: the chains
            // that publish `done` may be waiting for exactly that item.
            unsigned val;
            if (w4 == 0) {
                const unsigned *dl = a.done + (size_t)(a.split - 1) * CNT_PAD;
                if ((int)(__hip_atomic_load(dl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.done_target) >= 0) {
                    val = (unsigned)a.split;
                } else {
#ifdef PS_LA_COUNT
                    if (lane == 0) atomicAdd((unsigned *)a.done + (size_t)wk.stage * CNT_PAD + 1, 1u);
#endif
                    if (pending) {
                        signal_done(pending, lane);
                        pending = nullptr;
                    }
                    const unsigned *dp = a.done + (size_t)wk.stage * CNT_PAD;
                    unsigned have = __hip_atomic_load(dp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int spins = 0;
                    while ((int)(have - a.done_target) < 0) {
                        if (++spins > (WAIT_SPINS >> 4)) { if (lane == 0) *a.err = 1; break; }
                        __builtin_amdgcn_s_sleep(100);
                        have = __hip_atomic_load(dp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    val = (unsigned)wk.stage + 1u;
                }
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_store(&sGo[grp4], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                int spins = 0;
                while ((val = __hip_atomic_load(&sGo[grp4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < (unsigned)wk.stage + 1u) {
                    if (++spins > WAIT_SPINS) { if (lane == 0) *a.err = 1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            ready_upto = (int)val - 1;
            asm volatile("" ::: "memory");
        }
        if (ahead) {
            part = wk.NG == 10 ? nbr_tap<10, false, true>(wk, a, sCtxN, t, wk.cog * 16, col, valid, i, kk)
                               : nbr_tap<5, false, true>(wk, a, sCtxN, t, wk.cog * 16, col, valid, i, kk);
        } else if (r == 0) {
            part = wk.NG == 10 ? nbr_tap<10, true>(wk, a, sCtx, t, wk.cog * 16, col, valid, i, kk)
                               : nbr_tap<5, true>(wk, a, sCtx, t, wk.cog * 16, col, valid, i, kk);
        } else {
            part = wk.NG == 10 ? nbr_tap<10, false>(wk, a, sCtx, t, wk.cog * 16, col, valid, i, kk)
                               : nbr_tap<5, false>(wk, a, sCtx, t, wk.cog * 16, col, valid, i, kk);
        }
        if (w4 != 0) {
            if (r >= 2) spin_until(&sRd[grp4], r - 1);
            *(f32x4 *)(&sNP[par][grp4][w4][i][kk * 4]) = part;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&sArr[par][grp4], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (pending) {  // the operands of this round have arrived, so the older stores have left too
                signal_done(pending, lane);
                pending = nullptr;
            }
            spin_until(&sArr[par][grp4], 3 * (r / 2 + 1));
            if (valid) {
                const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
                f32x4 tot = zero + part;
#pragma unroll
                for (int w = 1; w < 4; ++w) tot = tot + *(const f32x4 *)(&sNP[par][grp4][w][i][kk * 4]);
                store_through((ahead ? a.nbr_next : a.nbr) + (((size_t)wk.stage * 2 + wk.half) * a.col_stride + col) * NBR_LD + wk.cog * 16 + kk * 4, tot);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&sRd[grp4], r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pending = (ahead ? a.cnt_next : a.cnt) + cnt_index(wk.stage, ctile);
        }
        ++r;
    }
    if (w4 == 0 && pending) signal_done(pending, lane);
}

struct ChainArgs {
    const int *ctl1;          // the chain role's per-stage control records (C1_CTL_DWORDS dwords each), read with scalar loads
    const float *nbr;         // neighbour slots of this launch, from the neighbour role
    const float *uinit_w, *uinit_b;
    const int32_t *codes_in;  // (F,L) current codes: the u_init gather reads earlier positions
    const StepCtx *ctx;       // records of this launch's columns (workgroup k of the chain role takes column k)
    const float *out_b;
    int H, W, L;
    int ncols, col_stride;    // columns of this launch / column capacity of the nbr buffer
    // end of the column
    int32_t *codes;           // (F,L) written for sampled locations, or null (logits only)
    const uint8_t *region;    // (F,L) by location
    const int32_t *forced;    // (F,L) by location or null
    const float *uniforms;    // (F,L) by location or null
    float *out_logits;        // (F,L,512) by location or null
    float *step_logits;       // (F,512) by frame or null
    float temperature;
    const unsigned *cnt;       // completion counters of the neighbour role (NbrArgs::cnt)
    unsigned tile_uses[MAX_TILES];  // launches of this handle so far that had a tile t, this one included: the counters are
                               // never reset, counter (k, t) stands at tile_uses[t] x (items of stage k per tile) when done
    int *err;                  // set to 1 if a bounded wait ran out (ps_pixelcnn_status)
    unsigned long long *trace; // optional [NST][10] shader-clock stamps of workgroup 0 (tuning aid)
    int debug;                 // tuning only (PS_COLUMN_DEBUG): 1 = chains do not wait for the neighbour slots, 2 = no chains,
                               // 3 = no neighbour role and no waiting
    // look-ahead form (k_column_la, chain_role<FPW, true>): the slots of the stages below `la_split` were computed by the launch in
    // front (use counts uses_lo), the others by this one (uses_hi); `nbr` / `cnt` are the halves of this launch's parity; the
    // columns publish, stage by stage, that the input of stage k is in memory (`done`) for the neighbour role's look-ahead items
    int la_split;
    unsigned uses_lo[MAX_TILES], uses_hi[MAX_TILES];
    unsigned *done;
    int publish_upto;
};

// categorical draw from logits / T by inverse CDF with one uniform (sample.py:60-66); lane l holds classes 8l..8l+7
__device__ __forceinline__ int draw_code(const float (&lg)[8], float temperature, float u, int lane)
{
    float x[8], m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) { x[k] = lg[k] / temperature; m = fmaxf(m, x[k]); }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float e[8], ls = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { e[k] = expf(x[k] - m); ls += e[k]; }
    float incl = ls;  // inclusive scan of the per-lane sums (classes are lane-major)
    for (int off = 1; off < 64; off <<= 1) {
        const float tv = __shfl_up(incl, off, 64);
        if (lane >= off) incl += tv;
    }
    const float total = __shfl(incl, 63, 64);
    const float target = u * total;
    float run = incl - ls;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { run += e[k]; cnt += run <= target ? 1 : 0; }  // classes whose cdf <= target
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    return min(cnt, NCLS - 1);
}

// Workgroup barrier that only drains LDS traffic.  __syncthreads() also waits for every outstanding
// global access (vmcnt(0)), which would serialise the weight / neighbour-slot prefetches of k_chain
// against its two barriers per stage; the data exchanged between the waves here lives in LDS only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ==========================================================================================
// chain role: the 33-stage chain of one frame on the vector ALU, one workgroup (= one CU) per frame.
// A 16-frame MFMA tile per CU would leave 15/16 of the chip idle at PixelSynth's frame counts; fp32 FMA on the
// VALU has the same peak as fp32 MFMA on gfx950, so every frame gets its own CU and nothing is padded.
// Thread t of a stage owns ONE chain (output o, accumulator j) of mfma_chunk5's order -- 16 or 32 dependent
// v_fma_f32 -- with its weights in registers (layout [step][chain][4], one coalesced 16-byte load per step, three
// buffers: fetched two stages ahead) and the input read from LDS as broadcasts.  The five chain values per output
// meet in LDS; the frame's post op (PONO, gate / skip / residual, concat-ELU) is done by ONE wave, one channel per
// lane with DPP reductions, exactly like k_post_grid.  Two LDS-only barriers per stage.
// ==========================================================================================
constexpr int C1_THREADS = 1024;
constexpr int C1_MAXCHAIN = 800;   // 5 x 160, or 5 x 80 + 5 x 80 (conv_input + nin_skip)
constexpr int SX_LD = 2 * NF;
constexpr int C1_OUT_STEPS = 12;   // nin_out: thread (o, part): part 0 = chains 0..2, part 1 = chains 3..4

template <int NGL, int FPW>
__device__ __forceinline__ void valu_chain(const f32x4 *w /*4 * NGL steps*/, const float *xbase, int j, float (&acc)[FPW])
{
#pragma unroll
    for (int f = 0; f < FPW; ++f) acc[f] = 0.0f;
#pragma unroll
    for (int gl = 0; gl < NGL; ++gl) {
        const int g = 5 * gl + j;
        f32x4 xv[FPW][4];
#pragma unroll
        for (int f = 0; f < FPW; ++f)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xv[f][kk] = *(const f32x4 *)(xbase + f * SX_LD + 16 * g + 4 * kk);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 wv = w[gl * 4 + c];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int f = 0; f < FPW; ++f) acc[f] = __builtin_fmaf(wv[kk], xv[f][kk][c], acc[f]);
        }
    }
}

#ifndef PS_WSPLIT
#define PS_WSPLIT 3
#endif
#define PS_WLOAD(p) (*PS_GC(f32x4, p))  // (nontemporal loads were measured 40 % slower: they lose the L2 residency)
// Always EXACTLY eight loads, whatever the stage and thread: s_waitcnt counts are static, so a path that issued
// fewer loads than another would force the compiler to wait for everything (vmcnt(0)) before the chain that
// consumes the PREVIOUS fetch -- i.e. to wait for the prefetch it has just issued.  Four-step stages and threads
// beyond the last chain re-read valid addresses instead.
__device__ __forceinline__ void load_chain_weights(const float *wv, int nchain, int nstep, int t, f32x4 (&w)[8])
{
    const float *base = wv + (size_t)min(t, nchain - 1) * 4;
    const size_t stride = (size_t)nchain * 4;
    const float *hi = nstep == 8 ? base + 4 * stride : base;
#pragma unroll
    for (int st = 0; st < 4; ++st) w[st] = PS_WLOAD(base + st * stride);
#pragma unroll
    for (int st = 0; st < 4; ++st) w[4 + st] = PS_WLOAD(hi + st * stride);
}
// the same fetch in two instalments, loads [LO, HI) of the eight (see chain_stage)
template <int LO, int HI>
__device__ __forceinline__ void load_chain_weights_part(const float *wv, int nchain, int nstep, int t, f32x4 (&w)[8])
{
    const float *base = wv + (size_t)min(t, nchain - 1) * 4;
    const size_t stride = (size_t)nchain * 4;
    const float *hi = nstep == 8 ? base + 4 * stride : base;
#pragma unroll
    for (int st = LO; st < HI; ++st) w[st] = PS_WLOAD((st < 4 ? base : hi) + (st & 3) * stride);
}

// Control record of one stage for the chain role, C1_CTL_DWORDS dwords in constant memory: record 0 describes the u0
// post op (norm_init), record 1 + s stage s and the post op that follows it, record NST the nin_out chains.
// Every role fetches its fields with scalar loads one stage ahead, so no wave ever waits on a descriptor.
constexpr int C1_CTL_DWORDS = 32;
enum { CTL_CO = 0, CTL_NCHAIN = 1, CTL_NG = 2, CTL_NSTEP = 3, CTL_WV = 4, CTL_BIAS = 6, CTL_KIND = 8, CTL_HAS_SKIP = 9,
       CTL_IN_FORM = 10, CTL_SAVE_SLOT = 11, CTL_SKIP_SLOT = 12, CTL_NBR_ITEMS = 13 /* of the stage, per tile */, CTL_BIAS2 = 14, CTL_R = 16, CTL_E = 18, CTL_X = 20,
       // throughput mode (k_column_tp): the centre tap / nin_skip in the MFMA layout [c/4][o][4], work items of the stage per tile
       CTL_WC = 22, CTL_WS = 24, CTL_TP_ITEMS = 26,
       CTL_TP_TYPE = 27, CTL_WTP = 28 /* the stage's weights in the chain role's own order [wave][unit][half][lane][4] */ };
typedef const __attribute__((address_space(4))) int *CtlInt;
typedef const __attribute__((address_space(4))) unsigned long long *CtlU64;
__device__ __forceinline__ int ctl_i(const int *ctl, int rec, int field) { return ((CtlInt)ctl)[rec * C1_CTL_DWORDS + field]; }
template <typename T>
__device__ __forceinline__ T *ctl_p(const int *ctl, int rec, int field)
{
    return (T *)((CtlU64)ctl)[(rec * C1_CTL_DWORDS + field) >> 1];
}
struct ChainCtl { int Co, nchain, NG, nstep; const float *wv; };
struct PostCtl { int Co, kind, has_skip, in_form, save_slot, nbr_items; const float *bias, *bias2; };
struct StoreCtl { int kind, skip_slot; float *R, *E, *X; };
__device__ __forceinline__ ChainCtl load_chain_ctl(const int *ctl, int rec)
{
    return ChainCtl{ctl_i(ctl, rec, CTL_CO), ctl_i(ctl, rec, CTL_NCHAIN), ctl_i(ctl, rec, CTL_NG), ctl_i(ctl, rec, CTL_NSTEP),
                    ctl_p<const float>(ctl, rec, CTL_WV)};
}
__device__ __forceinline__ PostCtl load_post_ctl(const int *ctl, int rec)
{
    return PostCtl{ctl_i(ctl, rec, CTL_CO), ctl_i(ctl, rec, CTL_KIND), ctl_i(ctl, rec, CTL_HAS_SKIP), ctl_i(ctl, rec, CTL_IN_FORM),
                   ctl_i(ctl, rec, CTL_SAVE_SLOT), ctl_i(ctl, rec, CTL_NBR_ITEMS), ctl_p<const float>(ctl, rec, CTL_BIAS),
                   ctl_p<const float>(ctl, rec, CTL_BIAS2)};
}
__device__ __forceinline__ StoreCtl load_store_ctl(const int *ctl, int rec)
{
    return StoreCtl{ctl_i(ctl, rec, CTL_KIND), ctl_i(ctl, rec, CTL_SKIP_SLOT), ctl_p<float>(ctl, rec, CTL_R),
                    ctl_p<float>(ctl, rec, CTL_E), ctl_p<float>(ctl, rec, CTL_X)};
}

__device__ __forceinline__ void store_through1(float *p, float v)
{
    asm volatile("global_store_dword %0, %1, off sc1" : : "v"(PS_G(float, p)), "v"(v) : "memory");
}
template <int FPW, bool LA = false>
__device__ __forceinline__ void chain_role(const ChainArgs &a, int wg)
{
    static_assert(FPW >= 1 && FPW <= 2, "waves 0..12 run the chains, wave 13 stores, the last FPW waves do the post ops");
    __shared__ __attribute__((aligned(16))) float sX[FPW][SX_LD];        // input of the centre taps
    __shared__ __attribute__((aligned(16))) float sSkip[FPW][SX_LD];     // concat_elu(u_k) feeding nin_skip
    __shared__ __attribute__((aligned(16))) float sP[FPW][C1_MAXCHAIN];  // chain values of the stage
    __shared__ __attribute__((aligned(16))) float sU[8][FPW][NF];        // u0..u7 of this location
    __shared__ __attribute__((aligned(16))) float sOut[FPW][3][NF];      // (u, elu(u), elu(-u)) on their way to the caches
    __shared__ __attribute__((aligned(16))) float sPL[FPW][5][NCLS];     // chain values of nin_out
    const int t = threadIdx.x, wave = uni(t >> 6), lane = t & 63;
    const int f0 = wg * FPW;  // first column of this workgroup (index into the launch's records)
    // Roles, each in its own wave-uniform branch (so their registers do not add up):
    //   waves 0..12         one chain per thread and stage
    //   wave 13             cache stores (finished values LDS -> R / E / X) and the nin_skip inputs
    //   wave 14             touches the control records ahead of everybody (scalar-cache prefetch)
    //   waves 16-FPW..15    post op of one frame each, two channels per lane (see pono_total)
    constexpr int NW = C1_THREADS / 64, STORE_WAVE = 13, CTL_WAVE = 14;
    const int pf = wave - (NW - FPW);  // frame slot of a post wave, negative otherwise
    const bool pwave = pf >= 0, swave = wave == STORE_WAVE;
    // store wave: one channel per lane, lane l also takes channel 64 + l for l < 16
    const bool hasB = lane < NF - 64;
    const int cA = lane, cB = 64 + (lane & (NF - 64 - 1));
    // chain role: 80-output stages hold chains t = j * 80 + o (then the nin_skip chains), 160-output ones j * 160 + o
    const int q80 = t / NF, j160 = t / (2 * NF);
    const int j80 = q80 >= 5 ? q80 - 5 : q80;
    if (t < FPW * SX_LD) { (&sX[0][0])[t] = 0.0f; (&sSkip[0][0])[t] = 0.0f; }  // absent frames feed zeros
    __syncthreads();

#ifdef PS_CHAIN_TRACE_BUILD  // tuning builds only: shader-clock stamps of workgroup 0, collected in LDS, dumped at the end
    __shared__ unsigned long long sTrace[NST][10];
#define PS_TRACE1(who, slot) do { if (who) sTrace[s][slot] = clock64(); } while (0)
#define PS_TRACE2(who, slot) do { if (who) sTrace[cur_stage][slot] = clock64(); } while (0)
#define PS_TRACE_MARK(who, k) do { if (who) sTrace[k][9] = clock64(); } while (0)  // [k][9]: 0 role start, 1 u0 done, 2 stages done, 3 draw done
#else
#define PS_TRACE1(who, slot) do { } while (0)
#define PS_TRACE2(who, slot) do { } while (0)
#define PS_TRACE_MARK(who, k) do { } while (0)
#endif
    f32x4 wo[C1_OUT_STEPS];
    const int opart = t >> 9;  // nin_out role: thread (o = t & 511, part): part 0 = chains 0..2, part 1 = chains 3..4
    auto load_out_weights = [&]() {
        const float *wo_base = ctl_p<const float>(a.ctl1, NST, CTL_WV) + (size_t)t * 4;
#pragma unroll
        for (int st = 0; st < 8; ++st) wo[st] = *PS_GC(f32x4, wo_base + (size_t)st * C1_THREADS * 4);
        if (opart == 0) {
#pragma unroll
            for (int st = 8; st < C1_OUT_STEPS; ++st) wo[st] = *PS_GC(f32x4, wo_base + (size_t)st * C1_THREADS * 4);
        }
    };
    // nin_out(elu(u)) (model.py:153); called at the end of every role's branch, so wo never crosses a join
    auto nin_out_chains = [&]() {
        const int o = t & (NCLS - 1);
#pragma unroll
        for (int cj = 0; cj < 3; ++cj) {
            if (opart == 1 && cj == 2) break;
            const int j = opart * 3 + cj;
            float acc[FPW];
            valu_chain<1, FPW>(&wo[cj * 4], &sX[0][0], j, acc);
#pragma unroll
            for (int f = 0; f < FPW; ++f) sPL[f][j][o] = acc[f];
        }
        lds_barrier();
    };

    if (pwave) {
        // ================= post waves: one frame each, two barriers per stage =================
        const int pfr = f0 + pf;   // column
        const bool pvalid = pfr < a.ncols;
        const bool own = lane < PONO_LANES;          // two channels per lane: 2 * lane, 2 * lane + 1 (see pono_total)
        const int c2 = own ? 2 * lane : 0;
        const float *nbr_f = a.nbr + (size_t)(pvalid ? pfr : 0) * NBR_LD + c2;
        const size_t nbr_half = (size_t)a.col_stride * NBR_LD, nbr_stage = 2 * nbr_half;
        const f32x2 zero2 = {0.0f, 0.0f};
        f32x2 ucur = zero2;
        [[maybe_unused]] int cur_stage = 0;  // (tuning builds: the stage the trace stamps of post_body belong to)
        // bias and neighbour-tap slots of a stage's post op: y = ((bias + NA) + centre) + NB; fetched one stage ahead.
        // Always exactly seven 8-byte loads from valid addresses, in every lane: static s_waitcnt counts (see
        // load_chain_weights); kinds without a gate half / skip re-read the main operands.
        struct Ops { f32x2 b, na, nb, bg, nag, nbg, b2; };
        // The neighbour slots are produced by other workgroups of this launch (nbr_role, other XCDs): they are read
        // with device-scope loads, and only once the stage's completion counter has reached this launch's target.
        auto fresh = [](const float *p) {
            const unsigned long long raw = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return f32x2{__uint_as_float((unsigned)raw), __uint_as_float((unsigned)(raw >> 32))};
        };
        auto plain = [](const float *p) { return *PS_GC(f32x2, p); };
        auto load_ops = [&](int s, const PostCtl &c, Ops &o) {
            const float *nb = nbr_f + (size_t)s * nbr_stage;
            const int gofs = c.kind == PRO_GATE ? NF : 0;
            const float *b2 = c.has_skip ? c.bias2 : c.bias;
            o.b = plain(c.bias + c2);
            o.na = fresh(nb);
            o.nb = fresh(nb + nbr_half);
            o.bg = plain(c.bias + gofs + c2);
            o.nag = fresh(nb + gofs);
            o.nbg = fresh(nb + nbr_half + gofs);
            o.b2 = plain(b2 + c2);
        };
        // completion counter of stage k; `have` is a value loaded earlier (normally already
        // past the target, so this costs nothing); bounded, so a lost neighbour workgroup cannot hang the GPU
        const int my_tile = (pvalid ? pfr : 0) >> 4;
        const unsigned my_uses = LA ? 0u : a.tile_uses[my_tile];
        const unsigned uses_lo = LA ? a.uses_lo[my_tile] : 0u, uses_hi = LA ? a.uses_hi[my_tile] : 0u;
        auto counter = [&](int k) { return __hip_atomic_load(a.cnt + cnt_index(k, my_tile), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        auto wait_counter = [&](unsigned have, int k, unsigned items_per_tile) {
            const unsigned need = (LA ? (k < a.la_split ? uses_lo : uses_hi) : my_uses) * items_per_tile;
            if (a.debug & 1) return;
            int spins = 0;
            while ((int)(have - need) < 0) {
                if (++spins > WAIT_SPINS) { if (lane == 0) *a.err = 1; break; }
                __builtin_amdgcn_s_sleep(2);
                have = counter(k);
            }
        };
        // PONO + finish + hand-off to the next stage.  Compiled once per (kind, skip, input form) combination that
        // occurs in the network, so the body is straight-line code; only save_slot stays a run-time value.
        auto post_and_emit = [&](const f32x2 &y, const f32x2 &g, const f32x2 &skip, auto KIND, auto SKIP, auto INFORM, int save_slot) {
            constexpr int kind = decltype(KIND)::value, in_form = decltype(INFORM)::value;
            constexpr bool has_skip = decltype(SKIP)::value;
            const float mean = pono_mean(pono_total(y, own));
            const f32x2 d = y - mean;
            const float inv = pono_inv(pono_total(d * d, own));
            if (!pvalid || !own) return;
            const f32x2 n = d * inv;
            f32x2 out;
            if (kind == PRO_CONVIN) out = post_finish<POST_CONVIN>(n, zero2, skip, has_skip, zero2);
            else if (kind == PRO_GATE) out = post_finish<POST_GATE>(n, g, zero2, false, ucur);
            else out = n;  // PRO_DIL, PRO_UINIT (norm_init)
            f32x2 ep, en;
            celu_pair2(out, ep, en);
            float *x = &sX[pf][c2];
            if (in_form == IN_CELU) { *(f32x2 *)x = ep; *(f32x2 *)(x + NF) = en; }
            else if (in_form == IN_RAW) *(f32x2 *)x = out;
            else *(f32x2 *)x = ep;
            *(f32x2 *)(&sOut[pf][1][c2]) = ep;
            *(f32x2 *)(&sOut[pf][2][c2]) = en;
            if (kind != PRO_CONVIN) {
                *(f32x2 *)(&sOut[pf][0][c2]) = out;
                ucur = out;
                if (save_slot >= 0) *(f32x2 *)(&sU[save_slot][pf][c2]) = out;
            }
        };
        // y (+ gate half, + nin_skip) of this stage from the chain values and the prefetched operands, then the post op
        auto post_body = [&](const Ops &o, auto KIND, auto SKIP, auto INFORM, int save_slot) {
            constexpr int kind = decltype(KIND)::value;
            constexpr bool has_skip = decltype(SKIP)::value;
            constexpr int Co = kind == PRO_GATE ? 2 * NF : NF;
            auto five = [](const float *p, int stride) {
                return chain_total(*(const f32x2 *)p, *(const f32x2 *)(p + stride), *(const f32x2 *)(p + 2 * stride),
                                   *(const f32x2 *)(p + 3 * stride), *(const f32x2 *)(p + 4 * stride));
            };
            const float *P = &sP[pf][c2];
            const f32x2 y = slot_sum2(o.b, o.na, five(P, Co), o.nb);
            f32x2 g = zero2, skip = zero2;
            if (kind == PRO_GATE) g = slot_sum2(o.bg, o.nag, five(P + NF, Co), o.nbg);
            if (has_skip) skip = five(P + 5 * Co, NF) + o.b2;
            PS_TRACE2(t == C1_THREADS - 64 && y.x != 12345.0f, 2);
            post_and_emit(y, g, skip, KIND, SKIP, INFORM, save_slot);
        };
        using std::integral_constant;
        unsigned cnt_nxt = 0;  // counter of the next stage, as loaded a stage earlier
        PostCtl cur = load_post_ctl(a.ctl1, 0), nxt = load_post_ctl(a.ctl1, 1), nn = load_post_ctl(a.ctl1, 2);
        auto post_stage = [&](int s, const Ops &ocur, Ops &onxt) {
            cur_stage = s;
            cur = nxt;                                              // record 1 + s
            nxt = nn;                                               // record 2 + s, requested a stage ago
            if (s + 2 < NST - 1) nn = load_post_ctl(a.ctl1, 3 + s);
            PS_TRACE1(t == C1_THREADS - 64, 0);
            // operands of the NEXT post op, issued while this wave waits for the chains: the vector-memory queue is
            // empty now, whereas after the barrier the chain waves fill it with the next stage's weights and any
            // load issued behind them would stall this wave (the critical path) for the whole burst.  The counter
            // of the stage after that is requested now and looked at a stage later.
            if (s + 1 < NST - 1) {
                wait_counter(cnt_nxt, s + 1, (unsigned)nxt.nbr_items);
                cnt_nxt = counter(min(s + 2, NST - 2));
                load_ops(s + 1, nxt, onxt);
            }
            lds_barrier();   // the chains of this stage are in sP
            PS_TRACE1(t == C1_THREADS - 64, 1);
            const integral_constant<bool, true> yes{};
            const integral_constant<bool, false> no{};
            const integral_constant<int, IN_CELU> celu{};
            if (cur.kind == PRO_CONVIN) {
                if (cur.has_skip) post_body(ocur, integral_constant<int, PRO_CONVIN>{}, yes, celu, -1);
                else post_body(ocur, integral_constant<int, PRO_CONVIN>{}, no, celu, -1);
            } else if (cur.kind == PRO_GATE) {
                if (cur.in_form == IN_CELU) post_body(ocur, integral_constant<int, PRO_GATE>{}, no, celu, cur.save_slot);
                else if (cur.in_form == IN_RAW) post_body(ocur, integral_constant<int, PRO_GATE>{}, no, integral_constant<int, IN_RAW>{}, cur.save_slot);
                else post_body(ocur, integral_constant<int, PRO_GATE>{}, no, integral_constant<int, IN_ELU>{}, cur.save_slot);
            } else {
                post_body(ocur, integral_constant<int, PRO_DIL>{}, no, celu, cur.save_slot);
            }
            PS_TRACE1(t == C1_THREADS - 64, 3);
            lds_barrier();
            PS_TRACE1(t == C1_THREADS - 64, 4);
        };
        PS_TRACE_MARK(lane == 0, 0);
        Ops oA, oB;
        const StepCtx *ctxp = a.ctx + (pvalid ? pfr : 0);
        const int q0 = ctxp->q, fr0 = ctxp->f;
        {   // u0 = norm_init(u_init): the gather over the (earlier) neighbours' codes; the record says where they are,
            // the codes themselves were written by earlier launches (sampled) or are the caller's (observed)
            float mA[9];
            int ncode[9], nl[9];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                mA[tp] = ctxp->m[0][tp];
                nl[tp] = ctxp->nloc[tp];
            }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) ncode[tp] = a.codes_in[(size_t)fr0 * a.L + max(nl[tp], 0)];  // all nine in flight
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) ncode[tp] = nl[tp] >= 0 ? ncode[tp] : UINIT_CLOSED;
            const f32x2 y = uinit_from_codes<f32x2>(ncode, mA, a.uinit_w, a.uinit_b, c2);
            post_and_emit(y, zero2, zero2, integral_constant<int, PRO_UINIT>{}, integral_constant<bool, false>{},
                          integral_constant<int, IN_CELU>{}, cur.save_slot);
            // the neighbour slots of stage 0 are first needed after the chains of stage 0
            wait_counter(counter(0), 0, (unsigned)nxt.nbr_items);
            cnt_nxt = counter(1);
            load_ops(0, nxt, oA);
            load_ops(0, nxt, oB);
            lds_barrier();
        }
        PS_TRACE_MARK(lane == 0, 1);
        for (int s = 0; s < NST - 3; s += 2) {
            post_stage(s, oA, oB);
            post_stage(s + 1, oB, oA);
        }
        post_stage(NST - 3, oA, oB);
        load_out_weights();  // (peeled: keeps these 48 registers out of the loop)
        post_stage(NST - 2, oB, oA);
        PS_TRACE_MARK(lane == 0, 2);
        nin_out_chains();

        // ---- end of the column: logits, categorical draw (sample.py:60-66)
        if (pvalid) {
            const int f = uni(fr0);
            const int fq = uni(q0);
            const size_t loc = (size_t)f * a.L + fq;
            float lg[8];
            {
                const float *Lp = &sPL[pf][0][lane * 8];
                const f32x4 lo = chain_total(*(const f32x4 *)Lp, *(const f32x4 *)(Lp + NCLS), *(const f32x4 *)(Lp + 2 * NCLS),
                                             *(const f32x4 *)(Lp + 3 * NCLS), *(const f32x4 *)(Lp + 4 * NCLS));
                const f32x4 hi = chain_total(*(const f32x4 *)(Lp + 4), *(const f32x4 *)(Lp + NCLS + 4), *(const f32x4 *)(Lp + 2 * NCLS + 4),
                                             *(const f32x4 *)(Lp + 3 * NCLS + 4), *(const f32x4 *)(Lp + 4 * NCLS + 4));
#pragma unroll
                for (int k = 0; k < 4; ++k) { lg[k] = lo[k] + a.out_b[lane * 8 + k]; lg[4 + k] = hi[k] + a.out_b[lane * 8 + 4 + k]; }
            }
            if (a.out_logits) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a.out_logits[loc * NCLS + lane * 8 + k] = lg[k];
            }
            if (a.step_logits) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a.step_logits[(size_t)f * NCLS + lane * 8 + k] = lg[k];
            }
            if (a.codes && a.region[loc]) {
                const int code = a.forced ? a.forced[loc] : draw_code(lg, a.temperature, a.uniforms[loc], lane);
                if (lane == 0) a.codes[loc] = code;
            }
        }
        PS_TRACE_MARK(lane == 0, 3);
    } else if (swave) {
        // ================= store wave: off everybody's critical path =================
        size_t off80[FPW], offR[FPW];
        bool fvalid[FPW];
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            fvalid[f] = f0 + f < a.ncols;
            const StepCtx *rec = a.ctx + (fvalid[f] ? f0 + f : 0);
            const size_t at = (size_t)rec->f * a.L + rec->q;
            off80[f] = at * NF;
            offR[f] = at * R_LD;
        }
        const int ch[2] = {cA, cB};
        auto store_outputs = [&](const StoreCtl &c) {  // what the post op of the record produced: LDS -> caches
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
                if (!fvalid[f]) continue;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (k == 1 && !hasB) break;
                    const float ep = sOut[f][1][ch[k]], en = sOut[f][2][ch[k]];
                    if (LA) {   // (write-through: the neighbour role of this very launch reads them for the next launch's columns)
                        if (c.kind == PRO_CONVIN) {
                            store_through1(c.X + 2 * off80[f] + ch[k], ep);
                            store_through1(c.X + 2 * off80[f] + NF + ch[k], en);
                        } else {
                            store_through1(c.R + offR[f] + ch[k], sOut[f][0][ch[k]]);
                            store_through1(c.E + 2 * off80[f] + ch[k], ep);
                            store_through1(c.E + 2 * off80[f] + NF + ch[k], en);
                        }
                    } else if (c.kind == PRO_CONVIN) {
                        *PS_G(float, c.X + 2 * off80[f] + ch[k]) = ep;
                        *PS_G(float, c.X + 2 * off80[f] + NF + ch[k]) = en;
                    } else {
                        *PS_G(float, c.R + offR[f] + ch[k]) = sOut[f][0][ch[k]];
                        *PS_G(float, c.E + 2 * off80[f] + ch[k]) = ep;
                        *PS_G(float, c.E + 2 * off80[f] + NF + ch[k]) = en;
                    }
                }
            }
        };
        auto stage_skip_input = [&](const StoreCtl &c) {  // concat_elu(u_k) for the nin_skip of the stage the record feeds
            if (c.skip_slot < 0) return;
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (k == 1 && !hasB) break;
                    float ep, en;
                    celu_pair(sU[c.skip_slot][f][ch[k]], ep, en);
                    sSkip[f][ch[k]] = ep;
                    sSkip[f][NF + ch[k]] = en;
                }
            }
        };
        StoreCtl sc = load_store_ctl(a.ctl1, 0);
        lds_barrier();
        store_outputs(sc);
        StoreCtl sn = load_store_ctl(a.ctl1, 1);
        for (int s = 0; s < NST - 2; ++s) {
            // (look-ahead form: everything but the stores of the LAST store_outputs -- the input of stage s: 2 or 3 stores per
            // channel pass, two passes per frame -- has been acknowledged once vmcnt is down to their number, so after the barrier
            // below the control wave may publish that the input of stage s - 1 is in memory)
            if (LA) {
                if (sc.kind == PRO_CONVIN) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * FPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(6 * FPW) : "memory");
            }
            sc = sn;                // record 1 + s
            lds_barrier();          // chains of stage s done: sSkip is free, the u_k were saved long ago
            stage_skip_input(sc);   // for stage s + 1, whose chains start after the next barrier
            sn = load_store_ctl(a.ctl1, 2 + s);  // waited for at the next barrier, under the post op
            lds_barrier();
            store_outputs(sc);
        }
        sc = sn;                    // record NST - 1
        load_out_weights();
        lds_barrier();
        lds_barrier();
        store_outputs(sc);
        nin_out_chains();
    } else if (wave == CTL_WAVE) {
        // ================= control-record prefetch: keeps the scalar cache ahead of every other wave =================
        // A record is first touched here, between the barriers of stage s (nobody waits for this wave then), three
        // stages before the chain waves and two before the post / store waves ask for it: their s_loads hit.
        int keep = 0;
        auto touch = [&](int rec) {
            rec = min(rec, NST);
            keep ^= ctl_i(a.ctl1, rec, 0) ^ ctl_i(a.ctl1, rec, 16);  // both 64-byte lines of the 128-byte record
        };
        for (int r = 0; r < 6; ++r) touch(r);
        lds_barrier();
        for (int s = 0; s < NST - 2; ++s) {
            lds_barrier();
            touch(6 + s);
            if (LA && s >= 1 && s - 1 < a.publish_upto && lane == 0)   // (see the store wave)
                __hip_atomic_fetch_add(a.done + (size_t)(s - 1) * CNT_PAD, (unsigned)min(FPW, a.ncols - f0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds_barrier();
        }
        load_out_weights();
        lds_barrier();
        lds_barrier();
        if (keep == 0x5eed1234) sP[0][0] = 0.0f;  // (keeps the loads alive)
        nin_out_chains();
    } else {
        // ================= chain waves: one chain per thread and stage =================
        // Three weight buffers: the fetch for stage s + 2 is issued between the barriers of stage s (under the post
        // op), so it has a whole stage to land and the chain of stage s + 1 never waits for memory.
        f32x4 wA[8], wB[8], wC[8];
        constexpr int WSPLIT = PS_WSPLIT;  // loads of a stage's eight that are issued ahead of the chains
        ChainCtl cc = load_chain_ctl(a.ctl1, 1), cn = load_chain_ctl(a.ctl1, 2), cnn = load_chain_ctl(a.ctl1, 3), c3 = cnn;
        auto chain_stage = [&](int s, const f32x4 (&wcur)[8], f32x4 (&wnn)[8], auto fetch, auto last) {
            PS_TRACE1(t == 0, 5);
            // first instalment of the fetch for stage s + 2: the queue is empty now (the second instalment of the
            // previous stage went out under its post op), so these issue while the chains below run
            if (fetch) load_chain_weights_part<0, WSPLIT>(cnn.wv, cnn.nchain, cnn.nstep, t, wnn);
            if (t < cc.nchain) {
                const bool main = cc.Co == 2 * NF || q80 < 5;
                const int j = cc.Co == 2 * NF ? j160 : j80;
                const float *xb = main ? &sX[0][0] : &sSkip[0][0];
                float acc[FPW];
                if (cc.NG == 10) valu_chain<2, FPW>(wcur, xb, j, acc);
                else valu_chain<1, FPW>(wcur, xb, j, acc);
#pragma unroll
                for (int f = 0; f < FPW; ++f) sP[f][t] = acc[f];
            }
            PS_TRACE1(t == 0 && sP[0][0] != 12345.0f, 6);
            lds_barrier();
            PS_TRACE1(t == 0, 7);
            // The vector-memory queue is shallow: issuing a stage's 13 x 8 KB takes the CU ~1700 cycles and blocks the
            // issuing wave, so it happens here, where this wave only waits for the post op.  Same for the scalar load
            // of the control record three stages ahead (it shares lgkmcnt with the LDS reads of the chain).
            if (fetch) {
                c3 = load_chain_ctl(a.ctl1, 4 + s);  // (records past NST - 1 are rotated in but never used as stages)
                load_chain_weights_part<WSPLIT, 8>(cnn.wv, cnn.nchain, cnn.nstep, t, wnn);
            }
            if (last) load_out_weights();
            lds_barrier();
            PS_TRACE1(t == 0, 8);
            cc = cn;
            cn = cnn;
            cnn = c3;
        };
        load_chain_weights(cc.wv, cc.nchain, cc.nstep, t, wA);
        load_chain_weights(cn.wv, cn.nchain, cn.nstep, t, wB);
        lds_barrier();
        const std::true_type yes{};
        const std::false_type no{};
        for (int s = 0; s < NST - 3; s += 3) {  // stages 0 .. 29
            chain_stage(s, wA, wC, yes, no);
            chain_stage(s + 1, wB, wA, yes, no);
            chain_stage(s + 2, wC, wB, yes, no);
        }
        chain_stage(NST - 3, wA, wC, no, no);   // stage 30 (its successor's weights were fetched during stage 29)
        chain_stage(NST - 2, wB, wA, no, yes);  // stage 31, then the nin_out weights
        nin_out_chains();
    }
#undef PS_TRACE1
#undef PS_TRACE_MARK
#undef PS_TRACE2

#ifdef PS_CHAIN_TRACE_BUILD
    __syncthreads();
    if (a.trace && wg == 0)
        for (int k = t; k < (NST - 1) * 10; k += C1_THREADS) a.trace[k] = (&sTrace[0][0])[k];
#endif
}

// ==========================================================================================
// k_column: ONE launch per wavefront of columns (or per order position: one column per frame).  Workgroup b runs on
// XCD b % 8 (observed; used for speed only):
//   chain role       workgroup = one column's 33-stage chain, on XCDs 0 .. chain_xcds-1 (32 CUs each, one 1024-thread
//                    workgroup per CU): their L2s keep the centre-tap weights from one launch to the next.
//   neighbour role   one workgroup per CU of the other XCDs (at most), walking the item list (nbr_role).
// Both start together: the chain only needs the neighbour slots of stage s when it reaches the post op of stage s,
// and by then the neighbour role is normally past that stage (its items are ordered by stage); completion counters
// per stage (device-scope atomics) and write-through stores carry the hand-off, every wait is bounded.
// The neighbour workgroups never share an XCD with the chain workgroups (the other blocks of the chain XCDs exit at once), a
// launch holds at most 32 chain workgroups per chain XCD, and all workgroups of a launch are resident together (one per CU at
// most).  On its own items the neighbour role waits for nothing, so waiting chains cannot keep it from finishing; its look-ahead
// items (k_column_la: the next launch's first stages) wait for the chains' `done` counters, which the chains publish before they
// can get to waiting for anything that comes after those items in a group's list -- a group publishes its previous item before it
// waits.  What the design does NOT cover is a second process running column launches on the same GPU (two launches can then hold
// each other's CUs until the bounded waits give up): one column-launching process per GPU (DESIGN, section 7).
// ==========================================================================================
__global__ __launch_bounds__(C1_THREADS) void k_column(NbrArgs na, ChainArgs ca)
{
    const int b = blockIdx.x, cx = na.chain_xcds, x = b & 7, row = b >> 3;
    const int chain_rows = (ca.ncols + cx - 1) / cx;  // rows of 8 blocks (one per XCD) that hold chain workgroups
    if (row < chain_rows) {
        if (x < cx) {
            const int col = row * cx + x;
            if (col < ca.ncols && (ca.debug & 3) != 2) chain_role<1>(ca, col);
        } else if ((ca.debug & 3) != 3) {
            nbr_role(na, row * (8 - cx) + (x - cx));
        }
    } else if ((ca.debug & 3) != 3) {
        nbr_role(na, chain_rows * (8 - cx) + (row - chain_rows) * 8 + x);
    }
}

// k_column_la: the same launch where the host knows what follows on the stream (a wavefront schedule): the neighbour role works
// a launch ahead for the first stages (nbr_role with w_from / w_upto set), the columns publish their stores (chain_role<1, true>).
__global__ __launch_bounds__(C1_THREADS) void k_column_la(NbrArgs na, ChainArgs ca)
{
    const int b = blockIdx.x, cx = na.chain_xcds, x = b & 7, row = b >> 3;
    const int chain_rows = (ca.ncols + cx - 1) / cx;  // rows of 8 blocks (one per XCD) that hold chain workgroups
    if (row < chain_rows) {
        if (x < cx) {
            const int col = row * cx + x;
            if (col < ca.ncols && (ca.debug & 3) != 2) chain_role<1, true>(ca, col);
        } else if ((ca.debug & 3) != 3) {
            nbr_role(na, row * (8 - cx) + (x - cx));
        }
    } else if ((ca.debug & 3) != 3) {
        nbr_role(na, chain_rows * (8 - cx) + (row - chain_rows) * 8 + x);
    }
}

// ==========================================================================================
// k_column_tp: the column launch in THROUGHPUT form, for wavefronts of more columns than k_column takes (views x samples
// in the hundreds).  Same arithmetic, same canonical accumulation order, bit-identical results; what changes is how the
// work is laid on the chip:
//   chain role      one 512-thread workgroup (one CU) per TILE OF 16 COLUMNS.  The centre taps are MFMA work now -- 16
//                   columns are the N of v_mfma_f32_16x16x4_f32 -- so a stage's 104 KB of centre-tap weights are fetched once
//                   per 16 columns instead of once per column, and a launch takes 64 tiles = 1024 columns.  A stage = MFMA
//                   phase (units of (16 output channels, accumulation chain j) = 8 dependent MFMAs, dealt round-robin to
//                   the 8 waves; operand B = the tile's input vectors in LDS, laid out [channel / 4][column][4] so that a
//                   wave reads a contiguous KB; operand A = weights from L2 into registers, fetched right after the previous
//                   MFMA phase) -> LDS barrier -> post phase (wave w does the post ops of columns w and w + 8, the very code
//                   of the latency form: PONO, gate / skip / residual, concat-ELU, cache stores) -> LDS barrier.
//   neighbour role  every other CU: one WAVE per work item (stage, slot NA|NB, 32 output channels) x 16-column tile, its
//                   four taps in sequence on two MFMA output tiles that share the gathered input rows (the registers for
//                   that are there at 8 waves per CU; the 1024-thread latency form has 128 per thread and splits the taps
//                   over four waves instead).  Items are walked stage-major, so that all CUs work on one stage's weights
//                   at a time; results leave write-through, a per-(stage, tile) counter publishes them.
// The hand-off (write-through stores -> device-scope counter -> device-scope loads, bounded waits) is the one of k_column.
// ==========================================================================================
// The chain role of k_column_tp: 0 = chain_role_tp (round 2: eight waves, (output tile, chain) units dealt round-robin, post op of
// two columns per wave after a trip of the chain values through LDS) -- the default; 1 = chain_role_tp2 (round 3: ten waves, one
// output tile with all its chains per wave, post op in the MFMA layout).  Both are bit-identical to the walk (the tests pass with
// either).  Measured at C5's 128 views, per launch, chain role alone / whole launch: 0: 125 / 148 us; 1: 126 / 156 us as first
// written (a third of the vector instructions, but ten waves leave the neighbour role 168 registers: 92 -> 98 us alone), and no
// arrangement of its memory requests got below that: next stage's operands and control values fetched under the post op 148
// alone, weights refilled in place under the MFMAs in one burst 158, one request per five MFMAs 131.  What a stage costs either
// way (~8 k cycles) is not instruction count: ~3.2-3.8 k cycles of MFMA on one CU's four pipes, ~2.5 k cycles in which the CU's
// vector-memory path (64 bytes a clock) carries the stage's 100 KB of centre-tap weights, ~2.5 k of post op and barriers, and
// because every wave of the workgroup is in the same phase at the same time the three do not overlap.
#ifndef PS_TP_CHAIN2
#define PS_TP_CHAIN2 0
#endif
#ifndef PS_TP_WAVES
#if PS_TP_CHAIN2
#define PS_TP_WAVES 10
#else
#define PS_TP_WAVES 8
#endif
#endif
constexpr int TP_WAVES = PS_TP_WAVES, TP_THREADS = 64 * TP_WAVES, TP_COLS = 16;   // 8 waves (256 registers per thread) or 16 (128)
constexpr int TP_NPC = TP_COLS / TP_WAVES;   // columns a wave does the post op of: 2 or 1
constexpr int TP_MAX_TILES = 64, TP_COL_CAP = TP_MAX_TILES * TP_COLS;   // 1024 columns per launch
constexpr int TP_MAXU = (50 + TP_WAVES - 1) / TP_WAVES;   // units per wave and stage at most: 7 or 4
constexpr int TP_MINU = (25 + TP_WAVES - 1) / TP_WAVES;   // ... of a 25-unit stage: 4 or 2
constexpr int XB_LD = 68;             // B-operand layout: dwords per 4-channel group (16 columns x 4 + 4 pad: conflict-free
constexpr int XB_SIZE = 40 * XB_LD;   //   for the post op's 8-byte writes and for the waves' 16-byte reads)
constexpr int SP_LD = 5 * 2 * NF + 4; // chain values of one column [j][o] (+ [j][80] of nin_skip): 800 + 4 pad
constexpr int SLOG_LD = NCLS + 4;     // logits of one column (aliases the chain values)
static_assert(TP_COLS * SLOG_LD <= TP_COLS * SP_LD, "logits alias the chain-value buffer");
__device__ __host__ __forceinline__ size_t tp_cnt_index(int stage, int tile) { return ((size_t)stage * TP_MAX_TILES + tile) * CNT_PAD; }
__device__ __forceinline__ int xb_index(int ch, int col) { return (ch >> 2) * XB_LD + col * 4 + (ch & 3); }

// a work item of the throughput neighbour role = (stage, slot NA|NB, T x 16 output channels from o0)
struct __attribute__((aligned(16))) NbrWorkTp {
    const float *w;   // packed weights of the conv [taps][NG*4][Co_pad][4]
    const float *in;  // cache the taps gather from
    int stage, half, o0, T;
    int NG, Co_pad, in_ld, kind /* 0 = und, 1 = dil */;
};
static_assert(sizeof(NbrWorkTp) == 48, "three 16-byte loads");

// Stage types of the chain role (what fixes a stage's unit list): conv_input, conv_input + nin_skip, conv_out, dilated conv
enum { TPT_CONVIN = 0, TPT_CONVIN_SKIP = 1, TPT_CONVOUT = 2, TPT_DIL = 3 };
__device__ __host__ constexpr int tpt_units(int type) { return type == TPT_CONVIN || type == TPT_DIL ? 25 : 50; }   // (tile, chain) units
__device__ __host__ constexpr int tpt_nu(int type) { return type == TPT_CONVIN || type == TPT_DIL ? TP_MINU : TP_MAXU; } // per wave, at most
__device__ __host__ constexpr int tpt_nh(int type) { return type == TPT_DIL ? 1 : 2; }                              // 16-byte weight loads per unit
// Unit u of wave w in a stage of a given type, everything that does not depend on the lane: where its B operands sit in the
// B-operand buffers (floats from sXb; nin_skip's units read sSb = sXb + XB_SIZE), where its chain values go in a column's
// row of sP.  Built on the host (build_stage_table), staged in LDS.
struct __attribute__((aligned(16))) TpUnit { int b0, b1, dst, pad; };

struct TpArgs {
    const TpUnit *units;      // [4 types][TP_WAVES][TP_MAXU]
    // neighbour role
    const NbrWorkTp *work;
    const ColTaps *taps;      // records of this launch's columns
    float *nbr;               // [NST][2][TP_COL_CAP][NBR_LD]
    unsigned *cnt;            // [NST][TP_MAX_TILES] padded completion counters (tp_cnt_index), never reset
    int nwork, tiles, nbr_wgs;
    int chain_xcds, fill_nbr;   // placement (k_column_tp): XCDs that hold the chain tiles; first neighbour index of their spare CUs or -1
    // chain role (fields as in ChainArgs)
    const int *ctl1;
    const float *uinit_w, *uinit_b;
    const int32_t *codes_in;
    const StepCtx *ctx;
    const float *out_w, *out_b;
    int L, ncols;
    int32_t *codes;
    const uint8_t *region;
    const int32_t *forced;
    const float *uniforms;
    float *out_logits, *step_logits;
    float temperature;
    // counter (k, t) stands at uses x (items of stage k per tile) when tile t's slots of stage k are there: uses_lo for the stages
    // below `split` (computed a launch AHEAD, see nbr_role_tp), uses_hi for the others (computed by this launch)
    unsigned tile_uses_lo[TP_MAX_TILES], tile_uses_hi[TP_MAX_TILES];
    int split;                // stages [0, split) of a launch are the previous launch's business when it could see this one coming
    // the neighbour role's share: work entries [w_from, nwork) for this launch's columns, then [0, w_upto) for the NEXT launch's
    int w_from, w_upto;
    const ColTaps *taps_next;
    int ncols_next, tiles_next;
    float *nbr_next;          // the other half of the double-buffered slots
    unsigned *cnt_next;       // and of the counters
    unsigned *done;           // [NST] padded: chain tiles that have published the input of stage k (all launches so far)
    unsigned done_target;     // what done[k] reads when every tile of THIS launch has
    int publish_upto;         // stages whose input the chain tiles publish (0: nobody looks)
    int nbr_map;              // item -> wave mapping of the neighbour role: 0 = the waves of a workgroup take consecutive items, 1 = consecutive items go to different workgroups
    int *err;
    int debug;
    unsigned long long *trace;   // tuning builds (-DPS_TP_TRACE_BUILD): [NST][8] shader-clock stamps of tile 0, wave 0
};

// AHEAD: the item belongs to the NEXT launch's columns.  Some of the rows it gathers were written by chain tiles of THIS launch,
// on other XCDs: they were stored write-through and their stage's `done` counter has been seen.  What remains is a stale copy
// in this CU's L1 or this XCD's L2, which can only be there if the line was read earlier in this launch (both are invalidated
// when a kernel starts): (1) rows of 80 floats (the dilated convs' input) share 128-byte lines with their neighbours' -- those
// items gather with device-scope loads (sc1), which go past both caches, at the price of no reuse between the ~10 items that
// read a row (all rows that way: the neighbour role took 163 instead of 95 us); (2) rows of 160 floats are whole lines, and the
// only reads of a row that is not finished were the dummy reads of closed taps (row 0) -- a closed lane now reads a row some
// other lane of the wave gathers anyway.
#ifndef PS_TP_POLL_SLEEP
#define PS_TP_POLL_SLEEP 100
#endif
template <int T, int NG, bool AHEAD>
__device__ __forceinline__ void nbr_item_tp(const NbrWorkTp &wk, const TpArgs &a, int ctile, int lane)
{
    const int i = lane & 15, kk = lane >> 4;
    const int col = ctile * TP_COLS + i;
    const bool valid = col < (AHEAD ? a.ncols_next : a.ncols);
    const ColTaps *const taps = AHEAD ? a.taps_next : a.taps;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 rows = {-1, -1, -1, -1};
    if (valid) rows = *PS_GC(i32x4, &taps[col].row[wk.kind][wk.half * 4]);
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void *)wk.in, 0, 0x7fffffff, 0x00020000);
    f32x4 tot[T];
#pragma unroll
    for (int u = 0; u < T; ++u) tot[u] = zero;
    const size_t gstride = (size_t)16 * wk.Co_pad;   // floats between channel groups of the packed weights
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
        const int row = rows[tq];
        const bool live = row >= 0;
        const unsigned long long open = __builtin_amdgcn_ballot_w64(live);
        if (open == 0ull) continue;   // (a closed tap is an exact zero)
        const int safe = __shfl(row, __builtin_ctzll(open), 64);   // closed lanes read (and drop) a row that is being read anyway
        const int t = wk.half * 5 + tq;
        const float *src = wk.in + (size_t)(live ? row : safe) * wk.in_ld + 4 * kk;
        const float *wbase = wk.w + (size_t)t * NG * gstride + ((size_t)kk * wk.Co_pad + wk.o0 + i) * 4;
        Acc5 acc[T];
#pragma unroll
        for (int u = 0; u < T; ++u) acc[u] = acc5_zero();
#pragma unroll
        for (int g0 = 0; g0 < NG; g0 += 5) {
            f32x4 bv[5];
            if (AHEAD && NG == 5) {
                const int voff = ((live ? row : safe) * wk.in_ld + 4 * kk) * 4;
#pragma unroll
                for (int g = 0; g < 5; ++g)
                    bv[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, voff + 64 * (g0 + g), 0, 16 /* sc1 */));
            } else {
#pragma unroll
                for (int g = 0; g < 5; ++g) bv[g] = *PS_GC(f32x4, src + 16 * (g0 + g));
            }
#pragma unroll
            for (int g = 0; g < 5; ++g) bv[g] = live ? bv[g] : zero;   // (mask values are 0 / 1: no multiply needed)
#pragma unroll
            for (int u = 0; u < T; ++u) {
                f32x4 av[5];
#pragma unroll
                for (int g = 0; g < 5; ++g) av[g] = *PS_GC(f32x4, wbase + (size_t)(g0 + g) * gstride + 64 * u);
                mfma_chunk5(av, bv, acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < T; ++u) tot[u] = tot[u] + chunk_total(acc[u]);
    }
    if (valid) {
        float *dst = (AHEAD ? a.nbr_next : a.nbr) + (((size_t)wk.stage * 2 + wk.half) * TP_COL_CAP + col) * NBR_LD + wk.o0 + kk * 4;
#pragma unroll
        for (int u = 0; u < T; ++u) store_through(dst + 16 * u, tot[u]);
    }
}

// The neighbour role of a launch, one launch ahead where it can be.  The NA / NB slots of a column only read finished columns of
// EARLIER launches and, of the launch in front of its own, what that launch's chain tiles have already stored -- never its own
// launch's results.  So the slots of the stages [0, split) of launch i + 1 are computed by the neighbour role of launch i, behind
// launch i's chain tiles (which publish, stage by stage, that the input of stage k is in memory: `done`), and launch i + 1 finds
// them ready: its chain tiles start without waiting for a cold neighbour role (work records, first weights into the XCDs' L2s,
// ~13-15 us at the head of every launch before).  The stages [split, NST) stay with the launch itself -- they are not needed
// before its chain has walked `split` stages, and the tail of the launch in front would otherwise hang on its last `done`s.
// A wave's items: this launch's own entries [w_from, nwork) x tiles first (they wait for nothing), then the next launch's
// [0, w_upto) x tiles_next, stage-major.  Slots and completion counters are double-buffered by launch parity.
__device__ __forceinline__ void nbr_role_tp(const TpArgs &a, int nb)
{
    const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int gw = a.nbr_map ? wave * a.nbr_wgs + nb : nb * TP_WAVES + wave, nw = a.nbr_wgs * TP_WAVES;
    const int n_own = (a.nwork - a.w_from) * a.tiles;
    const int nitems = n_own + a.w_upto * a.tiles_next;
    __shared__ unsigned sReadyTp;   // look-ahead stages some wave of this workgroup has seen published, + 1
    if (threadIdx.x == 0) sReadyTp = 0;
    __syncthreads();
    int ready_upto = -1;
    for (int item = gw; item < nitems; item += nw) {
        const bool ahead = item >= n_own;
        int witem, ctile;
        if (!ahead) {
            const int q = item / a.tiles;
            witem = a.w_from + q; ctile = item - q * a.tiles;
        } else {
            const int j = item - n_own;
            witem = j / a.tiles_next; ctile = j - witem * a.tiles_next;
        }
        NbrWorkTp wk;
        {   // wave-uniform record: scalar loads
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(4))) u32x4 *CU4;
            const CU4 p = (CU4)(a.work + witem);
            u32x4 r[3];
            r[0] = p[0]; r[1] = p[1]; r[2] = p[2];
            __builtin_memcpy(&wk, r, sizeof(wk));
        }
        if (ahead) {
            // The chain tiles of this launch must have stored the input of the item's stage (`done`; bounded wait, normally long
            // past).  What a wave learns it leaves in LDS for the others of its workgroup, it looks at the LAST look-ahead stage's
            // counter first (that settles the rest of the launch), and it polls slowly: every wave polling every 0.2 us hammers the
            // counters' lines, which the chain tiles' publishing atomics and device-scope loads then queue behind (k_column_la).
            if (wk.stage > ready_upto) {
                unsigned val = __hip_atomic_load(&sReadyTp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (val < (unsigned)wk.stage + 1u) {
                    const unsigned *dl = a.done + (size_t)(a.split - 1) * CNT_PAD;
                    if ((int)(__hip_atomic_load(dl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.done_target) >= 0) {
                        val = (unsigned)a.split;
                    } else {
                        const unsigned *dp = a.done + (size_t)wk.stage * CNT_PAD;
                        int spins = 0;
                        while ((int)(__hip_atomic_load(dp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.done_target) < 0 &&
                               __hip_atomic_load(&sReadyTp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)wk.stage + 1u) {
                            if (++spins > (WAIT_SPINS >> 4)) { if (lane == 0) *a.err = 1; break; }
                            __builtin_amdgcn_s_sleep(PS_TP_POLL_SLEEP);
                        }
                        val = (unsigned)wk.stage + 1u;
                    }
                    if (lane == 0) __hip_atomic_fetch_max(&sReadyTp, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                ready_upto = (int)val - 1;
            }
            asm volatile("" ::: "memory");
            if (wk.T == 2) {
                if (wk.NG == 10) nbr_item_tp<2, 10, true>(wk, a, ctile, lane); else nbr_item_tp<2, 5, true>(wk, a, ctile, lane);
            } else {
                if (wk.NG == 10) nbr_item_tp<1, 10, true>(wk, a, ctile, lane); else nbr_item_tp<1, 5, true>(wk, a, ctile, lane);
            }
            signal_done(a.cnt_next + tp_cnt_index(wk.stage, ctile), lane);
        } else {
            if (wk.T == 2) {
                if (wk.NG == 10) nbr_item_tp<2, 10, false>(wk, a, ctile, lane); else nbr_item_tp<2, 5, false>(wk, a, ctile, lane);
            } else {
                if (wk.NG == 10) nbr_item_tp<1, 10, false>(wk, a, ctile, lane); else nbr_item_tp<1, 5, false>(wk, a, ctile, lane);
            }
            signal_done(a.cnt + tp_cnt_index(wk.stage, ctile), lane);
        }
    }
}

#if !PS_TP_CHAIN2
__device__ __forceinline__ void chain_role_tp(const TpArgs &a, int tile)
{
    // (Both 16-byte-accessed buffers are DECLARED as 16-byte elements: behind a float array and a run-time index hipcc cannot
    // prove the alignment and splits every ds_read_b128 / ds_write_b128 into two ds_read2_b32 -- which, at a lane stride of
    // four dwords, is an 8-way bank conflict on every operand read: the MFMA phase took 2-3x its MFMA time.)
    __shared__ f32x4 sXS4[2 * XB_SIZE / 4];                                // B-operand layout: input of the centre taps, and behind
    float *const sXS = (float *)sXS4;                                      //   it concat_elu(u_k) feeding nin_skip
    float *const sXb = sXS, *const sSb = sXS + XB_SIZE;
    __shared__ f32x4 sP4[TP_COLS * SP_LD / 4];                             // chain values of the stage [col][j][o]; logits at the end
    float *const sP = (float *)sP4;
    __shared__ __attribute__((aligned(16))) float sU[8][TP_COLS][NF];      // u0..u7 of the tile's columns
    __shared__ __attribute__((aligned(16))) StepCtx sC[TP_COLS];
    __shared__ __attribute__((aligned(16))) int sCtl[(NST + 1) * C1_CTL_DWORDS];   // the control records (a scalar load from memory at
                                                                                  // every stage start cost ~1000 cycles of its ~8000)
    const int t = threadIdx.x, wave = uni(t >> 6), lane = t & 63, i = lane & 15, kk = lane >> 4;
    const int col0 = tile * TP_COLS;
    const int ncl = min(TP_COLS, a.ncols - col0);   // columns of this tile (>= 1)
    {
        const int nq = (int)(sizeof(StepCtx) / 16);
        for (int k = t; k < TP_COLS * nq; k += TP_THREADS) {
            const int c = min(k / nq, ncl - 1);     // absent columns repeat the last one (their results are dropped)
            ((uint4 *)sC)[k] = ((const uint4 *)(a.ctx + col0 + c))[k % nq];
        }
        for (int k = t; k < XB_SIZE; k += TP_THREADS) { sXb[k] = 0.0f; sSb[k] = 0.0f; }
        for (int k = t; k < (NST + 1) * C1_CTL_DWORDS / 4; k += TP_THREADS) ((uint4 *)sCtl)[k] = ((const uint4 *)a.ctl1)[k];
    }
    __syncthreads();
    auto li = [&](int rec, int field) { return uni(sCtl[rec * C1_CTL_DWORDS + field]); };
    auto lpf = [&](int rec, int field) {
        const unsigned lo = (unsigned)li(rec, field), hi = (unsigned)li(rec, field + 1);
        return (float *)(((unsigned long long)hi << 32) | lo);
    };
    auto post_ctl = [&](int rec) {
        return PostCtl{li(rec, CTL_CO), li(rec, CTL_KIND), li(rec, CTL_HAS_SKIP), li(rec, CTL_IN_FORM), li(rec, CTL_SAVE_SLOT),
                       li(rec, CTL_NBR_ITEMS), lpf(rec, CTL_BIAS), lpf(rec, CTL_BIAS2)};
    };
    auto store_ctl = [&](int rec) { return StoreCtl{li(rec, CTL_KIND), li(rec, CTL_SKIP_SLOT), lpf(rec, CTL_R), lpf(rec, CTL_E), lpf(rec, CTL_X)}; };
    // ---- post-op side: wave w owns columns w and w + 8, two channels per lane (see pono_total)
    const bool own = lane < PONO_LANES;
    const int c2 = own ? 2 * lane : 0;
    const f32x2 zero2 = {0.0f, 0.0f};
    bool pvalid[TP_NPC];
    int pcol[TP_NPC], pfr[TP_NPC];
    size_t ploc[TP_NPC];
    f32x2 ucur[TP_NPC];
#pragma unroll
    for (int k = 0; k < TP_NPC; ++k) ucur[k] = zero2;
#pragma unroll
    for (int k = 0; k < TP_NPC; ++k) {
        pcol[k] = wave + TP_WAVES * k;
        pvalid[k] = pcol[k] < ncl;
        pfr[k] = uni(sC[pcol[k]].f);
        ploc[k] = (size_t)pfr[k] * a.L + uni(sC[pcol[k]].q);
    }
    const size_t nbr_half = (size_t)TP_COL_CAP * NBR_LD, nbr_stage = 2 * nbr_half;
    const unsigned uses_lo = a.tile_uses_lo[tile], uses_hi = a.tile_uses_hi[tile];
    auto counter = [&](int k) { return __hip_atomic_load(a.cnt + tp_cnt_index(k, tile), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // `have`: the counter as requested a stage earlier (normally past the target already); bounded
    auto wait_counter = [&](unsigned have, int k, unsigned items_per_tile) {
        if (a.debug & 1) return;
        const unsigned need = (k < a.split ? uses_lo : uses_hi) * items_per_tile;
        int spins = 0;
        while ((int)(have - need) < 0) {
            if (++spins > WAIT_SPINS) { if (lane == 0) *a.err = 1; break; }
            __builtin_amdgcn_s_sleep(2);
            have = counter(k);
        }
        asm volatile("" ::: "memory");
    };
    auto fresh = [](const float *p) {
        const unsigned long long raw = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return f32x2{__uint_as_float((unsigned)raw), __uint_as_float((unsigned)(raw >> 32))};
    };
    auto plain = [](const float *p) { return *PS_GC(f32x2, p); };
    // PONO + finish of BOTH columns of this wave (independent instruction streams, interleaved by the compiler) and the
    // hand-off: next stage's input into the B-operand layout, values to the caches.  KIND / HAS_SKIP are compile-time.
    auto emit2 = [&](const f32x2 (&y)[TP_NPC], const f32x2 (&g)[TP_NPC], const f32x2 (&skip)[TP_NPC], auto KINDc, auto SKIPc, int in_form, int save_slot,
                     const StoreCtl &sc) {
        constexpr int kind = decltype(KINDc)::value;
        constexpr bool has_skip = decltype(SKIPc)::value;
        float mean[TP_NPC], inv[TP_NPC];
        f32x2 d[TP_NPC];
#pragma unroll
        for (int k = 0; k < TP_NPC; ++k) mean[k] = pono_mean(pono_total(y[k], own));
#pragma unroll
        for (int k = 0; k < TP_NPC; ++k) d[k] = y[k] - mean[k];
#pragma unroll
        for (int k = 0; k < TP_NPC; ++k) inv[k] = pono_inv(pono_total(d[k] * d[k], own));
        if (!own) return;
#pragma unroll
        for (int k = 0; k < TP_NPC; ++k) {
            const f32x2 n = d[k] * inv[k];
            f32x2 out;
            if (kind == PRO_CONVIN) out = post_finish<POST_CONVIN>(n, zero2, skip[k], has_skip, zero2);
            else if (kind == PRO_GATE) out = post_finish<POST_GATE>(n, g[k], zero2, false, ucur[k]);
            else out = n;  // PRO_DIL, PRO_UINIT (norm_init)
            f32x2 ep, en;
            celu_pair2(out, ep, en);
            const int col = pcol[k];
            if (!pvalid[k]) continue;
            if (in_form == IN_CELU) { *(f32x2 *)&sXb[xb_index(c2, col)] = ep; *(f32x2 *)&sXb[xb_index(NF + c2, col)] = en; }
            else if (in_form == IN_RAW) *(f32x2 *)&sXb[xb_index(c2, col)] = out;
            else *(f32x2 *)&sXb[xb_index(c2, col)] = ep;
            // (write-through: the neighbour role of this very launch reads them, on other XCDs, for the next launch's columns)
            if (kind == PRO_CONVIN) {
                store_through2(sc.X + ploc[k] * (2 * NF) + c2, ep);
                store_through2(sc.X + ploc[k] * (2 * NF) + NF + c2, en);
            } else {
                store_through2(sc.R + ploc[k] * R_LD + c2, out);
                store_through2(sc.E + ploc[k] * (2 * NF) + c2, ep);
                store_through2(sc.E + ploc[k] * (2 * NF) + NF + c2, en);
                ucur[k] = out;
                if (save_slot >= 0) *(f32x2 *)(&sU[save_slot][col][c2]) = out;
            }
        }
    };
    // concat_elu(u_k) of the saved u the NEXT stage's nin_skip reads
    auto stage_skip_input = [&](int skip_slot) {
        if (skip_slot < 0 || !own) return;
#pragma unroll
        for (int k = 0; k < TP_NPC; ++k) {
            const int col = pcol[k];
            f32x2 ep, en;
            celu_pair2(*(const f32x2 *)(&sU[skip_slot][col][c2]), ep, en);
            *(f32x2 *)&sSb[xb_index(c2, col)] = ep;
            *(f32x2 *)&sSb[xb_index(NF + c2, col)] = en;
        }
    };

    // ---- MFMA side.  Unit n of a stage = (16 output channels ot, accumulation chain j), n = w, w + 8, ... for wave w; main units
    // first, then nin_skip's.  Nothing about a unit is computed here: its B-operand and chain-value offsets come from the unit
    // table of the stage's type, its weights from the stage's copy in this role's own order [wave][unit][half][lane][4] -- one
    // base register, the rest immediates.
#ifndef PS_TP_EXP
#define PS_TP_EXP 0
#endif
#ifdef PS_TP_TRACE_BUILD
    int trace_s = 0;
    const int trace_wave = a.debug >> 8;   // PS_COLUMN_DEBUG = 256 * wave (+ mode): the wave whose stamps are kept
#define TP_STAMP(slot) do { if (a.trace && tile == 0 && t == 64 * trace_wave) a.trace[s * 8 + (slot)] = clock64(); } while (0)
#define TP_STAMP2(slot, dep) do { if (a.trace && tile == 0 && t == 64 * trace_wave && (dep)) a.trace[trace_s * 8 + (slot)] = clock64(); } while (0)
#else
#define TP_STAMP(slot) do { } while (0)
#define TP_STAMP2(slot, dep) do { } while (0)
#endif
    struct UnitW { f32x4 a0, a1; };
    const int lane_b = kk * XB_LD + i * 4, lane_d = i * SP_LD + kk * 4;
    // weights of this wave's units of the stage of record `rec` (type ty): NU x 2 16-byte loads from consecutive KBs
    auto weights_base = [&](int rec, int ty) {
        return lpf(rec, CTL_WTP) + ((size_t)wave * tpt_nu(ty) * 2 * 64 + lane) * 4;
    };
    // MFMA phase of a stage of type TY.  One weight buffer, refilled in place: a unit's first-half weights (a0: channel group j)
    // are dead once the first half has been issued, so the NEXT stage's a0 (type nty, base nbase) are requested into the same
    // registers between the MFMAs of the second half, and its a1 right after the second half -- each has more than half a stage
    // to arrive, and the requests go out while the matrix pipe works through MFMAs already issued.  The B operands of the
    // second half (group j + 5) take the registers of the first half's.  Units 0 .. NU-2 exist for every wave, unit NU-1 for
    // the first waves only (`last`).
    auto mfma_units = [&](auto TYc, UnitW (&W)[TP_MAXU], int nty, const float *nbase, auto &&after_first) {
        constexpr int TY = decltype(TYc)::value, NU = tpt_nu(TY), NH = tpt_nh(TY);
        const bool last = wave + TP_WAVES * (NU - 1) < tpt_units(TY);
        f32x4 b[NU], acc[NU];
        int b1i[NU], dst[NU];
        // where unit n = wave + TP_WAVES u finds its B operand and parks its chain value: scalar arithmetic on the (uniform) wave
        // index -- the table in LDS that used to hold these cost a round trip in front of the first B read of every stage
        constexpr int CoT = TY == TPT_CONVOUT ? 2 * NF : NF, UM = 5 * (CoT >> 4);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int n = min(wave + TP_WAVES * u, tpt_units(TY) - 1);   // absent units name a valid one (never stored)
            const bool skp = n >= UM;
            const int m = skp ? n - UM : n, ot = m / 5, j = m - 5 * ot;
            const int b0 = (skp ? XB_SIZE : 0) + 4 * j * XB_LD;
            const int b1 = b0 + (TY == TPT_DIL ? 0 : 20 * XB_LD);
            const int d0 = (skp ? 5 * CoT + j * NF : j * CoT) + ot * 16;
            b[u] = sXS4[(b0 + lane_b) >> 2];
            b1i[u] = (b1 + lane_b) >> 2;
            dst[u] = (d0 + lane_d) >> 2;
            acc[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        const int nnu = tpt_nu(nty);   // units per wave of the next stage: 4 or 7 (every unit has two KBs in the stage's copy)
        TP_STAMP2(3, b[0][0] != 12345.0f);
        // canonical order of a chain: group j (c = 0..3), then group j + 5; the units are independent accumulators
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int u = 0; u < NU - 1; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[u].a0[c], b[u][c], acc[u], 0, 0, 0);
            if (last) acc[NU - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[NU - 1].a0[c], b[NU - 1][c], acc[NU - 1], 0, 0, 0);
            if (c == 0) after_first();   // (the post op's operand requests go out while the matrix pipe has work queued)
        }
        TP_STAMP2(4, true);
        if (NH == 2) {
#pragma unroll
            for (int u = 0; u < NU; ++u) b[u] = sXS4[b1i[u]];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int u = 0; u < NU - 1; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[u].a1[c], b[u][c], acc[u], 0, 0, 0);
                if (last) acc[NU - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[NU - 1].a1[c], b[NU - 1][c], acc[NU - 1], 0, 0, 0);
#pragma unroll
                for (int u = 2 * c; u < 2 * c + 2; ++u)
                    if (u < TP_MAXU && u < nnu) W[u].a0 = *PS_GC(f32x4, nbase + (size_t)(2 * u) * 256);
            }
        } else {
#pragma unroll
            for (int u = 0; u < TP_MAXU; ++u)
                if (u < nnu) W[u].a0 = *PS_GC(f32x4, nbase + (size_t)(2 * u) * 256);
        }
        TP_STAMP2(5, true);
#pragma unroll
        for (int u = 0; u < TP_MAXU; ++u)
            if (u < nnu) W[u].a1 = *PS_GC(f32x4, nbase + (size_t)(2 * u + 1) * 256);
#pragma unroll
        for (int u = 0; u < NU - 1; ++u) sP4[dst[u]] = acc[u];
        if (last) sP4[dst[NU - 1]] = acc[NU - 1];
        TP_STAMP2(6, acc[0][0] != 12345.0f);
    };

    // One stage: operands of its post op requested, MFMA phase (next stage's weights requested underneath), barrier, post op of
    // this wave's two columns, barrier.  The stage's type fixes Co, NG, the unit list and the post op that follows it
    // (conv_input -> CONVIN with or without nin_skip, conv_out -> GATE, dilated conv -> DIL).
    unsigned cnt_have = 0;
    // Control records: a record is 32 dwords, ONE LDS read (a dword per lane) puts it in a register and v_readlane hands out its
    // fields -- a ds_read + v_readfirstlane per field were two dozen round trips at the head of every stage.  cvA holds the
    // record of the stage about to run, cvB the next one's (its type and weights are needed for the requests under this stage's
    // MFMAs); the one after that is read at the head of the post phase.
    auto read_rec = [&](int rec) { return sCtl[min(rec, NST) * C1_CTL_DWORDS + (lane & (C1_CTL_DWORDS - 1))]; };
    int cvA = read_rec(1), cvB = read_rec(2);
    auto fi = [](int cv, int field) { return __builtin_amdgcn_readlane(cv, field); };
    auto fpf = [&](int cv, int field) { return (float *)(((unsigned long long)(unsigned)fi(cv, field + 1) << 32) | (unsigned)fi(cv, field)); };
    auto run_stage = [&](int s, auto TYc, auto FIRSTc, UnitW (&W)[TP_MAXU]) {
        constexpr int TY = decltype(TYc)::value;
        constexpr bool first = decltype(FIRSTc)::value;   // stage 0: its MFMA phase goes ahead of the wait for the neighbour role's first items
        constexpr int kind = TY == TPT_CONVOUT ? PRO_GATE : TY == TPT_DIL ? PRO_DIL : PRO_CONVIN;
        constexpr bool has_skip = TY == TPT_CONVIN_SKIP;
        constexpr int Co = kind == PRO_GATE ? 2 * NF : NF;
        using std::integral_constant;
        TP_STAMP(0);
        const PostCtl pc{Co, kind, has_skip, fi(cvA, CTL_IN_FORM), fi(cvA, CTL_SAVE_SLOT), 0, fpf(cvA, CTL_BIAS), fpf(cvA, CTL_BIAS2)};
        const StoreCtl sc{kind, fi(cvA, CTL_SKIP_SLOT), fpf(cvA, CTL_R), fpf(cvA, CTL_E), fpf(cvA, CTL_X)};
        const unsigned items = (unsigned)fi(cvA, CTL_TP_ITEMS);
        const int nty = fi(cvB, CTL_TP_TYPE);
        const float *nbase = fpf(cvB, CTL_WTP) + ((size_t)wave * tpt_nu(nty) * 2 * 64 + lane) * 4;
        // operands of this stage's post op: y = ((bias + NA) + centre) + NB (+ gate half, + nin_skip bias); they land under the MFMAs
        f32x2 ob = zero2, obg = zero2, ob2 = zero2, ona[TP_NPC], onb[TP_NPC], onag[TP_NPC], onbg[TP_NPC];
        auto request_operands = [&]() {
            wait_counter(cnt_have, s, items);
            cnt_have = counter(min(s + 1, NST - 2));             // looked at a stage later
            TP_STAMP(1);
            ob = plain(pc.bias + c2);
            if (kind == PRO_GATE) obg = plain(pc.bias + NF + c2);
            if (has_skip) ob2 = plain(pc.bias2 + c2);
#pragma unroll
            for (int k = 0; k < TP_NPC; ++k) {
                const float *nb = a.nbr + (size_t)s * nbr_stage + (size_t)(col0 + (pvalid[k] ? pcol[k] : 0)) * NBR_LD + c2;
#if PS_TP_EXP == 1
                ona[k] = plain(nb);
                onb[k] = plain(nb + nbr_half);
                if (kind == PRO_GATE) { onag[k] = plain(nb + NF); onbg[k] = plain(nb + nbr_half + NF); }
#elif PS_TP_EXP == 2
                ona[k] = onb[k] = onag[k] = onbg[k] = zero2; (void)nb;
#else
                ona[k] = fresh(nb);
                onb[k] = fresh(nb + nbr_half);
                if (kind == PRO_GATE) { onag[k] = fresh(nb + NF); onbg[k] = fresh(nb + nbr_half + NF); }
#endif
            }
            TP_STAMP(2);
        };
#ifdef PS_TP_TRACE_BUILD
        trace_s = s;
#endif
        if (first) {
            mfma_units(TYc, W, nty, nbase, []() {});
            request_operands();
        } else {
            mfma_units(TYc, W, nty, nbase, request_operands);
        }
        // Publishing the input of THIS stage (stored by the post op in front of it, write-through): vmcnt retires in order, so once
        // nothing but the next stage's weight requests (the newest 2 x nnu operations) is outstanding, this wave's stores have
        // been acknowledged; after the barrier that holds for the workgroup.  (Stage 0 requested its operands last: it drains.)
        const bool publish = s < a.publish_upto;
        if (publish) {
            if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (tpt_nu(nty) == TP_MAXU) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * TP_MAXU) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * TP_MINU) : "memory");
        }
        lds_barrier();
        if (publish && t == 0) __hip_atomic_fetch_add(a.done + (size_t)s * CNT_PAD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        TP_STAMP(7);
        const int cvC = read_rec(3 + s);
        auto five = [](const float *p, int stride) {
            return chain_total(*(const f32x2 *)p, *(const f32x2 *)(p + stride), *(const f32x2 *)(p + 2 * stride),
                               *(const f32x2 *)(p + 3 * stride), *(const f32x2 *)(p + 4 * stride));
        };
        f32x2 y[TP_NPC], g[TP_NPC], skip[TP_NPC];
#pragma unroll
        for (int k = 0; k < TP_NPC; ++k) { g[k] = zero2; skip[k] = zero2; }
#pragma unroll
        for (int k = 0; k < TP_NPC; ++k) {
            const float *P = &sP[pcol[k] * SP_LD + c2];
            y[k] = slot_sum2(ob, ona[k], five(P, Co), onb[k]);
            if (kind == PRO_GATE) g[k] = slot_sum2(obg, onag[k], five(P + NF, Co), onbg[k]);
            if (has_skip) skip[k] = five(P + 5 * Co, NF) + ob2;
        }
        emit2(y, g, skip, integral_constant<int, kind>{}, integral_constant<bool, has_skip>{}, pc.in_form, pc.save_slot, sc);
        stage_skip_input(sc.skip_slot);
        cvA = cvB; cvB = cvC;
        lds_barrier();
    };
    auto dispatch_stage = [&](int s, UnitW (&W)[TP_MAXU]) {
        using std::integral_constant;
        const int ty = fi(cvA, CTL_TP_TYPE);
        const integral_constant<bool, false> no{};
        if (ty == TPT_CONVOUT) run_stage(s, integral_constant<int, TPT_CONVOUT>{}, no, W);
        else if (ty == TPT_CONVIN_SKIP) run_stage(s, integral_constant<int, TPT_CONVIN_SKIP>{}, no, W);
        else if (ty == TPT_CONVIN) run_stage(s, integral_constant<int, TPT_CONVIN>{}, no, W);
        else run_stage(s, integral_constant<int, TPT_DIL>{}, no, W);
    };

    // ================= u0 = norm_init(u_init): gather over the (earlier) neighbours' codes =================
    UnitW WA[TP_MAXU];
    {
        const PostCtl pc = post_ctl(0);
        const StoreCtl sc = store_ctl(0);
        {   // stage 0's weights (conv_input without nin_skip: TP_MINU units x 2 halves)
            const float *b0p = weights_base(1, li(1, CTL_TP_TYPE));
#pragma unroll
            for (int u = 0; u < TP_MINU; ++u) { WA[u].a0 = *PS_GC(f32x4, b0p + (size_t)(2 * u) * 256); WA[u].a1 = *PS_GC(f32x4, b0p + (size_t)(2 * u + 1) * 256); }
        }
        cnt_have = counter(0);
        f32x2 y[TP_NPC];
#pragma unroll
        for (int k = 0; k < TP_NPC; ++k) {
            const StepCtx &cx = sC[pcol[k]];
            float mA[9];
            int ncode[9], nl[9];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) { mA[tp] = cx.m[0][tp]; nl[tp] = cx.nloc[tp]; }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) ncode[tp] = a.codes_in[(size_t)pfr[k] * a.L + max(nl[tp], 0)];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) ncode[tp] = nl[tp] >= 0 ? ncode[tp] : UINIT_CLOSED;
            y[k] = uinit_from_codes<f32x2>(ncode, mA, a.uinit_w, a.uinit_b, c2);
        }
        f32x2 z2[TP_NPC];
#pragma unroll
        for (int k = 0; k < TP_NPC; ++k) z2[k] = zero2;
        emit2(y, z2, z2, std::integral_constant<int, PRO_UINIT>{}, std::integral_constant<bool, false>{}, pc.in_form, pc.save_slot, sc);
        stage_skip_input(sc.skip_slot);
    }
    lds_barrier();

    // ================= the 32 conv stages =================
    run_stage(0, std::integral_constant<int, TPT_CONVIN>{}, std::integral_constant<bool, true>{}, WA);   // (stage 0 is a conv_input without nin_skip)
    for (int s = 1; s < NST - 1; ++s) dispatch_stage(s, WA);   // (the last one requests nin_out's record: a dummy, dropped)
#undef TP_STAMP
#undef TP_STAMP2

    // ================= nin_out(elu(u)) (model.py:153): 32 output tiles x 5 chains of 4 MFMAs, logits, draw =================
    {
        const int g_ = 0; (void)g_;
        f32x4 bx[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) bx[j] = sXS4[((4 * j + kk) * XB_LD + i * 4) >> 2];
#pragma unroll
        for (int q = 0; q < NCLS / 16 / TP_WAVES; ++q) {
            const int ot = wave + TP_WAVES * q;
            f32x4 av[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) av[j] = *PS_GC(f32x4, a.out_w + ((size_t)(4 * j + kk) * NCLS + ot * 16 + i) * 4);
            Acc5 acc = acc5_zero();
            mfma_chunk5(av, bx, acc);
            sP4[(i * SLOG_LD + ot * 16 + kk * 4) >> 2] = chunk_total(acc);
        }
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < TP_NPC; ++k) {
        if (!pvalid[k]) continue;
        float lg[8];
        const float *Lp = &sP[pcol[k] * SLOG_LD + lane * 8];
        const f32x4 lo = sP4[(pcol[k] * SLOG_LD + lane * 8) >> 2], hi = sP4[((pcol[k] * SLOG_LD + lane * 8) >> 2) + 1];
        (void)Lp;
#pragma unroll
        for (int q = 0; q < 4; ++q) { lg[q] = lo[q] + a.out_b[lane * 8 + q]; lg[4 + q] = hi[q] + a.out_b[lane * 8 + 4 + q]; }
        const size_t loc = ploc[k];
        if (a.out_logits) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a.out_logits[loc * NCLS + lane * 8 + q] = lg[q];
        }
        if (a.step_logits) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a.step_logits[(size_t)pfr[k] * NCLS + lane * 8 + q] = lg[q];
        }
        if (a.codes && a.region[loc]) {
            const int code = a.forced ? a.forced[loc] : draw_code(lg, a.temperature, a.uniforms[loc], lane);
            if (lane == 0) a.codes[loc] = code;
        }
    }
}

#endif   // !PS_TP_CHAIN2

// ------------------------------------------------------------------------------------------
// chain_role_tp2 (round 3): the chain role of k_column_tp with the post op done IN THE MFMA LAYOUT.
// chain_role_tp deals the (output tile, accumulation chain) units of a stage round-robin to eight waves, parks the five chain
// values of every output in LDS, and after a barrier every wave does the post op of two columns, two channels per lane on 40
// lanes: ~300 vector instructions per wave and stage, the longest part of a stage (2.5-3.8 k of its 7-12 k cycles, with ~0.8 k
// more for the trip of the chain values through LDS).  Here a 640-thread workgroup has ten waves and wave w owns OUTPUT TILE w
// of the stage with all five of its chains (conv_out: 10 tiles of 160 channels; conv_input: tiles 0-4, and nin_skip's five
// tiles on waves 5-9), so that an output's chain values meet in registers: lane (kk, i) holds channels 16 w + 4 kk .. + 3 of
// column i -- the D layout of the MFMA -- and does everything elementwise right there: chain total, ((bias + NA) + centre) +
// NB with 16-byte operand loads, PONO's normalisation, gate / skip / residual, concat-ELU, and ONE 16-byte store per array
// straight into the B-operand layout of the next stage and into the caches.  What has to cross lanes are PONO's two sums over
// the 80 channels of a column: 4 -> 8 -> 16 channels inside the wave (lanes i, i + 16, i + 32, i + 48), five tile sums per
// column through 320 bytes of LDS, in pono_total's association order -- two more LDS barriers per stage, on almost no data.
// The gate half / the nin_skip slot reach the waves that hold y through 5 KB of LDS ahead of the first of them.
// Same arithmetic, same order, bit-identical results (the tests compare with the walk and with the reference's trace).
// ------------------------------------------------------------------------------------------
#if PS_TP_CHAIN2
__device__ __forceinline__ float xlane_add(float x, int mask) { return x + __shfl_xor(x, mask, 64); }
__device__ __forceinline__ void chain_role_tp2(const TpArgs &a, int tile)
{
    static_assert(TP_WAVES == 10, "one output tile of a 160-channel stage per wave");
    __shared__ f32x4 sXS4[2 * XB_SIZE / 4];                                // B-operand layout: input of the centre taps, and behind
    float *const sXS = (float *)sXS4;                                      //   it concat_elu(u_k) feeding nin_skip
    float *const sXb = sXS, *const sSb = sXS + XB_SIZE;
    __shared__ f32x4 sP4[TP_COLS * SLOG_LD / 4];                           // logits of the tile's columns (the end of the chain)
    __shared__ f32x4 sU4[8 * TP_COLS * NF / 4];                            // u0..u7 of the tile's columns [slot][col][80]
    __shared__ f32x4 sG4[5 * 64];                                          // gate half / nin_skip slot of the stage, [tile][lane]
    __shared__ float sStat[2][5][TP_COLS];                                 // PONO: per-tile sums of y and of (y - mean)^2 per column
    __shared__ __attribute__((aligned(16))) StepCtx sC[TP_COLS];
    __shared__ __attribute__((aligned(16))) int sCtl[(NST + 1) * C1_CTL_DWORDS];
    const int t = threadIdx.x, wave = uni(t >> 6), lane = t & 63, i = lane & 15, kk = lane >> 4;
    const int col0 = tile * TP_COLS;
    const int ncl = min(TP_COLS, a.ncols - col0);   // columns of this tile (>= 1)
    {
        const int nq = (int)(sizeof(StepCtx) / 16);
        for (int k = t; k < TP_COLS * nq; k += TP_THREADS) {
            const int c = min(k / nq, ncl - 1);     // absent columns repeat the last one (their results are dropped)
            ((uint4 *)sC)[k] = ((const uint4 *)(a.ctx + col0 + c))[k % nq];
        }
        for (int k = t; k < XB_SIZE; k += TP_THREADS) { sXb[k] = 0.0f; sSb[k] = 0.0f; }
        for (int k = t; k < (NST + 1) * C1_CTL_DWORDS / 4; k += TP_THREADS) ((uint4 *)sCtl)[k] = ((const uint4 *)a.ctl1)[k];
    }
    __syncthreads();
    auto li = [&](int rec, int field) { return uni(sCtl[rec * C1_CTL_DWORDS + field]); };
    auto lpf = [&](int rec, int field) {
        const unsigned lo = (unsigned)li(rec, field), hi = (unsigned)li(rec, field + 1);
        return (float *)(((unsigned long long)hi << 32) | lo);
    };
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool ywave = wave < 5;                       // holds y of output tile `wave` (channels 16 wave .. + 15)
    const int otw = ywave ? wave : wave - 5;           // the tile of y this wave pairs with (gate half / nin_skip slot)
    const bool cvalid = i < ncl;
    const int fr = sC[i].f;
    const size_t loc = (size_t)fr * a.L + sC[i].q;     // this lane's column
    const int ch = 16 * otw + 4 * kk;                  // this lane's channels (of y; the gate wave's are 80 + ch)
    const int xg = (ch >> 2) * XB_LD + i * 4;          // their place in the B-operand layout (floats)
    const size_t nbr_half = (size_t)TP_COL_CAP * NBR_LD, nbr_stage = 2 * nbr_half;
    const unsigned my_uses = a.tile_uses_hi[tile];   // (this variant predates the look-ahead: the host keeps split = 0 for it)
    // The counter is requested a stage before it is looked at.  It must stay a VECTOR value until then: a wave-uniform load is
    // turned into a scalar by v_readfirstlane where it is issued, i.e. the wave waits for it -- and for every load in front of
    // it -- on the spot (~1 k cycles per stage).  The lane offset below is zero, but not to the compiler.
    int vzero = 0;
    asm volatile("" : "+v"(vzero));
    auto counter = [&](int k) { return __hip_atomic_load(a.cnt + tp_cnt_index(k, tile) + vzero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto wait_counter = [&](unsigned have, int k, unsigned items_per_tile) {
        if (a.debug & 1) return;
        const unsigned need = my_uses * items_per_tile;
        int spins = 0;
        while (__builtin_amdgcn_ballot_w64((int)(have - need) < 0) != 0ull) {
            if (++spins > WAIT_SPINS) { if (lane == 0) *a.err = 1; break; }
            __builtin_amdgcn_s_sleep(2);
            have = counter(k);
        }
        asm volatile("" ::: "memory");
    };
    auto fresh4 = [](const float *p) {   // 16 bytes another workgroup of this launch wrote (device-scope loads, past this CU's L1)
        const unsigned long long lo = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load((const unsigned long long *)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return f32x4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                     __uint_as_float((unsigned)(hi >> 32))};
    };
    auto plain4 = [](const float *p) { return *PS_GC(f32x4, p); };
    // sum over the 80 channels of every column, in pono_total's order: 2 + 2 channels in the lane, 4 + 4 and 8 + 8 across the
    // lanes of the wave (tile sums), then R0 = t0 + t1, R1 = t2 + t3, R2 = t4 through LDS, total = R2 + (R1 + R0)
    auto tile_sum = [&](const f32x4 &v) {
        float q = (v[0] + v[1]) + (v[2] + v[3]);
        q = xlane_add(q, 16);
        q = xlane_add(q, 32);
        return q;
    };
    auto total_of = [&](int which) {
        const float t0 = sStat[which][0][i], t1 = sStat[which][1][i], t2 = sStat[which][2][i], t3 = sStat[which][3][i], t4 = sStat[which][4][i];
        return t4 + ((t2 + t3) + (t0 + t1));
    };
#ifdef PS_TP_TRACE_BUILD
    int trace_s = 0;
    const int trace_wave = a.debug >> 8;   // PS_COLUMN_DEBUG = 256 * wave (+ mode): the wave whose stamps are kept
#define TP2_STAMP(slot) do { if (a.trace && tile == 0 && t == 64 * trace_wave) a.trace[trace_s * 8 + (slot)] = clock64(); } while (0)
#else
#define TP2_STAMP(slot) do { } while (0)
#endif
    f32x4 ucur = zero;   // u of this lane's (column, channels): the residual input of the next gate
    // everything after the products: y = this lane's four channels of its column (y waves), aux = the gate half / nin_skip slot
    // (waves 5-9).  KIND / HAS_SKIP are compile-time; in_form, save_slot, skip_slot wave-uniform.
    // weights of this wave's tile of a stage: NG 16-byte loads from the standard packed layout [c / 4][o][4] (the centre tap of
    // the conv, or nin_skip's matrix for waves 5-9 of a conv_input with skip)
    f32x4 wl[10];
    const float *wcur_base = nullptr; // this wave's weights of the stage whose control values are in nctl ...
    size_t wcur_gs = 0;
    const float *wq_base = nullptr;   // ... and this wave's weights of the stage AFTER the one whose control values are in nctl
    size_t wq_gs = 0;                 //   (floats between channel groups): requested under that stage's own MFMAs
    auto weight_address = [&](int rec) {   // (every wave, idle ones and 5-group stages included, gets ten valid addresses: the request
        const int cv = sCtl[rec * C1_CTL_DWORDS + (lane & (C1_CTL_DWORDS - 1))];   // count stays the same on every path)
        auto fi = [&](int field) { return __builtin_amdgcn_readlane(cv, field); };
        auto fp = [&](int field) { return (const float *)(((unsigned long long)(unsigned)fi(field + 1) << 32) | (unsigned)fi(field)); };
        const int ty = fi(CTL_TP_TYPE);
        const int Co = ty == TPT_CONVOUT ? 2 * NF : NF;
        const float *w = (!ywave && ty == TPT_CONVIN_SKIP) ? fp(CTL_WS) : fp(CTL_WC);
        const int o = (ty == TPT_CONVOUT ? wave : otw) * 16 + i;
        wq_base = w + ((size_t)kk * Co + o) * 4;
        wq_gs = (size_t)16 * Co;
    };
    // The NEXT stage's control values and post-op operands are fetched at the head of this phase (prefetch_stage): a stage then
    // opens with its products, not with ~1.5 k cycles of descriptor reads, counter check and operand requests.
    struct StageCtl { int ty, in_form, save_slot, skip_slot; float *R, *E, *X; };
    StageCtl nctl{};
    f32x4 nob = zero, nona = zero, nonb = zero;
    unsigned cnt_have = 0;
    auto prefetch_stage = [&](int s) {   // stage s < NST - 1: its record is 1 + s
        const int rec = 1 + s;
        // the whole 32-dword record with ONE LDS read (a dword per lane), its fields by v_readlane: a ds_read + readfirstlane per
        // field were two dozen dependent round trips on the critical path of every stage
        const int cv = sCtl[rec * C1_CTL_DWORDS + (lane & (C1_CTL_DWORDS - 1))];
        auto fi = [&](int field) { return __builtin_amdgcn_readlane(cv, field); };
        auto fp = [&](int field) { return (float *)(((unsigned long long)(unsigned)fi(field + 1) << 32) | (unsigned)fi(field)); };
        nctl = StageCtl{fi(CTL_TP_TYPE), fi(CTL_IN_FORM), fi(CTL_SAVE_SLOT), fi(CTL_SKIP_SLOT), fp(CTL_R), fp(CTL_E), fp(CTL_X)};
        const float *bias = fp(CTL_BIAS), *bias2 = fp(CTL_BIAS2);
        wait_counter(cnt_have, s, (unsigned)fi(CTL_TP_ITEMS));
        cnt_have = counter(min(s + 1, NST - 2));             // looked at a stage later
        // (requested by every wave alike -- idle ones drop them -- so that the request count is the same on every path)
        const int och = nctl.ty == TPT_CONVOUT ? 16 * wave + 4 * kk : ch;   // channel of this lane in the stage's output (gate half: 80 + ch)
        nob = plain4((!ywave && nctl.ty == TPT_CONVIN_SKIP) ? bias2 + ch : bias + och);
        const float *nb = a.nbr + (size_t)s * nbr_stage + (size_t)(col0 + (cvalid ? i : 0)) * NBR_LD + och;
        nona = fresh4(nb);
        nonb = fresh4(nb + nbr_half);
        weight_address(2 + s);
    };
    auto finish = [&](const f32x4 &yv, const f32x4 &aux, bool has_aux, auto KINDc, auto SKIPc, int in_form, int save_slot, int skip_slot,
                      float *R, float *E, float *X, int next_stage) {
        constexpr int kind = decltype(KINDc)::value;
        constexpr bool has_skip = decltype(SKIPc)::value;
        if (next_stage < NST - 1) prefetch_stage(next_stage);
        if (!ywave && has_aux) sG4[otw * 64 + lane] = aux;
        if (ywave) {
            const float ts = tile_sum(yv);
            if (kk == 0) sStat[0][wave][i] = ts;
        }
        TP2_STAMP(4);
        lds_barrier();
        TP2_STAMP(5);
        f32x4 d = zero;
        if (ywave) {
            const float mean = pono_mean(total_of(0));
            d = yv - mean;
            const float ts = tile_sum(d * d);
            if (kk == 0) sStat[1][wave][i] = ts;
        }
        lds_barrier();
        TP2_STAMP(6);
        if (ywave) {
            const float inv = pono_inv(total_of(1));
            const f32x4 n = d * inv;
            f32x4 out = n;
            if (kind == PRO_CONVIN && has_skip) out = n + sG4[otw * 64 + lane];
            if (kind == PRO_GATE) {
                const f32x4 g = sG4[otw * 64 + lane];
                out = ucur + n * f32x4{sigmoid1(g[0]), sigmoid1(g[1]), sigmoid1(g[2]), sigmoid1(g[3])};
            }
            f32x4 ep, en;
#pragma unroll
            for (int e = 0; e < 4; ++e) { float p_, n_; celu_pair(out[e], p_, n_); ep[e] = p_; en[e] = n_; }
            if (cvalid) {
                if (in_form == IN_CELU) { sXS4[xg >> 2] = ep; sXS4[(xg + 20 * XB_LD) >> 2] = en; }
                else if (in_form == IN_RAW) sXS4[xg >> 2] = out;
                else sXS4[xg >> 2] = ep;
                if (kind == PRO_CONVIN) {
                    *PS_G(f32x4, X + loc * (2 * NF) + ch) = ep;
                    *PS_G(f32x4, X + loc * (2 * NF) + NF + ch) = en;
                } else {
                    *PS_G(f32x4, R + loc * R_LD + ch) = out;
                    *PS_G(f32x4, E + loc * (2 * NF) + ch) = ep;
                    *PS_G(f32x4, E + loc * (2 * NF) + NF + ch) = en;
                    if (save_slot >= 0) sU4[((save_slot * TP_COLS + i) * NF + ch) >> 2] = out;
                }
            }
            if (kind != PRO_CONVIN) ucur = out;
            if (skip_slot >= 0) {   // concat_elu(u_k) of the saved u the NEXT stage's nin_skip reads
                const f32x4 u = sU4[((skip_slot * TP_COLS + i) * NF + ch) >> 2];   // (written above when it is this very u: same lane, in order)
                f32x4 sp, sn;
#pragma unroll
                for (int e = 0; e < 4; ++e) { float p_, n_; celu_pair(u[e], p_, n_); sp[e] = p_; sn[e] = n_; }
                sXS4[(XB_SIZE + xg) >> 2] = sp;
                sXS4[(XB_SIZE + xg + 20 * XB_LD) >> 2] = sn;
            }
        }
        TP2_STAMP(7);
        lds_barrier();
    };
    // one conv stage.  TY fixes the products (which waves, how many groups) and the post op that follows.
    auto run_stage = [&](int s, auto TYc) {
        constexpr int TY = decltype(TYc)::value;
        constexpr int kind = TY == TPT_CONVOUT ? PRO_GATE : TY == TPT_DIL ? PRO_DIL : PRO_CONVIN;
        constexpr bool has_skip = TY == TPT_CONVIN_SKIP;
        constexpr bool second = TY == TPT_CONVOUT || TY == TPT_CONVIN_SKIP;   // waves 5-9 have products too
        constexpr int NGH = TY == TPT_DIL ? 1 : 2;
        using std::integral_constant;
        const bool active = ywave || second;
#ifdef PS_TP_TRACE_BUILD
        trace_s = s;
#endif
        TP2_STAMP(0);
        // this stage's control values and operands were fetched during the previous stage's post op
        const StageCtl c = nctl;
        const f32x4 ob = nob, ona = nona, onb = nonb;
        TP2_STAMP(1);
        f32x4 tsum = zero;
        TP2_STAMP(2);
        // Weights go through the CU's vector-memory path at 64 bytes a clock: 10 waves x 10 KB = ~1700 cycles per stage that only
        // the matrix phase is long enough to cover -- and only if every wave spreads its requests BETWEEN its own MFMAs (all waves
        // run the same phase at the same time: requests bunched behind the MFMAs, or in front of the post op, queue up there).
        // One register buffer, refilled in place, a 16-byte request per five MFMAs:
        //   first half of the products (groups 0-4, wl[0..4])   <-  wl[5..9] of THIS stage  (needed by the second half)
        //   second half (groups 5-9, wl[5..9])                  <-  wl[0..4] of the NEXT stage
        const float *wc = wcur_base, *wn = wq_base;     // this stage's / the next stage's weights of this wave
        const size_t gc = wcur_gs, gn = wq_gs;
        if (active) {
            const f32x4 *x4 = (!ywave && has_skip) ? sXS4 + XB_SIZE / 4 : sXS4;
            Acc5 acc = acc5_zero();
#pragma unroll
            for (int h = 0; h < NGH; ++h) {
                f32x4 bx[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) bx[j] = x4[((4 * (j + 5 * h) + kk) * XB_LD + i * 4) >> 2];
                f32x4 nw[5];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int j = 0; j < 5; ++j)
                        acc.v[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[j + 5 * h][c], bx[j][c], acc.v[j], 0, 0, 0);
                    // requests of this quarter: 2, 1, 1, 1
#pragma unroll
                    for (int q = (c == 0 ? 0 : c + 1); q < (c == 0 ? 2 : c + 2); ++q)
                        nw[q] = (h == 0 && NGH == 2) ? *PS_GC(f32x4, wc + (5 + q) * gc) : *PS_GC(f32x4, wn + q * gn);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < 5; ++q) wl[(h == 0 && NGH == 2) ? 5 + q : q] = nw[q];
            }
            tsum = chunk_total(acc);
        } else {   // no products in this stage: only the next stage's first half is due
#pragma unroll
            for (int q = 0; q < 5; ++q) wl[q] = *PS_GC(f32x4, wn + q * gn);
        }
        wcur_base = wn;
        wcur_gs = gn;
        f32x4 yv;
        if (!ywave && has_skip) yv = tsum + ob;                          // nin_skip slot + its bias (layers.py:155-156)
        else yv = ((ob + ona) + tsum) + onb;                             // y = ((bias + NA) + centre) + NB
#ifdef PS_TP_TRACE_BUILD
        if (a.trace && tile == 0 && t == 64 * trace_wave && yv[0] != 12345.678f) a.trace[trace_s * 8 + 3] = clock64();
#endif
        finish(yv, yv, second, integral_constant<int, kind>{}, integral_constant<bool, has_skip>{}, c.in_form, c.save_slot, c.skip_slot, c.R, c.E,
               c.X, s + 1);
    };
    // ================= u0 = norm_init(u_init): gather over the (earlier) neighbours' codes =================
    {
        weight_address(1);           // stage 0's first half: requested here, under the gather (the second under its own MFMAs)
#pragma unroll
        for (int j = 0; j < 5; ++j) wl[j] = *PS_GC(f32x4, wq_base + j * wq_gs);
        wcur_base = wq_base;
        wcur_gs = wq_gs;
        cnt_have = counter(0);
        f32x4 y = zero;
        if (ywave) {
            const StepCtx &cx = sC[i];
            float mA[9];
            int ncode[9], nl[9];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) { mA[tp] = cx.m[0][tp]; nl[tp] = cx.nloc[tp]; }
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) ncode[tp] = a.codes_in[(size_t)fr * a.L + max(nl[tp], 0)];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) ncode[tp] = nl[tp] >= 0 ? ncode[tp] : UINIT_CLOSED;
            y = uinit_from_codes<f32x4>(ncode, mA, a.uinit_w, a.uinit_b, ch);
        }
        finish(y, y, false, std::integral_constant<int, PRO_UINIT>{}, std::integral_constant<bool, false>{}, li(0, CTL_IN_FORM),
               li(0, CTL_SAVE_SLOT), li(0, CTL_SKIP_SLOT), lpf(0, CTL_R), lpf(0, CTL_E), lpf(0, CTL_X), 0);
    }
    // ================= the 32 conv stages =================
    for (int s = 0; s < NST - 1; ++s) {
        using std::integral_constant;
        const int ty = nctl.ty;
        if (ty == TPT_CONVOUT) run_stage(s, integral_constant<int, TPT_CONVOUT>{});
        else if (ty == TPT_CONVIN_SKIP) run_stage(s, integral_constant<int, TPT_CONVIN_SKIP>{});
        else if (ty == TPT_CONVIN) run_stage(s, integral_constant<int, TPT_CONVIN>{});
        else run_stage(s, integral_constant<int, TPT_DIL>{});
    }
    // ================= nin_out(elu(u)) (model.py:153): 32 output tiles x 5 chains of 4 MFMAs, logits, draw =================
    {
        f32x4 bx[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) bx[j] = sXS4[((4 * j + kk) * XB_LD + i * 4) >> 2];
        for (int ot = wave; ot < NCLS / 16; ot += TP_WAVES) {
            f32x4 av[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) av[j] = *PS_GC(f32x4, a.out_w + ((size_t)(4 * j + kk) * NCLS + ot * 16 + i) * 4);
            Acc5 acc = acc5_zero();
            mfma_chunk5(av, bx, acc);
            sP4[(i * SLOG_LD + ot * 16 + kk * 4) >> 2] = chunk_total(acc);
        }
    }
    lds_barrier();
    for (int col = wave; col < ncl; col += TP_WAVES) {
        float lg[8];
        const f32x4 lo = sP4[(col * SLOG_LD + lane * 8) >> 2], hi = sP4[((col * SLOG_LD + lane * 8) >> 2) + 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) { lg[q] = lo[q] + a.out_b[lane * 8 + q]; lg[4 + q] = hi[q] + a.out_b[lane * 8 + 4 + q]; }
        const int cf = uni(sC[col].f);
        const size_t cloc = (size_t)cf * a.L + uni(sC[col].q);
        if (a.out_logits) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a.out_logits[cloc * NCLS + lane * 8 + q] = lg[q];
        }
        if (a.step_logits) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a.step_logits[(size_t)cf * NCLS + lane * 8 + q] = lg[q];
        }
        if (a.codes && a.region[cloc]) {
            const int code = a.forced ? a.forced[cloc] : draw_code(lg, a.temperature, a.uniforms[cloc], lane);
            if (lane == 0) a.codes[cloc] = code;
        }
    }
}
#endif

// chain_xcds = 0: blocks [0, nbr_wgs) neighbour role (dispatched first: the chain tiles wait for their items), the blocks after them one chain tile each.
// chain_xcds = cx > 0 (speed only; block b runs on XCD b % 8): the chain tiles are the blocks on XCDs 0 .. cx-1, whose L2s then
// hold the 2.8 MB of centre-tap weights instead of sharing their bandwidth with the neighbour role's operand stream; every
// other block is a neighbour workgroup (the spare CUs of the chain XCDs too when fill is set).
#if PS_TP_CHAIN2
#define CHAIN_ROLE_TP chain_role_tp2
#else
#define CHAIN_ROLE_TP chain_role_tp
#endif
__global__ __launch_bounds__(TP_THREADS) void k_column_tp(TpArgs a)
{
    const int b = blockIdx.x, cx = a.chain_xcds;
    if (cx == 0) {
        if (b < a.nbr_wgs) { if ((a.debug & 3) != 3) nbr_role_tp(a, b); }
        else if ((a.debug & 3) != 2) CHAIN_ROLE_TP(a, b - a.nbr_wgs);
        return;
    }
    const int x = b & 7, slot = b >> 3;
    if (x < cx) {
        const int tile = slot * cx + x;
        if (tile < a.tiles) { if ((a.debug & 3) != 2) CHAIN_ROLE_TP(a, tile); }
        else if (a.fill_nbr >= 0 && (a.debug & 3) != 3) nbr_role_tp(a, a.fill_nbr + (tile - a.tiles));
    } else if ((a.debug & 3) != 3) {
        nbr_role_tp(a, slot * (8 - cx) + (x - cx));
    }
}

// (ot, j, nin_skip?) of unit n of a stage with Co output channels: main units tile-major, then nin_skip's
__device__ __host__ __forceinline__ void tp_unit_of(int n, int Co, int &ot, int &j, bool &skip)
{
    const int um = 5 * (Co >> 4);
    skip = n >= um;
    const int m = skip ? n - um : n;
    ot = m / 5;
    j = m - ot * 5;
}

// the centre tap (+ nin_skip) of a stage in the throughput chain role's own order: out[wave][unit][half][lane][4] =
// W[o = ot*16 + i][channels 16*(j + 5*half) + 4*kk .. +3] for lane (kk, i) -- what load h of unit u of wave w wants, KB by KB
__global__ void k_pack_tp(const float *wc, const float *ws, int Co, int NG, int type, float *out)
{
    const int NU = tpt_nu(type), total = tpt_units(type);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= TP_WAVES * NU * 2 * 64) return;
    const int lane = idx & 63, h = (idx >> 6) & 1, u = ((idx >> 7) % NU), w = (idx >> 7) / NU;
    const int n = w + TP_WAVES * u, i = lane & 15, kk = lane >> 4;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (n < total && (h == 0 || NG == 10)) {
        int ot, j; bool skip;
        tp_unit_of(n, Co, ot, j, skip);
        const float *wp = skip ? ws : wc;
        const int cw = skip ? NF : Co;
        v = *(const f32x4 *)(wp + ((size_t)(4 * (j + 5 * h) + kk) * cw + ot * 16 + i) * 4);
    }
    *(f32x4 *)(out + (size_t)idx * 4) = v;
}

// repack the centre tap (+ nin_skip) of a stage for the chain role: out[step][chain][4]
__global__ void k_pack_valu(const float *wc, const float *wskip, int Co, int nchain, int nstep, float *out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nstep * nchain * 4) return;
    const int kk = idx & 3, t = (idx >> 2) % nchain, st = (idx >> 2) / nchain;
    const int gl = st >> 2, c = st & 3;
    const bool main = t < 5 * Co;
    const int t2 = main ? t : t - 5 * Co, n = main ? Co : NF;
    const int j = t2 / n, o = t2 - j * n;
    const int ch = 16 * (5 * gl + j) + 4 * kk + c;
    const float *w = main ? wc : wskip;
    out[idx] = w[((size_t)(ch >> 2) * n + o) * 4 + (ch & 3)];
}

// nin_out for the chain role: out[step 0..11][thread 0..1023][4]; thread (o = t & 511, part = t >> 9)
__global__ void k_pack_valu_out(const float *wo /*[20][512][4]*/, float *out)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C1_OUT_STEPS * C1_THREADS * 4) return;
    const int kk = idx & 3, t = (idx >> 2) & (C1_THREADS - 1), st = (idx >> 2) / C1_THREADS;
    const int o = t & (NCLS - 1), part = t >> 9;
    const int cj = st >> 2, c = st & 3;
    if (part == 1 && cj == 2) { out[idx] = 0.0f; return; }
    const int ch = 16 * (part * 3 + cj) + 4 * kk + c;
    out[idx] = wo[((size_t)(ch >> 2) * NCLS + o) * 4 + (ch & 3)];
}

// kernel masks from generation orders, on the device (masking.py:287-370: tap open iff the neighbour precedes the location in
// the order; centre 0 for type A, 1 for type B).  One block per frame: ranks in LDS, then the 3 x 9 x L mask values.
__global__ __launch_bounds__(256) void k_order_masks(const int32_t *order_loc, int H, int W, float *m_init, float *m_und, float *m_dil, int32_t *status)
{
    extern __shared__ int sRank[];
    const int L = H * W, f = blockIdx.x;
    const int32_t *ol = order_loc + (size_t)f * L;
    for (int k = threadIdx.x; k < L; k += blockDim.x) sRank[k] = -1;
    __syncthreads();
    for (int k = threadIdx.x; k < L; k += blockDim.x) {
        const int q = ol[k];
        if (q < 0 || q >= L) { if (status) atomicOr(status, PS_STATUS_BAD_ORDER); continue; }
        sRank[q] = k;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 27 * L; k += blockDim.x) {
        const int kind = k / (9 * L), t = (k / L) % 9, q = k % L;
        const int dil = kind == 2 ? 2 : 1, r = q / W, c = q - r * W, rr = r + (t / 3 - 1) * dil, cc = c + (t % 3 - 1) * dil;
        float v;
        if (t == 4) v = kind == 0 ? 0.0f : 1.0f;
        else v = (rr >= 0 && rr < H && cc >= 0 && cc < W && sRank[rr * W + cc] >= 0 && sRank[rr * W + cc] < sRank[q]) ? 1.0f : 0.0f;
        float *dst = kind == 0 ? m_init : kind == 1 ? m_und : m_dil;
        dst[((size_t)f * 9 + t) * L + q] = v;
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
    float *nbr = nullptr;           // column mode: neighbour slots [NST][2][COL_CAP][160]
    float *col_logits = nullptr;
    StepCtx *ctx = nullptr;         // column records of a run, [maxF * L]
    int32_t *pstart = nullptr;      // (N_EVAL, F) first rank of the prefix anyone reads, per stage and frame (k_prefix_starts)
    int *ctl1 = nullptr;            // the same for the chain role (scalar-load records)
    unsigned *cnt = nullptr;        // [NST][MAX_TILES] padded completion counters of the neighbour role, never reset
    int *err = nullptr;             // device flag: a bounded wait of the chain role ran out
    unsigned tile_uses[MAX_TILES] = {};  // column launches so far that had tile t (target of the counters)
    NbrWork *work = nullptr;
    int nwork = 0;
    // throughput form (k_column_tp): launches of more than tp_min_cols columns
    NbrWorkTp *work_tp = nullptr;
    int nwork_tp = 0;
    TpUnit *units_tp = nullptr;     // unit tables of the four stage types [4][TP_WAVES][TP_MAXU]
    float *nbr_tp = nullptr;        // neighbour slots [NST][2][TP_COL_CAP][160]
    unsigned *cnt_tp = nullptr;     // [NST][TP_MAX_TILES] padded completion counters, never reset
    // look-ahead of the neighbour role (nbr_role_tp): slots, counters and their targets are double-buffered by launch parity
    unsigned tile_uses_tp_lo[2][TP_MAX_TILES] = {}, tile_uses_tp_hi[2][TP_MAX_TILES] = {};
    unsigned *done_tp = nullptr;    // [NST] padded: chain tiles that have published the input of stage k, never reset
    unsigned done_total = 0;        // what they stand at when every publishing launch so far is through
    int tp_ahead = 12;              // PS_TP_AHEAD: stages [0, tp_ahead) of a launch are computed by the launch in front of it (0: off)
    int tp_wsplit = 0;              // first entry of work_tp whose stage is >= tp_ahead
    const StepCtx *ahead_rec = nullptr;   // the launch the last one prepared: its first record, its columns, the parity it wrote to
    int ahead_n = 0, ahead_parity = 0;
    int tp_launch_no = 0;           // throughput-form launches of the current run so far (tuning: PS_TP_TRACE_LAUNCH)
    // the same look-ahead for the latency form (k_column_la; from one latency-form launch to the next): `nbr` and `cnt` hold two halves
    unsigned col_uses_lo[2][MAX_TILES] = {}, col_uses_hi[2][MAX_TILES] = {};
    unsigned *done_col = nullptr;   // [NST] padded: columns that have published the input of stage k, never reset
    unsigned done_col_total = 0;
    int col_ahead = 16;             // PS_COL_AHEAD: stages computed a launch ahead (0: off -- k_column as before)
    int col_wsplit = 0;             // first entry of `work` whose stage is >= col_ahead
    const StepCtx *col_ahead_rec = nullptr;
    int col_ahead_n = 0, col_ahead_parity = 0;
    ColTaps *taps = nullptr;        // neighbour rows of the columns of a run, [maxF * L]
    unsigned long long *tp_trace = nullptr;   // tuning builds: stamps of the last k_column_tp launch (ps_pixelcnn_debug_cache what 4)
    int n_cus = 256;                // compute units of the device: workgroups of a column launch that are resident together
    bool xcd_even = true;           // n_cus is an even share of the 8 XCDs of a whole MI355X (block b runs on XCD b % 8)
    int tp_min_cols = 2 * COL_CAP + 1;  // PS_TP_MIN_COLS: a wavefront of up to 256 columns is two latency-form launches (2 x 48 us) rather than one
                                        // throughput-form launch (130 us whatever its width); measured crossover 257 .. 385 columns
    int tp_xcds = -1;               // PS_TP_XCDS: 0 = chain tiles anywhere, -1 = on as few XCDs as hold them, n = on at least n XCDs
    int tp_fill = 0;                // PS_TP_FILL: neighbour workgroups on the spare CUs of the chain XCDs (off since the neighbour role
                                    // works a launch ahead: it has time to spare, and the chain tiles are faster with their XCDs' L2 to themselves)
    int col_cap = COL_CAP;          // columns per launch (PS_COL_CAP: tuning)
    int chain_xcds = 0;             // PS_CHAIN_XCDS: tuning (0 = automatic)
    int force_groups = 0;           // PS_NBR_GROUPS: tuning (0 = automatic)
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

// grid of k_gemm: (channel blocks x item blocks x slots) laid out XCD by XCD, see the kernel
// -> true when the post op `post` was done in the same launch (k_gemm_wg)
bool launch_gemm(GemmArgs &a, int item_blocks, hipStream_t st, const PostArgs *post = nullptr)
{
    const bool split = getenv("PS_GEMM_SPLIT_SLOTS") != nullptr;   // tuning: one wave per slot
    a.nx = (a.Co_pad + 16 * GEMM_T - 1) / (16 * GEMM_T);
    a.ny = item_blocks;
    a.tpx = (item_blocks + N_XCD - 1) / N_XCD;
    // one wave per slot while that is what it takes to fill the chip (4096 wave slots): a 16-view prefix is 2870 (tile,
    // channel block) pairs, one view 180 -- walking all slots in one wave would leave most of the SIMDs idle and make
    // each wave three times as long
    const char *mm = getenv("PS_GEMM_MERGE_MIN");   // (read per launch: the parity test switches forms inside one process)
    const int merge_min = mm ? atoi(mm) : 8192;
    a.zgrid = split || a.nx * a.ny < merge_min ? a.nslots : 1;
    if (a.zgrid != 1 || a.nslots < 3) a.sum_bias = nullptr;   // (only a wave that walks NA, C and NB can add them up)
    // the workgroup form (k_gemm_wg: input rows shared through LDS) from PS_GEMM_WG_MIN item tiles on, for the shapes of the
    // network's 3x3 convs; it produces the summed form (y in place of slot NA), bit-identical to k_gemm's
    const char *wm = getenv("PS_GEMM_WG_MIN");
    const int wg_min = wm ? atoi(wm) : 1024;
    const bool shape_ok = (a.Cin == 2 * NF || a.Cin == NF) && (a.Co_pad == NF || (a.Co_pad == 2 * NF && a.Cin == 2 * NF));
    if (a.sum_bias && a.zgrid == 1 && shape_ok && item_blocks >= wg_min && a.tiles_per_block == 1) {
        const int kind = a.Co_pad == 2 * NF ? GW_CONVOUT : a.Cin == 2 * NF ? GW_CONVIN : GW_DIL;
        // item tiles per workgroup (tuning: PS_WG_TI = "out,in,dil")
        int ti_of[3] = {1, 2, 2};   // (conv_out with 16 items per workgroup: 168 registers, three workgroups per CU -- 0.6 % of the 128-view step over {2, 2, 2})
        if (const char *e = getenv("PS_WG_TI")) sscanf(e, "%d,%d,%d", &ti_of[0], &ti_of[1], &ti_of[2]);
        const int TI = ti_of[kind], MI = 16 * TI;
        a.ny = (a.nitems + MI - 1) / MI;
        a.tpx = (a.ny + N_XCD - 1) / N_XCD;
        const dim3 grid((unsigned)(N_XCD * a.tpx)), block(GW_THREADS);
        a.wg_reverse = getenv("PS_WG_REVERSE") ? 1 : 0;
        const bool fuse = post && !getenv("PS_GEMM_WG_NOFUSE");   // (tuning / parity: products only, k_post_grid afterwards)
        PostArgs pp{};
        if (fuse) { pp = *post; pp.summed = 1; }
        const int fz = fuse ? 1 : 0;
        if (kind == GW_CONVOUT && TI == 1) hipLaunchKernelGGL((k_gemm_wg<GW_CONVOUT, 1>), grid, block, 0, st, a, pp, fz);
        else if (kind == GW_CONVOUT) hipLaunchKernelGGL((k_gemm_wg<GW_CONVOUT, 2>), grid, block, 0, st, a, pp, fz);
        else if (kind == GW_CONVIN && TI == 2) hipLaunchKernelGGL((k_gemm_wg<GW_CONVIN, 2>), grid, block, 0, st, a, pp, fz);
        else if (kind == GW_CONVIN) hipLaunchKernelGGL((k_gemm_wg<GW_CONVIN, 4>), grid, block, 0, st, a, pp, fz);
        else if (TI == 2) hipLaunchKernelGGL((k_gemm_wg<GW_DIL, 2>), grid, block, 0, st, a, pp, fz);
        else hipLaunchKernelGGL((k_gemm_wg<GW_DIL, 4>), grid, block, 0, st, a, pp, fz);
        return fuse;
    }
    hipLaunchKernelGGL(k_gemm, dim3((unsigned)(N_XCD * a.nx * a.tpx * a.zgrid)), dim3(64), 0, st, a);
    return false;
}

// ------------------------------------------------------------------------------------------
// whole-grid evaluation (reference-faithful forward; cache build before the column steps)
// logits: null (caches only), (F,512,H,W) when nchw, else (F*L,512) by location
// ------------------------------------------------------------------------------------------
// (with an order: the pass can be restricted to frames [f0, f0 + nf) of the F -- independent passes over disjoint frame ranges
// may run on different streams)
void run_grid(ps_pixelcnn *h, int F, const int32_t *codes, const Masks &m, float *logits, bool nchw, hipStream_t st,
              const int32_t *order = nullptr, int npre = -1, int f0 = 0, int nf = -1)
{
    if (nf < 0) nf = F;
    const ItemMap all_items{order, order ? npre : h->L, nullptr, f0};
    const int nitems = nf * all_items.npre;
    if (nitems <= 0) return;  // an AR run that starts at rank 0 has no prefix
    const int pblocks = (nitems + 3) / 4;
    // the prefix of an AR run: only the items somebody reads, stage by stage (k_prefix_starts).  PS_PREFIX_FULL=1: all of them.
    // (with out_logits the caller also gets the logits of the prefix locations: every item is needed then.  PS_PREFIX_CONE_FORCE
    // keeps the elimination on for the parity test, which compares the logits of the WALKED locations only.)
    const bool cone = order && (!logits || getenv("PS_PREFIX_CONE_FORCE")) && h->L <= STARTS_MAXL && !getenv("PS_PREFIX_FULL");
    if (cone) {
        StartsArgs sa{order, m.und, m.dil, h->H, h->W, h->L, npre, F, {}, {}, {}, {}, {}, h->pstart, f0};
        for (int g = 0; g < NGATED; ++g) { sa.g_in[g] = h->gated[g].node_in; sa.g_out[g] = h->gated[g].node_out; sa.g_skip[g] = h->gated[g].node_skip; }
        for (int d = 0; d < 4; ++d) { sa.d_in[d] = h->dil[d].node_in; sa.d_out[d] = h->dil[d].node_out; }
        hipLaunchKernelGGL(k_prefix_starts, dim3(nf), dim3(1024), 0, st, sa);
    }
    float *const part = h->partial + (size_t)4 * f0 * h->L * (2 * NF);
    ItemMap items = all_items;
    auto at_stage = [&](int stage_id) { items.start = cone ? h->pstart + (size_t)stage_id * F : nullptr; };
    // -> 0: raw slots in `partial`, 1: slots summed by the kernel, 2: the post op `post` done by the kernel as well
    auto gemm = [&](GemmArgs &a, const float *mask, const float *sum_bias = nullptr, const PostArgs *post = nullptr) {
        a.items = items;
        a.H = h->H; a.W = h->W; a.L = h->L; a.nitems = nitems;
        a.mask = mask; a.mask_fstride = (size_t)9 * h->L; a.tiles_per_block = 1;
        a.partial = h->partial + (size_t)4 * f0 * h->L * (2 * NF);   // (the frame range's own part of the scratch: passes over disjoint ranges may run side by side)
        a.sum_bias = sum_bias;
        const int tiles = (nitems + 15) / 16;
        if (launch_gemm(a, tiles, st, post)) return 2;
        return a.sum_bias != nullptr ? 1 : 0;
    };
    // PS_GEMM_FUSE=1: one launch per stage (k_stage_fused: products + post op, no partial sums in HBM).  Bit-identical
    // (tested) but measured SLOWER than k_gemm + k_post_grid at 128 frames -- conv_out 488 us against 347 + 29, conv_in 241
    // against 175 + 17: the five waves of a workgroup wait for each other and the matrix pipes idle under the post op --
    // so it is not the default.  (Read per call: the parity test switches forms inside one process.)
    const int tiles_all = (nitems + 15) / 16;
    const bool fused = getenv("PS_GEMM_FUSE") != nullptr;
    auto stage = [&](GemmArgs &a, const float *mask, PostArgs &p, int kind) {   // -> true when the post op is done too
        if (!fused) return false;
        a.items = items;
        a.H = h->H; a.W = h->W; a.L = h->L; a.nitems = nitems;
        a.mask = mask; a.mask_fstride = (size_t)9 * h->L; a.partial = nullptr; a.tiles_per_block = 1;
        a.sum_bias = p.bias;
        a.nx = (a.Co_pad + 31) / 32; a.ny = tiles_all; a.tpx = (tiles_all + N_XCD - 1) / N_XCD; a.zgrid = 1;
        p.summed = 1;
        const dim3 grid((unsigned)(N_XCD * a.tpx)), block(64 * a.nx);
        if (kind == POST_CONVIN) hipLaunchKernelGGL(k_stage_fused<POST_CONVIN>, grid, block, 0, st, a, p);
        else if (kind == POST_GATE) hipLaunchKernelGGL(k_stage_fused<POST_GATE>, grid, block, 0, st, a, p);
        else hipLaunchKernelGGL(k_stage_fused<POST_DIL>, grid, block, 0, st, a, p);
        return true;
    };
    {   // u_init + norm_init  (model.py:132)
        at_stage(0);
        UinitArgs u{items, codes, m.init, h->uinit_w, h->uinit_b, h->R[0], h->E[0], h->H, h->W, h->L, nitems};
        hipLaunchKernelGGL(k_uinit_grid, dim3(pblocks), dim3(256), 0, st, u);
    }
    auto gated = [&](int g) {
        const ps_pixelcnn::Gated &G = h->gated[g];
        GemmArgs a{};
        at_stage(1 + g);
        conv_taps(a, h->E[G.node_in], 2 * NF, G.w_in, 2 * NF, NF, 1);                 // conv_input (layers.py:153)
        if (G.node_skip >= 0) {                                                         // nin_skip   (layers.py:155-156)
            a.tap[9] = GemmTap{h->E[G.node_skip], G.w_skip, 0, 0, -1, 2 * NF};
            a.slot_first[4] = 10;
            a.nslots = 4;
        }
        PostArgs p{items, part, nitems, NF, h->L, G.node_skip >= 0, 0, G.b_in, G.b_skip, nullptr, nullptr, nullptr, h->X[g]};
        if (!stage(a, m.und, p, POST_CONVIN)) {
            p.summed = gemm(a, m.und, G.b_in, &p);
            if (p.summed < 2) hipLaunchKernelGGL(k_post_grid<POST_CONVIN>, dim3(pblocks), dim3(256), 0, st, p);
        }
        GemmArgs b{};
        at_stage(15 + g);
        conv_taps(b, h->X[g], 2 * NF, G.w_out, 2 * NF, 2 * NF, 1);                     // conv_out   (layers.py:159)
        PostArgs q{items, part, nitems, 2 * NF, h->L, 0, 0, G.b_out, nullptr, h->R[G.node_in], h->R[G.node_out],
                   h->E[G.node_out], nullptr};
        if (!stage(b, m.und, q, POST_GATE)) {                                           // gate + residual (:160-163)
            q.summed = gemm(b, m.und, G.b_out, &q);
            if (q.summed < 2) hipLaunchKernelGGL(k_post_grid<POST_GATE>, dim3(pblocks), dim3(256), 0, st, q);
        }
    };
    auto dilated = [&](int d) {
        const ps_pixelcnn::Dil &D = h->dil[d];
        GemmArgs a{};
        at_stage(29 + d);
        conv_taps(a, h->R[D.node_in], R_LD, D.w, NF, NF, 2);                            // model.py:138,148
        PostArgs p{items, part, nitems, NF, h->L, 0, 0, D.b, nullptr, nullptr, h->R[D.node_out], h->E[D.node_out], nullptr};
        if (!stage(a, m.dil, p, POST_DIL)) {
            p.summed = gemm(a, m.dil, D.b, &p);
            if (p.summed < 2) hipLaunchKernelGGL(k_post_grid<POST_DIL>, dim3(pblocks), dim3(256), 0, st, p);
        }
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
    hipLaunchKernelGGL(k_logits_grid, dim3(nitems), dim3(256), 0, st, items, part, h->out_b, nitems, h->L, nchw ? 1 : 0,
                       logits);
}

// ------------------------------------------------------------------------------------------
// the 33-stage description and the tables the two roles of k_column read (built once per handle)
// ------------------------------------------------------------------------------------------
int build_stage_table(ps_pixelcnn *h)
{
    std::vector<StageDesc> st;
    std::vector<NbrWork> work;
    std::vector<NbrWorkTp> work_tp;
    std::vector<int> tp_items;   // work items per tile of every stage (k_column_tp)
    int fine_stages = 0;         // PS_TP_FINE_STAGES: tuning (one output tile per item in the first stages: measured no faster)
    if (const char *cc = getenv("PS_TP_FINE_STAGES")) fine_stages = std::max(0, atoi(cc));
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
                for (int cog = 0; cog < Co / 16; ++cog) work.push_back(NbrWork{w, in, s, half, cog, NG, Co, in_ld, dil, mask_kind});
        if (has_nbr) {   // throughput form: two output tiles per item where the stage has them
            const int step = s < fine_stages ? 16 : 32;
            for (int half = 0; half < 2; ++half)
                for (int o0 = 0; o0 < Co; o0 += step)
                    work_tp.push_back(NbrWorkTp{w, in, s, half, o0, (step == 32 && o0 + 32 <= Co) ? 2 : 1, NG, Co, in_ld, mask_kind - 1});
            tp_items.push_back(2 * ((Co + step - 1) / step));
        } else {
            tp_items.push_back(0);
        }
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
        push(D.w, nullptr, h->R[D.node_in], R_LD, 5, NF, 2, 2, 4, 1, IN_RAW, -1);
        prev = Prev{PRO_DIL, D.b, nullptr, 0, h->R[D.node_out], h->E[D.node_out], nullptr, -1};
        prev.save = (D.node_out >= 1 && D.node_out <= 7) ? D.node_out : -1;
    };
    gated(0); gated(1); dilated(0); gated(2); gated(3); dilated(1); gated(4); gated(5);
    gated(6); gated(7); dilated(2); gated(8); gated(9); gated(10); dilated(3);
    gated(11); gated(12); gated(13);
    push(h->out_w, nullptr, nullptr, 0, 5, NCLS, 1, 1, 0, 0, IN_ELU, -1);  // nin_out(elu(u)), prologue = last gate
    if ((int)st.size() != NST) return ps::fail(PS_ERR_STATE, "stage table has %d entries, expected %d", (int)st.size(), NST);
    for (int k = 0; k < NST; ++k) {  // the centre taps again in the chain role's [step][chain][4] layout
        StageDesc &d = st[k];
        float *wv = nullptr;
        if (k == NST - 1) {
            const int n = C1_OUT_STEPS * C1_THREADS * 4;
            if (int rc = dev_alloc(h, &wv, (size_t)n)) return rc;
            hipLaunchKernelGGL(k_pack_valu_out, dim3((n + 255) / 256), dim3(256), 0, 0, d.w, wv);
            d.nchain = C1_THREADS; d.nstep = C1_OUT_STEPS;
        } else {
            d.nchain = 5 * d.Co_pad + (d.w_skip ? 5 * NF : 0);
            d.nstep = 4 * (d.NG / 5);
            const int n = d.nstep * d.nchain * 4;
            if (int rc = dev_alloc(h, &wv, (size_t)n)) return rc;
            hipLaunchKernelGGL(k_pack_valu, dim3((n + 255) / 256), dim3(256), 0, 0,
                               d.w + (size_t)d.center_tap * d.NG * 16 * d.Co_pad, d.w_skip, d.Co_pad, d.nchain, d.nstep, wv);
        }
        d.wv = wv;
    }
    // throughput form: the stages' weights in that chain role's own order, and the unit tables of the four stage types
    std::vector<float *> wtp(NST, nullptr);
    std::vector<int> tptype(NST, TPT_DIL);
    for (int k = 0; k < NST - 1; ++k) {
        const StageDesc &d = st[k];
        const int type = d.NG == 5 ? TPT_DIL : d.Co_pad == 2 * NF ? TPT_CONVOUT : d.w_skip ? TPT_CONVIN_SKIP : TPT_CONVIN;
        tptype[k] = type;
        const int n = TP_WAVES * tpt_nu(type) * 2 * 64;
        if (int rc = dev_alloc(h, &wtp[k], (size_t)n * 4)) return rc;
        hipLaunchKernelGGL(k_pack_tp, dim3((n + 255) / 256), dim3(256), 0, 0, d.w + (size_t)d.center_tap * d.NG * 16 * d.Co_pad, d.w_skip,
                           d.Co_pad, d.NG, type, wtp[k]);
    }
    wtp[NST - 1] = wtp[NST - 2];   // nin_out has its own loop: the record only has to name loadable memory (requested, dropped)
    {
        std::vector<TpUnit> units((size_t)4 * TP_WAVES * TP_MAXU, TpUnit{0, 0, 0, 0});
        for (int type = 0; type < 4; ++type) {
            const int Co = type == TPT_CONVOUT ? 2 * NF : NF, total = tpt_units(type);
            for (int w = 0; w < TP_WAVES; ++w)
                for (int u = 0; u < TP_MAXU; ++u) {
                    const int n = std::min(w + TP_WAVES * u, total - 1);   // absent units name a valid one (never stored)
                    int ot, j; bool skip;
                    tp_unit_of(n, Co, ot, j, skip);
                    TpUnit &e = units[((size_t)type * TP_WAVES + w) * TP_MAXU + u];
                    e.b0 = (skip ? XB_SIZE : 0) + 4 * j * XB_LD;
                    e.b1 = e.b0 + (type == TPT_DIL ? 0 : 20 * XB_LD);
                    e.dst = (skip ? 5 * Co + j * NF : j * Co) + ot * 16;
                }
        }
        if (int rc = dev_alloc(h, &h->units_tp, units.size())) return rc;
        PS_HIP_CHECK(hipMemcpy(h->units_tp, units.data(), units.size() * sizeof(TpUnit), hipMemcpyHostToDevice));
    }
    PS_HIP_CHECK(hipDeviceSynchronize());
    {   // the chain role's control records
        std::vector<int> ctl((size_t)(NST + 1) * C1_CTL_DWORDS, 0);
        auto put_p = [&](int rec, int field, const void *ptr) { memcpy(&ctl[(size_t)rec * C1_CTL_DWORDS + field], &ptr, 8); };
        auto put_post = [&](int rec, const StageDesc &nx) {  // the post op feeding stage `nx`
            int *c = &ctl[(size_t)rec * C1_CTL_DWORDS];
            c[CTL_KIND] = nx.pro; c[CTL_HAS_SKIP] = nx.pro == PRO_CONVIN && nx.p_has_skip; c[CTL_IN_FORM] = nx.in_form;
            c[CTL_SAVE_SLOT] = nx.save_slot; c[CTL_SKIP_SLOT] = nx.skip_slot;
            put_p(rec, CTL_BIAS, nx.pbias); put_p(rec, CTL_BIAS2, nx.pbias2);
            put_p(rec, CTL_R, nx.outR); put_p(rec, CTL_E, nx.outE); put_p(rec, CTL_X, nx.outX);
        };
        put_post(0, st[0]);
        for (int k = 0; k < NST; ++k) {
            int *c = &ctl[(size_t)(1 + k) * C1_CTL_DWORDS];
            c[CTL_CO] = st[k].Co_pad; c[CTL_NCHAIN] = st[k].nchain; c[CTL_NG] = st[k].NG; c[CTL_NSTEP] = st[k].nstep;
            c[CTL_NBR_ITEMS] = st[k].has_nbr ? 2 * (st[k].Co_pad / 16) : 0;
            c[CTL_TP_ITEMS] = tp_items[k];
            put_p(1 + k, CTL_WV, st[k].wv);
            // the centre tap (and nin_skip) in the MFMA layout [c/4][o][4]; nin_out's weights are that layout already
            put_p(1 + k, CTL_WC, k == NST - 1 ? st[k].w : st[k].w + (size_t)st[k].center_tap * st[k].NG * 16 * st[k].Co_pad);
            put_p(1 + k, CTL_WS, st[k].w_skip);
            c[CTL_TP_TYPE] = tptype[k];
            put_p(1 + k, CTL_WTP, getenv("PS_TP_EXP_HOTW") ? wtp[1] : wtp[k]);   // (timing experiment: every stage streams the same 112 KB)
            if (k + 1 < NST) put_post(1 + k, st[k + 1]);
        }
        if (int rc = dev_alloc(h, &h->ctl1, ctl.size())) return rc;
        PS_HIP_CHECK(hipMemcpy(h->ctl1, ctl.data(), ctl.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    if (int rc = dev_alloc(h, &h->work, work.size())) return rc;
    PS_HIP_CHECK(hipMemcpy(h->work, work.data(), work.size() * sizeof(NbrWork), hipMemcpyHostToDevice));
    h->nwork = (int)work.size();
    h->col_wsplit = 0;
    while (h->col_wsplit < h->nwork && work[h->col_wsplit].stage < h->col_ahead) ++h->col_wsplit;   // (entries are stage-major)
    PS_REQUIRE(h->nwork <= NWORK_MAX, "pixelcnn: %d neighbour work entries exceed the staging table", h->nwork);
    if (int rc = dev_alloc(h, &h->work_tp, work_tp.size())) return rc;
    PS_HIP_CHECK(hipMemcpy(h->work_tp, work_tp.data(), work_tp.size() * sizeof(NbrWorkTp), hipMemcpyHostToDevice));
    h->nwork_tp = (int)work_tp.size();
    h->tp_wsplit = 0;
    while (h->tp_wsplit < h->nwork_tp && work_tp[h->tp_wsplit].stage < h->tp_ahead) ++h->tp_wsplit;   // (entries are stage-major)
    return PS_OK;
}

// `ncols` independent columns (records rec[0..ncols)): neighbour taps of every conv and the centre-tap chains + draw,
// in launches of at most col_cap columns.
// next_rec / next_ncols: the columns of the launch that FOLLOWS on this stream, when the caller knows it (a wavefront schedule):
// the throughput form computes their first stages' neighbour slots a launch ahead (nbr_role_tp).
void run_columns(ps_pixelcnn *h, const StepCtx *rec, int ncols, const int32_t *codes, ChainArgs ca, hipStream_t st,
                 const StepCtx *next_rec = nullptr, int next_ncols = 0)
{
    ca.ctl1 = h->ctl1; ca.nbr = h->nbr;
    ca.uinit_w = h->uinit_w; ca.uinit_b = h->uinit_b; ca.codes_in = codes;
    ca.out_b = h->out_b;
    ca.H = h->H; ca.W = h->W; ca.L = h->L; ca.col_stride = COL_CAP;
    ca.cnt = h->cnt; ca.err = h->err;
    if (const char *dbg = getenv("PS_COLUMN_DEBUG")) ca.debug = atoi(dbg);
    if (ncols >= h->tp_min_cols) {   // throughput form: 16-column chain tiles, up to TP_COL_CAP columns per launch
        TpArgs ta{};
        ta.units = h->units_tp;
        ta.work = h->work_tp; ta.nwork = h->nwork_tp;
        ta.done = h->done_tp; ta.split = h->tp_ahead;
        const size_t nbr_half_buf = (size_t)NST * 2 * TP_COL_CAP * NBR_LD, cnt_half_buf = tp_cnt_index(NST, 0);
        ta.ctl1 = h->ctl1; ta.uinit_w = h->uinit_w; ta.uinit_b = h->uinit_b; ta.codes_in = codes;
        ta.out_w = h->out_w; ta.out_b = h->out_b; ta.L = h->L;
        ta.codes = ca.codes; ta.region = ca.region; ta.forced = ca.forced; ta.uniforms = ca.uniforms;
        ta.out_logits = ca.out_logits; ta.step_logits = ca.step_logits; ta.temperature = ca.temperature;
        ta.err = h->err; ta.debug = ca.debug;
        ta.trace = h->tp_trace;
        static const int nbr_map = getenv("PS_TP_NBR_MAP") ? atoi(getenv("PS_TP_NBR_MAP")) : 0;
        ta.nbr_map = nbr_map;
        static const int trace_sel = getenv("PS_TP_TRACE_LAUNCH") ? atoi(getenv("PS_TP_TRACE_LAUNCH")) : -1;   // tuning: stamps of that launch of the run only
        const int cap = std::min(TP_COL_CAP, std::max(TP_COLS, (h->n_cus / 2) * TP_COLS));   // at least half of the CUs to the neighbour role
        const ColTaps *taps = h->taps + (rec - h->ctx);
        for (int done = 0; done < ncols; done += cap) {
            const int n = std::min(cap, ncols - done);
            const int tiles = (n + TP_COLS - 1) / TP_COLS;
            ta.taps = taps + done; ta.ctx = rec + done; ta.ncols = n; ta.tiles = tiles;
            // did the launch in front prepare this one?  then its slots of the stages [0, split) are in the buffers of `par`
            const bool prepared = h->tp_ahead > 0 && h->ahead_rec == rec + done && h->ahead_n == n;
            const int par = prepared ? h->ahead_parity : 0;
            ta.nbr = h->nbr_tp + par * nbr_half_buf; ta.cnt = h->cnt_tp + par * cnt_half_buf;
            ta.nbr_next = h->nbr_tp + (par ^ 1) * nbr_half_buf; ta.cnt_next = h->cnt_tp + (par ^ 1) * cnt_half_buf;
            ta.w_from = prepared ? h->tp_wsplit : 0;
            // and what follows this one: the rest of an oversized wavefront, or the caller's next wavefront if it takes this form
            const StepCtx *nrec = nullptr;
            int nn = 0;
            if (done + cap < ncols) { nrec = rec + done + cap; nn = std::min(cap, ncols - done - cap); }
            else if (next_rec && next_ncols >= h->tp_min_cols) { nrec = next_rec; nn = std::min(cap, next_ncols); }
            const bool ahead = h->tp_ahead > 0 && nrec != nullptr && !(ca.debug & 2);
            ta.w_upto = ahead ? h->tp_wsplit : 0;
            ta.taps_next = ahead ? h->taps + (nrec - h->ctx) : ta.taps;
            ta.ncols_next = ahead ? nn : 0;
            ta.tiles_next = ahead ? (nn + TP_COLS - 1) / TP_COLS : 1;
            ta.publish_upto = ahead ? h->tp_ahead : 0;
            if (ahead) h->done_total += (unsigned)tiles;
            ta.done_target = h->done_total;
            static const int exp_mode = getenv("PS_TP_AHEAD_EXP") ? atoi(getenv("PS_TP_AHEAD_EXP")) : 0;   // timing experiments (results invalid)
            if (exp_mode == 1) ta.done_target = 0;                       // look-ahead items do not wait for the chain tiles
            if (exp_mode == 2) { ta.w_upto = 0; ta.w_from = 0; }         // chain tiles publish, nobody looks ahead
            for (int t = 0; t < tiles; ++t) {
                if (!prepared) h->tile_uses_tp_lo[par][t] += 1;
                h->tile_uses_tp_hi[par][t] += 1;
            }
            for (int t = 0; t < TP_MAX_TILES; ++t) { ta.tile_uses_lo[t] = h->tile_uses_tp_lo[par][t]; ta.tile_uses_hi[t] = h->tile_uses_tp_hi[par][t]; }
            if (ahead) for (int t = 0; t < ta.tiles_next; ++t) h->tile_uses_tp_lo[par ^ 1][t] += 1;
            h->ahead_rec = ahead ? nrec : nullptr; h->ahead_n = nn; h->ahead_parity = par ^ 1;
            int grid;
            // The XCD-affine layout assumes a whole MI355X (SPX mode: 8 XCDs x 32 CUs, block b on XCD b % 8) or an even share of its
            // XCDs (a stream confined to compute units [0, 8 k): k per XCD, ps_stream_create_cu_range).  On a partition
            // (CPX: 32 CUs = one XCD per device) or any other CU count that mapping means nothing: the plain layout is used --
            // neighbour blocks first in the grid, so they are dispatched ahead of the chain tiles that wait for them.  Either way
            // the grid holds at most one workgroup per CU and the waits are bounded (40000 polls with s_sleep, tens of ms):
            // kernels of OTHER streams that hold CUs for a while (bench.py / driver.py overlap the next batch's ~2 ms of splat
            // kernels with this run) delay a launch, they cannot starve it past the bound.
            const int rows = h->n_cus / 8;   // CUs per XCD
            if (h->tp_xcds != 0 && h->xcd_even && tiles <= 4 * rows) {
                const int cx = h->tp_xcds > 0 ? std::max(h->tp_xcds, (tiles + rows - 1) / rows) : (tiles + rows - 1) / rows;
                ta.chain_xcds = std::min(cx, 7);
                const int spare = ta.chain_xcds * rows - tiles;
                static const int grid_rows = getenv("PS_TP_GRID_ROWS") ? atoi(getenv("PS_TP_GRID_ROWS")) : 0;   // tuning: fewer neighbour workgroups
                const int use_rows = grid_rows > 0 ? std::min(rows, std::max(grid_rows, (tiles + ta.chain_xcds - 1) / ta.chain_xcds)) : rows;
                ta.nbr_wgs = (8 - ta.chain_xcds) * use_rows;
                ta.fill_nbr = -1;
                if (h->tp_fill && spare > 0 && use_rows == rows) { ta.fill_nbr = ta.nbr_wgs; ta.nbr_wgs += spare; }
                grid = use_rows * 8;
            } else {
                ta.chain_xcds = 0; ta.fill_nbr = -1;
                ta.nbr_wgs = std::max(1, std::min(h->n_cus - tiles, (h->nwork_tp * tiles + TP_WAVES - 1) / TP_WAVES));
                grid = ta.nbr_wgs + tiles;
            }
            ta.trace = (trace_sel < 0 || trace_sel == h->tp_launch_no) ? h->tp_trace : nullptr;
            h->tp_launch_no += 1;
            timed(h, st, TAG_CHAIN, [&]() { hipLaunchKernelGGL(k_column_tp, dim3(grid), dim3(TP_THREADS), 0, st, ta); });
        }
        return;
    }
    const size_t nbr_half_col = (size_t)NST * 2 * COL_CAP * NBR_LD, cnt_half_col = cnt_index(NST, 0);
    for (int done = 0; done < ncols; done += h->col_cap) {
        const int n = std::min(h->col_cap, ncols - done);
        const int tiles = (n + 15) / 16;
        // chain workgroups on XCDs 0 .. cx-1 of the first rows of 8 blocks, neighbour workgroups everywhere else, 256
        // blocks at most (one per CU, all resident)
        // (the launch holds one workgroup per CU at most: every workgroup is resident, which the in-launch waits rest on;
        // per_xcd = CUs per XCD of THIS device, 32 on a whole MI355X)
        const int per_xcd = h->n_cus / 8;
        const int cx = h->chain_xcds > 0 ? std::min(8, std::max(h->chain_xcds, (n + per_xcd - 1) / per_xcd)) : std::min(4, (n + per_xcd - 1) / per_xcd);
        const int chain_rows = (n + cx - 1) / cx;
        const int nbr_cus = chain_rows * (8 - cx) + (per_xcd - chain_rows) * 8;
        // the look-ahead, from one latency-form launch to the next (as in the throughput form above): was this launch prepared, and
        // what follows it -- the rest of an oversized wavefront or the caller's next wavefront, if that takes this form too
        const bool prepared = h->col_ahead > 0 && h->col_ahead_rec == rec + done && h->col_ahead_n == n;
        const int par = prepared ? h->col_ahead_parity : 0;
        const StepCtx *nrec = nullptr;
        int nn = 0;
        if (done + h->col_cap < ncols) { nrec = rec + done + h->col_cap; nn = std::min(h->col_cap, ncols - done - h->col_cap); }
        else if (next_rec && next_ncols > 0 && next_ncols < h->tp_min_cols) { nrec = next_rec; nn = std::min(h->col_cap, next_ncols); }
        const bool ahead = h->col_ahead > 0 && nrec != nullptr && !(ca.debug & 2);
        bool la = prepared || ahead;
        for (int t = 0; t < tiles && !la; ++t)   // (k_column keeps ONE use count per tile for all stages: should a prepared launch
            la = h->col_uses_lo[0][t] != h->col_uses_hi[0][t];   // ever not have followed, the two-count form takes over)
        static const int col_exp = getenv("PS_COL_AHEAD_EXP") ? atoi(getenv("PS_COL_AHEAD_EXP")) : 0;   // timing experiments (results invalid)
        // 2: the columns publish, nobody looks ahead; 3: look-ahead items, nobody publishes or waits
        const int w_from = prepared && col_exp != 2 ? h->col_wsplit : 0, w_upto = ahead && col_exp != 2 ? h->col_wsplit : 0;
        const int tiles_next = ahead ? (nn + 15) / 16 : 1;
        const int nitems = (h->nwork - w_from) * tiles + w_upto * tiles_next;
        const int groups = h->force_groups ? h->force_groups : (nitems > 2 * nbr_cus ? 4 : 2);
        const int nbr_wgs = std::min(nbr_cus, (nitems + groups - 1) / groups);
        NbrArgs na{h->work, rec + done, h->nbr + par * nbr_half_col, h->nwork, h->H, h->W, h->L, n, COL_CAP, tiles, cx,
                   h->cnt + par * cnt_half_col, nbr_wgs, groups, ca.debug, h->err};
        na.w_from = w_from; na.w_upto = w_upto;
        na.ctx_next = ahead ? nrec : rec + done; na.ncols_next = ahead ? nn : 0; na.tiles_next = tiles_next;
        na.nbr_next = h->nbr + (par ^ 1) * nbr_half_col; na.cnt_next = h->cnt + (par ^ 1) * cnt_half_col;
        na.done = h->done_col; na.split = std::max(1, h->col_ahead);
        if (ahead) h->done_col_total += (unsigned)n;   // (every column publishes once per stage)
        na.done_target = col_exp == 3 ? 0u : h->done_col_total;
        ca.ctx = rec + done; ca.ncols = n;
        ca.nbr = na.nbr; ca.cnt = na.cnt;
        const int in_chain_rows = chain_rows * (8 - cx);
        const int rows = nbr_wgs <= in_chain_rows ? chain_rows : chain_rows + (nbr_wgs - in_chain_rows + 7) / 8;
        if (la) {
            for (int t = 0; t < tiles; ++t) {
                if (!prepared) h->col_uses_lo[par][t] += 1;
                h->col_uses_hi[par][t] += 1;
            }
            for (int t = 0; t < MAX_TILES; ++t) { ca.uses_lo[t] = h->col_uses_lo[par][t]; ca.uses_hi[t] = h->col_uses_hi[par][t]; }
            if (ahead) for (int t = 0; t < tiles_next; ++t) h->col_uses_lo[par ^ 1][t] += 1;
            ca.la_split = h->col_ahead; ca.done = h->done_col; ca.publish_upto = ahead && col_exp != 3 ? h->col_ahead : 0;
            h->col_ahead_rec = ahead ? nrec : nullptr; h->col_ahead_n = nn; h->col_ahead_parity = par ^ 1;
            timed(h, st, TAG_CHAIN, [&]() { hipLaunchKernelGGL(k_column_la, dim3(rows * 8), dim3(C1_THREADS), 0, st, na, ca); });
        } else {   // a launch nobody prepared and that prepares nobody (a walk position by position): one set of use counts for all stages
            h->col_ahead_rec = nullptr;
            for (int t = 0; t < tiles; ++t) { h->col_uses_lo[0][t] += 1; h->col_uses_hi[0][t] += 1; }
            for (int t = 0; t < MAX_TILES; ++t) ca.tile_uses[t] = h->col_uses_hi[0][t];
            timed(h, st, TAG_CHAIN, [&]() { hipLaunchKernelGGL(k_column, dim3(rows * 8), dim3(C1_THREADS), 0, st, na, ca); });
        }
    }
}

CtxArgs make_ctx_args(ps_pixelcnn *h, const int32_t *order, const Masks &m, int F)
{
    CtxArgs cx{};
    cx.ctx = h->ctx; cx.taps = h->taps; cx.order = order;
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

int ps_order_masks_f32(const int32_t *order_loc, int F, int H, int W, float *mask_init, float *mask_undilated, float *mask_dilated,
                       int32_t *status, void *stream)
{
    PS_REQUIRE(order_loc && mask_init && mask_undilated && mask_dilated, "order_masks: null pointer");
    PS_REQUIRE(F > 0 && H > 0 && W > 0 && (size_t)H * W * sizeof(int) <= 64 * 1024, "order_masks: bad sizes");
    hipLaunchKernelGGL(k_order_masks, dim3(F), dim3(256), (size_t)H * W * sizeof(int), (hipStream_t)stream, order_loc, H, W, mask_init,
                       mask_undilated, mask_dilated, status);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_pixelcnn_create(const float *const *params, int n_params, int H, int W, int max_frames, ps_pixelcnn **out)
{
    PS_REQUIRE(params && out, "pixelcnn_create: null pointer");
    PS_REQUIRE(n_params == PS_PIXELCNN_NUM_PARAMS, "pixelcnn_create: expected %d tensors, got %d",
               PS_PIXELCNN_NUM_PARAMS, n_params);
    PS_REQUIRE(H > 0 && W > 0 && max_frames > 0, "pixelcnn_create: bad sizes");
    for (int i = 0; i < n_params; ++i) PS_REQUIRE(params[i], "pixelcnn_create: tensor %d is null", i);
    ps_pixelcnn *h = new ps_pixelcnn();
    h->H = H; h->W = W; h->L = H * W; h->maxF = max_frames;
    {   // device size first: the column launches keep one workgroup per CU, all resident
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            h->n_cus = cus;
        h->xcd_even = h->n_cus == 8 * 32;
        if (h->n_cus % 8 != 0 || h->n_cus < 16) {
            ps::fail(PS_ERR_STATE, "pixelcnn_create: %d compute units -- the column launches lay their workgroups out over 8 XCDs", h->n_cus);
            delete h;
            return PS_ERR_STATE;
        }
        h->col_cap = std::min(COL_CAP, (h->n_cus / 8) * 4);   // at most four XCDs of chains, the rest for the neighbour role
    }
    if (const char *cc = getenv("PS_COL_CAP")) h->col_cap = std::max(1, std::min(h->col_cap, atoi(cc)));
    if (const char *cc = getenv("PS_CHAIN_XCDS")) h->chain_xcds = std::max(0, std::min(8, atoi(cc)));
    if (const char *cc = getenv("PS_NBR_GROUPS")) h->force_groups = std::max(0, std::min(NBR_MAX_GROUPS, atoi(cc)));
    if (const char *cc = getenv("PS_TP_MIN_COLS")) h->tp_min_cols = std::max(1, atoi(cc));
    if (const char *cc = getenv("PS_TP_XCDS")) h->tp_xcds = atoi(cc);
    if (const char *cc = getenv("PS_TP_FILL")) h->tp_fill = atoi(cc);
    if (const char *cc = getenv("PS_TP_AHEAD")) h->tp_ahead = std::min(NST - 1, std::max(0, atoi(cc)));
    if (const char *cc = getenv("PS_COL_AHEAD")) h->col_ahead = std::min(NST - 4, std::max(0, atoi(cc)));
#if PS_TP_CHAIN2
    h->tp_ahead = 0;
#endif

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
        if ((rc = dev_alloc(h, &h->R[n], locs * R_LD))) return fail_out(rc);
        if ((rc = dev_alloc(h, &h->E[n], locs * 2 * NF))) return fail_out(rc);
    }
    for (int g = 0; g < NGATED; ++g)
        if ((rc = dev_alloc(h, &h->X[g], locs * 2 * NF))) return fail_out(rc);
    size_t pfloats = (size_t)4 * locs * 2 * NF;
    if (locs * NCLS > pfloats) pfloats = locs * NCLS;
    if ((rc = dev_alloc(h, &h->partial, pfloats))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->col_logits, (size_t)max_frames * NCLS))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->nbr, (size_t)2 * NST * 2 * COL_CAP * NBR_LD))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->done_col, (size_t)NST * CNT_PAD))) return fail_out(rc);
    if (hipMemset(h->done_col, 0, (size_t)NST * CNT_PAD * sizeof(unsigned)) != hipSuccess) {
        ps::fail(PS_ERR_HIP, "pixelcnn_create: hipMemset failed");
        return fail_out(PS_ERR_HIP);
    }
    if ((rc = dev_alloc(h, &h->ctx, locs))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->pstart, (size_t)N_EVAL * max_frames))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->taps, locs))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->nbr_tp, (size_t)2 * NST * 2 * TP_COL_CAP * NBR_LD))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->cnt_tp, 2 * tp_cnt_index(NST, 0)))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->done_tp, (size_t)NST * CNT_PAD))) return fail_out(rc);
    if (hipMemset(h->cnt_tp, 0, 2 * tp_cnt_index(NST, 0) * sizeof(unsigned)) != hipSuccess ||
        hipMemset(h->done_tp, 0, (size_t)NST * CNT_PAD * sizeof(unsigned)) != hipSuccess) {
        ps::fail(PS_ERR_HIP, "pixelcnn_create: hipMemset failed");
        return fail_out(PS_ERR_HIP);
    }
    if ((rc = dev_alloc(h, &h->cnt, 2 * cnt_index(NST, 0)))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->err, 1))) return fail_out(rc);
    if (hipMemset(h->cnt, 0, 2 * cnt_index(NST, 0) * sizeof(unsigned)) != hipSuccess || hipMemset(h->err, 0, sizeof(int)) != hipSuccess) {
        ps::fail(PS_ERR_HIP, "pixelcnn_create: hipMemset failed");
        return fail_out(PS_ERR_HIP);
    }
    if ((rc = build_stage_table(h))) return fail_out(rc);
    *out = h;
    return PS_OK;
}

void ps_pixelcnn_destroy(ps_pixelcnn *h)
{
    if (!h) return;
    for (void *p : h->allocs) (void)hipFree(p);
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
    if (step == first_step) run_grid(h, F, codes, m, nullptr, false, st, order, first_step);
    ChainArgs ca{};
    hipLaunchKernelGGL(k_ctx_build, dim3(F), dim3(32), 0, st, make_ctx_args(h, order, m, F), (const int32_t *)nullptr, F, step,
                       h->H, h->W, h->err);
    ca.step_logits = logits;
    ca.temperature = 1.0f;
    run_columns(h, h->ctx, F, codes, ca, st);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

// The AR run: whole-grid pass over the observed prefix, then the remaining columns -- wavefront by wavefront when the
// caller brings a schedule (ps_ar_wavefronts), else position by position (one column per frame and launch).
enum { AR_PREFIX = 1, AR_COLUMNS = 2 };
static int ar_run_impl(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region,
                       const float *mask_init, const float *mask_undilated, const float *mask_dilated,
                       const int32_t *forced, const float *uniforms, float temperature, int F, int first_step,
                       const int32_t *wave_cols, const int32_t *wave_start, int n_waves, float *out_logits, void *stream,
                       int phases = AR_PREFIX | AR_COLUMNS, int f0 = 0, int nf = -1)
{
    if (int rc = check_handle(h, F)) return rc;
    if (nf < 0) nf = F;
    PS_REQUIRE(codes && order && sample_region && mask_init && mask_undilated && mask_dilated, "pixelcnn_ar_run: null pointer");
    PS_REQUIRE(f0 >= 0 && nf >= 0 && f0 + nf <= F, "pixelcnn_ar_prefix: frames [%d, %d) outside the run's %d", f0, f0 + nf, F);
    PS_REQUIRE(!(phases & AR_COLUMNS) || (forced != nullptr) != (uniforms != nullptr), "pixelcnn_ar_run: give exactly one of forced / uniforms");
    PS_REQUIRE(first_step >= 0 && first_step <= h->L, "pixelcnn_ar_run: first_step out of range");
    PS_REQUIRE(temperature > 0.0f, "pixelcnn_ar_run: temperature must be > 0");
    const int nsteps = h->L - first_step;
    if (wave_cols) {
        PS_REQUIRE(wave_start && n_waves >= 0 && wave_start[0] == 0, "pixelcnn_ar_run_waves: bad schedule");
        for (int w = 0; w < n_waves; ++w)
            PS_REQUIRE(wave_start[w + 1] >= wave_start[w], "pixelcnn_ar_run_waves: wave_start must not decrease");
        PS_REQUIRE(wave_start[n_waves] == F * nsteps, "pixelcnn_ar_run_waves: the schedule holds %d columns, the run has %d",
                   wave_start[n_waves], F * nsteps);
    }
    hipStream_t st = (hipStream_t)stream;
    const Masks m{mask_init, mask_undilated, mask_dilated};
    if (phases & AR_PREFIX) {   // frames [f0, f0 + nf): sampled codes masked out, whole-grid pass over the observed prefix
        const size_t n = (size_t)nf * h->L, off = (size_t)f0 * h->L;
        if (n > 0) hipLaunchKernelGGL(k_mask_codes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, codes + off, sample_region + off, n);
        // whole-grid pass: exact for every location that precedes the first sampled one; with out_logits it also
        // yields their logits, by location (the walked positions are overwritten by the column steps)
        run_grid(h, F, codes, m, out_logits, false, st, order, first_step, f0, nf);
        PS_LAUNCH_CHECK();
    }
    if (!(phases & AR_COLUMNS)) return PS_OK;
    ChainArgs ca{};
    ca.codes = codes; ca.region = sample_region; ca.forced = forced; ca.uniforms = uniforms;
    ca.out_logits = out_logits; ca.temperature = temperature;
    const int total = F * nsteps;
    if (total > 0)
        hipLaunchKernelGGL(k_ctx_build, dim3(total), dim3(32), 0, st, make_ctx_args(h, order, m, F), wave_cols, total, first_step,
                           h->H, h->W, h->err);
    PS_LAUNCH_CHECK();
    // launches are enqueued eagerly: the host stays far ahead of the GPU (a hipGraph replay was measured slower, and
    // the completion-counter target changes with every launch anyway)
    h->tp_launch_no = 0;
    if (wave_cols) {
        for (int w = 0; w < n_waves; ++w) {
            if (wave_start[w + 1] <= wave_start[w]) continue;
            int nx = w + 1;
            while (nx < n_waves && wave_start[nx + 1] <= wave_start[nx]) ++nx;
            const int nnext = nx < n_waves ? wave_start[nx + 1] - wave_start[nx] : 0;
            run_columns(h, h->ctx + wave_start[w], wave_start[w + 1] - wave_start[w], codes, ca, st,
                        nnext > 0 ? h->ctx + wave_start[nx] : nullptr, nnext);
        }
    } else {
        for (int sidx = 0; sidx < nsteps; ++sidx) run_columns(h, h->ctx + (size_t)sidx * F, F, codes, ca, st);
    }
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_pixelcnn_ar_run(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region,
                       const float *mask_init, const float *mask_undilated, const float *mask_dilated,
                       const int32_t *forced, const float *uniforms, float temperature, int F, int first_step,
                       float *out_logits, void *stream)
{
    return ar_run_impl(h, codes, order, sample_region, mask_init, mask_undilated, mask_dilated, forced, uniforms, temperature, F,
                       first_step, nullptr, nullptr, 0, out_logits, stream);
}

int ps_pixelcnn_ar_run_waves(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region,
                             const float *mask_init, const float *mask_undilated, const float *mask_dilated,
                             const int32_t *forced, const float *uniforms, float temperature, int F, int first_step,
                             const int32_t *wave_cols, const int32_t *wave_start, int n_waves, float *out_logits,
                             void *stream)
{
    PS_REQUIRE(wave_cols && wave_start, "pixelcnn_ar_run_waves: null schedule");
    return ar_run_impl(h, codes, order, sample_region, mask_init, mask_undilated, mask_dilated, forced, uniforms, temperature, F,
                       first_step, wave_cols, wave_start, n_waves, out_logits, stream);
}

int ps_pixelcnn_ar_prefix(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region, const float *mask_init,
                          const float *mask_undilated, const float *mask_dilated, int F, int first_step, int frame_begin, int frame_end,
                          void *stream)
{
    return ar_run_impl(h, codes, order, sample_region, mask_init, mask_undilated, mask_dilated, nullptr, nullptr, 1.0f, F, first_step, nullptr,
                       nullptr, 0, nullptr, stream, AR_PREFIX, frame_begin, frame_end - frame_begin);
}

int ps_pixelcnn_ar_columns(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region, const float *mask_init,
                           const float *mask_undilated, const float *mask_dilated, const int32_t *forced, const float *uniforms,
                           float temperature, int F, int first_step, const int32_t *wave_cols, const int32_t *wave_start, int n_waves,
                           void *stream)
{
    PS_REQUIRE(wave_cols && wave_start, "pixelcnn_ar_columns: the wavefront schedule is required");
    return ar_run_impl(h, codes, order, sample_region, mask_init, mask_undilated, mask_dilated, forced, uniforms, temperature, F, first_step,
                       wave_cols, wave_start, n_waves, nullptr, stream, AR_COLUMNS);
}

int ps_pixelcnn_set_compute_units(ps_pixelcnn *h, int n_cus)
{
    PS_REQUIRE(h, "pixelcnn: null handle");
    int dev = 0, cus = 0;
    PS_HIP_CHECK(hipGetDevice(&dev));
    PS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (n_cus <= 0) n_cus = cus;
    PS_REQUIRE(n_cus % 8 == 0 && n_cus >= 16 && n_cus <= cus, "pixelcnn_set_compute_units: %d compute units (a multiple of 8 in [16, %d])", n_cus, cus);
    h->n_cus = n_cus;
    h->xcd_even = cus == 8 * 32 && n_cus % 8 == 0;
    h->col_cap = std::min(COL_CAP, (h->n_cus / 8) * 4);
    return PS_OK;
}

int ps_stream_create_cu_range(int first_cu, int n_cus, void **stream)
{
    PS_REQUIRE(stream, "stream_create_cu_range: null pointer");
    int dev = 0, cus = 0;
    PS_HIP_CHECK(hipGetDevice(&dev));
    PS_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    PS_REQUIRE(first_cu >= 0 && n_cus > 0 && first_cu + n_cus <= cus && cus <= 1024, "stream_create_cu_range: compute units [%d, %d) of %d",
               first_cu, first_cu + n_cus, cus);
    uint32_t mask[32] = {};
    for (int c = first_cu; c < first_cu + n_cus; ++c) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t st = nullptr;
    PS_HIP_CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t)((cus + 31) / 32), mask));
    *stream = (void *)st;
    return PS_OK;
}

int ps_stream_destroy(void *stream)
{
    if (stream) PS_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return PS_OK;
}

int ps_pixelcnn_time_ar_run_waves(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region,
                                  const float *mask_init, const float *mask_undilated, const float *mask_dilated,
                                  const float *uniforms, float temperature, int F, int first_step, const int32_t *wave_cols,
                                  const int32_t *wave_start, int n_waves, int *launches, float *total_ms,
                                  double *flops_per_column, void *stream)
{
    PS_REQUIRE(h && launches && total_ms, "pixelcnn_time_ar_run_waves: null pointer");
    std::vector<ps_pixelcnn::ProfRec> recs;
    h->prof = &recs;
    const int rc = ar_run_impl(h, codes, order, sample_region, mask_init, mask_undilated, mask_dilated, nullptr, uniforms,
                               temperature, F, first_step, wave_cols, wave_start, n_waves, nullptr, stream);
    h->prof = nullptr;
    (void)hipStreamSynchronize((hipStream_t)stream);
    *launches = 0;
    *total_ms = 0.0f;
    for (auto &r : recs) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        if (r.tag == TAG_CHAIN) { *launches += 1; *total_ms += ms; }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    if (flops_per_column) *flops_per_column = h->flops_nbr + h->flops_chain;
    return rc;
}

// tuning / debugging aid (tools/tp_debug.py): device address of an activation cache -- what 0: R[idx] (raw u of node idx,
// row stride 96), 1: E[idx] (concat_elu(u), 160), 2: X[idx] (inside gated resnet idx, 160); rows are (frame * L + location)
void *ps_pixelcnn_debug_cache(ps_pixelcnn *h, int what, int idx)
{
    if (!h) return nullptr;
    if (what == 0 && idx >= 0 && idx < NNODE) return h->R[idx];
    if (what == 1 && idx >= 0 && idx < NNODE) return h->E[idx];
    if (what == 2 && idx >= 0 && idx < NGATED) return h->X[idx];
    if (what == 8) return h->done_col; // tuning: the latency form's `done` counters [NST][CNT_PAD] (dword 1 of a row: look-ahead waits that had to wait)
    if (what == 3) return h->nbr_tp;   // neighbour slots of the last throughput launch [NST][2][1024][160]
    if (what == 5) return h->pstart;   // (33, F) int32 of the last AR run's prefix pass: first rank evaluated per stage and frame
#ifdef PS_WG_TRACE_BUILD
    if (what == 6) { void *p = nullptr; return hipGetSymbolAddress(&p, HIP_SYMBOL(g_wg_trace)) == hipSuccess ? p : nullptr; }
    if (what == 7) { void *p = nullptr; return hipGetSymbolAddress(&p, HIP_SYMBOL(g_wg_span)) == hipSuccess ? p : nullptr; }
#endif
    if (what == 4) {                   // tuning builds: allocate / return the stamp buffer [NST][8] of 64-bit clocks
        if (!h->tp_trace && dev_alloc(h, &h->tp_trace, (size_t)NST * 8) == PS_OK) (void)hipMemset(h->tp_trace, 0, NST * 8 * 8);
        return h->tp_trace;
    }
    return nullptr;
}

int ps_pixelcnn_status(ps_pixelcnn *h, void *stream)
{
    PS_REQUIRE(h, "pixelcnn_status: null handle");
    PS_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    int flag = 0;
    PS_HIP_CHECK(hipMemcpy(&flag, h->err, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) PS_HIP_CHECK(hipMemset(h->err, 0, sizeof(int)));  // reported once; the handle stays usable
    if (flag == 2) return ps::fail(PS_ERR_STATE, "pixelcnn: the wavefront schedule names columns outside this run");
    if (flag) return ps::fail(PS_ERR_STATE, "pixelcnn: a bounded in-launch wait ran out (neighbour slots never arrived)");
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
    ca.step_logits = h->col_logits;
    ca.temperature = 1.0f;
    hipLaunchKernelGGL(k_ctx_build, dim3(F), dim3(32), 0, st, make_ctx_args(h, order, Masks{mask_init, mask_undilated, mask_dilated}, F),
                       (const int32_t *)nullptr, F, step, h->H, h->W, h->err);
    run_columns(h, h->ctx, F, codes, ca, st);  // untimed warm-up
    if (const char *tp = getenv("PS_CHAIN_TRACE")) {  // tuning aid: per-stage shader-clock stamps of workgroup 0
        unsigned long long *d = nullptr;
        if (hipMalloc(&d, NST * 10 * 8) == hipSuccess) {
            (void)hipMemsetAsync(d, 0, NST * 10 * 8, st);
            ChainArgs ct = ca;
            ct.trace = d;
            run_columns(h, h->ctx, F, codes, ct, st);
            std::vector<unsigned long long> hst(NST * 10);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(hst.data(), d, NST * 10 * 8, hipMemcpyDeviceToHost);
            if (FILE *fp = fopen(tp, "w")) {
                for (int s2 = 0; s2 < NST - 1; ++s2)
                    fprintf(fp, "%d %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", s2, hst[s2 * 10], hst[s2 * 10 + 5],
                            hst[s2 * 10 + 6], hst[s2 * 10 + 7], hst[s2 * 10 + 8], hst[s2 * 10 + 1], hst[s2 * 10 + 2],
                            hst[s2 * 10 + 3], hst[s2 * 10 + 4]);
                fprintf(fp, "# marks (role start, u0 done, stages done, draw done): %llu %llu %llu %llu\n", hst[9], hst[19], hst[29], hst[39]);
                fclose(fp);
            }
            (void)hipFree(d);
        }
    }
    h->prof = &recs;
    for (int r = 0; r < reps; ++r) run_columns(h, h->ctx, F, codes, ca, st);
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
    a.items = ItemMap{nullptr, L, nullptr, 0};
    a.H = H; a.W = W; a.L = L; a.nitems = B * L; a.mask = mask; a.mask_fstride = mask_batch_stride;
    a.partial = partial; a.tiles_per_block = 2;
    const int tiles = (a.nitems + 15) / 16;
    launch_gemm(a, (tiles + 1) / 2, st);
    hipLaunchKernelGGL(k_reduce_nchw, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, partial, bias, B, Co, Cop, L, y);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // extern "C"
