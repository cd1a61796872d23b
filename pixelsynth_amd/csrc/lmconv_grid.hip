// lmconv_grid.hip -- whole-grid mode of the locally-masked PixelCNN engine: items = (frame, location) pairs of the full grid or of the
// observed prefix of every generation order.  The reference-faithful OurPixelCNN.forward (models/lmconv/model.py:110-155) and the
// cache build an AR run starts from; also the generic NCHW entry point of one locally masked convolution
// (models/lmconv/locally_masked_convolution.py:11-50).
//   k_gemm        one wave = 16 items x 32 output channels of one split-K slot (small launches)
//   k_gemm_wg     a workgroup = 16/32 items x ALL output channels, input rows shared through LDS, post op fused (large launches)
//   k_post_grid / k_uinit_grid / k_logits_grid, k_prefix_starts (which prefix items anybody reads)
#include "lmconv_handle.h"

namespace pslm {

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
// Items of a whole-grid pass: every (frame, location) pair, or -- with a generation order -- only the first
// `npre` locations of each frame in that order (the observed prefix an AR run starts from; later locations
// are produced by the column steps, and no earlier location ever reads them).
struct ItemMap {
    const int32_t *order;  // (F, L) location by rank, or null = all L locations in raster order
    int npre;              // locations per frame
    const int32_t *start;  // (F) or null: ranks below start[f] are NOT evaluated at this stage -- nothing reads them
                           // (k_prefix_starts); only with an order
    int f0;                // first frame of the pass (a pass over frames [f0, f0 + n): item 0 is rank 0 of frame f0)
    const int32_t *perm;   // or null: position p of the products' item list holds item perm[p] -- the items grouped by their set of
                           // open taps (k_perm_*), so that a tile of 16 / 32 items shares its taps; the post ops walk the items as they are
    const int2 *permq;     // the same list as (item, location) pairs: one load instead of the chain position -> item -> order -> location
    const int32_t *end;    // (F) or null: frame f's prefix ends at rank end[f] <= npre (per-frame prefixes: the ranks from there on are
                           // its columns'); only with an order
};
// item at position `pos` of the products' item list, -1 past its end
__device__ __forceinline__ int item_at(const ItemMap &m, int pos, int nitems)
{
    if (pos >= nitems) return -1;
    return m.perm ? m.perm[pos] : pos;
}
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
    if (!m.start && !m.end) return true;
    const int fl = item / m.npre, r = item - fl * m.npre;
    return (!m.start || r >= m.start[m.f0 + fl]) && (!m.end || r < m.end[m.f0 + fl]);
}

struct GemmArgs {
    GemmTap tap[MAX_TAPS];
    ItemMap items;
    int slot_first[5];  // slot s covers taps [slot_first[s], slot_first[s+1])
    int nslots, Cin, Co_pad, H, W, L, nitems, tiles_per_block;
    int nx, ny, tpx;    // launch geometry (launch_gemm): channel blocks, item blocks, item blocks per XCD
#ifdef PS_WG_TRACE_BUILD
    int trace_on;       // tuning builds: this launch writes the stamps
#endif
    int zgrid;          // slots along the grid (nslots), or 1 = every wave walks all slots
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
template <int T>
__device__ __forceinline__ void gemm_tiles(const GemmArgs &a, int o0, int z0, int z1, int first_tile)
{
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    const int ngroups = a.Cin >> 4;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int tt = 0; tt < a.tiles_per_block; ++tt) {
        const int tile = first_tile + tt;
        if (tile * 16 >= a.nitems) break;
        const int item = item_at(a.items, tile * 16 + i, a.nitems);
        const bool valid = item >= 0 && item_wanted(a.items, item);
        if (!__any(valid)) continue;   // a tile nobody reads at this stage
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
                for (int j = 0; j < 5; ++j) bv[j] = *(const f32x4 *)(src + 16 * (g + j));
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
                    for (int j = 0; j < 5; ++j) av[j] = wload(g + j, u);
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
        if (valid) {
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
// a.summed), `S` at the same pair of its nin_skip slot; `loc` = the item's cache row, `rin` = the residual input of a gate
// (R[node_in] at the location), `b2` = the nin_skip bias -- both fetched by the caller, who can do so for several items at once.
template <int KIND>
__device__ __forceinline__ void post_item_at(const PostArgs &a, size_t loc, int lane, const float *Pbase, size_t ss, const float *Sbase,
                                             const f32x2 &rin, const f32x2 &b2)
{
    const bool own = lane < PONO_LANES;
    const int c = own ? 2 * lane : 0;
    const float *P = Pbase + c;
    const f32x2 zero = {0.0f, 0.0f};
    auto ld = [](const float *p) { return *(const f32x2 *)p; };
    f32x2 g = zero, skip = zero;
    const f32x2 y = a.summed ? ld(P + SLOT_NA * ss)
                             : slot_sum2(ld(a.bias + c), ld(P + SLOT_NA * ss), ld(P + SLOT_C * ss), ld(P + SLOT_NB * ss));
    if (KIND == POST_GATE)
        g = a.summed ? ld(P + SLOT_NA * ss + NF)
                     : slot_sum2(ld(a.bias + NF + c), ld(P + SLOT_NA * ss + NF), ld(P + SLOT_C * ss + NF), ld(P + SLOT_NB * ss + NF));
    if (KIND == POST_CONVIN && a.has_skip) skip = ld(Sbase + c) + b2;
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
template <int KIND>
__device__ __forceinline__ void post_item(const PostArgs &a, int item, int lane, const float *Pbase, size_t ss, const float *Sbase)
{
    int f, q;
    item_loc(a.items, item, a.L, f, q);
    const size_t loc = (size_t)f * a.L + q;
    const int c = lane < PONO_LANES ? 2 * lane : 0;
    const f32x2 zero = {0.0f, 0.0f};
    const f32x2 rin = KIND == POST_GATE ? *(const f32x2 *)(a.Rin + loc * R_LD + c) : zero;
    const f32x2 b2 = KIND == POST_CONVIN && a.has_skip ? *(const f32x2 *)(a.bias2 + c) : zero;
    post_item_at<KIND>(a, loc, lane, Pbase, ss, Sbase, rin, b2);
}

// FOUR items per wave pass (round 6): the workgroup forms run the post op of 16 to 64 items per workgroup, and one item per pass
// keeps 40 of 64 lanes busy and makes a wave's 4 to 16 items one dependent chain after the other (two cross-lane reductions, an
// exponential, a reciprocal square root each).  Here item j of the pass sits in row j of the wave (lanes 16 j .. 16 j + 15) and lane l
// of the row carries the channel pairs l, l + 16 and l + 32 (< 40) -- the three ROWS of the canonical one-item layout (pono_total:
// lane p < 40 owns channels 2 p, 2 p + 1; butterfly over each row of 16 lanes; T = R2 + (R1 + R0)) held as three slots of one lane.
// The butterflies are the same DPP steps on the same positions within a row and the rows are added in the same order, so the
// statistics -- and with them every output -- are the canonical bits; the totals are row-local (no readlane).
// row: the item's y tile in LDS (y at [0, 80), the gate half or the nin_skip slot at [80, 160)); item_ok: the lane's item is evaluated
// (lane-level: the four items of a pass differ); rin / b2: the residual input (POST_GATE) and the nin_skip bias at the lane's three pairs.
__device__ __forceinline__ float pono_total_rows(const f32x2 (&v)[3], const bool (&ok)[3])
{
    float x[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        x[s] = ok[s] ? v[s].x + v[s].y : 0.0f;
        x[s] = dpp_xadd<0xB1>(x[s]);
        x[s] = dpp_xadd<0x4E>(x[s]);
        x[s] = dpp_xadd<0x141>(x[s]);
        x[s] = dpp_xadd<0x140>(x[s]);      // every lane of the row: R_s of ITS item
    }
    return x[2] + (x[1] + x[0]);
}
template <int KIND>
__device__ __forceinline__ void post_items4(const PostArgs &a, int lane, const float *row, bool item_ok, size_t loc, const f32x2 (&rin)[3],
                                            const f32x2 (&b2)[3])
{
    const int l = lane & 15;
    const f32x2 zero = {0.0f, 0.0f};
    auto ld = [](const float *p) { return *(const f32x2 *)p; };
    f32x2 y[3], g[3], skip[3], d[3], dd[3];
    bool ok[3];
    int c[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        ok[s] = l + 16 * s < PONO_LANES;
        c[s] = ok[s] ? 2 * (l + 16 * s) : 0;
        y[s] = ld(row + c[s]);
        g[s] = KIND == POST_GATE ? ld(row + NF + c[s]) : zero;
        skip[s] = KIND == POST_CONVIN && a.has_skip ? ld(row + NF + c[s]) + b2[s] : zero;
    }
    const float mean = pono_mean(pono_total_rows(y, ok));
#pragma unroll
    for (int s = 0; s < 3; ++s) { d[s] = y[s] - mean; dd[s] = d[s] * d[s]; }
    const float inv = pono_inv(pono_total_rows(dd, ok));
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (!ok[s] || !item_ok) continue;
        const f32x2 out = post_finish<KIND>(d[s] * inv, g[s], skip[s], a.has_skip != 0, rin[s]);
        if (KIND == POST_CONVIN) {
            f32x2 ep, en;
            celu_pair2(out, ep, en);
            *(f32x2 *)(a.Xout + loc * (2 * NF) + c[s]) = ep;
            *(f32x2 *)(a.Xout + loc * (2 * NF) + NF + c[s]) = en;
        } else {
            store_raw_celu2(a.Rout, a.Eout, loc, c[s], out);
        }
    }
}
// the post op of a workgroup's items parked in LDS (sY: a row of YLD floats per item), wave w taking items w, w + WAVES, ...: four
// per pass.  sItem / sLoc: item index (-1: not evaluated) and cache row per item.  Called by every thread of the workgroup BEFORE the
// barrier that publishes sY: the residual rows of all of a wave's items are requested first (post_prefetch), the passes run behind
// the barrier (post_run).
template <int KIND, int NPI, int WAVES>
struct Post4 {
    static_assert(NPI % 4 == 0, "four items per pass");
    static constexpr int NG = NPI / 4;
    f32x2 rin[NG][3], b2[3];
    int loc[NG], m[NG];
    bool ok[NG];
    __device__ __forceinline__ void prefetch(const PostArgs &pa, int wave, int lane, const int *sItem, const int *sLoc)
    {
        const int r = lane >> 4, l = lane & 15;
        const f32x2 z2 = {0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int c = l + 16 * s < PONO_LANES ? 2 * (l + 16 * s) : 0;
            b2[s] = KIND == POST_CONVIN && pa.has_skip ? *(const f32x2 *)(pa.bias2 + c) : z2;
        }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            m[gi] = wave + WAVES * (4 * gi + r);
            loc[gi] = sLoc[m[gi]];
            ok[gi] = sItem[m[gi]] >= 0;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int c = l + 16 * s < PONO_LANES ? 2 * (l + 16 * s) : 0;
                rin[gi][s] = KIND == POST_GATE ? *(const f32x2 *)(pa.Rin + (size_t)loc[gi] * R_LD + c) : z2;   // (row 0 for an absent item: loaded, dropped)
            }
        }
    }
    __device__ __forceinline__ void run(const PostArgs &pa, int lane, const float *sY, int yld) const
    {
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (!__any(ok[gi])) continue;     // (wave-uniform: none of the pass's four items is evaluated)
            post_items4<KIND>(pa, lane, sY + m[gi] * yld, ok[gi], (size_t)loc[gi], rin[gi], b2);
        }
    }
};

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
// stored in place of slot NA, the nin_skip slot raw.  Taken for launches of at least `gemm_wg_min` item tiles (struct Tuning).
// ------------------------------------------------------------------------------------------
constexpr int GW_WAVES = 4, GW_THREADS = 64 * GW_WAVES;
enum { GW_CONVOUT = 0, GW_CONVIN = 1, GW_DIL = 2 };
#ifdef PS_WG_TRACE_BUILD   // tuning builds: shader-clock stamps of wave 0 of the first 32 workgroups, per kernel variant
__device__ unsigned long long g_wg_trace[3][32][16];
__device__ unsigned long long g_wg_span[3][4096][2];   // wall clock (100 MHz) at the start and the end of every workgroup, + hw id
__device__ unsigned long long g_wg_chain[3][32][4][24];   // stamps inside the fourth open tap of the traced workgroups, per wave
// (32 workgroups spread over the grid of ONE large launch per kind and pass: every 128th)
#define WG_STAMP(k) do { if (a.trace_on && (y & 127) == 0 && (y >> 7) < 32 && tid == 0 && (k) < 16) g_wg_trace[KIND][y >> 7][(k)] = clock64(); } while (0)
void *wg_trace_symbol(int what)
{
    void *p = nullptr;
    const hipError_t e = what == 6 ? hipGetSymbolAddress(&p, HIP_SYMBOL(g_wg_trace)) : what == 7 ? hipGetSymbolAddress(&p, HIP_SYMBOL(g_wg_span)) : hipGetSymbolAddress(&p, HIP_SYMBOL(g_wg_chain));
    return e == hipSuccess ? p : nullptr;
}
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
    __shared__ int sLoc[MI];                 // its cache row (frame * L + location)
#ifdef PS_WG_TRACE_BUILD
    __shared__ unsigned long long sChain[4][24];   // tuning builds: stamps inside ONE tap (the fourth open one), per wave: phases of the tap loop
    int tapno = 0;                                 // [16..20]; with -DPS_WG_CHAIN_STAMPS also 5 chains x (start, MFMAs issued, next weights in) [0..15]
#ifdef PS_WG_CHAIN_STAMPS   // (the stamps inside the chains and the wait they need change the tap: off unless asked for)
#define CH_STAMP(k) do { if (tapno == 3 && lane == 0) sChain[wave][(k)] = clock64(); } while (0)
#else
#define CH_STAMP(k) do { if ((k) >= 16 && tapno == 3 && lane == 0) sChain[wave][(k)] = clock64(); } while (0)
#endif
#else
#define CH_STAMP(k) do { } while (0)
#endif
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15, kk = lane >> 4;
    const int xcd = blockIdx.x & (N_XCD - 1), tb = blockIdx.x >> 3;
    const int yy = xcd * a.tpx + tb;         // contiguous item ranges per XCD, as in k_gemm
    if (tb >= a.tpx || yy >= a.ny) return;
    const int y = yy;
    const int item0 = y * MI;
    const int ntaps = a.slot_first[a.nslots];
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    WG_STAMP(0);
#ifdef PS_WG_TRACE_BUILD
    auto span_end = [&]() {
        if (tid == 0 && y < 4096 && a.trace_on) {
            unsigned hw = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            g_wg_span[KIND][y][1] = (wall_clock64() << 16) | (hw & 0xffffu);
        }
    };
    if (tid == 0 && y < 4096 && a.trace_on) g_wg_span[KIND][y][0] = wall_clock64();
#else
    auto span_end = [&]() { };
#endif
    // ---- set-up: rows and mask values of every (tap, item) of the tile; wave w does taps w, w + 4, w + 8.
    // Where the products' items were sorted, k_perm_scatter left (item, location | tap set << 12 | fractional-masks flag << 21)
    // pairs: ONE load per lane instead of the chain position -> item -> order -> location -> nine mask values, which was 8-12 k
    // cycles at the head of an 80 k-cycle workgroup (tools/wg_trace.py, round 5); with 0 / 1 masks -- the reference's always are --
    // the tap set says what the mask values are.
    {
        const int m = lane & (MI - 1), pos = item0 + m;
        int item = -1, q = 0, pat = 0x1ff;
        bool frac = true;
        if (pos < a.nitems) {
            if (a.items.permq) {
                const int2 v = a.items.permq[pos];
                item = v.x; q = v.y & 4095; pat = (v.y >> 12) & 0x1ff; frac = (v.y >> 21) & 1;
            } else {
                int fq;
                item = item_at(a.items, pos, a.nitems);
                item_loc(a.items, item, a.L, fq, q);
            }
        }
        const bool valid = item >= 0 && item_wanted(a.items, item);
        const int f = a.items.f0 + (item >= 0 ? item / a.items.npre : 0), r = q / a.W, c = q - r * a.W;
        if (wave == 0 && lane < MI) { sItem[m] = valid ? item : -1; sLoc[m] = valid ? f * a.L + q : 0; }
        for (int t = wave; t < ntaps; t += GW_WAVES) {
            const GemmTap tp = a.tap[t];
            const int rr = r + tp.dr, cc = c + tp.dc;
            float mv = 0.0f;
            if (valid) {
                if (tp.mask_row < 0) mv = 1.0f;                                   // (nin_skip: the location itself, unmasked)
                else if (!frac) mv = (pat >> tp.mask_row) & 1 ? 1.0f : 0.0f;      // 0 / 1 masks: the tap set says it all
                else if (rr >= 0 && rr < a.H && cc >= 0 && cc < a.W) mv = a.mask[(size_t)f * a.mask_fstride + (size_t)tp.mask_row * a.L + q];
            }
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
    if (live == 0 && __ballot(sItem[lane & (MI - 1)] >= 0) == 0ull) { span_end(); return; }   // nothing of this tile is evaluated here
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
            CH_STAMP(3 * j);
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
            CH_STAMP(3 * j + 1);
#pragma unroll
            for (int h = 0; h < NGH; ++h)
#pragma unroll
                for (int n = 0; n < NAU; ++n) av[h][n] = an[h][n];
#ifdef PS_WG_CHAIN_STAMPS
            __builtin_amdgcn_s_waitcnt(0x0F70);   // (vmcnt(0): the next chain's weights are in -- what the first MFMA of the next chain waits for)
#endif
            CH_STAMP(3 * j + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        CH_STAMP(15);
#pragma unroll
        for (int k = 0; k < NTL; ++k)
            if (lv[k]) tot[k] = tot[k] + taptot[k];
    };
    auto tap_dispatch = [&](int t, int nxt, int buf) {
        if (has5 && tiles_of(t) == (1u << TI) - 1u) tap_products(t, nxt, buf, std::integral_constant<bool, true>{});
        else tap_products(t, nxt, buf, std::integral_constant<bool, false>{});
    };
    // The barrier of a tap exchanges LDS data only (the staged rows): it drains the LDS queue, not the vector-memory queue --
    // __syncthreads() would also sit out the weight requests of the next tap's first chain, issued just before it on purpose.
#ifndef PS_WG_TAP_SYNCTHREADS
    auto tap_barrier = [] { lds_barrier(); };
#else
    auto tap_barrier = [] { __syncthreads(); };
#endif
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
            CH_STAMP(16);
            const int nxt = next_live(t);
            if (nxt < ntaps) stage_load(nxt);    // in flight under the MFMAs
            CH_STAMP(17);
            tap_dispatch(t, nxt, buf);
            CH_STAMP(18);
            if (DB) {
                if (nxt < ntaps) stage_store(nxt, buf ^ 1);
                CH_STAMP(19);
                tap_barrier();                   // next tap's rows visible; everybody is done with this tap's
                CH_STAMP(20);
                buf ^= 1;
            } else {
                tap_barrier();                   // everybody is done with this tap's rows
                if (nxt < ntaps) stage_store(nxt, 0);
                tap_barrier();
            }
            cur = nxt;
            WG_STAMP(nstamp < 14 ? nstamp : 13);
#ifdef PS_WG_TRACE_BUILD
            ++tapno;
#endif
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
    WG_STAMP(14);
    if (fuse_post) {
        float *sY = (float *)sB;
        // the wave's items are m = wave, wave + 4, ...: what their post ops read from memory -- the residual input of a gate, a row
        // per item -- is requested for ALL of them here, ahead of the parking and the barrier (fetched item by item inside the loop,
        // each was a dependent round trip -- order look-up, then the row -- in front of its item: 9-19 k cycles of a workgroup's life)
        Post4<POSTK, MI / GW_WAVES, GW_WAVES> post;
        post.prefetch(pa, wave, lane, sItem, sLoc);
        // (conv_input with nin_skip: the skip slot was parked above, after the barrier of the last tap; here the taps are done too)
#pragma unroll
        for (int k = 0; k < NTL; ++k)
            if (k < NT4 || has5) *(f32x4 *)(sY + m_of(k) * YLD + o_of(k) + kk * 4) = ysum[k];
        __syncthreads();
        post.run(pa, lane, sY, YLD);
    }
    WG_STAMP(15);
#ifdef PS_WG_TRACE_BUILD
    __syncthreads();
    if (a.trace_on && (y & 127) == 0 && (y >> 7) < 32 && tid < 96) g_wg_chain[KIND][y >> 7][tid / 24][tid % 24] = sChain[tid / 24][tid % 24];
#endif
    span_end();
}

// ------------------------------------------------------------------------------------------
// k_gemm_ws: the whole-grid products with the WEIGHTS shared through LDS (round 5).
//
// k_gemm_wg fetches a wave's weights from L2 into registers for every 16 (or 32) items: at the full MFMA rate that stream alone
// is 32 B/clk per CU against the ~41 B/clk a CU's vector-memory path passes -- the kernel sat at 0.45-0.5 MFMA-busy whatever was
// done to its waves.  With the items grouped by open-tap set (k_perm_*), 64 consecutive items share their taps as well as 16 do,
// so the operands change places:
//   * a workgroup of four waves owns 64 items; wave w owns item tile w (16 items) and ALL output channels (10 or 5 MFMA tiles);
//   * operand A, the weights of one (tap, accumulation chain) -- 20 KB for conv_out -- is staged in LDS ONCE per workgroup, in
//     MFMA fragment order, two buffers: the chunk after the current one is requested into registers before the chain's MFMAs and
//     parked behind them, one LDS-only barrier per chunk.  A weight byte fetched from L2 now feeds 64 items instead of 16;
//   * operand B, a wave's own 16 input rows, goes straight from L2 to its registers a chain ahead (nobody else reads them);
//   * a tap that is closed for all 16 items of a wave is skipped by that wave (it keeps the barriers), a tap closed for all 64
//     is skipped by the workgroup; the post op runs in the tail exactly as in k_gemm_wg.
// Arithmetic and order are k_gemm's to the bit (five chains per tap, chain j = channel groups j, j + 5 in MFMA order, tap value
// (((a0 + a1) + a2) + a3) + a4, taps added in order into the slot, y = ((bias + NA) + C) + NB, the nin_skip slot raw).
// ------------------------------------------------------------------------------------------
#ifndef PS_WS_EXP   // tuning builds only (results INVALID): 1 no MFMAs, 2 no weight staging, 4 no input-row loads, 8 no chunk barriers, 16 no post op
#define PS_WS_EXP 0
#endif
constexpr int WS_WAVES = 4, WS_THREADS = 64 * WS_WAVES, WS_MI = 16 * WS_WAVES;
constexpr int ws_occ(int kind) { return kind == GW_CONVOUT ? 2 : 3; }
template <int KIND>
__attribute__((amdgpu_waves_per_eu(ws_occ(KIND), ws_occ(KIND))))
__global__ __launch_bounds__(WS_THREADS) void k_gemm_ws(GemmArgs a, PostArgs pa)
{
    constexpr int NGH = KIND == GW_DIL ? 1 : 2, NOT = KIND == GW_CONVOUT ? 10 : 5, CO = 16 * NOT, MI = WS_MI;
    constexpr int POSTK = KIND == GW_CONVOUT ? POST_GATE : KIND == GW_CONVIN ? POST_CONVIN : POST_DIL;
    constexpr int YLD = KIND == GW_DIL ? 84 : 168;
    constexpr int HALF = 4 * CO;                 // f32x4 of one 16-channel group of a tap's weights: [kk][o]
    constexpr int CHUNK = NGH * HALF;            // one (tap, chain) chunk: groups j and j + 5
    constexpr int SVN = (CHUNK + WS_THREADS - 1) / WS_THREADS;   // staging elements per thread (the last one may be absent)
    constexpr int NQ = NGH * NOT;                // A fragments of a chain, in the order they are used: q = h * NOT + tile
    constexpr int NB4 = 2 * CHUNK > MI * YLD / 4 ? 2 * CHUNK : MI * YLD / 4;   // (the post op's tile reuses the weight buffers)
    __shared__ f32x4 sA[NB4];
    __shared__ int sRow[MAX_TAPS * MI];
    __shared__ float sMv[MAX_TAPS * MI];
    __shared__ int sItem[MI], sLoc[MI];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15, kk = lane >> 4;
    const int xcd = blockIdx.x & (N_XCD - 1), tb = blockIdx.x >> 3;
    const int y = xcd * a.tpx + tb;              // contiguous item ranges per XCD, as in k_gemm
    if (tb >= a.tpx || y >= a.ny) return;
    const int item0 = y * MI, ntaps = a.slot_first[a.nslots];
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    {   // ---- set-up: rows and mask values of every (tap, item): lane = item, wave w does taps w, w + 4, w + 8.
        // Where the items were sorted, k_perm_scatter left (item, location | tap set << 12 | fractional-masks flag << 21) pairs: ONE load
        // per lane instead of the chain position -> item -> order -> location -> mask values (three dependent round trips at the head of a
        // workgroup whose partner on the compute unit starts at the same moment); with 0 / 1 masks the tap set says what the values are
        // (k_gemm_wg's set-up, round 5)
        const int pos = item0 + lane;
        int item = -1, q = 0, pat = 0x1ff;
        bool frac = true;
        if (pos < a.nitems) {
            if (a.items.permq) {
                const int2 v = a.items.permq[pos];
                item = v.x; q = v.y & 4095; pat = (v.y >> 12) & 0x1ff; frac = (v.y >> 21) & 1;
            } else {
                int fq;
                item = item_at(a.items, pos, a.nitems);
                item_loc(a.items, item, a.L, fq, q);
            }
        }
        const bool valid = item >= 0 && item_wanted(a.items, item);
        const int f = a.items.f0 + (item >= 0 ? item / a.items.npre : 0), r = q / a.W, c = q - r * a.W;
        if (wave == 0) { sItem[lane] = valid ? item : -1; sLoc[lane] = valid ? f * a.L + q : 0; }
        for (int t = wave; t < ntaps; t += WS_WAVES) {
            const GemmTap tp = a.tap[t];
            const int rr = r + tp.dr, cc = c + tp.dc;
            float mv = 0.0f;
            if (valid) {
                if (tp.mask_row < 0) mv = 1.0f;                                   // (nin_skip: the location itself, unmasked)
                else if (!frac) mv = (pat >> tp.mask_row) & 1 ? 1.0f : 0.0f;      // 0 / 1 masks: the tap set says it all
                else if (rr >= 0 && rr < a.H && cc >= 0 && cc < a.W) mv = a.mask[(size_t)f * a.mask_fstride + (size_t)tp.mask_row * a.L + q];
            }
            sRow[t * MI + lane] = mv != 0.0f ? (f * a.L + rr * a.W + cc) : -1;
            sMv[t * MI + lane] = mv;
        }
    }
    __syncthreads();
    // taps with an open item in the workgroup / in this wave's item tile (wave-uniform)
    unsigned wg_live = 0, my_live = 0;
    for (int t = 0; t < ntaps; ++t) {
        const unsigned long long b = __ballot(sRow[t * MI + lane] >= 0);
        if (b) wg_live |= 1u << t;
        if ((b >> (16 * wave)) & 0xFFFFull) my_live |= 1u << t;
    }
    if (wg_live == 0 && __ballot(sItem[lane] >= 0) == 0ull) return;   // nothing of this tile is evaluated here
    auto next_live = [&](int t) {   // first tap after t that is open somewhere in the workgroup, or ntaps
        int n = t + 1;
        while (n < ntaps && !((wg_live >> n) & 1u)) ++n;
        return n;
    };
    // ---- weights: element e = tid + 256 s of a chunk is (h, kk, o); it comes from [group j + 5 h][kk][o] of the tap's packed weights
    // and goes to [h][output tile][lane (o % 16, kk)] of the buffer -- the order the MFMA A fragments are read in
    int goff[SVN], loff[SVN];
#pragma unroll
    for (int sx = 0; sx < SVN; ++sx) {
        const int e = min(tid + WS_THREADS * sx, CHUNK - 1), h = e / HALF, r = e - h * HALF, k4 = r / CO, o = r - k4 * CO;
        goff[sx] = h * 5 * HALF + r;
        loff[sx] = h * HALF + (o >> 4) * 64 + k4 * 16 + (o & 15);
    }
    f32x4 sv[SVN];
    auto w_load = [&](int t, int j) {
        const f32x4 *w = (const f32x4 *)a.tap[t].w + j * HALF;
#pragma unroll
        for (int sx = 0; sx < SVN; ++sx) sv[sx] = w[goff[sx]];   // (the absent element of the last round re-reads a valid one)
    };
    auto w_store = [&](int buf) {
#pragma unroll
        for (int sx = 0; sx < SVN; ++sx)
            if (sx < SVN - 1 || CHUNK % WS_THREADS == 0 || tid + WS_THREADS * sx < CHUNK) sA[buf * CHUNK + loff[sx]] = sv[sx];
    };
    // ---- input rows of this wave's 16 items: lane (i, kk) carries channels 16 g + 4 kk .. + 3 of item i
    const float *bsrc = nullptr;     // row of the tap being loaded (the zero row where it is closed: loaded, dropped)
    float bmv = 0.0f;
    bool bopen = false;
    f32x4 bn[NGH];                   // a chain ahead
    auto b_tap = [&](int t) {        // the lane's row of tap t
        const int row = sRow[t * MI + wave * 16 + i];
        bmv = sMv[t * MI + wave * 16 + i];
        bopen = row >= 0;
        bsrc = (bopen ? a.tap[t].in + (size_t)row * a.tap[t].ld : (const float *)g_zero_row) + 4 * kk;
    };
    auto b_load = [&](int j) {
#pragma unroll
        for (int h = 0; h < NGH; ++h) bn[h] = *(const f32x4 *)(bsrc + 16 * (j + 5 * h));
    };
    f32x4 tot[NOT], ysum[NOT];
#pragma unroll
    for (int k = 0; k < NOT; ++k) { tot[k] = zero; ysum[k] = zero; }
    int cur = next_live(-1), buf = 0;
    if (cur < ntaps) {
        w_load(cur, 0);
        b_tap(cur);
        b_load(0);
        w_store(0);
    }
    lds_barrier();
    for (int slot = 0; slot < a.nslots; ++slot) {
        for (int t = a.slot_first[slot]; t < a.slot_first[slot + 1]; ++t) {
            if (t != cur) continue;              // closed for the whole workgroup: an exact zero, skipped by every wave
            const int nxt = next_live(t);
            const bool mine = (my_live >> t) & 1u;
            f32x4 taptot[NOT];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                // the chunk after this one into registers, under the MFMAs (of this tap again when there is none: same requests on every path)
                if (!(PS_WS_EXP & 2)) { if (j < 4) w_load(t, j + 1); else w_load(nxt < ntaps ? nxt : t, 0); }
                f32x4 bv[NGH];
#pragma unroll
                for (int h = 0; h < NGH; ++h) bv[h] = bopen ? bn[h] * bmv : zero;   // (x * 1.0f is x: 0 / 1 masks cost nothing)
                if (j == 4) b_tap(nxt < ntaps ? nxt : t);
                if (!(PS_WS_EXP & 4)) b_load(j < 4 ? j + 1 : 0);
                if (mine && !(PS_WS_EXP & 1)) {
                    const f32x4 *A = sA + buf * CHUNK + lane;
                    // the A fragments of the pair after this one are requested BEFORE this pair's MFMAs and nothing may sink them
                    // behind (left alone, the scheduler read every pair's fragments right in front of its MFMAs: an LDS round
                    // trip per 8 MFMAs)
                    f32x4 acc[NOT], w0 = A[0], w1 = A[64], w2 = zero, w3 = zero;
#pragma unroll
                    for (int q = 0; q < NQ; q += 2) {
                        if (q + 2 < NQ) w2 = A[((q + 2) / NOT) * HALF + ((q + 2) % NOT) * 64];
                        if (q + 3 < NQ) w3 = A[((q + 3) / NOT) * HALF + ((q + 3) % NOT) * 64];
                        __builtin_amdgcn_sched_barrier(0);
                        const int h0 = q / NOT, k0 = q % NOT, h1 = (q + 1) / NOT, k1 = (q + 1) % NOT;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            acc[k0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[c], bv[h0][c], h0 == 0 && c == 0 ? zero : acc[k0], 0, 0, 0);
                            if (q + 1 < NQ)
                                acc[k1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[c], bv[h1][c], h1 == 0 && c == 0 ? zero : acc[k1], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        w0 = w2;
                        w1 = w3;
                    }
#pragma unroll
                    for (int k = 0; k < NOT; ++k) taptot[k] = j == 0 ? acc[k] : taptot[k] + acc[k];
                }
                if (!(PS_WS_EXP & 2)) w_store(buf ^ 1);
                if (!(PS_WS_EXP & 8)) lds_barrier();                   // the next chunk is visible; everybody is done with this one
                buf ^= 1;
            }
            if (mine && !(PS_WS_EXP & 1)) {
#pragma unroll
                for (int k = 0; k < NOT; ++k) tot[k] = tot[k] + taptot[k];
            }
            cur = nxt;
        }
        if (slot == SLOT_SKIP) {                 // (the last slot: every chunk is consumed, the buffers are free)
            float *sY = (float *)sA;
#pragma unroll
            for (int k = 0; k < NOT; ++k) *(f32x4 *)(sY + (wave * 16 + i) * YLD + NF + 16 * k + kk * 4) = tot[k];
        } else {
#pragma unroll
            for (int k = 0; k < NOT; ++k) ysum[k] = (slot == SLOT_NA ? *(const f32x4 *)(a.sum_bias + 16 * k + kk * 4) : ysum[k]) + tot[k];
        }
#pragma unroll
        for (int k = 0; k < NOT; ++k) tot[k] = zero;
    }
    // ---- the post op of the stage on the workgroup's 64 items, 16 per wave (as in k_gemm_wg)
    {
        float *sY = (float *)sA;
        Post4<POSTK, MI / WS_WAVES, WS_WAVES> post;
        post.prefetch(pa, wave, lane, sItem, sLoc);
#pragma unroll
        for (int k = 0; k < NOT; ++k) *(f32x4 *)(sY + (wave * 16 + i) * YLD + 16 * k + kk * 4) = ysum[k];
        __syncthreads();
        if (!(PS_WS_EXP & 16)) post.run(pa, lane, sY, YLD);
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
struct StartsArgs {
    const int32_t *order;   // (F, L)
    const float *mask_und, *mask_dil;   // (F, 9, L): type B dilation 1 / dilation 2
    int H, W, L, npre, F;
    int g_in[NGATED], g_out[NGATED], g_skip[NGATED], d_in[4], d_out[4];
    int32_t *starts;        // (N_EVAL, F)
    int f0;                 // frames [f0, f0 + gridDim.x) of the F
    const int32_t *pend;    // (F) or null: the prefix of frame f is its ranks [0, pend[f]) instead of [0, npre)
};
constexpr int STARTS_MAXL = 4096;
__global__ __launch_bounds__(1024) void k_prefix_starts(StartsArgs a)
{
    __shared__ int rank[STARTS_MAXL];   // by location
    __shared__ int s1[STARTS_MAXL];     // by rank < npre: min rank among the open dilation-1 taps of ranks >= r (suffix minimum)
    __shared__ int s2[STARTS_MAXL];     //                 the same, dilation-2 taps of the dilated mask
    __shared__ int cmin[2];             // min rank the COLUMNS (ranks >= npre) read through dilation-1 / dilation-2 taps
    const int f = a.f0 + blockIdx.x, t = threadIdx.x, L = a.L, npre = a.pend ? a.pend[f] : a.npre;
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

// ------------------------------------------------------------------------------------------
// Items grouped by their set of open taps (round 5).  A tile of the products computes a tap for all its items as soon as ONE of
// them has it open; a location has 4.7 of its 9 taps open on average (of every adjacent pair exactly one precedes the other), a
// tile of 16 consecutive ranks of a frame 7.1 of 9, a tile of 32 already 8.1 -- a third of the MFMA work of the pass multiplied
// zeros.  The masks of a frame take about 40 distinct tap sets, so the items are SORTED by tap set (9 bits: tap t open and inside
// the grid) and the products walk that list: 4.8 taps per tile of 16, 4.85 per tile of 32.  A closed tap adds an exact zero, so
// which items share a tile changes no bit; the post ops and every cache row are addressed by the item itself, as before.
// Order inside a tap set: frame, then rank (neighbouring ranks are neighbouring locations: their input rows are the same lines).
// `nparts` > 1: one sort per share of the frames, so that the contiguous range of tiles an XCD takes (k_gemm / k_gemm_wg) reads
// the rows of ITS frames only.  Three small launches per pass and mask kind (dilation 1, dilation 2):
//   k_perm_sort     per frame: (tap set's place << 13 | rank << 1 | fractional masks) of its items, sorted (bitonic, LDS), and the run
//                   length of every tap set
//   k_perm_scan     first position of every (share, tap set, frame) run: exclusive scan over [share][tap set][frame], in tiles of
//                   1024 entries (the tiles' totals are scanned by every block of the next launch for itself)
//   k_perm_scatter  per frame: perm[first + index in the run] = item
// ------------------------------------------------------------------------------------------
struct PermArgs {
    const int32_t *order;               // (F, L) or null (raster)
    const float *mask[2];               // (F, 9, L): type B dilation 1 / dilation 2
    int H, W, L, npre, f0, nf, nparts;
    uint32_t *sorted[2];                // [nf][npre]
    int32_t *cnt[2];                    // [nparts][512][frames per share]
    int32_t *tsum[2];                   // totals of the table's tiles of 1024 entries
    int32_t *perm[2];                   // [nf * npre]
    int2 *permq[2];                     // the same as (item, location) pairs
    const int32_t *pend;                // (F) or null: ranks >= pend[f] of frame f are not part of its prefix (they sort behind everything)
};
constexpr int PERM_KEYS = 512;
// Sort key of a tap set: its place in the order (number of open taps, descending; then the 9-bit set).  The workgroups of a launch
// are dispatched in item order, so the tiles with the most taps -- the longest jobs -- start first and the launch's tail is made of
// the cheapest ones (longest-processing-time-first: a tile of 7 open taps costs twice one of 3, and 64-item workgroups fill the
// chip only two to three times over).
struct PermBins { unsigned short v[PERM_KEYS]; };
constexpr PermBins make_perm_bins()
{
    PermBins t{};
    int n = 0;
    for (int pc = 9; pc >= 0; --pc)
        for (int p = 0; p < PERM_KEYS; ++p)
            if (__builtin_popcount((unsigned)p) == pc) t.v[p] = (unsigned short)n++;
    return t;
}
__device__ const PermBins g_perm_bin = make_perm_bins();
constexpr PermBins make_perm_pats()   // the inverse: place -> tap set
{
    PermBins t{}, b = make_perm_bins();
    for (int p = 0; p < PERM_KEYS; ++p) t.v[b.v[p]] = (unsigned short)p;
    return t;
}
__device__ const PermBins g_perm_pat = make_perm_pats();
__device__ __forceinline__ int perm_fpp(const PermArgs &a) { return (a.nf + a.nparts - 1) / a.nparts; }   // frames per share
__device__ __forceinline__ size_t perm_cnt_index(const PermArgs &a, int fl, int key)
{
    const int fpp = perm_fpp(a), part = fl / fpp;
    return ((size_t)part * PERM_KEYS + key) * fpp + (fl - part * fpp);
}
__global__ __launch_bounds__(1024) void k_perm_sort(PermArgs a)
{
    __shared__ uint32_t s[STARTS_MAXL];
    __shared__ int hist[PERM_KEYS];
    const int fl = blockIdx.x, kind = blockIdx.y, f = a.f0 + fl, t = threadIdx.x, dil = kind + 1;
    const float *mask = a.mask[kind];
    int P = 2;
    while (P < a.npre) P <<= 1;
    for (int r = t; r < P; r += 1024) {
        uint32_t v = 0xFFFFFFFFu;
        if (r < a.npre) {
            const int q = a.order ? a.order[(size_t)f * a.L + r] : r, y = q / a.W, x = q - y * a.W;
            uint32_t key = 0, frac = 0;   // frac: a mask value that is neither 0 nor 1 (the reference's never are): the products load them
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + (tap / 3 - 1) * dil, xx = x + (tap % 3 - 1) * dil;
                if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
                    const float mvv = mask[((size_t)f * 9 + tap) * a.L + q];
                    if (mvv != 0.0f) key |= 1u << tap;
                    if (mvv != 0.0f && mvv != 1.0f) frac = 1;
                }
            }
#ifdef PS_PERM_PLAIN_BINS   // (tuning builds: the tap sets in the order of their 9-bit value)
            v = key << 13 | (uint32_t)r << 1 | frac;
#else
            v = (uint32_t)g_perm_bin.v[key] << 13 | (uint32_t)r << 1 | frac;
#endif
            // (per-frame prefixes: the ranks behind a frame's own end keep their place in the item space -- nobody evaluates them,
            // item_wanted -- and are put together behind every tap set, so that they fill whole tiles, which leave at once)
            if (a.pend && r >= a.pend[f]) v = (uint32_t)(PERM_KEYS - 1) << 13 | (uint32_t)r << 1;
        }
        s[r] = v;
    }
    for (int k = t; k < PERM_KEYS; k += 1024) hist[k] = 0;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < P; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t x = s[i], y = s[l];
                    if ((x > y) == ((i & k) == 0)) { s[i] = y; s[l] = x; }
                }
            }
            __syncthreads();
        }
    for (int i = t; i < a.npre; i += 1024) {
        a.sorted[kind][(size_t)fl * a.npre + i] = s[i];
        atomicAdd(&hist[s[i] >> 13], 1);
    }
    __syncthreads();
    for (int k = t; k < PERM_KEYS; k += 1024) a.cnt[kind][perm_cnt_index(a, fl, k)] = hist[k];
}
// block-wide exclusive scan of one value per thread (1024 threads); returns the exclusive prefix, *total = the block's sum
__device__ __forceinline__ int block_exscan_1024(int v, int *sh /*[1024]*/, int *total)
{
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int u = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += u;
        __syncthreads();
    }
    *total = sh[1023];
    return sh[t] - v;
}
// grid (tiles of 1024 table entries, mask kinds): run lengths -> exclusive prefix inside the tile, + the tile's total
__global__ __launch_bounds__(1024) void k_perm_scan(PermArgs a)
{
    __shared__ int sh[1024];
    const int kind = blockIdx.y, i = blockIdx.x * 1024 + threadIdx.x;
    const int n = a.nparts * PERM_KEYS * perm_fpp(a);
    int32_t *c = a.cnt[kind];
    int total;
    const int ex = block_exscan_1024(i < n ? c[i] : 0, sh, &total);
    if (i < n) c[i] = ex;
    if (threadIdx.x == 0) a.tsum[kind][blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void k_perm_scatter(PermArgs a)
{
    __shared__ int first[PERM_KEYS];
    __shared__ int sh[1024];
    __shared__ int tbase[1024];          // first position of every tile of the run-length table (its tiles' totals, scanned)
    const int fl = blockIdx.x, kind = blockIdx.y, t = threadIdx.x;
    const int ntiles = (a.nparts * PERM_KEYS * perm_fpp(a) + 1023) / 1024;   // <= 1024: maxF <= 2048 (checked by the caller)
    int total;
    tbase[t] = block_exscan_1024(t < ntiles ? a.tsum[kind][t] : 0, sh, &total);
    const uint32_t *s = a.sorted[kind] + (size_t)fl * a.npre;
    for (int i = t; i < a.npre; i += 1024) {
        const uint32_t key = s[i] >> 13;
        if (i == 0 || (s[i - 1] >> 13) != key) first[key] = i;
    }
    __syncthreads();
    for (int i = t; i < a.npre; i += 1024) {
        const uint32_t v = s[i], key = v >> 13;
        const size_t e = perm_cnt_index(a, fl, (int)key);
        const int r = (int)((v >> 1) & 4095u), pos = a.cnt[kind][e] + tbase[e >> 10] + i - first[key];
        const int q = a.order ? a.order[(size_t)(a.f0 + fl) * a.L + r] : r;
#ifdef PS_PERM_PLAIN_BINS
        const int pat = (int)key;
#else
        const int pat = g_perm_pat.v[key];
#endif
        a.perm[kind][pos] = fl * a.npre + r;
        a.permq[kind][pos] = int2{fl * a.npre + r, q | pat << 12 | (int)(v & 1u) << 21};   // (item, location | tap set | fractional masks)
    }
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
bool launch_gemm(GemmArgs &a, int item_blocks, hipStream_t st, const Tuning &tune, const PostArgs *post = nullptr, ps_pixelcnn *h = nullptr)
{
    a.nx = (a.Co_pad + 16 * GEMM_T - 1) / (16 * GEMM_T);
    a.ny = item_blocks;
    a.tpx = (item_blocks + N_XCD - 1) / N_XCD;
    // one wave per slot while that is what it takes to fill the chip (4096 wave slots): a 16-view prefix is 2870 (tile,
    // channel block) pairs, one view 180 -- walking all slots in one wave would leave most of the SIMDs idle and make
    // each wave three times as long
    // the workgroup forms (k_gemm_wg: input rows shared through LDS; k_gemm_ws: weights shared through LDS) from tune.gemm_wg_min item
    // tiles on, for the shapes of the network's 3x3 convs; they produce the summed form (y in place of slot NA), bit-identical to k_gemm's
    const bool shape_ok = (a.Cin == 2 * NF || a.Cin == NF) && (a.Co_pad == NF || (a.Co_pad == 2 * NF && a.Cin == 2 * NF));
    const bool wg_form = a.sum_bias && a.nslots >= 3 && shape_ok && item_blocks >= tune.gemm_wg_min && a.tiles_per_block == 1;
    a.zgrid = a.nx * a.ny < tune.gemm_merge_min && !wg_form ? a.nslots : 1;
    if (a.zgrid != 1 || a.nslots < 3) a.sum_bias = nullptr;   // (only a wave that walks NA, C and NB can add them up)
    if (wg_form) {
        const int kind = a.Co_pad == 2 * NF ? GW_CONVOUT : a.Cin == 2 * NF ? GW_CONVIN : GW_DIL;
        // item tiles per workgroup (conv_out with 16 items per workgroup: 168 registers, three workgroups per CU -- 0.6 % of the
        // 128-view step over {2, 2, 2})
        const int TI = kind == GW_CONVOUT ? tune.wg_ti_out : kind == GW_CONVIN ? tune.wg_ti_in : tune.wg_ti_dil, MI = 16 * TI;
        a.ny = (a.nitems + MI - 1) / MI;
        a.tpx = (a.ny + N_XCD - 1) / N_XCD;
        const dim3 grid((unsigned)(N_XCD * a.tpx)), block(GW_THREADS);
#ifdef PS_WG_TRACE_BUILD   // the stamps of ONE launch per kind and pass: PS_WG_TRACE_SEL-th of the large ones (default 5)
        static int seen[3] = {0, 0, 0};
        static const int sel = getenv("PS_WG_TRACE_SEL") ? atoi(getenv("PS_WG_TRACE_SEL")) : 5;
        const int per_pass = kind == GW_DIL ? 4 : 14;
        a.trace_on = a.ny >= 2048 && seen[kind]++ % per_pass == (sel < per_pass ? sel : per_pass - 1);
#endif
        if ((tune.gemm_ws >> kind & 1) && post != nullptr && item_blocks >= tune.gemm_ws_min) {   // weights shared through LDS, 64 items per workgroup (round 5); one bit per kind
            a.ny = (a.nitems + WS_MI - 1) / WS_MI;
            a.tpx = (a.ny + N_XCD - 1) / N_XCD;
            PostArgs pw = *post;
            pw.summed = 1;
            const dim3 gws((unsigned)(N_XCD * a.tpx)), bws(WS_THREADS);
            timed(h, st, TAG_GRID, kind == GW_CONVOUT ? LK_GEMM_WS_OUT : kind == GW_CONVIN ? LK_GEMM_WS_IN : LK_GEMM_WS_DIL, [&]() {
                if (kind == GW_CONVOUT) hipLaunchKernelGGL((k_gemm_ws<GW_CONVOUT>), gws, bws, 0, st, a, pw);
                else if (kind == GW_CONVIN) hipLaunchKernelGGL((k_gemm_ws<GW_CONVIN>), gws, bws, 0, st, a, pw);
                else hipLaunchKernelGGL((k_gemm_ws<GW_DIL>), gws, bws, 0, st, a, pw);
            });
            return true;
        }
        const bool fuse = post != nullptr;
        PostArgs pp{};
        if (fuse) { pp = *post; pp.summed = 1; }
        const int fz = fuse ? 1 : 0;
        timed(h, st, TAG_GRID, LK_GEMM_WG, [&]() {
            if (kind == GW_CONVOUT && TI == 1) hipLaunchKernelGGL((k_gemm_wg<GW_CONVOUT, 1>), grid, block, 0, st, a, pp, fz);
            else if (kind == GW_CONVOUT) hipLaunchKernelGGL((k_gemm_wg<GW_CONVOUT, 2>), grid, block, 0, st, a, pp, fz);
            else if (kind == GW_CONVIN && TI == 2) hipLaunchKernelGGL((k_gemm_wg<GW_CONVIN, 2>), grid, block, 0, st, a, pp, fz);
            else if (kind == GW_CONVIN) hipLaunchKernelGGL((k_gemm_wg<GW_CONVIN, 4>), grid, block, 0, st, a, pp, fz);
            else if (TI == 2) hipLaunchKernelGGL((k_gemm_wg<GW_DIL, 2>), grid, block, 0, st, a, pp, fz);
            else hipLaunchKernelGGL((k_gemm_wg<GW_DIL, 4>), grid, block, 0, st, a, pp, fz);
        });
        return fuse;
    }
    timed(h, st, TAG_GRID, LK_GEMM, [&]() { hipLaunchKernelGGL(k_gemm, dim3((unsigned)(N_XCD * a.nx * a.tpx * a.zgrid)), dim3(64), 0, st, a); });
    return false;
}

// ------------------------------------------------------------------------------------------
// whole-grid evaluation (reference-faithful forward; cache build before the column steps)
// logits: null (caches only), (F,512,H,W) when nchw, else (F*L,512) by location
// ------------------------------------------------------------------------------------------
// (with an order: the pass can be restricted to frames [f0, f0 + nf) of the F -- independent passes over disjoint frame ranges
// may run on different streams)
void run_grid(ps_pixelcnn *h, int F, const int32_t *codes, const Masks &m, float *logits, bool nchw, hipStream_t st,
              const int32_t *order, int npre, int f0, int nf, const int32_t *pend)
{
    if (nf < 0) nf = F;
    if (!order) pend = nullptr;
    ItemMap all_items{order, order ? npre : h->L, nullptr, f0};
    all_items.end = pend;
    const int nitems = nf * all_items.npre;
    if (nitems <= 0) return;  // an AR run that starts at rank 0 has no prefix
    const int pblocks = (nitems + 3) / 4;
    // the prefix of an AR run: only the items somebody reads, stage by stage (k_prefix_starts).  tune.prefix_full: all of them.
    // (with out_logits the caller also gets the logits of the prefix locations: every item is needed then.  tune.prefix_cone_force
    // keeps the elimination on for the parity test, which compares the logits of the WALKED locations only.)
    const bool cone = order && (!logits || h->tune.prefix_cone_force) && h->L <= STARTS_MAXL && !h->tune.prefix_full;
    if (cone) {
        StartsArgs sa{order, m.und, m.dil, h->H, h->W, h->L, npre, F, {}, {}, {}, {}, {}, h->pstart, f0, pend};
        for (int g = 0; g < NGATED; ++g) { sa.g_in[g] = h->gated[g].node_in; sa.g_out[g] = h->gated[g].node_out; sa.g_skip[g] = h->gated[g].node_skip; }
        for (int d = 0; d < 4; ++d) { sa.d_in[d] = h->dil[d].node_in; sa.d_out[d] = h->dil[d].node_out; }
        hipLaunchKernelGGL(k_prefix_starts, dim3(nf), dim3(1024), 0, st, sa);
    }
    // the products' item lists, grouped by open-tap set (one per mask kind); the frame range's own part of the scratch
    const int32_t *perm[2] = {nullptr, nullptr};
    const int2 *permq[2] = {nullptr, nullptr};
    if (h->tune.item_sort && h->L <= STARTS_MAXL && all_items.npre >= 2 && nf <= 2048) {
        PermArgs pa{};
        pa.order = order; pa.mask[0] = m.und; pa.mask[1] = m.dil;
        pa.H = h->H; pa.W = h->W; pa.L = h->L; pa.npre = all_items.npre; pa.f0 = f0; pa.nf = nf; pa.pend = pend;
        pa.nparts = h->tune.item_sort == 2 && nf >= 2 * N_XCD && nf % N_XCD == 0 ? N_XCD : 1;   // (even shares only: the table is [share][tap set][frame])
        const size_t locs = (size_t)h->maxF * h->L;
        for (int k = 0; k < 2; ++k) {
            pa.sorted[k] = h->perm_sorted + k * locs + (size_t)f0 * h->L;
            pa.perm[k] = h->perm + k * locs + (size_t)f0 * h->L;
            pa.permq[k] = h->permq + k * locs + (size_t)f0 * h->L;
            permq[k] = pa.permq[k];
            pa.cnt[k] = h->perm_cnt + ((size_t)k * h->maxF + f0) * PERM_KEYS;
            pa.tsum[k] = h->perm_tsum + (size_t)k * h->maxF + f0;
            perm[k] = pa.perm[k];
        }
        hipLaunchKernelGGL(k_perm_sort, dim3(nf, 2), dim3(1024), 0, st, pa);
        hipLaunchKernelGGL(k_perm_scan, dim3((pa.nparts * PERM_KEYS * ((nf + pa.nparts - 1) / pa.nparts) + 1023) / 1024, 2), dim3(1024), 0, st, pa);
        hipLaunchKernelGGL(k_perm_scatter, dim3(nf, 2), dim3(1024), 0, st, pa);
    }
    float *const part = h->partial + (size_t)4 * f0 * h->L * (2 * NF);
    ItemMap items = all_items;
    auto at_stage = [&](int stage_id) { items.start = cone ? h->pstart + (size_t)stage_id * F : nullptr; };
    // -> 0: raw slots in `partial`, 1: slots summed by the kernel, 2: the post op `post` done by the kernel as well
    auto gemm = [&](GemmArgs &a, const float *mask, const float *sum_bias = nullptr, const PostArgs *post = nullptr) {
        a.items = items;
        a.items.perm = mask == m.und ? perm[0] : mask == m.dil ? perm[1] : nullptr;
        a.items.permq = mask == m.und ? permq[0] : mask == m.dil ? permq[1] : nullptr;
        a.H = h->H; a.W = h->W; a.L = h->L; a.nitems = nitems;
        a.mask = mask; a.mask_fstride = (size_t)9 * h->L; a.tiles_per_block = 1;
        a.partial = h->partial + (size_t)4 * f0 * h->L * (2 * NF);   // (the frame range's own part of the scratch: passes over disjoint ranges may run side by side)
        a.sum_bias = sum_bias;
        const int tiles = (nitems + 15) / 16;
        if (launch_gemm(a, tiles, st, h->tune, post, h)) return 2;
        return a.sum_bias != nullptr ? 1 : 0;
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
        p.summed = gemm(a, m.und, G.b_in, &p);
        if (p.summed < 2) hipLaunchKernelGGL(k_post_grid<POST_CONVIN>, dim3(pblocks), dim3(256), 0, st, p);
        GemmArgs b{};
        at_stage(15 + g);
        conv_taps(b, h->X[g], 2 * NF, G.w_out, 2 * NF, 2 * NF, 1);                     // conv_out   (layers.py:159)
        PostArgs q{items, part, nitems, 2 * NF, h->L, 0, 0, G.b_out, nullptr, h->R[G.node_in], h->R[G.node_out],
                   h->E[G.node_out], nullptr};
        q.summed = gemm(b, m.und, G.b_out, &q);                                         // gate + residual (:160-163)
        if (q.summed < 2) hipLaunchKernelGGL(k_post_grid<POST_GATE>, dim3(pblocks), dim3(256), 0, st, q);
    };
    auto dilated = [&](int d) {
        const ps_pixelcnn::Dil &D = h->dil[d];
        GemmArgs a{};
        at_stage(29 + d);
        conv_taps(a, h->R[D.node_in], R_LD, D.w, NF, NF, 2);                            // model.py:138,148
        PostArgs p{items, part, nitems, NF, h->L, 0, 0, D.b, nullptr, nullptr, h->R[D.node_out], h->E[D.node_out], nullptr};
        p.summed = gemm(a, m.dil, D.b, &p);
        if (p.summed < 2) hipLaunchKernelGGL(k_post_grid<POST_DIL>, dim3(pblocks), dim3(256), 0, st, p);
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

}  // namespace pslm

using namespace pslm;

extern "C" {

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
    launch_gemm(a, (tiles + 1) / 2, st, Tuning{});
    hipLaunchKernelGGL(k_reduce_nchw, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, partial, bias, B, Co, Cop, L, y);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

}  // extern "C"
