// lmconv_column.hip -- column mode, LATENCY form: one location per frame per order position (the incremental AR step, models/lmconv/sample.py:54-66),
// one launch per wavefront of up to 128 independent columns.
//   neighbour role   every NEIGHBOUR-tap partial sum (slots NA, NB) of all 32 masked convs at once.  They only read finished columns of
//                    earlier order positions, so they do not depend on this position's chain and run fully parallel (MFMA; one wave =
//                    one tap of one stage x slot x 16 channels); for the first stages a launch AHEAD (k_column_la).
//   chain role       one workgroup (= one CU) per column walks the 33 stages in order.  Only the centre taps (1x1 products on the fresh
//                    activation) and the post ops are sequential; activations go stage to stage through LDS, weights stream from L2
//                    into registers.  Ends with the categorical draw.
// The throughput form of the same launch (16-column MFMA chain tiles, wavefronts of hundreds of columns) is lmconv_tp.hip.
#include "lmconv_handle.h"

namespace pslm {

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
constexpr int C1_MAXCHAIN = 800;   // 5 x 160, or 5 x 80 + 5 x 80 (conv_input + nin_skip)
constexpr int SX_LD = 2 * NF;

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

__device__ __forceinline__ int ctl_i(const int *ctl, int rec, int field) { return ((CtlInt)ctl)[rec * C1_CTL_DWORDS + field]; }
template <typename T>
__device__ __forceinline__ T *ctl_p(const int *ctl, int rec, int field)
{
    return (T *)((CtlU64)ctl)[(rec * C1_CTL_DWORDS + field) >> 1];
}
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
// each other's CUs until the bounded waits give up): one column-launching process per GPU.
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

void launch_pack_valu(const float *wc, const float *wskip, int Co, int nchain, int nstep, float *out)
{
    const int n = nstep * nchain * 4;
    hipLaunchKernelGGL(k_pack_valu, dim3((n + 255) / 256), dim3(256), 0, 0, wc, wskip, Co, nchain, nstep, out);
}
void launch_pack_valu_out(const float *wo, float *out)
{
    const int n = C1_OUT_STEPS * C1_THREADS * 4;
    hipLaunchKernelGGL(k_pack_valu_out, dim3((n + 255) / 256), dim3(256), 0, 0, wo, out);
}

// `ncols` independent columns (records rec[0..ncols)): neighbour taps of every conv and the centre-tap chains + draw, in launches of
// at most col_cap columns.  next_rec / next_ncols: the columns of the launch that FOLLOWS on this stream, when the caller knows it
// (a wavefront schedule): their first stages' neighbour slots are computed a launch ahead (nbr_role's look-ahead items).
void run_columns_la(ps_pixelcnn *h, const StepCtx *rec, int ncols, ChainArgs ca, hipStream_t st, const StepCtx *next_rec, int next_ncols)
{
    const int col_cap = h->tune.col_cap, col_ahead = h->tune.col_ahead;
    h->columns_launched = true;
    const size_t nbr_half_col = (size_t)NST * 2 * COL_CAP * NBR_LD, cnt_half_col = cnt_index(NST, 0);
    for (int done = 0; done < ncols; done += col_cap) {
        const int n = std::min(col_cap, ncols - done);
        const int tiles = (n + 15) / 16;
        // chain workgroups on XCDs 0 .. cx-1 of the first rows of 8 blocks, neighbour workgroups everywhere else, 256
        // blocks at most (one per CU, all resident)
        // (the launch holds one workgroup per CU at most: every workgroup is resident, which the in-launch waits rest on;
        // per_xcd = CUs per XCD of THIS device, 32 on a whole MI355X)
        const int per_xcd = h->n_cus / 8;
        const int cx = h->tune.chain_xcds > 0 ? std::min(8, std::max(h->tune.chain_xcds, (n + per_xcd - 1) / per_xcd)) : std::min(4, (n + per_xcd - 1) / per_xcd);
        const int chain_rows = (n + cx - 1) / cx;
        const int nbr_cus = chain_rows * (8 - cx) + (per_xcd - chain_rows) * 8;
        // the look-ahead, from one latency-form launch to the next (as in the throughput form above): was this launch prepared, and
        // what follows it -- the rest of an oversized wavefront or the caller's next wavefront, if that takes this form too
        const bool prepared = col_ahead > 0 && h->col_ahead_rec == rec + done && h->col_ahead_n == n;
        const int par = prepared ? h->col_ahead_parity : 0;
        const StepCtx *nrec = nullptr;
        int nn = 0;
        if (done + col_cap < ncols) { nrec = rec + done + col_cap; nn = std::min(col_cap, ncols - done - col_cap); }
        else if (next_rec && next_ncols > 0 && next_ncols < h->tune.tp_min_cols) { nrec = next_rec; nn = std::min(col_cap, next_ncols); }
        const bool ahead = col_ahead > 0 && nrec != nullptr && !(ca.debug & 2);
        bool la = prepared || ahead;
        for (int t = 0; t < tiles && !la; ++t)   // (k_column keeps ONE use count per tile for all stages: should a prepared launch
            la = h->col_uses_lo[0][t] != h->col_uses_hi[0][t];   // ever not have followed, the two-count form takes over)
        const int w_from = prepared ? h->col_wsplit : 0, w_upto = ahead ? h->col_wsplit : 0;
        const int tiles_next = ahead ? (nn + 15) / 16 : 1;
        const int nitems = (h->nwork - w_from) * tiles + w_upto * tiles_next;
        const int groups = h->tune.nbr_groups ? h->tune.nbr_groups : (nitems > 2 * nbr_cus ? 4 : 2);
        const int nbr_wgs = std::min(nbr_cus, (nitems + groups - 1) / groups);
        NbrArgs na{h->work, rec + done, h->nbr + par * nbr_half_col, h->nwork, h->H, h->W, h->L, n, COL_CAP, tiles, cx,
                   h->cnt + par * cnt_half_col, nbr_wgs, groups, ca.debug, h->err};
        na.w_from = w_from; na.w_upto = w_upto;
        na.ctx_next = ahead ? nrec : rec + done; na.ncols_next = ahead ? nn : 0; na.tiles_next = tiles_next;
        na.nbr_next = h->nbr + (par ^ 1) * nbr_half_col; na.cnt_next = h->cnt + (par ^ 1) * cnt_half_col;
        na.done = h->done_col; na.split = std::max(1, col_ahead);
        if (ahead) h->done_col_total += (unsigned)n;   // (every column publishes once per stage)
        na.done_target = h->done_col_total;
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
            ca.la_split = col_ahead; ca.done = h->done_col; ca.publish_upto = ahead ? col_ahead : 0;
            h->col_ahead_rec = ahead ? nrec : nullptr; h->col_ahead_n = nn; h->col_ahead_parity = par ^ 1;
            timed(h, st, TAG_CHAIN, LK_COLUMN_LA, [&]() { hipLaunchKernelGGL(k_column_la, dim3(rows * 8), dim3(C1_THREADS), 0, st, na, ca); });
        } else {   // a launch nobody prepared and that prepares nobody (a walk position by position): one set of use counts for all stages
            h->col_ahead_rec = nullptr;
            for (int t = 0; t < tiles; ++t) { h->col_uses_lo[0][t] += 1; h->col_uses_hi[0][t] += 1; }
            for (int t = 0; t < MAX_TILES; ++t) ca.tile_uses[t] = h->col_uses_hi[0][t];
            timed(h, st, TAG_CHAIN, LK_COLUMN, [&]() { hipLaunchKernelGGL(k_column, dim3(rows * 8), dim3(C1_THREADS), 0, st, na, ca); });
        }
    }
}

}  // namespace pslm
