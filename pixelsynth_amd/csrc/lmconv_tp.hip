// lmconv_tp.hip -- column mode, THROUGHPUT form (see lmconv_column.hip for the latency form and the vocabulary).
#include "lmconv_handle.h"

namespace pslm {

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
constexpr int TP_WAVES = 8, TP_THREADS = 64 * TP_WAVES;   // 8 waves: 256 registers per thread
constexpr int TP_MAXU = (50 + TP_WAVES - 1) / TP_WAVES;   // units per wave and stage at most: 7
constexpr int TP_MINU = (25 + TP_WAVES - 1) / TP_WAVES;   // ... of a 25-unit stage: 4
constexpr int XB_LD = 68;             // B-operand layout: dwords per 4-channel group (16 columns x 4 + 4 pad: conflict-free
constexpr int XB_SIZE = 40 * XB_LD;   //   for the post op's 8-byte writes and for the waves' 16-byte reads)
constexpr int SP_LD = 5 * 2 * NF + 4; // chain values of one column [j][o] (+ [j][80] of nin_skip): 800 + 4 pad
constexpr int SLOG_LD = NCLS + 4;     // logits of one column (aliases the chain values)
static_assert(TP_COLS * SLOG_LD <= TP_COLS * SP_LD, "logits alias the chain-value buffer");
__device__ __forceinline__ int xb_index(int ch, int col) { return (ch >> 2) * XB_LD + col * 4 + (ch & 3); }

__device__ __host__ constexpr int tpt_units(int type) { return type == TPT_CONVIN || type == TPT_DIL ? 25 : 50; }   // (tile, chain) units
__device__ __host__ constexpr int tpt_nu(int type) { return type == TPT_CONVIN || type == TPT_DIL ? TP_MINU : TP_MAXU; } // per wave, at most
__device__ __host__ constexpr int tpt_nh(int type) { return type == TPT_DIL ? 1 : 2; }                              // 16-byte weight loads per unit

struct TpArgs {
    // neighbour role
    const NbrWorkTp *work;
    const ColTaps *taps;      // records of this launch's columns
    float *nbr;               // [NST][2][TP_COL_CAP][NBR_LD]
    unsigned *cnt;            // [NST][TP_MAX_TILES] padded completion counters (tp_cnt_index), never reset
    int nwork, tiles, nbr_wgs;   // (tiles: the neighbour role's 16-column tiles)
    int ctiles;                 // chain tiles: of 16 columns (k_column_tp) or of 8 (k_column_tp8)
    int chain_xcds, fill_nbr, fill_cnt;   // placement (k_column_tp): XCDs that hold the chain tiles; first neighbour index of their spare CUs (or -1) and how many of them work
    // stage-affine neighbour XCDs (0: all XCDs walk all stages): nx neighbour XCDs; XCD xi owns entries xent[xi * TP_XENT_MAX + k], k < xlen[xi],
    // of which the first xlo[xi] lie below the look-ahead depth
    int affine_nx;
    const int *xent;
    int xlen[8], xlo[8];
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
    int *err;
    int debug;
    // neighbour role, items dealt on demand (round 6): one never-reset device counter per share (stage-affine XCD, or [0] for all), each on
    // its own line; dq_base[x] = what counter x stood at when this launch was enqueued (null: items dealt round-robin, as before)
    unsigned *dq;
    unsigned dq_base[8];
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
#ifdef PS_TUNING_BUILD   // timing experiments (results INVALID): column_debug bits 1024 / 2048 / 4096 / 8192 = no weight loads / no row loads / no MFMAs / no stores
    const bool dbg_noav = (a.debug & 1024) != 0, dbg_nobv = (a.debug & 2048) != 0, dbg_nomfma = (a.debug & 4096) != 0, dbg_nost = (a.debug & 8192) != 0;
#else
    constexpr bool dbg_noav = false, dbg_nobv = false, dbg_nomfma = false, dbg_nost = false;
#endif
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
            if (dbg_nobv) {
#pragma unroll
                for (int g = 0; g < 5; ++g) bv[g] = f32x4{(float)row, 1.0f, 2.0f, (float)g};
            } else if (AHEAD && NG == 5) {
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
                for (int g = 0; g < 5; ++g) av[g] = dbg_noav ? f32x4{(float)u, 1.0f, (float)g, 3.0f} : *PS_GC(f32x4, wbase + (size_t)(g0 + g) * gstride + 64 * u);
                if (!dbg_nomfma) mfma_chunk5(av, bv, acc[u]);
                else {
#pragma unroll
                    for (int g = 0; g < 5; ++g) acc[u].v[g] = acc[u].v[g] + av[g] * bv[g];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < T; ++u) tot[u] = tot[u] + chunk_total(acc[u]);
    }
    if (valid) {
        float *dst = (AHEAD ? a.nbr_next : a.nbr) + (((size_t)wk.stage * 2 + wk.half) * TP_COL_CAP + col) * NBR_LD + wk.o0 + kk * 4;
#pragma unroll
        for (int u = 0; u < T; ++u) if (!dbg_nost || tot[u][0] == 12345.0f) store_through(dst + 16 * u, tot[u]);
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
    // all XCDs together: wave gw of nw takes items gw, gw + nw, ... of (entries [w_from, nwork) x tiles, then [0, w_upto) x tiles_next);
    // stage-affine: the waves of neighbour XCD xi do the same over the entries that XCD owns (the waves of a workgroup take consecutive
    // items either way: they share the weights in L1 / L2)
    const int nx = a.affine_nx, xi = nx ? nb % nx : 0;
    const int gw = (nx ? nb / nx : nb) * TP_WAVES + wave, nw = (nx ? a.nbr_wgs / nx : a.nbr_wgs) * TP_WAVES;
    const int e_own0 = nx ? (a.w_from ? a.xlo[xi] : 0) : a.w_from, e_own1 = nx ? a.xlen[xi] : a.nwork;
    const int e_ahead1 = nx ? (a.w_upto ? a.xlo[xi] : 0) : a.w_upto;
    const int *xe = nx ? a.xent + (size_t)xi * TP_XENT_MAX : nullptr;
    const int n_own = (e_own1 - e_own0) * a.tiles;
    const int nitems = n_own + e_ahead1 * a.tiles_next;
    __shared__ unsigned sReadyTp;   // look-ahead stages some wave of this workgroup has seen published, + 1
    if (threadIdx.x == 0) sReadyTp = 0;
    __syncthreads();
    int ready_upto = -1;
    // Which items a wave takes: round-robin (wave gw of nw takes gw, gw + nw, ...), or ON DEMAND -- the items differ by a factor of ten
    // (0 to 4 open taps, 5 or 10 channel groups, one or two output tiles), a wave gets about nine of them, and the launch is as long as
    // the unluckiest wave's share; with a device-scope counter per share every wave takes the next item when it is through with its own
    // (the next index is requested while the current item runs, so the round trip is hidden).  The order in which items START is the
    // table's either way: stage-major, own entries before the next launch's.  Every wave requests exactly one index past the end: the
    // host knows what the counter stands at afterwards (run_columns_tp).
    unsigned *const dq = a.dq ? a.dq + (size_t)xi * CNT_PAD : nullptr;
    const unsigned dq_base = a.dq_base[xi];
    auto dq_next = [&]() {
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(dq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;      // (lane 0's value; made uniform where it is consumed)
    };
    unsigned fetched = dq ? dq_next() : 0u;
    for (int item = dq ? (int)((unsigned)uni((int)fetched) - dq_base) : gw; item < nitems;) {
        if (dq) fetched = dq_next();
        const bool ahead = item >= n_own;
        int witem, ctile;
        if (!ahead) {
            const int q = item / a.tiles;
            witem = e_own0 + q; ctile = item - q * a.tiles;
        } else {
            const int j = item - n_own;
            witem = j / a.tiles_next; ctile = j - witem * a.tiles_next;
        }
        if (nx) witem = xe[witem];
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
        item = dq ? (int)((unsigned)uni((int)fetched) - dq_base) : item + nw;
    }
}

// NPC = columns a wave does the post op of: 2 = tiles of 16 columns, 1 = tiles of 8 (the MFMA phase is what it is for 16 -- the N of the
// MFMA --, the post phase and the stores are half: for launches whose tiles then still fit three XCDs; run_columns_tp)
template <int NPC>
__device__ __forceinline__ void chain_role_tp(const TpArgs &a, int ctile)
{
    constexpr int CT = NPC * TP_WAVES;   // columns of a chain tile
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
#ifdef PS_TUNING_BUILD   // timing experiments (results INVALID): column_debug bits 16 / 32 / 64 / 128 = no cache stores / no post-op operand requests / no weight refill / no post op at all
    const bool dbg_nostore = (a.debug & 16) != 0, dbg_noops = (a.debug & 32) != 0, dbg_norefill = (a.debug & 64) != 0, dbg_nopost = (a.debug & 128) != 0;
#else
    constexpr bool dbg_nostore = false, dbg_noops = false, dbg_norefill = false, dbg_nopost = false;
#endif
    const int col0 = ctile * CT;
    const int ncl = min(CT, a.ncols - col0);   // columns of this tile (>= 1)
    const int tile = col0 / TP_COLS;           // the neighbour role's 16-column tile these columns lie in: its counters, its use counts
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
    bool pvalid[NPC];
    int pcol[NPC], pfr[NPC];
    size_t ploc[NPC];
    f32x2 ucur[NPC];
#pragma unroll
    for (int k = 0; k < NPC; ++k) ucur[k] = zero2;
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        pcol[k] = wave + TP_WAVES * k;
        pvalid[k] = pcol[k] < ncl;
        pfr[k] = uni(sC[pcol[k]].f);
        ploc[k] = (size_t)pfr[k] * a.L + uni(sC[pcol[k]].q);
    }
    const size_t nbr_half = (size_t)TP_COL_CAP * NBR_LD, nbr_stage = 2 * nbr_half;
    const unsigned uses_lo = a.tile_uses_lo[tile], uses_hi = a.tile_uses_hi[tile];
    // The counter is requested a stage before it is looked at.  It must stay a VECTOR value until then (round 4, read off the ISA): a
    // wave-uniform load is turned into a scalar by v_readfirstlane where it is ISSUED, i.e. the wave waits for it on the spot -- and,
    // vmcnt retiring in order, for every store and weight request in front of it; and a polling loop that reloads the value makes
    // the compiler wait for ALL outstanding memory operations at the loop's head, so the first look is peeled off the loop.  The
    // lane offset below is zero, but not to the compiler.
    int vzero = 0;
    asm volatile("" : "+v"(vzero));
    auto counter = [&](int k) { return __hip_atomic_load(a.cnt + tp_cnt_index(k, tile) + vzero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // `have`: the counter as requested a stage earlier (normally past the target already); bounded
    auto wait_counter = [&](unsigned have, int k, unsigned items_per_tile) {
        if (a.debug & 1) return;
        const unsigned need = (k < a.split ? uses_lo : uses_hi) * items_per_tile;
        if (__builtin_amdgcn_ballot_w64((int)(have - need) < 0) != 0ull) {
            int spins = 0;
            do {
                if (++spins > WAIT_SPINS) { if (lane == 0) *a.err = 1; break; }
                __builtin_amdgcn_s_sleep(2);
                have = counter(k);
            } while (__builtin_amdgcn_ballot_w64((int)(have - need) < 0) != 0ull);
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
    auto emit2 = [&](const f32x2 (&y)[NPC], const f32x2 (&g)[NPC], const f32x2 (&skip)[NPC], auto KINDc, auto SKIPc, int in_form, int save_slot,
                     const StoreCtl &sc) {
        constexpr int kind = decltype(KINDc)::value;
        constexpr bool has_skip = decltype(SKIPc)::value;
        float mean[NPC], inv[NPC];
        f32x2 d[NPC];
#pragma unroll
        for (int k = 0; k < NPC; ++k) mean[k] = pono_mean(pono_total(y[k], own));
#pragma unroll
        for (int k = 0; k < NPC; ++k) d[k] = y[k] - mean[k];
#pragma unroll
        for (int k = 0; k < NPC; ++k) inv[k] = pono_inv(pono_total(d[k] * d[k], own));
        if (!own) return;
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
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
                if (!dbg_nostore) {
                    store_through2(sc.X + ploc[k] * (2 * NF) + c2, ep);
                    store_through2(sc.X + ploc[k] * (2 * NF) + NF + c2, en);
                }
            } else {
                if (!dbg_nostore) {
                    store_through2(sc.R + ploc[k] * R_LD + c2, out);
                    store_through2(sc.E + ploc[k] * (2 * NF) + c2, ep);
                    store_through2(sc.E + ploc[k] * (2 * NF) + NF + c2, en);
                }
                ucur[k] = out;
                if (save_slot >= 0) *(f32x2 *)(&sU[save_slot][col][c2]) = out;
            }
        }
    };
    // concat_elu(u_k) of the saved u the NEXT stage's nin_skip reads
    auto stage_skip_input = [&](int skip_slot) {
        if (skip_slot < 0 || !own) return;
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
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
#ifdef PS_TP_TRACE_BUILD
    int trace_s = 0;
    const int trace_wave = a.debug >> 8;   // column_debug = 256 * wave (+ mode): the wave whose stamps are kept
#define TP_STAMP(slot) do { if (a.trace && ctile == 0 && t == 64 * trace_wave) a.trace[s * 8 + (slot)] = clock64(); } while (0)
#define TP_STAMP2(slot, dep) do { if (a.trace && ctile == 0 && t == 64 * trace_wave && (dep)) a.trace[trace_s * 8 + (slot)] = clock64(); } while (0)
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
                    if (u < TP_MAXU && u < nnu && !dbg_norefill) W[u].a0 = *PS_GC(f32x4, nbase + (size_t)(2 * u) * 256);
            }
        } else {
#pragma unroll
            for (int u = 0; u < TP_MAXU; ++u)
                if (u < nnu && !dbg_norefill) W[u].a0 = *PS_GC(f32x4, nbase + (size_t)(2 * u) * 256);
        }
        TP_STAMP2(5, true);
#pragma unroll
        for (int u = 0; u < TP_MAXU; ++u)
            if (u < nnu && !dbg_norefill) W[u].a1 = *PS_GC(f32x4, nbase + (size_t)(2 * u + 1) * 256);
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
        f32x2 ob = zero2, obg = zero2, ob2 = zero2, ona[NPC], onb[NPC], onag[NPC], onbg[NPC];
        auto request_operands = [&]() {
            wait_counter(cnt_have, s, items);
            cnt_have = counter(min(s + 1, NST - 2));             // looked at a stage later
            TP_STAMP(1);
            if (dbg_noops) return;
            ob = plain(pc.bias + c2);
            if (kind == PRO_GATE) obg = plain(pc.bias + NF + c2);
            if (has_skip) ob2 = plain(pc.bias2 + c2);
#pragma unroll
            for (int k = 0; k < NPC; ++k) {
                const float *nb = a.nbr + (size_t)s * nbr_stage + (size_t)(col0 + (pvalid[k] ? pcol[k] : 0)) * NBR_LD + c2;
                ona[k] = fresh(nb);
                onb[k] = fresh(nb + nbr_half);
                if (kind == PRO_GATE) { onag[k] = fresh(nb + NF); onbg[k] = fresh(nb + nbr_half + NF); }
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
        f32x2 y[NPC], g[NPC], skip[NPC];
#pragma unroll
        for (int k = 0; k < NPC; ++k) { g[k] = zero2; skip[k] = zero2; }
        if (dbg_nopost) { cvA = cvB; cvB = cvC; lds_barrier(); return; }
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
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
        f32x2 y[NPC];
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
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
        f32x2 z2[NPC];
#pragma unroll
        for (int k = 0; k < NPC; ++k) z2[k] = zero2;
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
    for (int k = 0; k < NPC; ++k) {
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


// chain_xcds = 0: blocks [0, nbr_wgs) neighbour role (dispatched first: the chain tiles wait for their items), the blocks after them one chain tile each.
// chain_xcds = cx > 0 (speed only; block b runs on XCD b % 8): the chain tiles are the blocks on XCDs 0 .. cx-1, whose L2s then
// hold the 2.8 MB of centre-tap weights instead of sharing their bandwidth with the neighbour role's operand stream; every
// other block is a neighbour workgroup (the spare CUs of the chain XCDs too when fill is set).
template <int NPC>
__device__ __forceinline__ void column_tp_body(const TpArgs &a)
{
    const int b = blockIdx.x, cx = a.chain_xcds;
    if (cx == 0) {
        if (b < a.nbr_wgs) { if ((a.debug & 3) != 3) nbr_role_tp(a, b); }
        else if ((a.debug & 3) != 2) chain_role_tp<NPC>(a, b - a.nbr_wgs);
        return;
    }
    const int x = b & 7, slot = b >> 3;
    if (x < cx) {
        const int tile = slot * cx + x;
        if (tile < a.ctiles) { if ((a.debug & 3) != 2) chain_role_tp<NPC>(a, tile); }
        else if (tile - a.ctiles < a.fill_cnt && (a.debug & 3) != 3) nbr_role_tp(a, a.fill_nbr + (tile - a.ctiles));
    } else if ((a.debug & 3) != 3) {
        nbr_role_tp(a, slot * (8 - cx) + (x - cx));
    }
}
__global__ __launch_bounds__(TP_THREADS) void k_column_tp(TpArgs a) { column_tp_body<2>(a); }    // chain tiles of 16 columns
__global__ __launch_bounds__(TP_THREADS) void k_column_tp8(TpArgs a) { column_tp_body<1>(a); }   // chain tiles of 8 columns

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

int tp_weights_floats(int type) { return TP_WAVES * tpt_nu(type) * 2 * 64 * 4; }
void launch_pack_tp(const float *wc, const float *ws, int Co, int NG, int type, float *out)
{
    const int n = TP_WAVES * tpt_nu(type) * 2 * 64;
    hipLaunchKernelGGL(k_pack_tp, dim3((n + 255) / 256), dim3(256), 0, 0, wc, ws, Co, NG, type, out);
}

// `ncols` independent columns (records rec[0 .. ncols)) as throughput-form launches: 16-column chain tiles, up to TP_COL_CAP columns
// per launch.  next_rec / next_ncols: the columns of the launch that FOLLOWS on this stream, when the caller knows it (a wavefront
// schedule): their first stages' neighbour slots are computed a launch ahead (nbr_role_tp).
void run_columns_tp(ps_pixelcnn *h, const StepCtx *rec, int ncols, const ChainArgs &ca, hipStream_t st, const StepCtx *next_rec, int next_ncols)
{
    {
        const int tp_ahead = h->tune.tp_ahead;
        h->columns_launched = true;
        TpArgs ta{};
        ta.work = h->work_tp; ta.nwork = h->nwork_tp;
        ta.done = h->done_tp; ta.split = tp_ahead;
        const size_t nbr_half_buf = (size_t)NST * 2 * TP_COL_CAP * NBR_LD, cnt_half_buf = tp_cnt_index(NST, 0);
        ta.ctl1 = h->ctl1; ta.uinit_w = h->uinit_w; ta.uinit_b = h->uinit_b; ta.codes_in = ca.codes_in;
        ta.out_w = h->out_w; ta.out_b = h->out_b; ta.L = h->L;
        ta.codes = ca.codes; ta.region = ca.region; ta.forced = ca.forced; ta.uniforms = ca.uniforms;
        ta.out_logits = ca.out_logits; ta.step_logits = ca.step_logits; ta.temperature = ca.temperature;
        ta.err = h->err; ta.debug = ca.debug;
        ta.trace = h->tp_trace;
#ifdef PS_TP_TRACE_BUILD
        static const int trace_sel = getenv("PS_TP_TRACE_LAUNCH") ? atoi(getenv("PS_TP_TRACE_LAUNCH")) : -1;   // tuning builds: stamps of that launch of the run only
#else
        const int trace_sel = -1;
#endif
        const int cap = std::min(TP_COL_CAP, std::max(TP_COLS, (h->n_cus / 2) * TP_COLS));   // at least half of the CUs to the neighbour role
        const ColTaps *taps = h->taps + (rec - h->ctx);
        for (int done = 0; done < ncols; done += cap) {
            const int n = std::min(cap, ncols - done);
            const int tiles = (n + TP_COLS - 1) / TP_COLS;
            // chain tiles of 8 columns (k_column_tp8) while they fit tp_ct8_xcds XCDs: the post phase and the stores of a stage are half
            const int rows_ = h->n_cus / 8;
            const bool ct8 = h->tune.tp_ct8_xcds > 0 && h->tune.tp_xcds != 0 && h->xcd_even && (n + 7) / 8 <= h->tune.tp_ct8_xcds * rows_ && n <= h->tune.tp_ct8_cols;
            const int ctiles = ct8 ? (n + 7) / 8 : tiles;
            ta.taps = taps + done; ta.ctx = rec + done; ta.ncols = n; ta.tiles = tiles; ta.ctiles = ctiles;
            // did the launch in front prepare this one?  then its slots of the stages [0, split) are in the buffers of `par`
            const bool prepared = tp_ahead > 0 && h->ahead_rec == rec + done && h->ahead_n == n;
            const int par = prepared ? h->ahead_parity : 0;
            ta.nbr = h->nbr_tp + par * nbr_half_buf; ta.cnt = h->cnt_tp + par * cnt_half_buf;
            ta.nbr_next = h->nbr_tp + (par ^ 1) * nbr_half_buf; ta.cnt_next = h->cnt_tp + (par ^ 1) * cnt_half_buf;
            ta.w_from = prepared ? h->tp_wsplit : 0;
            // and what follows this one: the rest of an oversized wavefront, or the caller's next wavefront if it takes this form
            const StepCtx *nrec = nullptr;
            int nn = 0;
            if (done + cap < ncols) { nrec = rec + done + cap; nn = std::min(cap, ncols - done - cap); }
            else if (next_rec && next_ncols >= h->tune.tp_min_cols) { nrec = next_rec; nn = std::min(cap, next_ncols); }
            const bool ahead = tp_ahead > 0 && nrec != nullptr && !(ca.debug & 2);
            ta.w_upto = ahead ? h->tp_wsplit : 0;
            ta.taps_next = ahead ? h->taps + (nrec - h->ctx) : ta.taps;
            ta.ncols_next = ahead ? nn : 0;
            ta.tiles_next = ahead ? (nn + TP_COLS - 1) / TP_COLS : 1;
            ta.publish_upto = ahead ? tp_ahead : 0;
            if (ahead) h->done_total += (unsigned)ctiles;
            ta.done_target = h->done_total;
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
            if (h->tune.tp_xcds != 0 && h->xcd_even && ctiles <= 4 * rows) {
                const int cx = h->tune.tp_xcds > 0 ? std::max(h->tune.tp_xcds, (ctiles + rows - 1) / rows) : (ctiles + rows - 1) / rows;
                ta.chain_xcds = std::min(cx, 7);
                const int spare = ta.chain_xcds * rows - ctiles;
                const int use_rows = rows;
                ta.nbr_wgs = (8 - ta.chain_xcds) * use_rows;
                ta.fill_nbr = -1; ta.fill_cnt = 0;
                const bool affine = h->tune.tp_affine && h->tp_xent && h->nwork_tp <= TP_XENT_MAX;   // (a longer work table never filled tp_xent)
                // spare CUs of the chain XCDs as neighbour workgroups; stage-affine: as many as deal evenly to the neighbour XCDs' shares
                // (workgroup nb works for share nb % nx wherever it runs -- these fetch their weights through a chain XCD's L2)
                const int fill = !h->tune.tp_fill || use_rows != rows ? 0 : affine ? spare / (8 - ta.chain_xcds) * (8 - ta.chain_xcds) : spare;
                if (fill > 0) { ta.fill_nbr = ta.nbr_wgs; ta.fill_cnt = fill; ta.nbr_wgs += fill; }
                ta.affine_nx = 0;
                if (affine) {   // (neighbour block nb sits on neighbour XCD nb % nx)
                    const int nx = 8 - ta.chain_xcds;
                    ta.affine_nx = nx;
                    ta.xent = h->tp_xent + (size_t)nx * 8 * TP_XENT_MAX;
                    for (int x = 0; x < 8; ++x) { ta.xlen[x] = h->tp_xlen[nx][x]; ta.xlo[x] = h->tp_xlo[nx][x]; }
                }
                grid = use_rows * 8;
            } else {
                ta.chain_xcds = 0; ta.fill_nbr = -1; ta.fill_cnt = 0; ta.affine_nx = 0;
                ta.nbr_wgs = std::max(1, std::min(h->n_cus - ctiles, (h->nwork_tp * tiles + TP_WAVES - 1) / TP_WAVES));
                grid = ta.nbr_wgs + ctiles;
            }
            // items on demand (tune.tp_dequeue): where every share's counter stands now, and where it will stand when this launch is through --
            // a share's items + one request past the end by each of its waves (exactly what nbr_role_tp does)
            ta.dq = nullptr;
            if (h->tune.tp_dequeue && h->dq_tp) {
                ta.dq = h->dq_tp;
                const int nx = ta.affine_nx;
                for (int x = 0; x < (nx ? nx : 1); ++x) {
                    const int e_own0 = nx ? (ta.w_from ? ta.xlo[x] : 0) : ta.w_from, e_own1 = nx ? ta.xlen[x] : ta.nwork;
                    const int e_ahead1 = nx ? (ta.w_upto ? ta.xlo[x] : 0) : ta.w_upto;
                    const int nitems = (e_own1 - e_own0) * ta.tiles + e_ahead1 * ta.tiles_next;
                    int wgs = ta.nbr_wgs;
                    if (nx) { wgs = 0; for (int nb = 0; nb < ta.nbr_wgs; ++nb) wgs += nb % nx == x; }
                    ta.dq_base[x] = h->dq_total[x];
                    h->dq_total[x] += (unsigned)(nitems + wgs * TP_WAVES);
                }
            }
            ta.trace = (trace_sel < 0 || trace_sel == h->tp_launch_no) ? h->tp_trace : nullptr;
            h->tp_launch_no += 1;
            timed(h, st, TAG_CHAIN, ct8 ? LK_COLUMN_TP8 : LK_COLUMN_TP, [&]() {
                if (ct8) hipLaunchKernelGGL(k_column_tp8, dim3(grid), dim3(TP_THREADS), 0, st, ta);
                else hipLaunchKernelGGL(k_column_tp, dim3(grid), dim3(TP_THREADS), 0, st, ta);
            });
        }
    }
}

}  // namespace pslm
