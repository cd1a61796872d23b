// lmconv_device.h -- device-side vocabulary shared by the translation units of the locally-masked PixelCNN engine
// (lmconv_grid.hip: whole-grid pass; lmconv_column.hip: latency form of the column launch; lmconv_tp.hip: throughput form;
// lmconv.hip: the handle and the C ABI).  The canonical arithmetic lives here: every kernel walks taps, 80-channel chunks and
// accumulation chains in ONE order and reduces PONO's statistics in ONE association order, so whole-grid and column evaluation
// agree bit for bit.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "ps_common.h"

namespace pslm {

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
constexpr int N_XCD = 8;      // gfx950: 8 XCDs, workgroup ids are dealt round-robin over them
constexpr int N_EVAL = 1 + 2 * NGATED + 4;   // stages of the whole-grid pass that have a prefix start rank (k_prefix_starts): 33

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

enum { PRO_UINIT = 0, PRO_CONVIN = 1, PRO_GATE = 2, PRO_DIL = 3 };
enum { IN_CELU = 0, IN_RAW = 1, IN_ELU = 2 };
constexpr int NST = 33;       // 14 x (conv_input, conv_out) + 4 dilated convs + nin_out
constexpr int NBR_LD = 2 * NF;

// Completion counters of the neighbour role: one per (stage, 16-column tile of the launch), each on its own 128-byte
// line -- several thousand items finish per launch, and atomics on one line are served one after the other by the
// memory side (counters packed in two lines made the neighbour role atomics-bound and every chain's polls queue behind
// them: 128 columns 84 -> 72 us, docs/LAB_NOTEBOOK.md).  A chain only watches the counters of its own tile.
constexpr int COL_CAP = 128;  // columns per launch: 4 chain XCDs x 32 CUs (larger wavefronts are split)
constexpr int MAX_TILES = COL_CAP / 16, CNT_PAD = 32 /* dwords */;
constexpr int C1_THREADS = 1024;    // latency-form workgroups
constexpr int C1_OUT_STEPS = 12;    // nin_out in the latency-form chain role: thread (o, part): part 0 = chains 0..2, part 1 = chains 3..4
constexpr int NBR_MAX_GROUPS = 4;   // work items a latency-form neighbour workgroup runs at a time, at most
constexpr int NWORK_MAX = 512;      // work-table entries the latency-form neighbour role can stage in LDS (this network: 460)
__device__ __host__ __forceinline__ size_t cnt_index(int stage, int tile) { return ((size_t)stage * MAX_TILES + tile) * CNT_PAD; }
// throughput form (k_column_tp): chain tiles of 16 columns -- the N of v_mfma_f32_16x16x4_f32 --, 64 of them per launch
constexpr int TP_COLS = 16, TP_MAX_TILES = 64, TP_COL_CAP = TP_MAX_TILES * TP_COLS;   // 1024 columns per launch
__device__ __host__ __forceinline__ size_t tp_cnt_index(int stage, int tile) { return ((size_t)stage * TP_MAX_TILES + tile) * CNT_PAD; }

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
// Bound of the in-launch waits on the neighbour role's completion counters: a hang guard, not a schedule.  A wait is normally
// over before it starts; it lasts when workgroups of the launch are not resident yet because kernels of ANOTHER stream hold their
// CUs (bench.py / driver.py run the next batch's splat under this batch's AR run: a stream of 64-thread workgroups can keep a
// 512- or 1024-thread workgroup that needs most of a CU's LDS waiting for as long as that kernel lasts, milliseconds).  Round 2's
// bounds (20 000 / 40 000 polls of >= 128 clocks: a few ms) were inside that range and expired now and then (one bench run in
// six); 2^24 polls are seconds -- still finite, so a lost workgroup ends as an error from ps_pixelcnn_status, not as a hung GPU.
constexpr int WAIT_SPINS = 1 << 24;

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
struct ChainCtl { int Co, nchain, NG, nstep; const float *wv; };
struct PostCtl { int Co, kind, has_skip, in_form, save_slot, nbr_items; const float *bias, *bias2; };
struct StoreCtl { int kind, skip_slot; float *R, *E, *X; };

__device__ __forceinline__ void store_through1(float *p, float v)
{
    asm volatile("global_store_dword %0, %1, off sc1" : : "v"(PS_G(float, p)), "v"(v) : "memory");
}

}  // namespace pslm
