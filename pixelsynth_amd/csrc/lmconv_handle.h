// lmconv_handle.h -- the ps_pixelcnn handle and what the translation units of the engine share on the HOST side: the tables a handle
// owns, its tuning values, and the launchers each unit exports to the C ABI in lmconv.hip.
#pragma once
#include "lmconv_device.h"
#include "../../include/pixelsynth_hip_debug.h"

namespace pslm {

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

// a work item of the throughput neighbour role = (stage, slot NA|NB, T x 16 output channels from o0)
struct __attribute__((aligned(16))) NbrWorkTp {
    const float *w;   // packed weights of the conv [taps][NG*4][Co_pad][4]
    const float *in;  // cache the taps gather from
    int stage, half, o0, T;
    int NG, Co_pad, in_ld, kind /* 0 = und, 1 = dil */;
};
static_assert(sizeof(NbrWorkTp) == 48, "three 16-byte loads");

// Stage types of the throughput chain role (what fixes a stage's unit list): conv_input, conv_input + nin_skip, conv_out, dilated conv
enum { TPT_CONVIN = 0, TPT_CONVIN_SKIP = 1, TPT_CONVOUT = 2, TPT_DIL = 3 };

constexpr int TP_XENT_MAX = 256;   // work-table entries of the throughput form (this network: 248)

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
    int debug;                 // tuning builds only (Tuning::column_debug): 1 = chains do not wait for the neighbour slots, 2 = no chains,
                               // 3 = no neighbour role and no waiting
    // look-ahead form (k_column_la, chain_role<FPW, true>): the slots of the stages below `la_split` were computed by the launch in
    // front (use counts uses_lo), the others by this one (uses_hi); `nbr` / `cnt` are the halves of this launch's parity; the
    // columns publish, stage by stage, that the input of stage k is in memory (`done`) for the neighbour role's look-ahead items
    int la_split;
    unsigned uses_lo[MAX_TILES], uses_hi[MAX_TILES];
    unsigned *done;
    int publish_upto;
};

// Tuning values of a handle.  Defaults are what was measured best on one MI355X; ps_pixelcnn_create overrides
// them ONCE from PS_<NAME> environment variables (tuning_table in lmconv.hip -- the only place the engine reads the environment),
// and ps_pixelcnn_set_tuning (include/pixelsynth_hip_debug.h) sets them by name on a live handle: the parity tests run every launch
// form of the whole-grid pass inside one process.  Nothing here changes results -- every form is bit-identical (tested).
struct Tuning {
    int gemm_merge_min = 8192;   // (tile, channel block) pairs from which one k_gemm wave walks all slots of its tile
    int gemm_wg_min = 256;       // item tiles from which the whole-grid products take a workgroup form: k_gemm_wg (rows through LDS, post op fused --
                                 // 16 views: 4.06 against 4.16 ms per step through k_gemm + k_post_grid, round 6) ...
    int gemm_ws_min = 1024;      // ... and from which on the forms of `gemm_ws` (weights through LDS, 64 items per workgroup; 16 views: 4.42 ms)
    int wg_ti_out = 1, wg_ti_in = 2, wg_ti_dil = 2;   // item tiles per k_gemm_wg workgroup: conv_out / conv_input / dilated
    int gemm_ws = 7;             // bits 0 / 1 / 2 = conv_out / conv_input / dilated: the workgroup form with the WEIGHTS shared through LDS (k_gemm_ws, 64 items per workgroup) instead of k_gemm_wg --
                                 // bit-identical; slower on the round-5 prefix of one first step per batch (236 / 157 / 93 us against 219 / 146 / 89),
                                 // faster once the pass takes every frame up to ITS first sampled position (13.7 -> 12.9 ms per step): the default since
    int item_sort = 2;           // whole-grid products: items grouped by their set of open taps (round 5).  0 = natural (frame, rank)
                                 // order, 1 = one sort over all frames, 2 = one per XCD share of the frames (the rows of a frame stay in one L2)
    int prefix_full = 0;         // 1: the prefix pass evaluates every item (no dependency-cone elimination)
    int prefix_cone_force = 0;   // 1: keep the elimination on when the caller asks for logits (parity tests: walked locations only)
    int tp_ahead = 16;           // stages [0, tp_ahead) of a throughput-form launch are computed by the launch in front of it (0: off)
    int col_ahead = 16;          // the same for the latency form (0: off -- k_column as before)
    int tp_min_cols = 2 * COL_CAP + 1;   // a wavefront of up to 256 columns is two latency-form launches rather than one throughput-form launch
    int tp_xcds = -1;            // 0 = chain tiles anywhere, -1 = on as few XCDs as hold them, n = on at least n XCDs
    int tp_fill = 0;             // neighbour workgroups on the spare CUs of the chain XCDs
    int tp_affine = 1;           // throughput form: every neighbour XCD owns a fixed share of the STAGES (its ~3 MB of their weights stay in its L2
                                 // from launch to launch) instead of all XCDs walking all stages together: 378 -> 162 MB per launch at the L2's memory
                                 // side, the launch as long as before (122 us at 128 views; 151 -> 155 us at 256, where the neighbour role is the bound)
    int tp_dequeue = 1;          // throughput form, neighbour role: items taken on demand from a device counter per share instead of dealt round-robin
                                 // (round 6: k_column_tp8 115.5 -> 113.7 us per launch, 12.31 -> 12.23 ms per step; scheduling only, bit-identical)
    int tp_ct8_cols = 1024;      // ... and that have at most this many columns (beyond, the neighbour role on the fewer CUs left to it is the bound)
    int tp_ct8_xcds = 3;         // throughput form: chain tiles of 8 columns (k_column_tp8) for launches whose tiles then fit this many XCDs (0: never)
    int col_cap = COL_CAP;       // columns per latency-form launch
    int chain_xcds = 0;          // latency form: XCDs that hold chain workgroups (0 = automatic)
    int nbr_groups = 0;          // latency form: work items a neighbour workgroup runs at a time (0 = automatic)
    int column_debug = 0;        // settable in tuning builds only (-DPS_TUNING_BUILD): timing experiments whose results are INVALID
};

}  // namespace pslm

// ------------------------------------------------------------------------------------------
// the handle
// ------------------------------------------------------------------------------------------
struct ps_pixelcnn {
    int H = 0, W = 0, L = 0, maxF = 0;
    std::vector<void *> allocs;
    struct Gated {
        float *w_in, *b_in, *w_out, *b_out, *w_skip, *b_skip;
        int node_in, node_skip, node_out;
    } gated[pslm::NGATED];
    struct Dil { float *w, *b; int node_in, node_out; } dil[4];
    float *uinit_w = nullptr, *uinit_b = nullptr, *out_w = nullptr, *out_b = nullptr;
    float *R[pslm::NNODE], *E[pslm::NNODE], *X[pslm::NGATED];
    float *partial = nullptr;       // whole-grid slots [4][maxF*L][160]
    float *nbr = nullptr;           // column mode: neighbour slots [2][NST][2][COL_CAP][160]
    float *col_logits = nullptr;
    pslm::StepCtx *ctx = nullptr;   // column records of a run, [maxF * L]
    int32_t *pstart = nullptr;      // (N_EVAL, F) first rank of the prefix anyone reads, per stage and frame (k_prefix_starts)
    // items of a whole-grid pass grouped by open-tap set (k_perm_*, lmconv_grid.hip): [2 mask kinds][maxF * L] each
    int32_t *perm = nullptr;        // position -> natural item index (frame-local: fl * npre + rank)
    int2 *permq = nullptr;          // the same as (item, location) pairs
    uint32_t *perm_sorted = nullptr;   // scratch: (key << 12 | rank) of every frame, sorted
    int32_t *perm_cnt = nullptr;    // scratch: [2][512 * maxF] run lengths -> first positions
    int32_t *perm_tsum = nullptr;   // scratch: [2][maxF] totals of that table's tiles of 1024 entries
    int *ctl1 = nullptr;            // the chain roles' control records
    unsigned *cnt = nullptr;        // [2][NST][MAX_TILES] padded completion counters of the neighbour role, never reset
    int *err = nullptr;             // device flag: a bounded wait of a column launch ran out
    pslm::NbrWork *work = nullptr;
    int nwork = 0;
    std::vector<int> work_stage;    // stage of every entry of `work` / `work_tp` (entries are stage-major): where a look-ahead depth splits them
    // throughput form (k_column_tp): launches of at least tune.tp_min_cols columns
    pslm::NbrWorkTp *work_tp = nullptr;
    int nwork_tp = 0;
    std::vector<int> work_tp_stage;
    std::vector<double> work_tp_cost;   // relative MFMA work of every entry (output tiles x channel groups)
    float *nbr_tp = nullptr;        // neighbour slots [2][NST][2][TP_COL_CAP][160]
    unsigned *cnt_tp = nullptr;     // [2][NST][TP_MAX_TILES] padded completion counters, never reset
    // look-ahead of the neighbour role (nbr_role_tp): slots, counters and their targets are double-buffered by launch parity
    unsigned tile_uses_tp_lo[2][pslm::TP_MAX_TILES] = {}, tile_uses_tp_hi[2][pslm::TP_MAX_TILES] = {};
    unsigned *done_tp = nullptr;    // [NST] padded: chain tiles that have published the input of stage k, never reset
    unsigned *dq_tp = nullptr;      // [8] padded: items the neighbour role's shares have taken (tune.tp_dequeue), never reset
    unsigned dq_total[8] = {};      // what they will stand at when every launch enqueued so far is through
    unsigned done_total = 0;        // what they stand at when every publishing launch so far is through
    int tp_wsplit = 0;              // first entry of work_tp whose stage is >= tune.tp_ahead
    // stage-affine neighbour XCDs (tune.tp_affine): for every count nx of neighbour XCDs, the work-table entries of XCD xi in table
    // order -- device [9][8][TP_XENT_MAX]; how many there are and how many of them lie below tp_wsplit (host)
    int *tp_xent = nullptr;
    int tp_xlen[9][8] = {}, tp_xlo[9][8] = {};
    const pslm::StepCtx *ahead_rec = nullptr;   // the launch the last one prepared: its first record, its columns, the parity it wrote to
    int ahead_n = 0, ahead_parity = 0;
    int tp_launch_no = 0;           // throughput-form launches of the current run so far (tuning builds: which launch is traced)
    // the same look-ahead for the latency form (k_column_la; from one latency-form launch to the next): `nbr` and `cnt` hold two halves
    unsigned col_uses_lo[2][pslm::MAX_TILES] = {}, col_uses_hi[2][pslm::MAX_TILES] = {};
    unsigned *done_col = nullptr;   // [NST] padded: columns that have published the input of stage k, never reset
    unsigned done_col_total = 0;
    int col_wsplit = 0;             // first entry of `work` whose stage is >= tune.col_ahead
    const pslm::StepCtx *col_ahead_rec = nullptr;
    int col_ahead_n = 0, col_ahead_parity = 0;
    bool columns_launched = false;  // a column launch has used the never-reset counters: the look-ahead depths are fixed from then on
    pslm::ColTaps *taps = nullptr;  // neighbour rows of the columns of a run, [maxF * L]
    unsigned long long *tp_trace = nullptr;   // tuning builds: stamps of the last k_column_tp launch (ps_pixelcnn_debug_cache what 4)
    int n_cus = 256;                // compute units of the device: workgroups of a column launch that are resident together
    bool xcd_even = true;           // n_cus is an even share of the 8 XCDs of a whole MI355X (block b runs on XCD b % 8)
    pslm::Tuning tune;
    int env_col_cap = 0;            // col_cap as asked for (environment at creation or set_tuning); re-applied when the compute-unit count changes
    // bench.py profiling aid (ps_pixelcnn_time_column_step): event pair around every launch, by kernel tag
    struct ProfRec { int tag; hipEvent_t e0, e1; int wave; int kind; };
    std::vector<ProfRec> *prof = nullptr;
    int prof_wave = 0;            // the wavefront whose launches are being enqueued (timed runs)
    // which kernel a launch was (pslm::LaunchKind): counted always (ps_pixelcnn_launch_counts: tests assert which forms a run took),
    // event-timed between ps_pixelcnn_profile_begin / _end (bench.py: `roofline.kernels`)
    long long launch_count[16] = {};
    std::vector<ProfRec> prof_own;
    double flops_nbr = 0.0, flops_chain = 0.0, wbytes_nbr = 0.0, wbytes_chain = 0.0;  // dense work of one step, per frame
};

namespace pslm {

template <typename T>
inline int dev_alloc(ps_pixelcnn *h, T **p, size_t count)
{
    void *d = nullptr;
    PS_HIP_CHECK(hipMalloc(&d, count * sizeof(T)));
    h->allocs.push_back(d);
    *p = (T *)d;
    return PS_OK;
}

inline int upload(ps_pixelcnn *h, float **p, const float *src, size_t count)
{
    if (int rc = dev_alloc(h, p, count)) return rc;
    PS_HIP_CHECK(hipMemcpy(*p, src, count * sizeof(float), hipMemcpyHostToDevice));
    return PS_OK;
}

struct Masks { const float *init, *und, *dil; };

enum { TAG_NBR = 0, TAG_CHAIN = 1, TAG_GRID = 2 };
// the kernels that carry the matrix work of the path, as launch_count / the profile see them (names: lmconv.hip launch_kind_name)
enum LaunchKind { LK_COLUMN = 0, LK_COLUMN_LA, LK_COLUMN_TP, LK_COLUMN_TP8, LK_GEMM, LK_GEMM_WG, LK_GEMM_WS_OUT, LK_GEMM_WS_IN, LK_GEMM_WS_DIL, LK_N };

template <typename Fn>
inline void timed(ps_pixelcnn *h, hipStream_t st, int tag, int kind, Fn &&launch)
{
    if (h) h->launch_count[kind] += 1;
    if (!h || !h->prof) { launch(); return; }
    ps_pixelcnn::ProfRec r{tag, nullptr, nullptr, h->prof_wave, kind};
    (void)hipEventCreate(&r.e0);
    (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, st);
    launch();
    (void)hipEventRecord(r.e1, st);
    h->prof->push_back(r);
}

inline int pad16(int v) { return (v + 15) / 16 * 16; }

// ---- what the translation units export to each other (all in namespace pslm) ----
// lmconv_grid.hip: whole-grid evaluation (reference-faithful forward; cache build before the column steps).  logits: null (caches only),
// (F,512,H,W) when nchw, else (F*L,512) by location.  With an order the pass covers ranks [0, npre) of frames [f0, f0 + nf) -- with
// pend (device, (F)) ranks [0, pend[f]) of frame f, pend[f] <= npre (per-frame prefixes).
void run_grid(ps_pixelcnn *h, int F, const int32_t *codes, const Masks &m, float *logits, bool nchw, hipStream_t st,
              const int32_t *order = nullptr, int npre = -1, int f0 = 0, int nf = -1, const int32_t *pend = nullptr);
// lmconv_column.hip: `ncols` independent columns (records rec[0 .. ncols)) as latency-form launches of at most col_cap columns;
// next_rec / next_ncols: the columns of the launch that FOLLOWS on this stream, when the caller knows it (look-ahead)
void run_columns_la(ps_pixelcnn *h, const StepCtx *rec, int ncols, ChainArgs ca, hipStream_t st, const StepCtx *next_rec, int next_ncols);
void launch_pack_valu(const float *wc, const float *wskip, int Co, int nchain, int nstep, float *out);
void launch_pack_valu_out(const float *wo, float *out);
// lmconv_tp.hip: the same as throughput-form launches (16-column chain tiles, up to TP_COL_CAP columns per launch)
void run_columns_tp(ps_pixelcnn *h, const StepCtx *rec, int ncols, const ChainArgs &ca, hipStream_t st, const StepCtx *next_rec, int next_ncols);
int tp_weights_floats(int type);   // floats of a stage's centre-tap (+ nin_skip) weights in the throughput chain role's own order
void launch_pack_tp(const float *wc, const float *ws, int Co, int NG, int type, float *out);
#ifdef PS_WG_TRACE_BUILD
void *wg_trace_symbol(int what);   // k_gemm_wg's stamp arrays (lmconv_grid.hip), tuning builds
#endif

}  // namespace pslm
