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
#include "lmconv_handle.h"

namespace pslm {

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
// host: weight packing
// ------------------------------------------------------------------------------------------

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

// ------------------------------------------------------------------------------------------
// tuning values by name (Tuning, lmconv_handle.h): the environment is read HERE, once per handle, and nowhere else in the engine
// ------------------------------------------------------------------------------------------
struct TuningEntry { const char *key; int Tuning::*field; int lo, hi; };
const TuningEntry tuning_table[] = {
    {"gemm_merge_min", &Tuning::gemm_merge_min, 0, 1 << 30}, {"gemm_wg_min", &Tuning::gemm_wg_min, 1, 1 << 30}, {"gemm_ws_min", &Tuning::gemm_ws_min, 1, 1 << 30},
    {"wg_ti_out", &Tuning::wg_ti_out, 1, 2}, {"wg_ti_in", &Tuning::wg_ti_in, 2, 4}, {"wg_ti_dil", &Tuning::wg_ti_dil, 2, 4},
    {"item_sort", &Tuning::item_sort, 0, 2}, {"gemm_ws", &Tuning::gemm_ws, 0, 7},
    {"prefix_full", &Tuning::prefix_full, 0, 1}, {"prefix_cone_force", &Tuning::prefix_cone_force, 0, 1},
    {"tp_ahead", &Tuning::tp_ahead, 0, NST - 2}, {"col_ahead", &Tuning::col_ahead, 0, NST - 4},
    {"tp_min_cols", &Tuning::tp_min_cols, 1, 1 << 30}, {"tp_xcds", &Tuning::tp_xcds, -1, 7}, {"tp_fill", &Tuning::tp_fill, 0, 1},
    {"tp_affine", &Tuning::tp_affine, 0, 1}, {"tp_dequeue", &Tuning::tp_dequeue, 0, 1}, {"tp_ct8_xcds", &Tuning::tp_ct8_xcds, 0, 4}, {"tp_ct8_cols", &Tuning::tp_ct8_cols, 0, 1024},
    {"col_cap", &Tuning::col_cap, 1, COL_CAP}, {"chain_xcds", &Tuning::chain_xcds, 0, 8}, {"nbr_groups", &Tuning::nbr_groups, 0, NBR_MAX_GROUPS},
#ifdef PS_TUNING_BUILD   // timing experiments whose results are INVALID (1: chains do not wait for the neighbour slots, 2: no chains, 3: no
    {"column_debug", &Tuning::column_debug, 0, 1 << 20},   // neighbour role and no waiting; + 256 x the traced wave): tuning builds only
#endif
};
const TuningEntry *find_tuning(const char *key)
{
    for (const TuningEntry &e : tuning_table)
        if (strcmp(e.key, key) == 0) return &e;
    return nullptr;
}
// a value inside [lo, hi] that no launch form exists for: k_gemm_wg has conv_input / dilated kernels for 2 and 4 item tiles only
bool tuning_value_ok(const TuningEntry &e, int v)
{
    if (v < e.lo || v > e.hi) return false;
    if (e.field == &Tuning::wg_ti_in || e.field == &Tuning::wg_ti_dil) return v == 2 || v == 4;
    return true;
}
// PS_<KEY in upper case>, e.g. PS_TP_AHEAD=8 (a value no form exists for is ignored)
void tuning_from_env(Tuning &t)
{
    for (const TuningEntry &e : tuning_table) {
        std::string name = "PS_";
        for (const char *c = e.key; *c; ++c) name += (char)toupper(*c);
        if (const char *v = getenv(name.c_str())) {
            const int x = std::min(e.hi, std::max(e.lo, atoi(v)));
            if (tuning_value_ok(e, x)) t.*(e.field) = x;
        }
    }
}
// where the look-ahead depths split the (stage-major) work tables
void apply_look_ahead(ps_pixelcnn *h)
{
    h->col_wsplit = 0;
    while (h->col_wsplit < h->nwork && h->work_stage[h->col_wsplit] < h->tune.col_ahead) ++h->col_wsplit;
    h->tp_wsplit = 0;
    while (h->tp_wsplit < h->nwork_tp && h->work_tp_stage[h->tp_wsplit] < h->tune.tp_ahead) ++h->tp_wsplit;
    // Stage-affine neighbour XCDs: the stages dealt to nx XCDs, heaviest first to the least loaded, separately below and above the
    // look-ahead depth (a launch computes the stages >= tp_ahead for itself and the stages below it for the next launch: both parts
    // should be even); an XCD's entries stay in table order, so it still walks its stages in the order the chain tiles need them.
    if (h->tp_xent && h->nwork_tp <= TP_XENT_MAX) {
        std::vector<int> tab((size_t)9 * 8 * TP_XENT_MAX, 0);
        std::vector<double> cost(NST, 0.0);
        for (int e = 0; e < h->nwork_tp; ++e) cost[h->work_tp_stage[e]] += h->work_tp_cost[e];
        for (int nx = 1; nx <= 8; ++nx) {
            std::vector<int> owner(NST, 0);
            for (int part = 0; part < 2; ++part) {
                std::vector<int> st;
                for (int k = 0; k < NST; ++k)
                    if (cost[k] > 0.0 && (k < h->tune.tp_ahead) == (part == 0)) st.push_back(k);
                std::stable_sort(st.begin(), st.end(), [&](int x, int y) { return cost[x] > cost[y]; });
                std::vector<double> load(nx, 0.0);
                for (int k : st) {
                    const int x = (int)(std::min_element(load.begin(), load.end()) - load.begin());
                    owner[k] = x;
                    load[x] += cost[k];
                }
            }
            for (int x = 0; x < 8; ++x) { h->tp_xlen[nx][x] = 0; h->tp_xlo[nx][x] = 0; }
            for (int e = 0; e < h->nwork_tp; ++e) {
                const int x = owner[h->work_tp_stage[e]];
                tab[((size_t)nx * 8 + x) * TP_XENT_MAX + h->tp_xlen[nx][x]++] = e;
                if (e < h->tp_wsplit) h->tp_xlo[nx][x] += 1;
            }
        }
        (void)hipMemcpy(h->tp_xent, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice);
    }
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
            const int step = 32;   // (one output tile per item in the first stages, and all 80 channels of a half per item, were measured slower)
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
            launch_pack_valu_out(d.w, wv);
            d.nchain = C1_THREADS; d.nstep = C1_OUT_STEPS;
        } else {
            d.nchain = 5 * d.Co_pad + (d.w_skip ? 5 * NF : 0);
            d.nstep = 4 * (d.NG / 5);
            const int n = d.nstep * d.nchain * 4;
            if (int rc = dev_alloc(h, &wv, (size_t)n)) return rc;
            launch_pack_valu(d.w + (size_t)d.center_tap * d.NG * 16 * d.Co_pad, d.w_skip, d.Co_pad, d.nchain, d.nstep, wv);
        }
        d.wv = wv;
    }
    // throughput form: the stages' weights in that chain role's own order
    std::vector<float *> wtp(NST, nullptr);
    std::vector<int> tptype(NST, TPT_DIL);
    for (int k = 0; k < NST - 1; ++k) {
        const StageDesc &d = st[k];
        const int type = d.NG == 5 ? TPT_DIL : d.Co_pad == 2 * NF ? TPT_CONVOUT : d.w_skip ? TPT_CONVIN_SKIP : TPT_CONVIN;
        tptype[k] = type;
        if (int rc = dev_alloc(h, &wtp[k], (size_t)tp_weights_floats(type))) return rc;
        launch_pack_tp(d.w + (size_t)d.center_tap * d.NG * 16 * d.Co_pad, d.w_skip, d.Co_pad, d.NG, type, wtp[k]);
    }
    wtp[NST - 1] = wtp[NST - 2];   // nin_out has its own loop: the record only has to name loadable memory (requested, dropped)
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
            put_p(1 + k, CTL_WTP, wtp[k]);
            if (k + 1 < NST) put_post(1 + k, st[k + 1]);
        }
        if (int rc = dev_alloc(h, &h->ctl1, ctl.size())) return rc;
        PS_HIP_CHECK(hipMemcpy(h->ctl1, ctl.data(), ctl.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    if (int rc = dev_alloc(h, &h->work, work.size())) return rc;
    PS_HIP_CHECK(hipMemcpy(h->work, work.data(), work.size() * sizeof(NbrWork), hipMemcpyHostToDevice));
    h->nwork = (int)work.size();
    for (const NbrWork &w : work) h->work_stage.push_back(w.stage);
    PS_REQUIRE(h->nwork <= NWORK_MAX, "pixelcnn: %d neighbour work entries exceed the staging table", h->nwork);
    if (int rc = dev_alloc(h, &h->work_tp, work_tp.size())) return rc;
    PS_HIP_CHECK(hipMemcpy(h->work_tp, work_tp.data(), work_tp.size() * sizeof(NbrWorkTp), hipMemcpyHostToDevice));
    h->nwork_tp = (int)work_tp.size();
    for (const NbrWorkTp &w : work_tp) { h->work_tp_stage.push_back(w.stage); h->work_tp_cost.push_back((double)w.T * w.NG); }
    if (int rc = dev_alloc(h, &h->tp_xent, (size_t)9 * 8 * TP_XENT_MAX)) return rc;
    apply_look_ahead(h);
    return PS_OK;
}

// `ncols` independent columns (records rec[0..ncols)): neighbour taps of every conv and the centre-tap chains + draw.  A wavefront of
// at least tune.tp_min_cols columns takes the throughput form (lmconv_tp.hip), a smaller one the latency form (lmconv_column.hip).
// next_rec / next_ncols: the columns of the launch that FOLLOWS on this stream, when the caller knows it (a wavefront schedule).
void run_columns(ps_pixelcnn *h, const StepCtx *rec, int ncols, const int32_t *codes, ChainArgs ca, hipStream_t st,
                 const StepCtx *next_rec = nullptr, int next_ncols = 0)
{
    ca.ctl1 = h->ctl1; ca.nbr = h->nbr;
    ca.uinit_w = h->uinit_w; ca.uinit_b = h->uinit_b; ca.codes_in = codes;
    ca.out_b = h->out_b;
    ca.H = h->H; ca.W = h->W; ca.L = h->L; ca.col_stride = COL_CAP;
    ca.cnt = h->cnt; ca.err = h->err;
    ca.debug = h->tune.column_debug;   // (0 outside tuning builds)
    if (ncols >= h->tune.tp_min_cols) run_columns_tp(h, rec, ncols, ca, st, next_rec, next_ncols);
    else run_columns_la(h, rec, ncols, ca, st, next_rec, next_ncols);
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

}  // namespace pslm

using namespace pslm;

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
    }
    tuning_from_env(h->tune);
    h->env_col_cap = h->tune.col_cap;
    h->tune.col_cap = std::min(h->env_col_cap, (h->n_cus / 8) * 4);   // at most four XCDs of chains, the rest for the neighbour role

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
    if ((rc = dev_alloc(h, &h->perm, 2 * locs))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->permq, 2 * locs))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->perm_sorted, 2 * locs))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->perm_cnt, (size_t)2 * 512 * max_frames))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->perm_tsum, (size_t)2 * max_frames))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->taps, locs))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->nbr_tp, (size_t)2 * NST * 2 * TP_COL_CAP * NBR_LD))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->cnt_tp, 2 * tp_cnt_index(NST, 0)))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->done_tp, (size_t)NST * CNT_PAD))) return fail_out(rc);
    if ((rc = dev_alloc(h, &h->dq_tp, (size_t)8 * CNT_PAD))) return fail_out(rc);
    if (hipMemset(h->cnt_tp, 0, 2 * tp_cnt_index(NST, 0) * sizeof(unsigned)) != hipSuccess || hipMemset(h->dq_tp, 0, (size_t)8 * CNT_PAD * sizeof(unsigned)) != hipSuccess ||
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
    for (auto &r : h->prof_own) {   // (a launch profile that was begun and never ended: its events go with the handle)
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
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
constexpr int AR_ALL_FRAMES = INT32_MIN;   // (a sentinel of its own: a negative frame count from a caller is an error, not "all")
static int ar_run_impl(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region,
                       const float *mask_init, const float *mask_undilated, const float *mask_dilated,
                       const int32_t *forced, const float *uniforms, float temperature, int F, int first_step,
                       const int32_t *wave_cols, const int32_t *wave_start, int n_waves, float *out_logits, void *stream,
                       int phases = AR_PREFIX | AR_COLUMNS, int f0 = 0, int nf = AR_ALL_FRAMES,
                       const int32_t *first_steps = nullptr, int max_first_step = -1)
{
    // first_steps (device, (F)) / max_first_step: PER-FRAME prefixes -- frame f's whole-grid pass covers its ranks [0, first_steps[f]),
    // first_step <= first_steps[f] <= max_first_step, and the schedule holds its columns from first_steps[f] on (ps_ar_wavefronts_frames)
    const bool per_frame = first_steps != nullptr;
    if (int rc = check_handle(h, F)) return rc;
    if (nf == AR_ALL_FRAMES) nf = F;
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
        // (a whole run walks every column once; ps_pixelcnn_ar_columns alone also takes PART of them -- callers that run the narrow last
        // wavefronts of one batch inside the launches of the next batch's first ones)
        PS_REQUIRE(phases == AR_COLUMNS || per_frame ? wave_start[n_waves] <= F * nsteps : wave_start[n_waves] == F * nsteps,
                   "pixelcnn_ar_run_waves: the schedule holds %d columns, the run has %d", wave_start[n_waves], F * nsteps);
    }
    hipStream_t st = (hipStream_t)stream;
    const Masks m{mask_init, mask_undilated, mask_dilated};
    if (phases & AR_PREFIX) {   // frames [f0, f0 + nf): sampled codes masked out, whole-grid pass over the observed prefix
        const size_t n = (size_t)nf * h->L, off = (size_t)f0 * h->L;
        if (n > 0) hipLaunchKernelGGL(k_mask_codes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, codes + off, sample_region + off, n);
        // whole-grid pass: exact for every location that precedes the first sampled one; with out_logits it also
        // yields their logits, by location (the walked positions are overwritten by the column steps)
        PS_REQUIRE(!per_frame || (max_first_step >= first_step && max_first_step <= h->L && !out_logits),
                   "pixelcnn_ar_prefix: per-frame first steps need first_step <= max_first_step <= L (and no logits)");
        run_grid(h, F, codes, m, out_logits, false, st, order, per_frame ? max_first_step : first_step, f0, nf, first_steps);
        PS_LAUNCH_CHECK();
    }
    if (!(phases & AR_COLUMNS)) return PS_OK;
    ChainArgs ca{};
    ca.codes = codes; ca.region = sample_region; ca.forced = forced; ca.uniforms = uniforms;
    ca.out_logits = out_logits; ca.temperature = temperature;
    const int total = wave_cols ? wave_start[n_waves] : F * nsteps;
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
            h->prof_wave = w;
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
    PS_REQUIRE(frame_begin >= 0 && frame_begin <= frame_end && frame_end <= F, "pixelcnn_ar_prefix: frames [%d, %d) are not a range of the run's %d",
               frame_begin, frame_end, F);
    return ar_run_impl(h, codes, order, sample_region, mask_init, mask_undilated, mask_dilated, nullptr, nullptr, 1.0f, F, first_step, nullptr,
                       nullptr, 0, nullptr, stream, AR_PREFIX, frame_begin, frame_end - frame_begin);
}

int ps_pixelcnn_ar_prefix_frames(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region, const float *mask_init,
                                 const float *mask_undilated, const float *mask_dilated, int F, const int32_t *first_steps, int min_first_step,
                                 int max_first_step, int frame_begin, int frame_end, void *stream)
{
    PS_REQUIRE(first_steps, "pixelcnn_ar_prefix_frames: null first_steps");
    PS_REQUIRE(frame_begin >= 0 && frame_begin <= frame_end && frame_end <= F, "pixelcnn_ar_prefix: frames [%d, %d) are not a range of the run's %d",
               frame_begin, frame_end, F);
    return ar_run_impl(h, codes, order, sample_region, mask_init, mask_undilated, mask_dilated, nullptr, nullptr, 1.0f, F, min_first_step, nullptr,
                       nullptr, 0, nullptr, stream, AR_PREFIX, frame_begin, frame_end - frame_begin, first_steps, max_first_step);
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
    h->tune.col_cap = std::min(h->env_col_cap, (h->n_cus / 8) * 4);   // (the cap asked for at creation stays in force)
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
    return ps_pixelcnn_time_ar_run_waves_range(h, codes, order, sample_region, mask_init, mask_undilated, mask_dilated, uniforms, temperature, F,
                                               first_step, wave_cols, wave_start, n_waves, 0, n_waves, nullptr, -1, launches, total_ms, flops_per_column, stream);
}

// ... counting only the launches of the wavefronts [wave_from, wave_to) (bench.py: the launches of one steady-state step of the
// pipelined form inside a two-batch run)
int ps_pixelcnn_time_ar_run_waves_range(ps_pixelcnn *h, int32_t *codes, const int32_t *order, const uint8_t *sample_region,
                                        const float *mask_init, const float *mask_undilated, const float *mask_dilated,
                                        const float *uniforms, float temperature, int F, int first_step, const int32_t *wave_cols,
                                        const int32_t *wave_start, int n_waves, int wave_from, int wave_to, const int32_t *first_steps,
                                        int max_first_step, int *launches, float *total_ms, double *flops_per_column, void *stream)
{
    PS_REQUIRE(h && launches && total_ms, "pixelcnn_time_ar_run_waves: null pointer");
    std::vector<ps_pixelcnn::ProfRec> recs;
    h->prof = &recs;
    const int rc = ar_run_impl(h, codes, order, sample_region, mask_init, mask_undilated, mask_dilated, nullptr, uniforms,
                               temperature, F, first_step, wave_cols, wave_start, n_waves, nullptr, stream, AR_PREFIX | AR_COLUMNS, 0,
                               AR_ALL_FRAMES, first_steps, max_first_step);
    h->prof = nullptr;
    (void)hipStreamSynchronize((hipStream_t)stream);
    *launches = 0;
    *total_ms = 0.0f;
    for (auto &r : recs) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        if (r.tag == TAG_CHAIN && r.wave >= wave_from && r.wave < wave_to) { *launches += 1; *total_ms += ms; }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    if (flops_per_column) *flops_per_column = h->flops_nbr + h->flops_chain;
    return rc;
}

// Which kernels carried a run's matrix work: cumulative launch counts of the handle by kind (pslm::LaunchKind), always on ...
static const char *const launch_kind_names[LK_N] = {"k_column", "k_column_la", "k_column_tp", "k_column_tp8", "k_gemm", "k_gemm_wg",
                                                    "k_gemm_ws<0>", "k_gemm_ws<1>", "k_gemm_ws<2>"};
int ps_pixelcnn_launch_kinds(void) { return LK_N; }
const char *ps_pixelcnn_launch_kind_name(int kind) { return kind >= 0 && kind < LK_N ? launch_kind_names[kind] : nullptr; }
int ps_pixelcnn_launch_counts(ps_pixelcnn *h, long long *counts, int n)
{
    PS_REQUIRE(h && counts && n >= 0, "pixelcnn_launch_counts: null pointer");
    for (int k = 0; k < n; ++k) counts[k] = k < LK_N ? h->launch_count[k] : 0;
    return PS_OK;
}
// ... and, between _begin and _end, a HIP event pair around every such launch on the stream it goes to (whatever entry point enqueues
// it, on whatever stream): _end synchronises the device and returns launches and summed duration by kind.
int ps_pixelcnn_profile_begin(ps_pixelcnn *h)
{
    PS_REQUIRE(h, "pixelcnn_profile_begin: null pointer");
    if (h->prof) return ps::fail(PS_ERR_STATE, "pixelcnn_profile_begin: a timed run is in progress");
    h->prof_own.clear();
    h->prof = &h->prof_own;
    return PS_OK;
}
int ps_pixelcnn_profile_end(ps_pixelcnn *h, int n, int *launches, float *total_ms)
{
    PS_REQUIRE(h && launches && total_ms && n >= 0, "pixelcnn_profile_end: null pointer");
    if (h->prof != &h->prof_own) return ps::fail(PS_ERR_STATE, "pixelcnn_profile_end: no profile was begun");
    h->prof = nullptr;
    PS_HIP_CHECK(hipDeviceSynchronize());
    for (int k = 0; k < n; ++k) { launches[k] = 0; total_ms[k] = 0.0f; }
    for (auto &r : h->prof_own) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        if (r.kind >= 0 && r.kind < n) { launches[r.kind] += 1; total_ms[r.kind] += ms; }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    h->prof_own.clear();
    return PS_OK;
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
    if (what == 6 || what == 7 || what == 9) return wg_trace_symbol(what);   // (lmconv_grid.hip)
#endif
    if (what == 4) {                   // tuning builds: allocate / return the stamp buffer [2][NST][8] of 64-bit clocks
        if (!h->tp_trace && dev_alloc(h, &h->tp_trace, (size_t)3 * NST * 8) == PS_OK) (void)hipMemset(h->tp_trace, 0, 3 * NST * 8 * 8);
        return h->tp_trace;
    }
    return nullptr;
}

int ps_pixelcnn_set_tuning(ps_pixelcnn *h, const char *key, int value)
{
    PS_REQUIRE(h && key, "pixelcnn_set_tuning: null pointer");
    const TuningEntry *e = find_tuning(key);
    PS_REQUIRE(e, "pixelcnn_set_tuning: no tuning value named '%s'", key);
    PS_REQUIRE(value >= e->lo && value <= e->hi, "pixelcnn_set_tuning: %s = %d outside [%d, %d]", key, value, e->lo, e->hi);
    PS_REQUIRE(tuning_value_ok(*e, value), "pixelcnn_set_tuning: no launch form for %s = %d", key, value);
    const bool depth = e->field == &Tuning::tp_ahead || e->field == &Tuning::col_ahead;
    // the never-reset completion counters count items per stage under ONE look-ahead depth: it is fixed by the first column launch
    if (depth && h->columns_launched && h->tune.*(e->field) != value)
        return ps::fail(PS_ERR_STATE, "pixelcnn_set_tuning: %s can only be set before the handle's first column launch", key);
    h->tune.*(e->field) = value;
    if (e->field == &Tuning::col_cap) { h->env_col_cap = value; h->tune.col_cap = std::min(value, (h->n_cus / 8) * 4); }
    if (depth) apply_look_ahead(h);
    return PS_OK;
}

int ps_pixelcnn_get_tuning(ps_pixelcnn *h, const char *key, int *value)
{
    PS_REQUIRE(h && key && value, "pixelcnn_get_tuning: null pointer");
    const TuningEntry *e = find_tuning(key);
    PS_REQUIRE(e, "pixelcnn_get_tuning: no tuning value named '%s'", key);
    *value = h->tune.*(e->field);
    return PS_OK;
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
#ifdef PS_CHAIN_TRACE_BUILD
    if (const char *tp = getenv("PS_CHAIN_TRACE")) {  // tuning builds: per-stage shader-clock stamps of workgroup 0
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
#endif
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

}  // extern "C"
