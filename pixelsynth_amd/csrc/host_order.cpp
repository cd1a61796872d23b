// host_order.cpp -- host-side integer glue between the splat and the AR sampler:
// block reduction of the background mask, 5x5 chamfer distance transforms, the greedy frontier
// generation order and the per-location 3x3 kernel masks.
//
// Replaces (behind the C ABI): ZbufferModelPts.get_masks_for_batch (models/z_buffermodel.py:641-701),
// get_custom_order.custom_idx (models/lmconv/get_custom_order.pyx:4-124, the reference's Cython
// native module) and masking.kernel_masks / get_unfolded_masks (models/lmconv/masking.py:287-349).
// Pure C++17, no GPU work: the data is a 32x32 grid and the algorithm is a sequential heap walk.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <queue>
#include <tuple>
#include <utility>
#include <string>
#include <cstdlib>
#include <thread>
#include <vector>

#include "ps_common.h"

namespace {

// cv2.distanceTransform(src, DIST_L2, 5): two-pass chamfer, 16.16 fixed point, metrics
// (1, 1.4, 2.1969); portable OpenCV definition ("Parity unpinned").
struct Chamfer5 {
    static constexpr uint32_t INIT = 0x7FFFFFFF >> 2;
    // forward-pass neighbourhood (dy, dx, metric id); the backward pass mirrors it
    static constexpr int NB[8][3] = {{-2, -1, 2}, {-2, 1, 2}, {-1, -2, 2}, {-1, -1, 1},
                                     {-1, 0, 0},  {-1, 1, 1}, {-1, 2, 2},  {0, -1, 0}};
    static void run(const uint8_t *src, int H, int W, float *out)
    {
        const uint32_t metric[3] = {65536u, (uint32_t)std::lrint((double)1.4f * 65536.0),
                                    (uint32_t)std::lrint((double)2.1969f * 65536.0)};
        const int B = 2, step = W + 2 * B;
        std::vector<uint32_t> buf((size_t)step * (H + 2 * B), INIT);
        auto at = [&](int y, int x) -> uint32_t & { return buf[(size_t)(y + B) * step + (x + B)]; };
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                if (!src[(size_t)y * W + x]) { at(y, x) = 0; continue; }
                uint32_t best = 0xFFFFFFFFu;
                for (auto &nb : NB) best = std::min(best, at(y + nb[0], x + nb[1]) + metric[nb[2]]);
                at(y, x) = best;
            }
        for (int y = H - 1; y >= 0; --y)
            for (int x = W - 1; x >= 0; --x) {
                uint32_t best = at(y, x);
                if (best > metric[0]) {
                    for (auto &nb : NB) best = std::min(best, at(y - nb[0], x - nb[1]) + metric[nb[2]]);
                    at(y, x) = best;
                }
                out[(size_t)y * W + x] = (float)std::min(best, INIT) * (1.0f / 65536.0f);
            }
    }
};
constexpr int Chamfer5::NB[8][3];

void custom_order(int rows, int cols, int64_t *d, int32_t *order)
{
    const int L = rows * cols;
    for (int i = 0; i < L; ++i) d[i] *= 10000;                                    // .pyx:26
    const int start = (int)(std::max_element(d, d + L) - d);                      // first argmax, .pyx:55-56
    int c = start % rows, r = (start - c) / rows;
    using Item = std::tuple<int64_t, int, int>;                                   // (-distance, r, c)
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> frontier;
    std::vector<uint8_t> seen((size_t)L, 0);
    seen[(size_t)r * cols + c] = 1;
    int n = 0;
    order[2 * n] = r; order[2 * n + 1] = c; ++n;
    static const int STEP[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}};              // Up, Down, Left, Right .pyx:65-80
    while (n < L) {
        for (auto &s : STEP) {
            const int rr = r + s[0], cc = c + s[1];
            if (rr < 0 || rr >= rows || cc < 0 || cc >= cols || seen[(size_t)rr * cols + cc]) continue;
            seen[(size_t)rr * cols + cc] = 1;
            frontier.emplace(-d[(size_t)rr * cols + cc], rr, cc);
        }
        std::tie(std::ignore, r, c) = frontier.top();
        frontier.pop();
        order[2 * n] = r; order[2 * n + 1] = c; ++n;
    }
}

}  // namespace

extern "C" {

int ps_abi_version(void) { return PS_ABI_VERSION; }
// What this library was built from, for whoever reports a number measured through it (bench.py prints it): pixelsynth_amd/build.py
// passes its extra flags in (-DPS_BUILD_EXTRA_FLAGS="..."); a product build has none.  Tuning / trace / experiment builds
// (-DPS_TUNING_BUILD, -DPS_*_TRACE_BUILD, -DPS_WS_EXP=..., ...: results INVALID or timing perturbed) say so here.
#ifndef PS_BUILD_EXTRA_FLAGS
#define PS_BUILD_EXTRA_FLAGS ""
#endif
#define PS_STR2(x) #x
#define PS_STR(x) PS_STR2(x)
const char *ps_build_info(void)
{
    static const std::string info = [] {
        const std::string extra = PS_BUILD_EXTRA_FLAGS;
        return std::string("libpixelsynth_hip abi ") + PS_STR(PS_ABI_VERSION) + "; hipcc --offload-arch=gfx950 -O3 -ffp-contract=off; "
               + (extra.empty() ? "product build (no extra flags)" : "NON-PRODUCT build, extra flags: " + extra);
    }();
    return info.c_str();
}
const char *ps_last_error(void) { return ps::last_error_ref().c_str(); }

int ps_custom_order(int rows, int cols, int64_t *distances, int32_t *order)
{
    PS_REQUIRE(distances && order, "custom_order: null pointer");
    PS_REQUIRE(rows > 0 && rows == cols, "custom_order: rows == cols > 0 required (reference asserts it)");
    custom_order(rows, cols, distances, order);
    return PS_OK;
}

int ps_generation_order(const uint8_t *bg, int S, int G, int32_t *order, uint8_t *bg_blocks, int64_t *distances)
{
    PS_REQUIRE(bg && order && bg_blocks, "generation_order: null pointer");
    PS_REQUIRE(S > 0 && G > 0 && S % G == 0, "generation_order: S must be a multiple of G");
    const int blk = S / G, L = G * G;
    std::vector<uint8_t> fgb((size_t)L), bgb((size_t)L);
    for (int by = 0; by < G; ++by)
        for (int bx = 0; bx < G; ++bx) {
            int ones = 0;
            for (int y = 0; y < blk; ++y)
                for (int x = 0; x < blk; ++x) ones += bg[(size_t)(by * blk + y) * S + bx * blk + x] != 0;
            // AvgPool2d(blk) then .astype(uint8): 1 only when the block mean is exactly 1 (:646-647,668-669)
            bgb[(size_t)by * G + bx] = ones == blk * blk;
            fgb[(size_t)by * G + bx] = ones == 0;
        }
    std::vector<float> fd((size_t)L), bd((size_t)L);
    Chamfer5::run(fgb.data(), G, G, fd.data());
    Chamfer5::run(bgb.data(), G, G, bd.data());
    std::vector<int64_t> D((size_t)L);
    for (int i = 0; i < L; ++i) D[i] = (int64_t)((double)fd[i] - (double)bd[i]);   // float64 subtract, trunc (:675)
    if (distances) std::memcpy(distances, D.data(), sizeof(int64_t) * (size_t)L);
    custom_order(G, G, D.data(), order);
    std::memcpy(bg_blocks, bgb.data(), (size_t)L);
    return PS_OK;
}

int ps_kernel_masks_f32(const int32_t *order, int L, int nrows, int ncols, int k, int dilation, int mask_type_b,
                        float *masks);

int ps_ar_plan(const uint8_t *bg, int B, int S, int G, int32_t *order_loc, uint8_t *region, float *mask_init,
               float *mask_undilated, float *mask_dilated, int32_t *first_step)
{
    PS_REQUIRE(bg && order_loc && region, "ar_plan: null pointer");
    PS_REQUIRE((mask_init != nullptr) == (mask_undilated != nullptr) && (mask_init != nullptr) == (mask_dilated != nullptr),
               "ar_plan: give all three masks or none (ps_order_masks_f32 builds them on the device)");
    PS_REQUIRE(B > 0 && S > 0 && G > 0 && S % G == 0, "ar_plan: bad sizes");
    const int L = G * G;
    // frames are independent: one worker per frame (up to 64, half the host's cores), each with its own scratch; the first failure wins
    std::vector<int> first_b((size_t)B, L), rc_b((size_t)B, PS_OK);
    std::vector<std::string> msg_b((size_t)B);   // the error channel is thread-local: a worker's message is carried over by hand
    auto one_frame_impl = [&](int b) {
        std::vector<int32_t> order((size_t)L * 2);
        uint8_t *reg = region + (size_t)b * L;
        if ((rc_b[b] = ps_generation_order(bg + (size_t)b * S * S, S, G, order.data(), reg, nullptr))) return;
        int32_t *ol = order_loc + (size_t)b * L;
        int first = L;
        for (int i = 0; i < L; ++i) {
            ol[i] = order[2 * i] * G + order[2 * i + 1];
            if (reg[ol[i]] && i < first) first = i;
        }
        first_b[b] = first;
        if (!mask_init) return;
        const size_t mo = (size_t)b * 9 * L;
        if ((rc_b[b] = ps_kernel_masks_f32(order.data(), L, G, G, 3, 1, 0, mask_init + mo))) return;
        if ((rc_b[b] = ps_kernel_masks_f32(order.data(), L, G, G, 3, 1, 1, mask_undilated + mo))) return;
        rc_b[b] = ps_kernel_masks_f32(order.data(), L, G, G, 3, 2, 1, mask_dilated + mo);
    };
    auto one_frame = [&](int b) {
        one_frame_impl(b);
        if (rc_b[b]) msg_b[b] = ps::last_error_ref();
    };
    // frames are independent.  One process per GPU on an 8-GPU node shares the host: WORLD_SIZE / LOCAL_WORLD_SIZE (set by
    // torch.distributed.run) divide the cores -- that default is worked out once per process; PS_PLAN_THREADS overrides it and is
    // read on every call (one getenv per plan of ~2 ms), so that a thread-count sweep inside one process measures what it says
    // (16 at most by default: 128 frames take 1.8 ms on 16 threads and 3.5 ms on 64 -- thread start-up, measured with tools/plan_time.py)
    static const int default_threads = []() {
        const int hw = (int)std::thread::hardware_concurrency();
        int share = 1;
        if (const char *e = getenv("LOCAL_WORLD_SIZE")) share = std::max(1, atoi(e));
        else if (const char *e2 = getenv("WORLD_SIZE")) share = std::max(1, atoi(e2));
        return std::max(1, std::min(16, hw > 0 ? hw / (2 * share) : 16));
    }();
    int plan_threads = default_threads;
    if (const char *e = getenv("PS_PLAN_THREADS")) plan_threads = std::max(1, atoi(e));
    const int nthreads = std::min(B, plan_threads);
    if (nthreads <= 1) {
        one_frame(0);
    } else {
        std::vector<std::thread> pool;
        for (int w = 1; w < nthreads; ++w)
            pool.emplace_back([&, w]() { for (int b = w; b < B; b += nthreads) one_frame(b); });
        for (int b = 0; b < B; b += nthreads) one_frame(b);
        for (auto &th : pool) th.join();
    }
    int first = L;
    for (int b = 0; b < B; ++b) {
        if (rc_b[b]) return ps::fail(rc_b[b], "ar_plan: frame %d: %s", b, msg_b[b].c_str());
        first = std::min(first, first_b[b]);
    }
    if (first_step) *first_step = first;
    return PS_OK;
}

// fs[b]: the first order position of frame b that is walked as a column (everything before it is the whole-grid pass's)
static int wavefronts_impl(const int32_t *order_loc, int B, int H, int W, const int32_t *fs, int max_cols, int32_t *cols,
                           int32_t *wave_start, int32_t *n_waves)
{
    PS_REQUIRE(order_loc && cols && wave_start && n_waves && fs, "ar_wavefronts: null pointer");
    PS_REQUIRE(B > 0 && H > 0 && W > 0 && max_cols >= 0, "ar_wavefronts: bad sizes");
    const int L = H * W;
    size_t ncolumns = 0;
    std::vector<size_t> base((size_t)B + 1, 0);   // columns of the frames in front of frame b
    for (int b = 0; b < B; ++b) {
        PS_REQUIRE(fs[b] >= 0 && fs[b] <= L, "ar_wavefronts: first_step out of range");
        ncolumns += (size_t)(L - fs[b]);
        base[(size_t)b + 1] = ncolumns;
    }
    // A column reads the columns of the locations that are a 3x3 tap neighbour at dilation 1 or 2 AND earlier in the
    // order (exactly the open taps of the three kernel masks); columns before first_step are done by the whole-grid pass.
    std::vector<int32_t> rank((size_t)B * L);
    for (int b = 0; b < B; ++b) {
        const int32_t *ol = order_loc + (size_t)b * L;
        int32_t *rk = rank.data() + (size_t)b * L;
        std::fill(rk, rk + L, -1);
        for (int i = 0; i < L; ++i) {
            PS_REQUIRE(ol[i] >= 0 && ol[i] < L && rk[ol[i]] < 0, "ar_wavefronts: frame %d: order is not a permutation", b);
            rk[ol[i]] = i;
        }
    }
    auto for_neighbours = [&](int q, auto &&fn) {
        const int r = q / W, c = q - r * W;
        for (int dil = 1; dil <= 2; ++dil)
            for (int t = 0; t < 9; ++t) {
                if (t == 4) continue;
                const int rr = r + (t / 3 - 1) * dil, cc = c + (t % 3 - 1) * dil;
                if (rr >= 0 && rr < H && cc >= 0 && cc < W) fn(rr * W + cc);
            }
    };
    if (max_cols == 0) {
        // pure dependency levels: wave of a column = 1 + the latest wave among the columns it reads
        std::vector<int32_t> wave(ncolumns), lvl((size_t)L);
        int deepest = 0;
        for (int b = 0; b < B; ++b) {
            const int32_t *ol = order_loc + (size_t)b * L, *rk = rank.data() + (size_t)b * L;
            std::fill(lvl.begin(), lvl.end(), 0);
            for (int i = fs[b]; i < L; ++i) {
                int dep = 0;
                for_neighbours(ol[i], [&](int p) { if (rk[p] < i) dep = std::max(dep, lvl[p]); });
                lvl[ol[i]] = dep + 1;
                wave[base[b] + (size_t)(i - fs[b])] = dep;  // 0-based wave index
                deepest = std::max(deepest, dep + 1);
            }
        }
        // counting sort by wave; within a wave by frame, then by position
        std::vector<int32_t> count((size_t)deepest + 1, 0);
        for (int32_t w : wave) count[(size_t)w + 1] += 1;
        for (int w = 0; w < deepest; ++w) count[(size_t)w + 1] += count[w];
        for (int w = 0; w <= deepest; ++w) wave_start[w] = count[w];
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < L - fs[b]; ++k) {
                const int32_t at = count[wave[base[b] + (size_t)k]]++;
                cols[2 * (size_t)at] = b;
                cols[2 * (size_t)at + 1] = fs[b] + k;
            }
        *n_waves = deepest;
        return PS_OK;
    }
    // Waves of at most max_cols columns (what one launch takes): list scheduling over the ready columns, those with the
    // longest chain of dependants first.  A level that does not fit is not cut in two launches of its own: what is
    // left over shares the next wave with the columns that have become ready meanwhile.
    std::vector<int32_t> height((size_t)B * L, 0), indeg((size_t)B * L, 0);
    // ready columns by height (a bucket queue: heights are at most L); within a height, first come first served
    std::vector<std::vector<std::pair<int32_t, int32_t>>> bucket((size_t)L + 2);
    int top = 0;
    size_t nready = 0;
    auto push = [&](int h, int b, int i) {
        bucket[(size_t)h].emplace_back(b, i);
        top = std::max(top, h);
        ++nready;
    };
    auto frame_graph = [&](int b) {  // heights (longest chain of dependants) and in-degrees of one frame's columns
        const int32_t *ol = order_loc + (size_t)b * L, *rk = rank.data() + (size_t)b * L;
        int32_t *hb = height.data() + (size_t)b * L, *db = indeg.data() + (size_t)b * L;
        for (int i = L - 1; i >= fs[b]; --i) {
            int h = 0, d = 0;
            for_neighbours(ol[i], [&](int p) {
                if (rk[p] > i) h = std::max(h, (int)hb[p]);
                else if (rk[p] >= fs[b]) ++d;
            });
            hb[ol[i]] = h + 1;
            db[ol[i]] = d;
        }
    };
    for (int b = 0; b < B; ++b) frame_graph(b);  // (threads were tried: their start-up costs more than the 0.3 ms they share out)
    for (int b = 0; b < B; ++b) {
        const int32_t *ol = order_loc + (size_t)b * L;
        const int32_t *hb = height.data() + (size_t)b * L, *db = indeg.data() + (size_t)b * L;
        for (int i = fs[b]; i < L; ++i)
            if (db[ol[i]] == 0) push(hb[ol[i]], b, i);
    }
    size_t at = 0;
    int nw = 0;
    std::vector<std::pair<int32_t, int32_t>> taken;
    std::vector<std::pair<uint32_t, std::pair<int32_t, int32_t>>> keyed;
    wave_start[0] = 0;
    while (nready > 0) {
        taken.clear();
        while (nready > 0 && (int)taken.size() < max_cols) {
            while (bucket[(size_t)top].empty()) --top;
            taken.push_back(bucket[(size_t)top].back());
            bucket[(size_t)top].pop_back();
            --nready;
        }
        // Within a wave the order of the columns is free.  A launch works on them in tiles of 16, and a neighbour tap that is
        // closed for every column of a tile is skipped for the whole tile: columns with the same set of open taps (the same
        // sixteen bits: 8 neighbours at dilation 1, 8 at dilation 2) are put next to each other.
        if (taken.size() > 16) {
            keyed.clear();
            for (const auto &k : taken) {
                const int b = k.first, i = k.second;
                const int32_t *ol = order_loc + (size_t)b * L, *rk = rank.data() + (size_t)b * L;
                const int q = ol[i], r = q / W, c = q - r * W;
                uint32_t sig = 0;
                for (int dil = 1; dil <= 2; ++dil)
                    for (int t = 0; t < 9; ++t) {
                        if (t == 4) continue;
                        const int rr = r + (t / 3 - 1) * dil, cc = c + (t % 3 - 1) * dil;
                        sig = (sig << 1) | ((rr >= 0 && rr < H && cc >= 0 && cc < W && rk[rr * W + cc] < i) ? 1u : 0u);
                    }
                keyed.emplace_back(sig, k);
            }
            std::stable_sort(keyed.begin(), keyed.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
            for (size_t k = 0; k < keyed.size(); ++k) taken[k] = keyed[k].second;
        }
        for (const auto &k : taken) {
            cols[2 * at] = k.first;
            cols[2 * at + 1] = k.second;
            ++at;
        }
        for (const auto &k : taken) {  // their dependants may run from the NEXT wave on
            const int b = k.first, i = k.second;
            const int32_t *ol = order_loc + (size_t)b * L, *rk = rank.data() + (size_t)b * L;
            int32_t *hb = height.data() + (size_t)b * L, *db = indeg.data() + (size_t)b * L;
            for_neighbours(ol[i], [&](int p) {
                if (rk[p] > i && --db[p] == 0) push(hb[p], b, rk[p]);
            });
        }
        wave_start[++nw] = (int32_t)at;
    }
    PS_REQUIRE(at == ncolumns, "ar_wavefronts: internal error, %zu of %zu columns scheduled", at, ncolumns);
    *n_waves = nw;
    return PS_OK;
}

int ps_ar_wavefronts_capped(const int32_t *order_loc, int B, int H, int W, int first_step, int max_cols, int32_t *cols,
                            int32_t *wave_start, int32_t *n_waves)
{
    PS_REQUIRE(B > 0, "ar_wavefronts: bad sizes");
    const std::vector<int32_t> fs((size_t)B, first_step);
    return wavefronts_impl(order_loc, B, H, W, fs.data(), max_cols, cols, wave_start, n_waves);
}

int ps_ar_wavefronts_frames(const int32_t *order_loc, int B, int H, int W, const int32_t *first_steps, int max_cols, int32_t *cols,
                            int32_t *wave_start, int32_t *n_waves)
{
    return wavefronts_impl(order_loc, B, H, W, first_steps, max_cols, cols, wave_start, n_waves);
}

int ps_ar_wavefronts(const int32_t *order_loc, int B, int H, int W, int first_step, int32_t *cols, int32_t *wave_start,
                     int32_t *n_waves)
{
    return ps_ar_wavefronts_capped(order_loc, B, H, W, first_step, 0, cols, wave_start, n_waves);
}

int ps_kernel_masks_f32(const int32_t *order, int L, int nrows, int ncols, int k, int dilation, int mask_type_b,
                        float *masks)
{
    PS_REQUIRE(order && masks, "kernel_masks: null pointer");
    PS_REQUIRE(k > 0 && (k & 1) && dilation > 0, "kernel_masks: k must be odd, dilation > 0");
    PS_REQUIRE(L == nrows * ncols, "kernel_masks: order must visit every location once (L=%d, grid=%d)", L,
               nrows * ncols);
    // rank[q] = position of location q in the generation order; a tap is open iff its neighbour
    // was generated earlier (masking.py:318-334 builds the same relation with a growing set).
    std::vector<int32_t> rank((size_t)L, -1);
    for (int i = 0; i < L; ++i) {
        const int r = order[2 * i], c = order[2 * i + 1];
        PS_REQUIRE(r >= 0 && r < nrows && c >= 0 && c < ncols, "kernel_masks: order entry %d out of the grid", i);
        PS_REQUIRE(rank[(size_t)r * ncols + c] < 0, "kernel_masks: location (%d,%d) visited twice", r, c);
        rank[(size_t)r * ncols + c] = i;
    }
    const int h = k / 2;
    for (int dr = -h; dr <= h; ++dr)
        for (int dc = -h; dc <= h; ++dc) {
            float *m = masks + (size_t)((dr + h) * k + (dc + h)) * L;
            for (int r = 0; r < nrows; ++r)
                for (int c = 0; c < ncols; ++c) {
                    const int q = r * ncols + c;
                    if (dr == 0 && dc == 0) { m[q] = mask_type_b ? 1.0f : 0.0f; continue; }
                    const int rr = r + dr * dilation, cc = c + dc * dilation;
                    const bool in = rr >= 0 && rr < nrows && cc >= 0 && cc < ncols;
                    m[q] = (in && rank[(size_t)rr * ncols + cc] < rank[q]) ? 1.0f : 0.0f;
                }
        }
    return PS_OK;
}

}  // extern "C"
