// splat.hip -- depth un-projection + camera transform + soft z-buffer splat for gfx950 (MI355X).
//
// Replaces, behind the C ABI of include/pixelsynth_hip.h:
//   PtsManipulator.project_pts / project_pts_cumulative   models/projection/z_buffer_manipulator.py:50-83, 221-266
//   RasterizePointsXYsBlending.forward                    models/layers/z_buffer_layers.py:55-131
//   + PyTorch3D rasterize_points / compositing it calls   (third party, semantics in DESIGN.md section 2)
//
// Pipeline (all on the caller's stream, no host sync, no allocation):
//   k_project        1 thread / point: p = grid*depth, X = K (RT2 RT1inv) Kinv p, divide, EPS rule
//   k_bin_count      conservative pixel bbox of each point's disc -> 8x8-pixel tiles, per-tile counts
//   k_scan           exclusive scan of each frame's tile counters (one workgroup per frame; every frame owns a
//                    fixed slice of the key array, so no cross-frame scan)
//   k_bin_fill       64-bit keys (z bits << 32 | point index) appended to each touched tile's list
//   k_sort_small/big per-tile sort of the keys (normalised bitonic network; LDS, or global for huge
//                    lists).  Keys are unique, so the order -- ascending (z, index), PyTorch3D's CPU
//                    tie-break -- is deterministic whatever order the atomics filled the list in.
//   k_composite      one wave64 per tile, one lane per pixel: walk the sorted list front to back,
//                    exact strict disc test, alpha from dist^2, blend on the fly, stop at K hits.
//                    Nothing of size (S,S,K) is ever materialised unless the debug outputs are asked for.
//   k_dilate         k x k binary dilation of the "no hit" mask (LDS tile + halo, separable max)
//
// Integer/index paths are bit-exact against oracle/ (this file is built with -ffp-contract=off: the
// disc test dx*dx+dy*dy < r*r and the z ordering must not be perturbed by FMA contraction).
#include <cmath>

#include "ps_common.h"

namespace {

constexpr int TILE = 8;                  // pixels per tile edge: 64 pixels = one wave64
constexpr float PS_EPS = 1e-2f;          // z_buffer_manipulator.py:8
constexpr uint32_t CULLED = 0xFFFFFFFFu;
constexpr int SORT_SMALL_CAP = 512;      // keys sorted in LDS by one wave (4 KB: eight waves per SIMD)
constexpr int SORT_BIG_CAP = 8192;       // keys sorted in LDS by a 1024-thread workgroup

// ------------------------------------------------------------------------------------------
// projection
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void mat4_vec(const float *M, const float *v, float *o)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = M[i * 4 + 0] * v[0];
        acc = acc + M[i * 4 + 1] * v[1];
        acc = acc + M[i * 4 + 2] * v[2];
        acc = acc + M[i * 4 + 3] * v[3];
        o[i] = acc;
    }
}

// sampler of one projected homogeneous point; X[2] is EPS-overwritten like the reference (:70-74)
__device__ __forceinline__ void finish_point(float *X, float &sx, float &sy, float &sz)
{
    const bool bad = fabsf(X[2]) < PS_EPS;
    if (bad) X[2] = PS_EPS;
    float x = X[0] / (-X[2]);
    float y = X[1] / (-X[2]);
    float z = X[2];
    if (bad) { x = -10.0f; y = -10.0f; z = -10.0f; }
    sx = x * 1.0f;
    sy = y * -1.0f;
    sz = z * -1.0f;
}

// LAYOUT 0: sampler (B,3,NT) as the reference returns it; LAYOUT 1: (B,NT,3) with x,y negated, i.e.
// exactly what PyTorch3D's rasterizer is handed after z_buffer_layers.py:71-72.
// PRIOR: the source is a homogeneous prior cloud (B,4,n) transformed by K (RT2 RT3inv)  (:244-247).
template <int LAYOUT, bool PRIOR>
__global__ __launch_bounds__(256) void k_project(const float *__restrict__ src,
                                                 const int32_t *__restrict__ new_index,
                                                 const float *__restrict__ K,
                                                 const float *__restrict__ Kinv,
                                                 const float *__restrict__ RTa_inv,
                                                 const float *__restrict__ RT2, int W, int n,
                                                 int out_n, int out_off, float *__restrict__ out,
                                                 float *__restrict__ cloud, uint32_t *__restrict__ zero_words = nullptr,
                                                 unsigned zero_count = 0)
{
    __shared__ float sRT[16], sK[16], sKinv[16];
    const int b = blockIdx.y;
    // (the fused project + splat call: the binning counters of the splat that follows are cleared here, not by a memset of their own)
    for (unsigned i = (blockIdx.y * gridDim.x + blockIdx.x) * 256u + threadIdx.x; i < zero_count; i += gridDim.x * gridDim.y * 256u)
        zero_words[i] = 0u;
    if (threadIdx.x < 16) {
        const int i = threadIdx.x >> 2, j = threadIdx.x & 3;
        const float *A = RT2 + b * 16, *Bm = RTa_inv + b * 16;
        float acc = A[i * 4 + 0] * Bm[0 * 4 + j];
        acc = acc + A[i * 4 + 1] * Bm[1 * 4 + j];
        acc = acc + A[i * 4 + 2] * Bm[2 * 4 + j];
        acc = acc + A[i * 4 + 3] * Bm[3 * 4 + j];
        sRT[threadIdx.x] = acc;
        sK[threadIdx.x] = K[b * 16 + threadIdx.x];
        if (!PRIOR) sKinv[threadIdx.x] = Kinv[b * 16 + threadIdx.x];
    }
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    float X[4];
    if (PRIOR) {
        float p[4], w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = src[((size_t)b * 4 + r) * n + t];
        mat4_vec(sRT, p, w);
        mat4_vec(sK, w, X);
    } else {
        const int g = new_index ? new_index[(size_t)b * n + t] : t;
        const int gx = g % W, gy = g / W;
        const float den = (float)(W - 1);
        const float xs = (float)gx / den * 2.0f - 1.0f;  // linspace(0,W-1,W)/(W-1)*2-1  (:38-39)
        const float ys = (float)gy / den * 2.0f - 1.0f;
        const float d = src[(size_t)b * n + t];
        float p[4] = {xs * d, (-ys) * d, -1.0f * d, 1.0f};
        float c[4], w[4];
        mat4_vec(sKinv, p, c);
        mat4_vec(sRT, c, w);
        mat4_vec(sK, w, X);
    }
    float sx, sy, sz;
    finish_point(X, sx, sy, sz);
    const int o = out_off + t;
    if (LAYOUT == 0) {
        out[((size_t)b * 3 + 0) * out_n + o] = sx;
        out[((size_t)b * 3 + 1) * out_n + o] = sy;
        out[((size_t)b * 3 + 2) * out_n + o] = sz;
    } else {
        float *q = out + ((size_t)b * out_n + o) * 3;
        q[0] = -sx;
        q[1] = -sy;
        q[2] = sz;
    }
    if (cloud) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cloud[((size_t)b * 4 + r) * out_n + o] = X[r];
    }
}

// z_buffer_layers.py:71-72 -- the reference negates x,y of the caller's tensor in place
__global__ void k_negate_xy(float *pts, size_t npts)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    pts[i * 3 + 1] = -pts[i * 3 + 1];
    pts[i * 3 + 0] = -pts[i * 3 + 0];
}

// ------------------------------------------------------------------------------------------
// binning
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float pix_to_ndc(int i, int S) { return -1.0f + (2 * i + 1.0f) / S; }

// Conservative range of pixel indices whose centre can pass the strict disc test, per axis.
// PyTorch3D tests output pixel xi against PixToNdc(S-1-xi); c is the point in reversed-index units.
__device__ __forceinline__ bool axis_range(float p, int S, float hw, int &lo_px, int &hi_px)
{
    const float c = ((p + 1.0f) * S - 1.0f) * 0.5f;
    float lo = c - hw, hi = c + hw;
    if (!(hi >= 0.0f) || !(lo <= (float)(S - 1))) return false;  // also rejects NaN / inf
    lo = fmaxf(lo, 0.0f);
    hi = fminf(hi, (float)(S - 1));
    const int ilo = (int)ceilf(lo), ihi = (int)floorf(hi);
    if (ilo > ihi) return false;
    lo_px = S - 1 - ihi;
    hi_px = S - 1 - ilo;
    return true;
}

__device__ __forceinline__ uint32_t point_bbox(const float *p, int S, float hw)
{
    const float px = p[0], py = p[1], pz = p[2];
    if (!(pz >= 0.0f)) return CULLED;  // PyTorch3D skips pz < 0; NaN z is skipped too (documented)
    int x0, x1, y0, y1;
    if (!axis_range(px, S, hw, x0, x1) || !axis_range(py, S, hw, y0, y1)) return CULLED;
    return (uint32_t)(x0 / TILE) | ((uint32_t)(y0 / TILE) << 8) | ((uint32_t)(x1 / TILE) << 16) |
           ((uint32_t)(y1 / TILE) << 24);
}

// Tiles per frame up to which a workgroup aggregates its tile counters in LDS before touching the global ones.
// The 256 points of a workgroup are neighbours in the source image, so they land on a few dozen tiles: one
// global atomic per (workgroup, touched tile) instead of one per (point, tile).
constexpr int LDS_TILES = 4096;
constexpr int MAX_TPP = 9;  // tiles per point kept in LDS ranks (larger footprints take the direct path)

__global__ __launch_bounds__(256) void k_bin_count(const float *__restrict__ pts, int N, int S,
                                                   float hw, int tilesX, int NT,
                                                   uint32_t *__restrict__ bbox,
                                                   uint32_t *__restrict__ tile_count)
{
    __shared__ uint32_t cnt[LDS_TILES];
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const bool agg = NT <= LDS_TILES;
    if (agg) {
        for (int t = threadIdx.x; t < NT; t += 256) cnt[t] = 0;
        __syncthreads();
    }
    if (n < N) {
        const uint32_t bb = point_bbox(pts + ((size_t)b * N + n) * 3, S, hw);
        bbox[(size_t)b * N + n] = bb;
        if (bb != CULLED) {
            const int tx0 = bb & 255, ty0 = (bb >> 8) & 255, tx1 = (bb >> 16) & 255, ty1 = bb >> 24;
            for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx) {
                    if (agg) atomicAdd(&cnt[ty * tilesX + tx], 1u);
                    else atomicAdd(&tile_count[(size_t)b * (NT + 1) + ty * tilesX + tx], 1u);
                }
        }
    }
    if (agg) {
        __syncthreads();
        for (int t = threadIdx.x; t < NT; t += 256)
            if (cnt[t]) atomicAdd(&tile_count[(size_t)b * (NT + 1) + t], cnt[t]);
    }
}

// exclusive scan of one frame's NT tile counters in place (one workgroup per frame); the offsets are then
// shifted by the frame's slice of the key array (frame_cap keys per frame), so no cross-frame scan is needed.
// counters layout: (B, NT + 1); entry NT receives the end of the frame's last list.
__global__ __launch_bounds__(1024) void k_scan(uint32_t *__restrict__ counters, int NT, uint32_t frame_cap)
{
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    uint32_t *c = counters + (size_t)blockIdx.x * (NT + 1);
    const uint32_t base = blockIdx.x * frame_cap;
    const int per = (NT + 1023) / 1024;
    const int beg = t * per, end = min(NT, beg + per);
    uint32_t s = 0;
    for (int i = beg; i < end; ++i) s += c[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = base + part[t] - s;
    for (int i = beg; i < end; ++i) {
        const uint32_t v = c[i];
        c[i] = run;
        run += v;
    }
    if (t == 1023) c[NT] = base + part[1023];
}

__global__ __launch_bounds__(256) void k_bin_fill(const float *__restrict__ pts, int N, int tilesX,
                                                  int NT, const uint32_t *__restrict__ bbox,
                                                  const uint32_t *__restrict__ tile_off,
                                                  uint32_t *__restrict__ tile_cursor,
                                                  uint64_t *__restrict__ keys)
{
    __shared__ uint32_t cnt[LDS_TILES];       // per-tile count of this workgroup, then its base in the tile's list
    __shared__ uint16_t rank[256][MAX_TPP];   // rank of each (point, tile) inside the workgroup's share
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    uint32_t bb = CULLED;
    if (n < N) bb = bbox[(size_t)b * N + n];
    const int tx0 = bb & 255, ty0 = (bb >> 8) & 255, tx1 = (bb >> 16) & 255, ty1 = bb >> 24;
    const bool live = bb != CULLED;
    // same-footprint decision for the whole workgroup (uniform branch): LDS aggregation or the direct path
    const bool fits = !live || (tx1 - tx0 + 1) * (ty1 - ty0 + 1) <= MAX_TPP;
    const bool agg = NT <= LDS_TILES && __syncthreads_and(fits);
    uint64_t key = 0;
    if (live) {
        const float z = pts[((size_t)b * N + n) * 3 + 2];
        const uint32_t zkey = (z == 0.0f) ? 0u : __float_as_uint(z);  // -0.0 == +0.0 in the tuple compare
        key = ((uint64_t)zkey << 32) | (uint32_t)n;
    }
    if (!agg) {
        if (live)
            for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx) {
                    const int t = ty * tilesX + tx;
                    keys[tile_off[(size_t)b * (NT + 1) + t] + atomicAdd(&tile_cursor[(size_t)b * NT + t], 1u)] = key;
                }
        return;
    }
    for (int t = threadIdx.x; t < NT; t += 256) cnt[t] = 0;
    __syncthreads();
    if (live) {
        int k = 0;
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) rank[threadIdx.x][k++] = (uint16_t)atomicAdd(&cnt[ty * tilesX + tx], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NT; t += 256)
        if (cnt[t]) cnt[t] = tile_off[(size_t)b * (NT + 1) + t] + atomicAdd(&tile_cursor[(size_t)b * NT + t], cnt[t]);
    __syncthreads();
    if (live) {
        int k = 0;
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) keys[cnt[ty * tilesX + tx] + rank[threadIdx.x][k++]] = key;
    }
}

// ------------------------------------------------------------------------------------------
// per-tile sort: normalised bitonic network (every comparator ascending), so elements beyond n act
// as +inf without being stored.  `a` is LDS or global memory.
// ------------------------------------------------------------------------------------------
// steps are separated by a workgroup barrier -- or, for a single wave (THREADS == 64), by nothing but a compiler fence: the
// LDS executes one wave's accesses in order
template <int THREADS>
__device__ __forceinline__ void sort_step_sync()
{
    if (THREADS == 64) asm volatile("" ::: "memory");
    else __syncthreads();
}

template <int THREADS, typename Ptr>
__device__ __forceinline__ void bitonic_sort(Ptr a, int n)
{
    int lp = 0;
    while ((1 << lp) < n) ++lp;
    const int half = (1 << lp) >> 1;
    for (int lk = 1; lk <= lp; ++lk) {            // k = 2^lk: size of the bitonic blocks being merged
        for (int lj = lk - 1; lj >= 0; --lj) {    // j = 2^lj: compare distance (powers of two: shifts, no divisions)
            const int j = 1 << lj;
            for (int p = threadIdx.x; p < half; p += THREADS) {
                const int blk = p >> lj, w = p & (j - 1);
                int i, l;
                if (lj == lk - 1) {               // first step of a merge: mirror inside the block of size k
                    i = (blk << lk) + w;
                    l = (blk << lk) + (1 << lk) - 1 - w;
                } else {
                    i = (blk << (lj + 1)) + w;
                    l = i + j;
                }
                if (l < n) {
                    const uint64_t x = a[i], y = a[l];
                    if (x > y) { a[i] = y; a[l] = x; }
                }
            }
            sort_step_sync<THREADS>();
        }
    }
}

// one wave per tile (grid: tiles x frames); lists longer than SORT_SMALL_CAP are queued for k_sort_big.
// The keys live in REGISTERS, KPL per lane (element e = lane * KPL + r), padded with +inf: the comparators of a bitonic network at
// distance j < KPL are compare-exchanges between two registers of a lane, the others an exchange with lane ^ (j / KPL) -- two
// cross-lane moves, a 64-bit compare and two selects per key and step, no address arithmetic, no loop: about a third of the
// instructions of the same network run through LDS (round 2: one wave, indices and bounds computed per comparator).  Keys are unique,
// so the result is THE sorted list whatever the network.
// lane ^ M's value: a DPP quad permutation for M = 1, 2 (no LDS instruction at all), ds_swizzle's xor mode inside 32 lanes for
// M = 4, 8, 16 (no address register), ds_bpermute for 32
template <int M>
__device__ __forceinline__ uint32_t xor_lane32(uint32_t v)
{
    if (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if (M < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (M << 10) | 0x1F);          // bit-mask mode: and 0x1f, or 0, xor M
    else return (uint32_t)__shfl_xor((int)v, M, 64);
}
template <int M>
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v)
{
    return ((uint64_t)xor_lane32<M>((uint32_t)(v >> 32)) << 32) | xor_lane32<M>((uint32_t)v);
}
template <int KPL>
__device__ __forceinline__ void wave_sort(uint64_t (&v)[KPL], int lane)
{
    constexpr int N = 64 * KPL;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j < KPL) {
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    const int p = r ^ j;
                    if (p < r) continue;
                    const bool up = ((lane * KPL + r) & k) == 0;      // ascending block?  (bit k is the same for r and p: k > j)
                    const bool sw = (v[r] > v[p]) == up;
                    const uint64_t a = sw ? v[p] : v[r], b = sw ? v[r] : v[p];
                    v[r] = a; v[p] = b;
                }
            } else {
                constexpr int lj_max = 32;
                const int lj = j / KPL;
                const bool lower = (lane & lj) == 0;
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    uint64_t o;
                    switch (lj) {   // (compile-time: the loops are unrolled)
                    case 1: o = shfl_xor64<1>(v[r]); break;
                    case 2: o = shfl_xor64<2>(v[r]); break;
                    case 4: o = shfl_xor64<4>(v[r]); break;
                    case 8: o = shfl_xor64<8>(v[r]); break;
                    case 16: o = shfl_xor64<16>(v[r]); break;
                    default: o = shfl_xor64<lj_max>(v[r]); break;
                    }
                    const bool up = ((lane * KPL + r) & k) == 0;
                    v[r] = ((v[r] < o) == (lower == up)) ? v[r] : o;   // the lower lane of an ascending pair keeps the smaller key
                }
            }
        }
    }
}
template <int KPL>
__device__ __forceinline__ void sort_tile_regs(uint64_t *__restrict__ keys, uint32_t n, int lane)
{
    uint64_t v[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const uint32_t e = (uint32_t)lane * KPL + r;
        v[r] = e < n ? keys[e] : ~0ull;
    }
    wave_sort<KPL>(v, lane);
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const uint32_t e = (uint32_t)lane * KPL + r;
        if (e < n) keys[e] = v[r];
    }
}
__global__ __launch_bounds__(64) void k_sort_small(uint64_t *__restrict__ keys,
                                                   const uint32_t *__restrict__ tile_off, int NT,
                                                   uint32_t *__restrict__ worklist)
{
    const uint32_t t = blockIdx.y * (NT + 1) + blockIdx.x;  // index into the (B, NT+1) offset table
    const uint32_t beg = tile_off[t], n = tile_off[t + 1] - beg;
    if (n < 2) return;
    if (n > SORT_SMALL_CAP) {
        if (threadIdx.x == 0) worklist[1 + atomicAdd(&worklist[0], 1u)] = t;
        return;
    }
    static_assert(SORT_SMALL_CAP == 512, "eight keys per lane at most");
    const int lane = threadIdx.x;
    if (n <= 64) sort_tile_regs<1>(keys + beg, n, lane);
    else if (n <= 128) sort_tile_regs<2>(keys + beg, n, lane);
    else if (n <= 256) sort_tile_regs<4>(keys + beg, n, lane);
    else sort_tile_regs<8>(keys + beg, n, lane);
}

__global__ __launch_bounds__(1024) void k_sort_big(uint64_t *__restrict__ keys,
                                                   const uint32_t *__restrict__ tile_off,
                                                   const uint32_t *__restrict__ worklist)
{
    __shared__ uint64_t s[SORT_BIG_CAP];
    const uint32_t cnt = worklist[0];
    for (uint32_t w = blockIdx.x; w < cnt; w += gridDim.x) {
        const uint32_t t = worklist[1 + w];
        const uint32_t beg = tile_off[t], n = tile_off[t + 1] - beg;
        if (n <= SORT_BIG_CAP) {
            for (uint32_t i = threadIdx.x; i < n; i += 1024) s[i] = keys[beg + i];
            __syncthreads();
            bitonic_sort<1024>(s, (int)n);
            for (uint32_t i = threadIdx.x; i < n; i += 1024) keys[beg + i] = s[i];
            __syncthreads();
        } else {
            bitonic_sort<1024>(keys + beg, (int)n);  // rare: degenerate clouds piling into one tile
        }
    }
}

// ------------------------------------------------------------------------------------------
// composite: one wave per 8x8 tile, one lane per pixel
// ------------------------------------------------------------------------------------------
constexpr int CG = 4;  // channels accumulated per wave; grid.z walks channel groups
#ifndef PS_COMPOSITE_WAVES
#define PS_COMPOSITE_WAVES 8
#endif
#ifndef PS_COMPOSITE_PK
#define PS_COMPOSITE_PK 1
#endif

// Correctly rounded square root of d in [1e-3, 1] (the clamped dist^2 / r^2 of a hit): the hardware's 1-ulp v_sqrt_f32 and the check of
// its two neighbours that sqrtf itself expands to -- without sqrtf's rescaling of inputs below 2^-96 and its pass-through of 0 / inf,
// 7 of its 16 instructions, in a loop body of ~46.  Equal to sqrtf bit for bit on the whole range (tools/sqrt_probe.hip checks
// every float in [2^-20, 2]).
__device__ __forceinline__ float sqrt_rn_unit(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float em = __builtin_fmaf(-sm, s, x), ep = __builtin_fmaf(-sp, s, x);
    float r = em <= 0.0f ? sm : s;
    r = ep > 0.0f ? sp : r;
    return r;
}

__device__ __forceinline__ float bcast(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Two phases per batch of 64 sorted records (one record staged per lane, shared through LDS):
//   1. every lane runs the exact strict disc test against all 64 records (wave-uniform LDS broadcasts) and
//      keeps one bit per record -- cheap, branch-free;
//   2. every lane walks only ITS OWN hits (ascending bit order = ascending (z, idx)), fetching the record from
//      LDS: alpha from dist^2, front-to-back blend, stop at K hits.  ~50 hits per pixel instead of ~256
//      predicated blend bodies per pixel.
struct __attribute__((aligned(16))) SplatRec { float x, y, z; uint32_t n; float f[CG]; };

// NC: channels a wave carries -- CG, or 3 when the features have exactly three (RGB: every configuration of the reference): the
// fourth accumulator of a group of four costs one LDS read and two vector instructions per hit of a walk of ~46
template <int MODE, bool DEBUG_OUT, bool RECIP, int NC = CG>
__attribute__((amdgpu_waves_per_eu(PS_COMPOSITE_WAVES, PS_COMPOSITE_WAVES)))
__global__ __launch_bounds__(64) void k_composite(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ tile_off,
    const float *__restrict__ pts, const float *__restrict__ feat, int N, int C, int S, int tilesX,
    int NT, float r2, float denom, float tau, int K, float *__restrict__ out_feat,
    uint8_t *__restrict__ bg0, int32_t *__restrict__ out_idx, float *__restrict__ out_zbuf,
    float *__restrict__ out_dist)
{
    // records of a batch, structure of arrays: in the walk every lane fetches a DIFFERENT record, and 32-byte structs put
    // records j and j + 4 on the same banks (up to 16 lanes per bank); 4-byte columns put j and j + 32 there (2 lanes).
    // Round 3, measured and not kept (all 19 parity checks green; 32 frames, this kernel 405 us as it is): the record as two
    // 16-byte halves / as 8-byte pairs with TWO hits fetched and their alphas computed per trip of the walk (independent work
    // over the LDS round trip) -- both need more than the 64 registers of eight waves per SIMD (spilled there: 3.0-3.2 ms); with
    // 96 / 103 registers at five / four waves per SIMD 417 / 435 us; the hit mask as two 32-bit words: 405 us.  The walk is not
    // short of independent instructions; what it pays for is the LDS itself (about 900 LDS instructions per tile, half of them
    // the broadcast reads of the test phase) at eight waves per SIMD.
    // What did pay (second half of round 3; 32 frames: 432 -> 362 us): the kernel is bound by vector-ALU ISSUE (the eight waves of a SIMD
    // together execute 2.5 instructions' worth per instruction slot), so instruction count is the lever -- the test phase two records
    // per packed instruction with the hit mask built by a carry chain (10 -> 5.5 instructions per record), and the square root of a hit
    // without sqrtf's handling of ranges its argument cannot be in (16 -> 9 instructions of the walk's ~46 per hit).
    __shared__ __attribute__((aligned(8))) float sx[64], sy[64], sz[64], sf[NC][64];
    __shared__ uint32_t sn[64];
    const int tile = blockIdx.x, b = blockIdx.y, c0 = blockIdx.z * CG;
    const int lane = threadIdx.x;
    const int tx = tile % tilesX, ty = tile / tilesX;
    const int xi = tx * TILE + (lane & 7), yi = ty * TILE + (lane >> 3);
    const bool valid = xi < S && yi < S;
    const float xf = pix_to_ndc(S - 1 - xi, S), yf = pix_to_ndc(S - 1 - yi, S);
    const uint32_t beg = tile_off[(size_t)b * (NT + 1) + tile], end = tile_off[(size_t)b * (NT + 1) + tile + 1];
    const int ncg = min(NC, C - c0);
    const size_t pix = ((size_t)b * S + yi) * S + xi;

    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0f;
    float cum = 1.0f, tsum = 0.0f;
    int cnt = 0;
    // Alpha compositing front to back: what the hits behind the current one can still add is bounded by the transmittance,
    // sum_i cum_i a_i |f_i| <= cum * max|f| (the weights telescope to cum - cum_end).  Once cum < 2^-23 that is below one ulp of a
    // unit-magnitude result, and a lane stops walking (round 6: ~40 instead of ~50 hits per pixel at r = 4 px, where a hit's mean alpha
    // is 1/3).  Only the features are affected, within the 1e-6 the parity tests state for them; the background mask needs ONE hit, and
    // the mode that emits idx / zbuf / dist (DEBUG_OUT: the bit-exact K-nearest lists) walks everything.
    constexpr bool EARLY = MODE == PS_ACC_ALPHACOMPOSITE && !DEBUG_OUT;
    constexpr float CUM_EPS = 1.1920929e-07f;   // 2^-23
    auto open_lane = [&]() { return cnt < K && (!EARLY || cum >= CUM_EPS); };

    for (int pass = (MODE == PS_ACC_WSUMNORM ? 0 : 1); pass < 2; ++pass) {
        cnt = 0;
        if (MODE == PS_ACC_WSUMNORM && pass == 1 && tsum < 1e-4f) tsum = 1e-4f;
        // software pipeline: the record of batch i+1 is fetched (two dependent global round trips: key -> point,
        // features) while batch i is tested and blended
        auto fetch = [&](uint32_t base) {
            SplatRec r;
            r.x = INFINITY; r.y = 0.0f; r.z = 0.0f; r.n = 0;  // lanes past the end carry x = +inf: never a hit
#pragma unroll
            for (int c = 0; c < NC; ++c) r.f[c] = 0.0f;
            if (base < end && lane < (int)min(64u, end - base)) {
                r.n = (uint32_t)keys[base + lane];
                const float *p = pts + ((size_t)b * N + r.n) * 3;
                r.x = p[0];
                r.y = p[1];
                r.z = p[2];
                if (pass == 1) {
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        if (c < ncg) r.f[c] = feat[((size_t)b * C + c0 + c) * N + r.n];
                }
            }
            return r;
        };
        SplatRec rnext = fetch(beg);
        for (uint32_t base = beg; base < end; base += 64) {
            const SplatRec r = rnext;
            rnext = fetch(base + 64);
            __syncthreads();  // the previous batch's phase 2 is done with rec[]
            sx[lane] = r.x; sy[lane] = r.y; sz[lane] = r.z; sn[lane] = r.n;
#pragma unroll
            for (int c = 0; c < NC; ++c) sf[c][lane] = r.f[c];
            __syncthreads();
            // phase 1: hit bits
            uint64_t hits = 0;
            if (valid && open_lane()) {
#if PS_COMPOSITE_PK
                // Two records per packed instruction (v_pk_add / v_pk_mul: the same IEEE operations, two at a time), and the mask
                // built by the carry chain: v_cmp writes vcc, v_addc computes h + h + vcc = (h << 1) | hit -- one instruction per
                // record instead of select + shift + or.  Records go in descending order, so record j ends up at bit j of its
                // 32-bit word; a block of 16 with no record in it (wave-uniform) is sixteen zero bits.
                const int nrec = (int)min(64u, end - base);
                typedef float pk2 __attribute__((ext_vector_type(2)));
                const pk2 xf2 = {xf, xf}, yf2 = {yf, yf};
                const float r2v = r2;
                uint32_t hw[2] = {0u, 0u};
#pragma unroll
                for (int blk = 3; blk >= 0; --blk) {
                    uint32_t h = hw[blk >> 1];
                    if (blk * 16 < nrec) {
#pragma unroll
                        for (int jj = 14; jj >= 0; jj -= 2) {
                            const int j = blk * 16 + jj;                       // records j + 1, then j
                            const pk2 xs = *(const pk2 *)&sx[j], ys = *(const pk2 *)&sy[j];
                            const pk2 dx = xs - xf2, dy = ys - yf2;
                            const pk2 d2 = dx * dx + dy * dy;
                            asm("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(h) : "v"(d2.y), "v"(r2v) : "vcc");
                            asm("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(h) : "v"(d2.x), "v"(r2v) : "vcc");
                        }
                    } else {
                        h <<= 16;
                    }
                    hw[blk >> 1] = h;
                }
                hits = ((uint64_t)hw[1] << 32) | hw[0];
#else
                const int nrec = (int)min(64u, end - base);   // (wave-uniform) records staged in this batch, in blocks of 16
                for (int j0 = 0; j0 < nrec; j0 += 16) {
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) {
                        const int j = j0 + jj;
                        const float dx = sx[j] - xf, dy = sy[j] - yf;
                        const float d2 = dx * dx + dy * dy;
                        hits |= (uint64_t)(d2 < r2) << j;
                    }
                }
#endif
            }
            // phase 2: this pixel's hits, front to back
            while (hits && open_lane()) {
                const int j = __builtin_ctzll(hits);
                hits &= hits - 1;
                SplatRec h;
                h.x = sx[j]; h.y = sy[j]; h.z = sz[j]; h.n = sn[j];
#pragma unroll
                for (int c = 0; c < NC; ++c) h.f[c] = sf[c][j];
                const float dx = h.x - xf, dy = h.y - yf;
                const float d2 = dx * dx + dy * dy;
                float d = RECIP ? d2 * denom : d2 / denom;
                d = __builtin_amdgcn_fmed3f(d, 1e-3f, 1.0f);   // = fminf(fmaxf(d, 1e-3f), 1.0f) for every d that is a number, one instruction
                // The product route (EARLY) takes the hardware's square root as it comes (1 ulp; sqrt_rn_unit's correction is 7 of the walk's
                // ~38 vector instructions): with the early-out and the fused sum the features stay within 3e-7 x max |feature| of the
                // oracle's (measured on pile-ups and image-like clouds; 1.8e-7 with the exact root), inside the 1e-6 the parity tests
                // state.  The list-emitting route -- the one the bit-exact checks go through -- keeps the correctly rounded root.
                float a = 1.0f - (EARLY ? __builtin_amdgcn_sqrtf(d) : sqrt_rn_unit(d));
                if (tau != 1.0f) a = powf(a, tau);
                if (MODE == PS_ACC_WSUMNORM && pass == 0) {
                    tsum = tsum + a;
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const float f = h.f[c];
                        // PyTorch3D: cum_alpha * alpha * feature.  The product route (EARLY: features at the stated tolerance, no K-nearest
                        // lists) takes the sum as one fused multiply-add -- what nvcc makes of the reference's own line, and three instead of
                        // five vector instructions per hit for RGB; the list-emitting route keeps the separately rounded product and sum.
                        if (MODE == PS_ACC_ALPHACOMPOSITE) acc[c] = EARLY ? __builtin_fmaf(cum * a, f, acc[c]) : acc[c] + cum * a * f;
                        else if (MODE == PS_ACC_WSUM) acc[c] = acc[c] + f * a;
                        else acc[c] = acc[c] + f * a / tsum;
                    }
                    if (MODE == PS_ACC_ALPHACOMPOSITE) cum = cum * (1.0f - a);
                    if (DEBUG_OUT && blockIdx.z == 0) {
                        if (out_idx) out_idx[pix * K + cnt] = (int32_t)(b * N + h.n);
                        if (out_zbuf) out_zbuf[pix * K + cnt] = h.z;
                        if (out_dist) out_dist[pix * K + cnt] = d2;
                    }
                }
                ++cnt;
            }
            if (__all(!open_lane() || !valid)) break;
        }
    }
    if (!valid) return;
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (c < ncg) out_feat[((size_t)b * C + c0 + c) * S * S + (size_t)yi * S + xi] = acc[c];
    if (blockIdx.z == 0) {
        bg0[pix] = (uint8_t)(cnt == 0);
        if (DEBUG_OUT) {
            for (int k = cnt; k < K; ++k) {
                if (out_idx) out_idx[pix * K + k] = -1;
                if (out_zbuf) out_zbuf[pix * K + k] = -1.0f;
                if (out_dist) out_dist[pix * K + k] = -1.0f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// background-mask dilation: conv2d(ones k x k, zero pad k//2) > 0   (z_buffer_layers.py:105-110)
// ------------------------------------------------------------------------------------------
constexpr int DT = 32;        // output tile edge
constexpr int DMAXH = 15;     // supports ksize <= 31
__global__ __launch_bounds__(256) void k_dilate(const uint8_t *__restrict__ bg0, int S, int h,
                                                uint8_t *__restrict__ bg)
{
    __shared__ uint8_t in[DT + 2 * DMAXH][DT + 2 * DMAXH + 2];
    __shared__ uint8_t row[DT + 2 * DMAXH][DT];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * DT, y0 = blockIdx.y * DT;
    const int E = DT + 2 * h;
    for (int i = threadIdx.x; i < E * E; i += 256) {
        const int ly = i / E, lx = i % E;
        const int y = y0 + ly - h, x = x0 + lx - h;
        in[ly][lx] = (y >= 0 && y < S && x >= 0 && x < S) ? bg0[((size_t)b * S + y) * S + x] : 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < E * DT; i += 256) {
        const int ly = i / DT, lx = i % DT;
        uint8_t m = 0;
        for (int d = 0; d <= 2 * h; ++d) m |= in[ly][lx + d];
        row[ly][lx] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < DT * DT; i += 256) {
        const int ly = i / DT, lx = i % DT;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= S || x >= S) continue;
        uint8_t m = 0;
        for (int d = 0; d <= 2 * h; ++d) m |= row[ly + d][lx];
        bg[((size_t)b * S + y) * S + x] = m ? 1 : 0;
    }
}

// The same dilation on BIT ROWS (round 6; S a multiple of 64, at most 2048): a wave's 64 one-byte loads of a row segment are one coalesced
// 64-byte request and its ballot is the segment as a 64-bit word; a band of 16 output rows keeps its 16 + 2 h input rows as words in LDS,
// the horizontal pass is shifts across word boundaries (OR over -h .. h), the vertical one an OR over 2 h + 1 rows, and a word goes back
// out as 64 coalesced bytes.  k_dilate moved every byte through LDS twice, a byte per access (57 us alone per 128 frames of 256 x 256,
// ~110 beside the prefix pass: the splat's kernels add to the step one for one).
constexpr int DB_ROWS = 16, DB_MAXW = 2048 / 64;   // (bands of 16 rows: 16 x B workgroups fill the chip from 16 frames on; 64-row bands ran 32 frames on 128 workgroups: C2 0.407 -> 0.442 ms)
__global__ __launch_bounds__(256) void k_dilate_bits(const uint8_t *__restrict__ bg0, int S, int h, uint8_t *__restrict__ bg)
{
    __shared__ unsigned long long m[DB_ROWS + 2 * DMAXH][DB_MAXW], mh[DB_ROWS + 2 * DMAXH][DB_MAXW];
    const int b = blockIdx.y, y0 = blockIdx.x * DB_ROWS, W64 = S >> 6, R = DB_ROWS + 2 * h;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint8_t *src = bg0 + (size_t)b * S * S;
    for (int q = wave; q < R * W64; q += 4) {          // (row, word) pairs: one 64-byte load and one ballot each
        const int r = q / W64, k = q - r * W64, y = y0 + r - h;
        const uint8_t v = (y >= 0 && y < S) ? src[(size_t)y * S + 64 * k + lane] : 0;
        const unsigned long long bits = __ballot(v != 0);
        if (lane == 0) m[r][k] = bits;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < R * W64; q += 256) {  // horizontal: bit i = pixel 64 k + i; out bit i = OR of the bits i - h .. i + h
        const int r = q / W64, k = q - r * W64;
        const unsigned long long x = m[r][k], left = k > 0 ? m[r][k - 1] : 0ull, right = k + 1 < W64 ? m[r][k + 1] : 0ull;
        unsigned long long o = x;
        for (int d = 1; d <= h; ++d) o |= (x >> d) | (right << (64 - d)) | (x << d) | (left >> (64 - d));
        mh[r][k] = o;
    }
    __syncthreads();
    uint8_t *dst = bg + (size_t)b * S * S;
    for (int q = wave; q < DB_ROWS * W64; q += 4) {     // vertical, and a word back out as 64 bytes
        const int r = q / W64, k = q - r * W64, y = y0 + r;
        if (y >= S) continue;                           // (wave-uniform)
        unsigned long long o = 0;
        for (int d = 0; d <= 2 * h; ++d) o |= mh[r + d][k];
        dst[(size_t)y * S + 64 * k + lane] = (uint8_t)((o >> lane) & 1ull);
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct SplatPlan {
    int tilesX, NT, max_tiles_pp;
    float hw;
    size_t off_pts, off_bbox, off_count, off_cursor, off_work, off_bg0, off_keys, total;
};

SplatPlan make_plan(int B, int N, int S, double radius_px)
{
    SplatPlan p;
    p.tilesX = (S + TILE - 1) / TILE;
    p.NT = p.tilesX * p.tilesX;
    p.hw = (float)(radius_px * 1.0001 + 0.01);
    const int span_px = (int)(2.0 * p.hw) + 2;
    const int span_t = (span_px + TILE - 2) / TILE + 1;
    p.max_tiles_pp = span_t * span_t;
    size_t o = 0;
    const size_t M = (size_t)B * p.NT;
    p.off_pts = o;    o = ps::align_up(o + (size_t)B * N * 3 * sizeof(float), 256);
    p.off_bbox = o;   o = ps::align_up(o + (size_t)B * N * sizeof(uint32_t), 256);
    p.off_count = o;  o = ps::align_up(o + (M + B) * sizeof(uint32_t), 256);   // (B, NT+1) counters -> offsets
    p.off_cursor = o; o = ps::align_up(o + M * sizeof(uint32_t), 256);
    p.off_work = o;   o = ps::align_up(o + (M + 1) * sizeof(uint32_t), 256);
    p.off_bg0 = o;    o = ps::align_up(o + (size_t)B * S * S, 256);
    p.off_keys = o;   o = ps::align_up(o + (size_t)B * N * p.max_tiles_pp * sizeof(uint64_t), 256);
    p.total = o;
    return p;
}

template <int MODE>
void launch_composite(bool debug, bool recip, dim3 grid, hipStream_t st, const uint64_t *keys,
                      const uint32_t *tile_off, const float *pts, const float *feat, int N, int C,
                      int S, int tilesX, int NT, float r2, float denom, float tau, int K,
                      float *out_feat, uint8_t *bg0, int32_t *out_idx, float *out_zbuf,
                      float *out_dist)
{
#define PS_COMPOSITE(DBG, RCP, NCH)                                                                \
    hipLaunchKernelGGL((k_composite<MODE, DBG, RCP, NCH>), grid, dim3(64), 0, st, keys, tile_off,  \
                       pts, feat, N, C, S, tilesX, NT, r2, denom, tau, K, out_feat, bg0, out_idx,  \
                       out_zbuf, out_dist)
    if (debug) { if (recip) PS_COMPOSITE(true, true, CG); else PS_COMPOSITE(true, false, CG); }
    else if (C == 3) { if (recip) PS_COMPOSITE(false, true, 3); else PS_COMPOSITE(false, false, 3); }
    else       { if (recip) PS_COMPOSITE(false, true, CG); else PS_COMPOSITE(false, false, CG); }
#undef PS_COMPOSITE
}

// rasterize + composite + dilate on pts (B,N,3) already in PyTorch3D orientation (x,y negated)
int splat_core(const float *pts, const float *feat, int B, int N, int C, int S, double radius_px,
               int K, float tau, int rad_pow, int accumulation, int bg_ksize, float *out_feat,
               uint8_t *out_bg, int32_t *out_idx, float *out_zbuf, float *out_dist, char *ws,
               const SplatPlan &p, hipStream_t st, bool counters_cleared = false)
{
    uint32_t *bbox = (uint32_t *)(ws + p.off_bbox);
    uint32_t *tile_off = (uint32_t *)(ws + p.off_count);
    uint32_t *cursor = (uint32_t *)(ws + p.off_cursor);
    uint32_t *work = (uint32_t *)(ws + p.off_work);
    uint8_t *bg0 = (uint8_t *)(ws + p.off_bg0);
    uint64_t *keys = (uint64_t *)(ws + p.off_keys);

    // radius exactly as the reference computes it: python double, then float at the PyTorch3D call
    const double radius = radius_px / (double)S * 2.0;            // z_buffer_layers.py:77
    const float rf = (float)radius;
    const float r2 = rf * rf;
    const float denom = (float)pow(radius, (double)rad_pow);       // :89
    int e = 0;
    const bool pow2 = std::frexp(denom, &e) == 0.5f;               // then d2/denom == d2*(1/denom) exactly
    const float denom_arg = pow2 ? 1.0f / denom : denom;

    // counters (count + cursor + worklist are adjacent) start from zero every call
    // (the fused project + splat call has them cleared by its projection kernel)
    if (!counters_cleared) PS_HIP_CHECK(hipMemsetAsync(ws + p.off_count, 0, p.off_bg0 - p.off_count, st));
    const dim3 gpt((N + 255) / 256, B);
    hipLaunchKernelGGL(k_bin_count, gpt, dim3(256), 0, st, pts, N, S, p.hw, p.tilesX, p.NT, bbox, tile_off);
    hipLaunchKernelGGL(k_scan, dim3(B), dim3(1024), 0, st, tile_off, p.NT, (uint32_t)((size_t)N * p.max_tiles_pp));
    hipLaunchKernelGGL(k_bin_fill, gpt, dim3(256), 0, st, pts, N, p.tilesX, p.NT, bbox, tile_off, cursor, keys);
    hipLaunchKernelGGL(k_sort_small, dim3(p.NT, B), dim3(64), 0, st, keys, tile_off, p.NT, work);
    hipLaunchKernelGGL(k_sort_big, dim3(128), dim3(1024), 0, st, keys, tile_off, work);
    const dim3 gc(p.NT, B, (C + CG - 1) / CG);
    const bool debug = out_idx || out_zbuf || out_dist;
    switch (accumulation) {
    case PS_ACC_ALPHACOMPOSITE:
        launch_composite<PS_ACC_ALPHACOMPOSITE>(debug, pow2, gc, st, keys, tile_off, pts, feat, N, C, S, p.tilesX,
                                                p.NT, r2, denom_arg, tau, K, out_feat, bg0, out_idx, out_zbuf, out_dist);
        break;
    case PS_ACC_WSUM:
        launch_composite<PS_ACC_WSUM>(debug, pow2, gc, st, keys, tile_off, pts, feat, N, C, S, p.tilesX, p.NT, r2,
                                      denom_arg, tau, K, out_feat, bg0, out_idx, out_zbuf, out_dist);
        break;
    default:
        launch_composite<PS_ACC_WSUMNORM>(debug, pow2, gc, st, keys, tile_off, pts, feat, N, C, S, p.tilesX, p.NT,
                                          r2, denom_arg, tau, K, out_feat, bg0, out_idx, out_zbuf, out_dist);
        break;
    }
    if (S % 64 == 0 && S <= 64 * DB_MAXW) {
        hipLaunchKernelGGL(k_dilate_bits, dim3((S + DB_ROWS - 1) / DB_ROWS, B), dim3(256), 0, st, bg0, S, bg_ksize / 2, out_bg);
    } else {
        const dim3 gd((S + DT - 1) / DT, (S + DT - 1) / DT, B);
        hipLaunchKernelGGL(k_dilate, gd, dim3(256), 0, st, bg0, S, bg_ksize / 2, out_bg);
    }
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int check_splat_args(int B, int N, int C, int S, double radius_px, int K, int accumulation, int bg_ksize)
{
    PS_REQUIRE(B > 0 && N > 0 && C > 0 && S > 1 && K > 0, "splat: B,N,C,K must be > 0 and S > 1");
    PS_REQUIRE(S <= 2048, "splat: image size %d > 2048 unsupported", S);
    PS_REQUIRE(radius_px > 0 && radius_px <= 64, "splat: radius_px %.3f out of range (0,64]", radius_px);
    PS_REQUIRE(accumulation >= 0 && accumulation <= 2, "splat: unknown accumulation %d", accumulation);
    PS_REQUIRE(bg_ksize >= 1 && (bg_ksize & 1) && bg_ksize <= 2 * DMAXH + 1, "splat: bg_ksize %d must be odd and <= %d",
               bg_ksize, 2 * DMAXH + 1);
    PS_REQUIRE((size_t)B * N < 0x7FFFFFFFull, "splat: B*N overflows the packed int32 index");
    PS_REQUIRE((size_t)B * N * 16 < 0xFFFFFFFFull, "splat: B*N too large for 32-bit key offsets");
    return PS_OK;
}

}  // namespace

extern "C" {

size_t ps_splat_workspace_bytes(int B, int N, int S, double radius_px)
{
    if (B <= 0 || N <= 0 || S <= 1 || radius_px <= 0) return 0;
    return make_plan(B, N, S, radius_px).total;
}

int ps_project_pts_f32(const float *depth, const float *K, const float *Kinv, const float *RT1inv,
                       const float *RT2, int B, int W, float *sampler, void *stream)
{
    PS_REQUIRE(depth && K && Kinv && RT1inv && RT2 && sampler, "project_pts: null pointer");
    PS_REQUIRE(B > 0 && W > 1, "project_pts: B > 0 and W > 1 required");
    const int N = W * W;
    hipLaunchKernelGGL((k_project<0, false>), dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, depth,
                       (const int32_t *)nullptr, K, Kinv, RT1inv, RT2, W, N, N, 0, sampler, (float *)nullptr);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_project_pts_cumulative_f32(const float *depth_new, const int32_t *new_index, const float *prior,
                                  const float *K, const float *Kinv, const float *RT1inv, const float *RT2,
                                  const float *RT3inv, int B, int W, int n_new, int n_prior, float *sampler,
                                  float *cloud, void *stream)
{
    PS_REQUIRE(K && Kinv && RT1inv && RT2 && sampler && cloud, "project_pts_cumulative: null pointer");
    PS_REQUIRE(B > 0 && W > 1 && n_new >= 0 && n_prior >= 0 && n_new + n_prior > 0, "project_pts_cumulative: bad sizes");
    PS_REQUIRE(new_index || n_new == W * W || n_new == 0, "project_pts_cumulative: n_new != W*W needs new_index");
    PS_REQUIRE(n_prior == 0 || (prior && RT3inv), "project_pts_cumulative: prior cloud needs RT3inv");
    const int NTt = n_new + n_prior;
    hipStream_t st = (hipStream_t)stream;
    if (n_new > 0) {
        PS_REQUIRE(depth_new, "project_pts_cumulative: depth_new is null");
        hipLaunchKernelGGL((k_project<0, false>), dim3((n_new + 255) / 256, B), dim3(256), 0, st, depth_new, new_index,
                           K, Kinv, RT1inv, RT2, W, n_new, NTt, 0, sampler, cloud);
    }
    if (n_prior > 0)
        hipLaunchKernelGGL((k_project<0, true>), dim3((n_prior + 255) / 256, B), dim3(256), 0, st, prior,
                           (const int32_t *)nullptr, K, Kinv, RT3inv, RT2, W, n_prior, NTt, n_new, sampler, cloud);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_splat_f32(float *pts, const float *feat, int B, int N, int C, int S, double radius_px, int K, float tau,
                 int rad_pow, int accumulation, int bg_ksize, float *out_feat, uint8_t *out_bg, int32_t *out_idx,
                 float *out_zbuf, float *out_dist, void *workspace, size_t workspace_bytes, void *stream)
{
    PS_REQUIRE(pts && feat && out_feat && out_bg && workspace, "splat: null pointer");
    if (int rc = check_splat_args(B, N, C, S, radius_px, K, accumulation, bg_ksize)) return rc;
    const SplatPlan p = make_plan(B, N, S, radius_px);
    if (workspace_bytes < p.total)
        return ps::fail(PS_ERR_WORKSPACE, "splat: workspace %zu < required %zu bytes", workspace_bytes, p.total);
    hipStream_t st = (hipStream_t)stream;
    const size_t npts = (size_t)B * N;
    hipLaunchKernelGGL(k_negate_xy, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, pts, npts);
    return splat_core(pts, feat, B, N, C, S, radius_px, K, tau, rad_pow, accumulation, bg_ksize, out_feat, out_bg,
                      out_idx, out_zbuf, out_dist, (char *)workspace, p, st);
}

int ps_project_splat_f32(const float *depth, const float *feat, const float *K, const float *Kinv,
                         const float *RT1inv, const float *RT2, int B, int C, int S, double radius_px, int Kpp,
                         float tau, int rad_pow, int accumulation, int bg_ksize, float *out_feat, uint8_t *out_bg,
                         void *workspace, size_t workspace_bytes, void *stream)
{
    PS_REQUIRE(depth && feat && K && Kinv && RT1inv && RT2 && out_feat && out_bg && workspace,
               "project_splat: null pointer");
    const int N = S * S;
    if (int rc = check_splat_args(B, N, C, S, radius_px, Kpp, accumulation, bg_ksize)) return rc;
    const SplatPlan p = make_plan(B, N, S, radius_px);
    if (workspace_bytes < p.total)
        return ps::fail(PS_ERR_WORKSPACE, "project_splat: workspace %zu < required %zu bytes", workspace_bytes, p.total);
    hipStream_t st = (hipStream_t)stream;
    float *pts = (float *)((char *)workspace + p.off_pts);
    hipLaunchKernelGGL((k_project<1, false>), dim3((N + 255) / 256, B), dim3(256), 0, st, depth,
                       (const int32_t *)nullptr, K, Kinv, RT1inv, RT2, S, N, N, 0, pts, (float *)nullptr,
                       (uint32_t *)((char *)workspace + p.off_count), (unsigned)((p.off_bg0 - p.off_count) / sizeof(uint32_t)));
    return splat_core(pts, feat, B, N, C, S, radius_px, Kpp, tau, rad_pow, accumulation, bg_ksize, out_feat, out_bg,
                      nullptr, nullptr, nullptr, (char *)workspace, p, st, true);
}

// ------------------------------------------------------------------------------------------
// Hard z-buffer of DepthManipulator.project_zbuffer (models/projection/depth_manipulator.py:37-104).
//   k_zb_project   :43-66 -- one thread per source pixel: p = grid * depth (w = 1), X = K (RT2 RT1inv) Kinv p in the
//                  association order of the reference's products (mul, add, ascending k; this unit is built with
//                  -ffp-contract=off), EPS rule, the literal (sampler + 1) * 128 pixel mapping, .long().clamp(0, 255) and the
//                  out-of-range flag (:86-87).  Everything by ORIGINAL point position; the sort by z stays with the caller.
//   k_zb_test / k_zb_write   the z-test: the reference scatters the sorted points with an indexed assignment, so that of
//                  several points on one pixel the LAST one in sorted order stays (sequential semantics of torch's CPU
//                  index_put_).  Here: atomic max of the sorted position per pixel, then every winner writes its values.
//                  With `order` (the caller's argsort) the per-point data is read through it and the values are built in
//                  place: v0 = grid_x[order[n]] + flag[n], v1 = -grid_y[order[n]] + flag[n] -- the flag by ORIGINAL position
//                  on values in SORTED order, a quirk of the reference that is kept (:88-97).
// A pixel outside the image raises bit 1 of the caller's status word (ps_read_status) and the point is dropped.
// ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_zb_project(const float *__restrict__ depth, const float *__restrict__ grid,
                                                    const float *__restrict__ K, const float *__restrict__ Kinv,
                                                    const float *__restrict__ RT1inv, const float *__restrict__ RT2, int N,
                                                    float *__restrict__ zproj, int32_t *__restrict__ ys, int32_t *__restrict__ xs,
                                                    float *__restrict__ flag)
{
    __shared__ float sRT[16], sK[16], sKinv[16];
    const int b = blockIdx.y;
    if (threadIdx.x < 16) {
        const int i = threadIdx.x >> 2, j = threadIdx.x & 3;
        const float *A = RT2 + b * 16, *Bm = RT1inv + b * 16;
        float acc = A[i * 4 + 0] * Bm[0 * 4 + j];
        acc = acc + A[i * 4 + 1] * Bm[1 * 4 + j];
        acc = acc + A[i * 4 + 2] * Bm[2 * 4 + j];
        acc = acc + A[i * 4 + 3] * Bm[3 * 4 + j];
        sRT[threadIdx.x] = acc;
        sK[threadIdx.x] = K[b * 16 + threadIdx.x];
        sKinv[threadIdx.x] = Kinv[b * 16 + threadIdx.x];
    }
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float d = depth[(size_t)b * N + n];
    float p[4] = {grid[n] * d, grid[N + n] * d, grid[2 * (size_t)N + n] * d, 1.0f}, c[4], w[4], X[4];
    mat4_vec(sKinv, p, c);
    mat4_vec(sRT, c, w);
    mat4_vec(sK, w, X);
    const bool bad = fabsf(X[2]) < PS_EPS;
    float sx = X[0] / -X[2], sy = X[1] / -X[2];
    if (bad) { sx = -10.0f; sy = -10.0f; }
    sy = -sy;
    const float tx = (sx + 1.0f) * 128.0f, ty = (sy + 1.0f) * 128.0f;
    // .long() truncates towards zero, then clamp(0, 255); NaN -> 0 like the reference's LONG_MIN
    auto pixel = [](float t) { return !(t > 0.0f) ? 0 : t >= 255.0f ? 255 : (int)t; };
    const size_t o = (size_t)b * N + n;
    zproj[o] = X[2];
    xs[o] = pixel(tx);
    ys[o] = pixel(ty);
    flag[o] = (tx < 0.0f || tx > 255.0f || ty < 0.0f || ty > 255.0f) ? 4.0f : 0.0f;
}
__global__ void k_zb_reset(int32_t *winner, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) winner[i] = -1;
}
__global__ void k_zb_test(const int64_t *order, const int32_t *ys, const int32_t *xs, int B, int N, int H, int W, int32_t *winner,
                          int32_t *status)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * N) return;
    const int b = (int)(i / N), n = (int)(i - (size_t)b * N);
    size_t src = i;
    if (order) {
        const int64_t p = order[i];
        if (p < 0 || p >= N) { if (status) atomicOr(status, PS_STATUS_BAD_PIXEL); return; }
        src = (size_t)b * N + (size_t)p;
    }
    const int y = ys[src], x = xs[src];
    if (y < 0 || y >= H || x < 0 || x >= W) { if (status) atomicOr(status, PS_STATUS_BAD_PIXEL); return; }
    atomicMax(&winner[((size_t)b * H + y) * W + x], n);
}
__global__ void k_zb_write(const int64_t *order, const int32_t *ys, const int32_t *xs, const float *v0, const float *v1,
                           const float *grid, const float *flag, int B, int N, int H, int W, const int32_t *winner, float *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * N) return;
    const int b = (int)(i / N), n = (int)(i - (size_t)b * N);
    size_t src = i;
    int64_t p = n;
    if (order) {
        p = order[i];
        if (p < 0 || p >= N) return;
        src = (size_t)b * N + (size_t)p;
    }
    const int y = ys[src], x = xs[src];
    if (y < 0 || y >= H || x < 0 || x >= W) return;
    if (winner[((size_t)b * H + y) * W + x] != n) return;
    float a0, a1;
    if (order) { a0 = grid[p] + flag[i]; a1 = -grid[N + p] + flag[i]; }   // (flag by original position n, values by sorted source p)
    else { a0 = v0[i]; a1 = v1[i]; }
    out[(((size_t)b * 2 + 0) * H + y) * W + x] = a0;
    out[(((size_t)b * 2 + 1) * H + y) * W + x] = a1;
}
}  // namespace

int ps_zbuffer_project_f32(const float *depth, const float *grid, const float *K, const float *Kinv, const float *RT1inv,
                           const float *RT2, int B, int W, float *zproj, int32_t *ys, int32_t *xs, float *flag, void *stream)
{
    PS_REQUIRE(depth && grid && K && Kinv && RT1inv && RT2 && zproj && ys && xs && flag, "zbuffer_project: null pointer");
    PS_REQUIRE(B > 0 && W > 1, "zbuffer_project: B > 0 and W > 1 required");
    const int N = W * W;
    hipLaunchKernelGGL(k_zb_project, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, depth, grid, K, Kinv, RT1inv, RT2,
                       N, zproj, ys, xs, flag);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

static int zb_scatter(const int64_t *order, const int32_t *ys, const int32_t *xs, const float *v0, const float *v1, const float *grid,
                      const float *flag, int B, int N, int H, int W, float *out, int32_t *winner, int32_t *status, hipStream_t st)
{
    const size_t np = (size_t)B * H * W, nn = (size_t)B * N;
    hipLaunchKernelGGL(k_zb_reset, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, winner, np);
    hipLaunchKernelGGL(k_zb_test, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, order, ys, xs, B, N, H, W, winner, status);
    hipLaunchKernelGGL(k_zb_write, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, order, ys, xs, v0, v1, grid, flag, B, N, H, W,
                       winner, out);
    PS_LAUNCH_CHECK();
    return PS_OK;
}

int ps_zbuffer_scatter_f32(const int32_t *ys, const int32_t *xs, const float *v0, const float *v1, int B, int N, int H, int W,
                           float *out, int32_t *winner, int32_t *status, void *stream)
{
    PS_REQUIRE(ys && xs && v0 && v1 && out && winner, "zbuffer_scatter: null pointer");
    PS_REQUIRE(B > 0 && N > 0 && H > 0 && W > 0, "zbuffer_scatter: bad sizes");
    return zb_scatter(nullptr, ys, xs, v0, v1, nullptr, nullptr, B, N, H, W, out, winner, status, (hipStream_t)stream);
}

int ps_zbuffer_scatter_sorted_f32(const int64_t *order, const int32_t *ys, const int32_t *xs, const float *grid, const float *flag,
                                  int B, int N, int H, int W, float *out, int32_t *winner, int32_t *status, void *stream)
{
    PS_REQUIRE(order && ys && xs && grid && flag && out && winner, "zbuffer_scatter_sorted: null pointer");
    PS_REQUIRE(B > 0 && N > 0 && H > 0 && W > 0, "zbuffer_scatter_sorted: bad sizes");
    return zb_scatter(order, ys, xs, nullptr, nullptr, grid, flag, B, N, H, W, out, winner, status, (hipStream_t)stream);
}

int ps_read_status(int32_t *status, void *stream)
{
    PS_REQUIRE(status, "read_status: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int32_t v = 0;
    PS_HIP_CHECK(hipMemcpyAsync(&v, status, sizeof(v), hipMemcpyDeviceToHost, st));
    PS_HIP_CHECK(hipStreamSynchronize(st));
    if (!v) return PS_OK;
    PS_HIP_CHECK(hipMemsetAsync(status, 0, sizeof(v), st));
    return ps::fail(PS_ERR_STATE, "device status 0x%x:%s%s", (unsigned)v,
                    (v & PS_STATUS_BAD_ORDER) ? " a generation order that is no permutation of the grid (ps_order_masks_f32);" : "",
                    (v & PS_STATUS_BAD_PIXEL) ? " points outside the image or the sort order dropped by the z-buffer scatter "
                                                "(ps_zbuffer_scatter_f32: the reference clamps before it scatters);" : "");
}

}  // extern "C"
