// kernels_tile.h -- the tile kernel: everything between the tile lists and the finished pixels.
//
// Design (MI355X-first, not a port of the reference's per-face loops): the frame is cut into
// 16x16-pixel tiles; one 256-thread workgroup owns one tile from the first triangle to the uint8
// pixels.  The tile's z-buffer and winner map live in LDS, each pixel's stencil count in a
// register of the thread that owns the pixel, so the frame's z / stencil / float colour -- the
// buffers the reference's three loops hammer (obj/core.py:588-591) -- never exist in HBM unless a
// caller asks to read them back.  Measured on MI355X (tools/micro/atomic_bench.hip): scattered
// 64-bit global atomics run at 24 G/s, LDS atomics at 1 900 G/s.
//
// Phases of a tile (k_tile):
//   1. big (triangle, tile) pairs -- a floor triangle -- one PIXEL per thread, the records staged
//      64 at a time in LDS and read as broadcasts (obj/triangular.py:72-118);
//   2. small pairs -- a dense mesh's triangles cover a handful of samples -- four lanes per
//      TRIANGLE (two where the tile lists more than a round of those): they share out the few samples of
//      the pixel box and do an LDS atomicMin on the order-preserving key of z, then (second sweep, from
//      the keys the first left in LDS) an LDS atomicMax of the face index where its z is the tile's final z.  The reference's sequential rule (a fragment writes when
//      zbuf >= z, so the last face in order wins ties) is reproduced order-free: smallest z, and
//      among equal z the largest face index;
//   3. the tile's shadow quads against the final z, QUAD_BATCH records staged in LDS per round,
//      one pixel per thread: stencil +-1 in a register (obj/triangular.py:335-368);
//   4. deferred shading of the winner (kernels_shade.h), finalise, uint8 store.
//
//   k_reduce_tile_stats   sums the per-tile fragment counts (only when statistics are asked for)
#pragma once

#include "kernels_bin.h"
#include "kernels_shade.h"

namespace mr {

// Order-preserving key of a non-NaN double: unsigned comparison of keys == comparison of values.
__device__ __forceinline__ unsigned long long z_key(double z)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(z);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double z_unkey(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// Coverage and depth of a sample as the tile kernel's pair loops evaluate them.  The reference's (N,K)@(K,) products
// round in one of two orders (rows_dot2 / rows_dot3 in rast_math.h: a dot for one row, a gemv for more), chosen per face
// by TF_SINGLE_BOX and TF_SINGLE_Z.  Evaluating both and selecting cost four float64 operations and four selects per
// sample on top of the four that count, and two and two more per depth; here the ORDER is folded into the operands,
// once per (lane, triangle):  gemv2(a0, a1, b0, b1) = chain2(a1, a0, b1, b0) and gemv3(a0, a1, a2, ...) =
// chain3(a1, a0, a2, ...), so one chain serves both with the first two operand pairs swapped.  Same operations on the
// same values as tri_bary / rows_dot3: bit-identical.  MAYBE_SINGLE == false: the caller knows neither flag is set.
template <bool MAYBE_SINGLE>
struct PairMath {
    double a0, a1, p0, p1, q0, q1, k0, k1, k2;
    float d00, d01, d11, inv_den;
    bool sbox, sz;
    __device__ __forceinline__ explicit PairMath(const TriRec &t)
    {
        sbox = MAYBE_SINGLE && (t.flags & TF_SINGLE_BOX) != 0;
        sz = MAYBE_SINGLE && (t.flags & TF_SINGLE_Z) != 0;
        a0 = sbox ? t.ax : t.ay;   a1 = sbox ? t.ay : t.ax;
        p0 = sbox ? t.v0x : t.v0y; p1 = sbox ? t.v0y : t.v0x;
        q0 = sbox ? t.v1x : t.v1y; q1 = sbox ? t.v1y : t.v1x;
        k0 = sz ? t.zl0 : t.zl1;   k1 = sz ? t.zl1 : t.zl0;   k2 = t.zl2;
        d00 = t.d00; d01 = t.d01; d11 = t.d11; inv_den = t.inv_den;
    }
    // tri_bary (obj/transformation.py:18-31) at the sample (x, y), given widened to double
    __device__ __forceinline__ void bary(double x, double y, float &u, float &v, float &w) const
    {
        const double r0 = (sbox ? x : y) - a0, r1 = (sbox ? y : x) - a1;
        const float d20 = (float)fma(r1, p1, r0 * p0);
        const float d21 = (float)fma(r1, q1, r0 * q0);
        v = (d11 * d20 - d01 * d21) * inv_den;
        w = (d00 * d21 - d01 * d20) * inv_den;
        u = 1.0f - v - w;
    }
    __device__ __forceinline__ void bary(int x, int y, float &u, float &v, float &w) const
    {
        bary((double)(sbox ? x : y), (double)(sbox ? y : x), true, u, v, w);
    }
    // bar @ zlin (obj/triangular.py:97)
    __device__ __forceinline__ double depth(float u, float v, float w) const
    {
        const float m0 = sz ? u : v, m1 = sz ? v : u;
        return fma((double)w, k2, fma((double)m1, k1, (double)m0 * k0));
    }
private:
    __device__ __forceinline__ void bary(double c0, double c1, bool, float &u, float &v, float &w) const   // operands in chain order already
    {
        const double r0 = c0 - a0, r1 = c1 - a1;
        const float d20 = (float)fma(r1, p1, r0 * p0);
        const float d21 = (float)fma(r1, q1, r0 * q0);
        v = (d11 * d20 - d01 * d21) * inv_den;
        w = (d00 * d21 - d01 * d20) * inv_den;
        u = 1.0f - v - w;
    }
};

// One small (triangle, tile) pair shared by SMALL_LANES neighbouring lanes: the samples of the
// pixel box that lie in the tile are dealt to them round-robin (a tile of a dense mesh lists
// 50-200 such pairs of 1-24 samples each; one lane per pair left three of the four wavefronts
// idle behind long serial walks).  SWEEP 0: atomicMin/Max of z into the tile's LDS z-buffer.
// SWEEP 1: the same samples again; where this face's z is the final z, atomicMax of the face
// index.  Pairs that need the per-fragment clip test never come here (pair_class).
constexpr int SMALL_LANES = 4;

// What a lane's first sweep leaves for its second (see the winners' sweep in k_tile): the z keys and pixels
// of its first SWEEP_CACHE_K samples that had a depth, and how many samples it walked in all.
constexpr int SWEEP_CACHE_K = 2, SWEEP_CACHE_ROUNDS = 2;
struct SweepCache {
    unsigned long long key[SWEEP_CACHE_K];
    uint32_t meta;                    // pixel of sample s in bits 8s..8s+7, "sample s is cached" in bit 16+s, samples walked from bit 20
};

// the frame's limits a tile clamps its pixel boxes to (values, not the frame constants: see kernargs())
struct TileBounds { int width, band_y0, band_y1; };

// SWEEP 0 with `cache`: fills it.  SWEEP 1 with `first` > 0: skips the lane's first `first` samples (the cache
// answered for them).
template <int SWEEP>
__device__ __forceinline__ void small_pair(const TileBounds &tb, const TriRec &t, int gx, int gy, bool rh,
                                           unsigned long long *s_key, int *s_win, int sub, int lanes, unsigned int &frags,
                                           SweepCache *cache = nullptr, int first = 0)
{
    const int x0 = max((int)t.x0, gx), x1 = min((int)t.x1, min(gx + TILE_W, tb.width));
    const int y0 = max(max((int)t.y0, gy), tb.band_y0), y1 = min(min((int)t.y1, gy + TILE_H), tb.band_y1);
    const PairMath<true> pm(t);
    // Model.depth_test == False (obj/triangular.py:117): the face's fragments are tested, never written to z.
    // The pixel then shows the LAST face in order among those that pass against the final z (see k_tile).
    const bool nodepth = ((t.flags >> 8) & FF_NO_DEPTH) != 0;
    const int bw = x1 - x0;
    if (cache) { cache->meta = 0; }
    if (bw <= 0) return;
    int px = x0 + sub, py = y0;
    while (px >= x1) { px -= bw; ++py; }
    int walked = 0;
    while (py < y1) {
        if (SWEEP == 0 || walked >= first) {
            float u, v, w;
            pm.bary(px, py, u, v, w);
            bool ok = u >= 0 && v >= 0 && w >= 0;
            if (ok && SWEEP == 0) ++frags;
            if (ok && (SWEEP == 1 || !nodepth)) {
                const double z = pm.depth(u, v, w);
                if (z == z) {                            // a NaN depth never passes the reference's test
                    const int p = (py - gy) * TILE_W + (px - gx);
                    const unsigned long long k = z_key(z);
                    if (SWEEP == 0) {
                        if (rh) atomicMin(&s_key[p], k); else atomicMax(&s_key[p], k);
                        if (cache) {
#pragma unroll
                            for (int c = 0; c < SWEEP_CACHE_K; ++c)
                                if (walked == c) { cache->key[c] = k; cache->meta |= ((uint32_t)p << (8 * c)) | (1u << (16 + c)); }
                        }
                    } else if (nodepth ? (rh ? k <= s_key[p] : k >= s_key[p]) : s_key[p] == k) {
                        atomicMax(&s_win[p], t.face);
                    }
                }
            }
        }
        ++walked;
        px += lanes;
        while (px >= x1) { px -= bw; ++py; }
    }
    if (cache) cache->meta |= (uint32_t)walked << 20;
}

// Row of the device's output buffer that screen row py (in local tile row l) lands in.  A band of
// rows is stored top row first (the flip of obj/core.py:640).  In the striped layout (multi-GPU,
// interleaved tile rows) every device's buffer holds out_tile_rows blocks of TILE_H rows, its
// highest tile row first, rows inside a block top-down; the host un-permutes after the all-gather.
__device__ __forceinline__ int out_row(const FrameConst &fc, int py, int l)
{
    if (fc.out_tile_rows > 0)
        return (fc.out_tile_rows - 1 - l) * TILE_H + (tile_row_frame(fc, l) * TILE_H + TILE_H - 1 - py);
    return fc.band_y1 - 1 - py;
}

constexpr int QUAD_STAGE_U4 = 12;     // uint4 pieces staged per quad: 64-byte header + 4 edges
static_assert(offsetof(QuadRec, e) == 64 && sizeof(QuadEdge) == 32, "QuadRec layout");
static_assert(QUAD_BATCH == WAVE, "lane j of every wavefront classifies quad j of the batch");

struct QuadHead {                     // the first 64 bytes of a QuadRec, as staged
    double nx, ny, nz, d;
    int16_t x0, x1, y0, y1;
    int32_t n, is_front, edge;
    uint32_t pad[3];
};
static_assert(sizeof(QuadHead) == 64, "QuadHead mirrors QuadRec's header");

constexpr int MAT_LDS = 8;            // scenes with up to this many materials keep them in LDS
static_assert(sizeof(Material) % 8 == 0 && MAT_LDS * sizeof(Material) / 8 <= TILE_PX, "material staging");

// Heaviest tiles first.  A tile under the mesh AND its shadow volume takes 10-80x the time of a floor
// tile.  In plain row-major order such tiles start in the middle of the launch and finish 40 us after
// everything else, and even with only the heaviest moved to the front the launch ended on the 10-20 us
// tiles that happened to start last.  Frames of a sequence resemble each other, so every tile leaves the
// class of its estimated cost (one byte, a plain store) for the slot's NEXT frame, whose k_bin_work sorts
// the tiles by class (order_tiles_block in kernels_bin.h: a counting sort in one workgroup, beside the
// binning) and whose tile workgroup b renders entry b of that order: longest-processing-time first.  The
// order is a permutation of the tiles by construction, and a grid with any tile of unknown class (a first
// frame, a new tile grid) is rendered in row-major order: a stale history costs time, never correctness.
// (A loop "list entry, then own tile" inside one workgroup doubled the kernel's register budget -- the
// compiler hoists the frame constants' register copies out of it; per-class lists appended to with atomics
// held every workgroup for the 2 us of a device-scope atomic's round trip: 82 -> 119 us.)
// cost thresholds of classes 1..7 (tile_cost units, ~0.1 us); the rest is class 8
#ifndef MR_CLASS_SCALE
#define MR_CLASS_SCALE 100            // per cent: tools/ab.sh "-DMR_CLASS_SCALE=70" ... moves all seven limits at once
#endif
__device__ __forceinline__ int tile_class(uint32_t cost)
{
    const uint32_t c = cost * 100u / (uint32_t)MR_CLASS_SCALE;
    return c >= 500u ? 1 : c >= 350u ? 2 : c >= 250u ? 3 : c >= 170u ? 4 : c >= 110u ? 5 : c >= 70u ? 6 : c >= 40u ? 7 : 8;
}
static_assert(ORDER_CLASSES == 8, "tile_class");
// On a device that owns few tiles (a rank of a multi-GPU split: all its tiles are resident at once and
// the launch lasts as long as its slowest tile) the heaviest class is not only started first: those
// tiles' shadow quads -- walked one after the other, up to a hundred and more where a mesh's
// outline and its shadow volume overlap -- are SHARED OUT over HEAVY_SPLIT workgroups.  Each of them
// rasterises the tile (same lists, same order-free arithmetic: the same z and winners), counts its share
// of the quads, and leaves its stencil counts in a scratch slot; the workgroup that arrives last (a
// counter, no waiting) adds the others' and shades.  Only frames rendered for the frame's sake are split
// (no MR_FRAME_COUNTERS), and only in the k_tile<true> instantiation: on a device that owns the whole
// frame the launch is bound by the tiles' total work, which the repeated rasterisation only adds to
// (measured on MI355X in round 2, c4: 86 -> 90 us whole frame; a rank of 8: 56 -> 38 us).  Round 3: with the
// heaviest tiles started first a whole frame's launch lasts exactly as long as its single heaviest tile (c4:
// 63.7 of 64.4 us, 46 of them its 111 shadow quads), so a lone whole frame now shares out its few heaviest tiles
// too -- a dozen on c4, chosen by a higher threshold (TileArgs::split_cost / split_quads) -- and nothing else.
#ifndef MR_HEAVY0_MAX
#define MR_HEAVY0_MAX 128
#endif
constexpr int HEAVY_SPLIT = 4, HEAVY0_MAX = MR_HEAVY0_MAX;
#ifndef MR_TILE_WAVES
#define MR_TILE_WAVES 5
#endif
constexpr int K_TILE_WAVES = MR_TILE_WAVES;              // wavefronts per SIMD the tile kernel's register budget allows
constexpr int SPLIT_FRONT = HEAVY_SPLIT * HEAVY0_MAX;     // extra workgroups of a k_tile<true> launch

// estimated cost of a tile in ~0.1 us from its list lengths: the least-squares fit of the measured tile times of
// c4 on MI355X with six wavefronts per SIMD, 20.1 + 1.63 small + 29.8 big + 3.19 quads (tools/diag_tiles.py), rounded
__device__ __forceinline__ uint32_t tile_cost(uint32_t n_small, uint32_t n_big, uint32_t n_quad)
{
    return 20u + 2u * n_small + 30u * n_big + 3u * n_quad;
}

struct TileArgs {
    const TriClip *clips;
    const QuadRec *quads;
    uint32_t *bin_count;          // cursors = list lengths; zeroed again at the end of the tile
    const uint32_t *items[BIN_CLASSES];
    uint32_t cap[BIN_CLASSES];
    const uint8_t *tap_mask;      // [tiles of the whole frame] or null: only the tiles marked here write the taps below (the overlay's tiles)
    double *zbuf;                 // optional taps (MR_FRAME_KEEP_BUFFERS): may be null
    int32_t *winner, *stencil;
    uint32_t *tile_stats;
    Counters *ctr, *next_ctr;     // this frame's counters; the next frame's (cleared here)
    Sticky *sticky;               // overflow verdicts of the slot's earlier frames (see Sticky)
    int32_t *split_sten;          // [HEAVY0_MAX][HEAVY_SPLIT][TILE_PX] stencil counts of a split tile's parts
    uint32_t *split_arrive;       // [HEAVY0_MAX] parts that have left theirs (zero between frames)
    const uint32_t *order;        // ORDER_HEAD words, then the tiles in the order to render them (k_bin_work); null: row-major
    uint8_t *tile_class;          // [n_tiles] what this frame leaves for the next: 1 + the tile's cost class
    uint32_t split_cost, split_quads;   // k_tile<true>: a tile this costly, with this many shadow quads, is shared out next frame ...
    uint32_t split_max;                 // ... if no more than this many are (<= HEAVY0_MAX)
};

struct TileKernArgs { FrameConst fc; TileArgs ta; ShadeArgs sh; };

// One workgroup per tile, one pixel per thread.  Tiles are dealt to workgroups in plain
// row-major order, i.e. round-robin over the XCDs: heavy tiles cluster on the screen, and an
// XCD-contiguous mapping (tried first) left six of the eight XCDs idle behind the two that
// owned the mesh and its shadow.
template <bool SPLIT>      // SPLIT: the heaviest tiles' shadow quads are shared out over HEAVY_SPLIT workgroups (see HEAVY_SPLIT)
__global__ void __launch_bounds__(TILE_PX, K_TILE_WAVES)
k_tile(const TileKernArgs)            // read through kernargs<TileKernArgs>(), phase by phase
{
    __shared__ unsigned long long s_key[TILE_PX];
    __shared__ int s_win[TILE_PX];
    __shared__ unsigned int s_cnt[TILE_STATS];
    __shared__ uint4 s_quad[QUAD_BATCH * QUAD_STAGE_U4];
    __shared__ uint32_t s_id[QUAD_BATCH];
    __shared__ float s_gamma[GAMMA_LUT_SIZE];
    __shared__ unsigned long long s_mat[MAT_LDS * sizeof(Material) / 8];

    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int lp = tid;                                   // pixel of this thread inside the tile

    // ---- 0. which tile, and what it lists
    int tile = 0, part = 0, n_parts = 1, entry = 0, n_tiles, ltr, gx, gy;
    bool rh, counters, live, mat_lds, taps;
    uint32_t n_small_raw, n_big_raw, n_quad_raw, n_small, n_big, n_quad, cost;
    unsigned long long t_start;
    {
        const TileKernArgs &ka = kernargs<TileKernArgs>();
        const FrameConst &fc = ka.fc;
        const TileArgs &ta = ka.ta;
        const ShadeArgs &sh = ka.sh;
        n_tiles = fc.tiles_x * fc.tiles_y;
        rh = fc.system == 1;
        counters = (fc.flags & MR_FRAME_COUNTERS) != 0;

        if (blockIdx.x == 0) {
            // The next frame's counter block is the previous frame's (double-buffered by parity; that frame is
            // complete: same stream).  Its overflow verdicts move to the slot's sticky record before the block is
            // cleared; nobody touches either before this kernel ends.
            if (tid == 0) {
                const Counters &old = *ta.next_ctr;
                uint32_t longest = 0;                     // the fullest stretch of the work list decides what the list needs
                for (int s = 0; s < WORK_SHARDS; ++s) longest = max(longest, old.work[s].n);
                if (old.overflow | old.n_quads | longest) {
                    Sticky &st = *ta.sticky;
                    st.overflow |= old.overflow;
#pragma unroll
                    for (int c = 0; c < BIN_CLASSES; ++c) st.max_list[c] = max(st.max_list[c], old.max_list[c]);
                    st.n_work = max(st.n_work, longest * (uint32_t)WORK_SHARDS);
                    st.n_quads = max(st.n_quads, old.n_quads);
                    st.n_quads_drawn = max(st.n_quads_drawn, old.n_quads_drawn);
                }
            }
            __syncthreads();
            uint32_t *words = reinterpret_cast<uint32_t *>(ta.next_ctr);
            for (int i = tid; i < (int)(sizeof(Counters) / 4); i += TILE_PX) words[i] = 0u;
        }

        // heaviest tiles first: entry blockIdx.x of the order k_setup left (see tile_class)
        uint32_t idx = blockIdx.x;
        if (SPLIT) {
            // the class whose quads are shared out -- when it is a handful of tiles, i.e. when the launch would end on them
            // alone; a frame with hundreds of them keeps the device full to the end, and rasterising each four times
            // only adds to that (c3: 209 qualify, k_tile 76 -> 84 us with the split; c5: 143, 253 -> 270; c4: some 40, 66 -> 63)
            const uint32_t n_heavy = ta.order ? ta.order[0] : 0u;
            const uint32_t n0 = n_heavy <= ta.split_max ? n_heavy : 0u;
            if (idx < (uint32_t)SPLIT_FRONT) {
                entry = (int)idx / HEAVY_SPLIT;
                part = (int)idx % HEAVY_SPLIT;
                n_parts = counters ? 1 : HEAVY_SPLIT;
                if (entry >= (int)n0 || part >= n_parts) return;
                idx = (uint32_t)entry;
            } else {
                idx = idx - (uint32_t)SPLIT_FRONT + n0;
            }
        }
        if (idx >= (uint32_t)n_tiles) return;
        tile = ta.order ? (int)ta.order[ORDER_HEAD + idx] : (int)idx;
        ltr = tile / fc.tiles_x;                          // local tile row
        gx = (tile % fc.tiles_x) * TILE_W; gy = tile_row_frame(fc, ltr) * TILE_H;
        const int px = gx + (lp & (TILE_W - 1)), py = gy + lp / TILE_W;
        live = px < fc.width && py >= fc.band_y0 && py < fc.band_y1;
        t_start = __builtin_amdgcn_s_memrealtime();

        // list lengths (a list that ran over its capacity is truncated; the host grows it and re-renders)
        n_small_raw = ta.bin_count[tile]; n_big_raw = ta.bin_count[n_tiles + tile]; n_quad_raw = ta.bin_count[2 * n_tiles + tile];
        n_small = min(n_small_raw, ta.cap[0]); n_big = min(n_big_raw, ta.cap[1]); n_quad = min(n_quad_raw, ta.cap[2]);
        // the heaviest tiles are the frame's critical path: their wavefronts go first wherever they
        // compete with a lighter tile's for a SIMD's issue slots
        cost = tile_cost(n_small_raw, n_big_raw, n_quad_raw);
        if (cost >= 500u) __builtin_amdgcn_s_setprio(3);
        else if (cost >= 250u) __builtin_amdgcn_s_setprio(1);

        // Nothing listed for this tile (a third of a typical frame): its pixels show the background, which
        // the host has finalised already, or the skybox (no lists to walk, no barriers but the gamma table's).
        // Shadow quads over it only matter to the counters.
        const bool sky_tile = (fc.flags & MR_FRAME_SKYBOX) && sh.sky;
        taps = (sh.frame || ta.zbuf) && (!ta.tap_mask || ta.tap_mask[tile_row_frame(fc, ltr) * fc.tiles_x + tile % fc.tiles_x]);
        if (n_small_raw == 0 && n_big_raw == 0 && (n_quad_raw == 0 || !counters) && (sky_tile || (fc.background_u8 >> 24)) &&
            !taps) {
            if (part != 0) return;                        // (a tile that was heavy a frame ago: one part will do)
            uint8_t *o = sh.out + ((size_t)out_row(fc, py, ltr) * fc.width + px) * 3;
            if (sky_tile) {
                s_gamma[tid] = sh.gamma_lut[tid];
                if (tid == 0) s_gamma[GAMMA_LUT_SIZE - 1] = sh.gamma_lut[GAMMA_LUT_SIZE - 1];
                float rgb[3] = { 0.f, 0.f, 0.f };
                if (live) sky_color(fc, sh.sky, px, py, rgb);
                __syncthreads();
                if (live) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) o[j] = gamma_u8(rgb[j], s_gamma);
                }
            } else if (live) {
                o[0] = (uint8_t)fc.background_u8; o[1] = (uint8_t)(fc.background_u8 >> 8); o[2] = (uint8_t)(fc.background_u8 >> 16);
            }
            if (tid == 0) {
                uint32_t *rec = ta.tile_stats + (size_t)tile * TILE_REC;
                for (int k = 0; k < TILE_REC; ++k) rec[k] = 0;
                rec[7] = n_quad_raw;
                rec[8] = rec[10] = rec[11] = (uint32_t)t_start;
                rec[9] = (uint32_t)__builtin_amdgcn_s_memrealtime();
                ta.bin_count[2 * n_tiles + tile] = 0;
                ta.tile_class[tile] = (uint8_t)ORDER_CLASSES;
            }
            return;
        }
        if (tid < TILE_STATS) s_cnt[tid] = 0;
        s_gamma[tid] = sh.gamma_lut[tid];
        if (tid == 0) s_gamma[GAMMA_LUT_SIZE - 1] = sh.gamma_lut[GAMMA_LUT_SIZE - 1];
        mat_lds = fc.n_materials <= MAT_LDS;
        if (mat_lds && tid < fc.n_materials * (int)(sizeof(Material) / 8))
            s_mat[tid] = reinterpret_cast<const unsigned long long *>(sh.materials)[tid];
    }
    // The pixel's coordinates are recomputed from the thread index where a phase needs them (two instructions),
    // from a copy of the index the compiler cannot connect to the other phases': kept in registers from the first
    // phase to the last, px, py and their float64 images held six of the 80 registers a wavefront may use when
    // six share a SIMD.
    auto my_pixel = [&](int &px, int &py) {
        int l = lp;
        asm volatile("" : "+v"(l));
        px = gx + (l & (TILE_W - 1)); py = gy + l / TILE_W;
    };

    // ---- 1. big pairs, one pixel per thread (obj/triangular.py:72-118)
    double zbest = rh ? INFINITY : -INFINITY;
    int best = -1;
    {
        const TileKernArgs &ka = kernargs<TileKernArgs>();
        const FrameConst &fc = ka.fc;
        const TileArgs &ta = ka.ta;
        const ShadeArgs &sh = ka.sh;
        // (what the loops below read of the arguments is fetched here, once: a load through the opaque pointer is
        // not hoisted out of a loop by the compiler when it sits under a condition)
        const TriRec *__restrict__ tris = sh.tris;
        const TriClip *__restrict__ clips = ta.clips;
        const uint32_t *__restrict__ small_items = ta.items[0] + (size_t)tile * ta.cap[0];
        const uint32_t *__restrict__ big_items = ta.items[1] + (size_t)tile * ta.cap[1];
        const bool same_clip = fc.same_clip != 0, has_no_depth = fc.has_no_depth != 0;
        const TileBounds tb = { fc.width, fc.band_y0, fc.band_y1 };
        int px, py;
        my_pixel(px, py);
        const double dpx = (double)px, dpy = (double)py;
        unsigned int frags = 0;
        // The records of up to 64 pairs are copied to LDS once per workgroup, 16 bytes per lane and step
        // (the staging area of the shadow quads, not in use yet), and every lane then reads the pair it is
        // testing at the same LDS address (a broadcast read).  Each wavefront fetching the records itself,
        // one per lane, and handing them round with v_readlane cost ~45 vector instructions per pair and
        // wavefront on top of the arithmetic; scalar loads cost a dependent memory round trip per pair.
        constexpr int TRI_U4 = (int)(sizeof(TriRec) / 16);
        static_assert(WAVE * TRI_U4 <= QUAD_BATCH * QUAD_STAGE_U4, "big-pair records are staged in the quad area");
        // LATE == false: the faces that write z.  LATE == true (only in scenes that have a model with
        // depth_test == False, after the tile's final z is known): those that do not -- they own the pixel
        // when they pass against the final z and come later in face order than the current owner.
        auto big_pairs = [&](const bool late) {
            for (uint32_t base = 0; base < n_big; base += WAVE) {
                const int n = (int)min((uint32_t)WAVE, n_big - base);
                if (base || late) __syncthreads();        // the staging area is free
                for (int i = tid; i < n * TRI_U4; i += TILE_PX) {
                    const int q = i / TRI_U4, piece = i - q * TRI_U4;
                    s_quad[i] = reinterpret_cast<const uint4 *>(tris + big_items[base + q])[piece];
                }
                __syncthreads();
                for (int j = 0; j < n; ++j) {
                    const TriRec &t = *reinterpret_cast<const TriRec *>(s_quad + j * TRI_U4);
                    const uint32_t flags = t.flags;
                    const bool nodepth = ((flags >> 8) & FF_NO_DEPTH) != 0;
                    if (late && !nodepth) continue;
                    // the pair is the same for every lane, so are its flags: the usual face (neither of the two
                    // summation-order flags) takes a copy of the arithmetic without their selects (see PairMath)
                    auto pair = [&](auto maybe_single) {
                        const PairMath<decltype(maybe_single)::value> pm(t);
                        const int f = t.face;
                        bool in = live && px >= t.x0 && px < t.x1 && py >= t.y0 && py < t.y1;
                        float u, v, w;
                        pm.bary(dpx, dpy, u, v, w);
                        in = in && u >= 0 && v >= 0 && w >= 0;
                        const unsigned long long m = __ballot(in);
                        if (!m) return;
                        if (!late) frags += (unsigned int)__popcll(m);
                        if (nodepth && !late) return;
                        if (flags & TF_CLIP) {
                            if (in) {
                                const TriClip &c = clips[f];
                                double p[3];
                                persp_bary(t.dp, u, v, w, (flags & TF_SINGLE_BOX) != 0, p);
                                in = inside_clip(p, c.clip) && (same_clip || inside_clip(p, c.clipd));
                            }
                        }
                        const double z = pm.depth(u, v, w);
                        if (late) {
                            if (in && (rh ? z <= zbest : z >= zbest) && f > best) best = f;
                        } else {
                            // sequential rule "zbuf >= z writes" == smallest z, and among equal z the latest face
                            const bool closer = rh ? (z < zbest) : (z > zbest);
                            if (in && (closer || (z == zbest && f > best))) { zbest = z; best = f; }
                        }
                    };
                    if (flags & (TF_SINGLE_BOX | TF_SINGLE_Z)) pair(std::true_type{});
                    else pair(std::false_type{});
                }
            }
        };
        big_pairs(false);
        if (n_small) { s_key[lp] = z_key(zbest); s_win[lp] = -1; }     // the small pairs' LDS z-buffer starts from the big pairs' z
        __syncthreads();                                  // s_cnt is zeroed, the LDS tables are loaded
        if (lane == 0 && frags) atomicAdd(&s_cnt[0], frags);
        if (n_small) {
            // ---- 2. small pairs, SMALL_LANES lanes per triangle, z into the LDS z-buffer
            // The second sweep (who owns the final z?) needs every sample's z key again.  For the first
            // SWEEP_CACHE_ROUNDS rounds a lane leaves the keys of its first SWEEP_CACHE_K samples in LDS (the staging
            // area of the shadow quads, not in use yet: 12 KB, exactly), with the face; the second sweep then
            // compares those without the record, the barycentrics or the depth -- a mesh triangle of a few pixels is
            // all cache.  Lanes with more samples, later rounds and faces that do not write z walk again.
            unsigned long long (*c_key)[SWEEP_CACHE_K][TILE_PX] =
                reinterpret_cast<unsigned long long (*)[SWEEP_CACHE_K][TILE_PX]>(s_quad);
            uint32_t (*c_meta)[TILE_PX] = reinterpret_cast<uint32_t (*)[TILE_PX]>(
                reinterpret_cast<unsigned long long *>(s_quad) + SWEEP_CACHE_ROUNDS * SWEEP_CACHE_K * TILE_PX);
            uint32_t (*c_face)[TILE_PX] = c_meta + SWEEP_CACHE_ROUNDS;
            static_assert(SWEEP_CACHE_ROUNDS * TILE_PX * (SWEEP_CACHE_K * 8 + 8) <= (int)sizeof(s_quad), "sweep cache fits the quad area");
            unsigned int sfrags = 0;
            int round = 0;
            // SMALL_LANES lanes per pair; two where the list is longer than one round of those (a round is two
            // dependent memory round trips whatever it holds: 65 pairs in two rounds cost twice what 64 do), eight
            // where it is at most half a round (the lanes would idle otherwise; c2's larger triangles: -2 %)
            const int spl = n_small > (uint32_t)(TILE_PX / SMALL_LANES) ? SMALL_LANES / 2
                          : n_small > (uint32_t)(TILE_PX / (2 * SMALL_LANES)) ? SMALL_LANES : 2 * SMALL_LANES;
            const uint32_t per_round = (uint32_t)(TILE_PX / spl), my_pair = (uint32_t)(tid / spl);
            const int sub = tid % spl;
            for (uint32_t i = my_pair; i < n_small; i += per_round, ++round) {
                const TriRec t = tris[small_items[i]];
                if (round < SWEEP_CACHE_ROUNDS) {
                    SweepCache sc;
                    small_pair<0>(tb, t, gx, gy, rh, s_key, s_win, sub, spl, sfrags, &sc);
                    const bool nodepth = ((t.flags >> 8) & FF_NO_DEPTH) != 0;
#pragma unroll
                    for (int c = 0; c < SWEEP_CACHE_K; ++c) c_key[round][c][tid] = sc.key[c];
                    c_meta[round][tid] = nodepth ? (sc.meta & 0xfff00000u) : sc.meta;      // nothing cached for a face that writes no z
                    c_face[round][tid] = (uint32_t)t.face | (nodepth ? 0x80000000u : 0u);
                } else {
                    small_pair<0>(tb, t, gx, gy, rh, s_key, s_win, sub, spl, sfrags);
                }
            }
            if (sfrags) atomicAdd(&s_cnt[0], sfrags);
            __syncthreads();

            // winners: a big pair keeps its face where its z survived (an atomic like the small pairs' sweep: the
            // largest face index among those at the final z, in any order)
            const unsigned long long kfinal = s_key[lp];
            if (best >= 0 && kfinal == z_key(zbest)) atomicMax(&s_win[lp], best);
            round = 0;
            for (uint32_t i = my_pair; i < n_small; i += per_round, ++round) {
                int first = 0;
                if (round < SWEEP_CACHE_ROUNDS) {
                    const uint32_t meta = c_meta[round][tid], cf = c_face[round][tid];
                    const int walked = (int)(meta >> 20);
                    const bool nodepth = (cf >> 31) != 0;
#pragma unroll
                    for (int c = 0; c < SWEEP_CACHE_K; ++c)
                        if ((meta >> (16 + c)) & 1u) {
                            const int p = (int)((meta >> (8 * c)) & 0xffu);
                            if (s_key[p] == c_key[round][c][tid]) atomicMax(&s_win[p], (int)(cf & 0x7fffffffu));
                        }
                    if (!nodepth && walked <= SWEEP_CACHE_K) continue;     // the cache answered for every sample of this lane
                    first = nodepth ? 0 : SWEEP_CACHE_K;
                }
                const TriRec t = tris[small_items[i]];
                small_pair<1>(tb, t, gx, gy, rh, s_key, s_win, sub, spl, sfrags, nullptr, first);
            }
            __syncthreads();
            best = s_win[lp];
            zbest = z_unkey(kfinal);
        }
        if (has_no_depth) big_pairs(true);
    }
    const bool covered = live && best >= 0;
    const unsigned long long t_raster = __builtin_amdgcn_s_memrealtime();

    // ---- 3. shadow quads against the final z: stencil +-1 (obj/triangular.py:335-368).  The
    // batch's records (header and first four edges: 192 bytes each) are copied to LDS once, 16 bytes
    // per lane, and every lane then reads the quad it is testing at the same LDS address (a
    // broadcast read).  Adds commute, so the order of the quads never mattered
    // (obj/triangular.py:365-368); the count stays in this thread's register.
    int sten = 0;
    unsigned int qfrags = 0, qupd = 0;
    // without the counters only the frame is the contract: the stencil matters where a triangle was drawn
    if (n_quad && (counters || __syncthreads_or(covered))) {
        const TileKernArgs &ka = kernargs<TileKernArgs>();
        const FrameConst &fc = ka.fc;
        const TileArgs &ta = ka.ta;
        const uint32_t *__restrict__ quad_items = ta.items[2] + (size_t)tile * ta.cap[2];
        const QuadRec *__restrict__ quads = ta.quads;
        const double f_plus_n = fc.f_plus_n, f_minus_n = fc.f_minus_n, two_nf = fc.two_nf;     // fetched once, see above
        int px, py;
        my_pixel(px, py);
        const double dpx = (double)px, dpy = (double)py;
        // most and least favourable covered z of this wavefront's strip (see the depth verdicts below)
        double zlim = rh ? -INFINITY : INFINITY, zhard = rh ? INFINITY : -INFINITY;
        if (covered) zlim = zhard = zbest;
#pragma unroll
        for (int off = WAVE / 2; off; off >>= 1) {
            const double o = __shfl_xor(zlim, off), h = __shfl_xor(zhard, off);
            zlim = rh ? fmax(zlim, o) : fmin(zlim, o);
            zhard = rh ? fmin(zhard, h) : fmax(zhard, h);
        }
        const uint32_t q_begin = SPLIT ? (uint32_t)((unsigned long long)part * n_quad / n_parts) : 0u,
                       q_end = SPLIT ? (uint32_t)((unsigned long long)(part + 1) * n_quad / n_parts) : n_quad;    // this part's share
        for (uint32_t qbase = q_begin; qbase < q_end; qbase += QUAD_BATCH) {
            const int n = (int)min((uint32_t)QUAD_BATCH, q_end - qbase);
            if (qbase != q_begin) __syncthreads();        // the previous batch has been read
            // Staging also folds two per-quad facts into the copy: a back-facing quad's edge vectors are
            // negated (the rounded cross product changes sign exactly, so "inner side" is "> 0" for every
            // staged quad), and the last 16 bytes of the header, unused here, receive f_plus_n * nz and
            // two_nf * nz of the depth test below.
            for (int i = tid; i < n * QUAD_STAGE_U4; i += TILE_PX) {
                const int q = i / QUAD_STAGE_U4, piece = i - q * QUAD_STAGE_U4;
                const uint32_t id = quad_items[qbase + q];
                const QuadRec *src = quads + id;
                uint4 val = reinterpret_cast<const uint4 *>(src)[piece];
                if (piece == 3) {
                    const double a0 = f_plus_n * src->nz, b0 = two_nf * src->nz;
                    val = make_uint4((uint32_t)__double2loint(a0), (uint32_t)__double2hiint(a0),
                                     (uint32_t)__double2loint(b0), (uint32_t)__double2hiint(b0));
                } else if (piece >= 5 && (piece & 1) && !src->is_front) {
                    val.y ^= 0x80000000u; val.w ^= 0x80000000u;
                }
                s_quad[i] = val;
                if (piece == 0) s_id[q] = id;
            }
            __syncthreads();

            // Lane j classifies quad j against this wavefront's 16x4 pixel strip with the same
            // corner argument as quad_touches_tile: per edge, the rounded cross product is monotone
            // in x and in y, so over the strip it is extreme at a corner.  No corner on the inner
            // side of some edge -> no sample of the strip is inside (skip the quad); all four
            // corners on the inner side of every edge -> every sample is inside (skip the
            // per-pixel edge tests).  Exact, no margins.
            // Per edge too: an edge whose inner side holds all four corners needs no per-pixel test
            // in this strip (bit i of emask clear); typically one edge of a quad crosses a strip.
            bool q_reject = lane >= n;
            int emask = 0;
            if (!q_reject) {
                const QuadHead &h = *reinterpret_cast<const QuadHead *>(s_quad + lane * QUAD_STAGE_U4);
                const QuadEdge *e = reinterpret_cast<const QuadEdge *>(s_quad + lane * QUAD_STAGE_U4 + 4);
                const double xa = (double)gx, xb = (double)(gx + TILE_W - 1);
                const double ya = (double)(gy + (tid / WAVE) * (WAVE / TILE_W)), yb = ya + (double)(WAVE / TILE_W - 1);
                if (h.n > 4) emask |= 16;                   // edges beyond the staged four: always per pixel
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < 3 || h.n > 3) {
                        const double px0 = (xa - e[i].sx) * e[i].ey, px1 = (xb - e[i].sx) * e[i].ey;
                        const double py0 = (ya - e[i].sy) * e[i].ex, py1 = (yb - e[i].sy) * e[i].ex;
                        const double c00 = px0 - py0, c10 = px1 - py0, c01 = px0 - py1, c11 = px1 - py1;
                        const bool any = c00 > 0 || c10 > 0 || c01 > 0 || c11 > 0;
                        const bool all = c00 > 0 && c10 > 0 && c01 > 0 && c11 > 0;
                        q_reject = q_reject || !any;
                        if (!all) emask |= 1 << i;
                    }
                }
            }
            // Depth verdicts for the whole strip.  The quad's plane depth is affine over the screen, so
            // over the strip -t/nz is extreme at a corner, and linearize_z is monotone while its
            // denominator stays positive: with 0 < den_lo <= den <= den_hi over the strip (bounds pushed
            // outwards by a slack that dwarfs the rounding of the per-pixel expression and of the
            // approximate reciprocal used here) the depth two_nf / den lies in [two_nf / den_hi,
            // two_nf / den_lo], and comparisons against it are decided without dividing.
            //   all pass: even the quad's least favourable depth passes against the least favourable
            //             covered z of the strip (uncovered pixels pass anyway: their z is +-inf) ->
            //             the per-pixel depth arithmetic is skipped, the verdict is the same;
            //   none can: (only when the frame is all that is asked for, no MR_FRAME_COUNTERS: the
            //             stencil matters where a triangle was drawn) even its most favourable depth
            //             loses against the most favourable covered z -> the quad is skipped.
            bool q_allpass = false;
            if (!q_reject) {
                const QuadHead &h = *reinterpret_cast<const QuadHead *>(s_quad + lane * QUAD_STAGE_U4);
                const double xa = (double)gx, xb = (double)(gx + TILE_W - 1);
                const double ya = (double)(gy + (tid / WAVE) * (WAVE / TILE_W)), yb = ya + (double)(WAVE / TILE_W - 1);
                const double t00 = (h.nx * xa + h.ny * ya) + h.d, t10 = (h.nx * xb + h.ny * ya) + h.d;
                const double t01 = (h.nx * xa + h.ny * yb) + h.d, t11 = (h.nx * xb + h.ny * yb) + h.d;
                const double tmin = fmin(fmin(t00, t10), fmin(t01, t11)), tmax = fmax(fmax(t00, t10), fmax(t01, t11));
                const double inz = approx_rcp(h.nz);
                const double za = -tmin * inz, zb = -tmax * inz;
                const double slack = 1e-12 * fmax(fabs(za), fabs(zb)) +
                                     1e-15 * ((fabs(h.nx) * xb + fabs(h.ny) * yb) + fabs(h.d)) * fabs(inz);
                const double zs_lo = fmin(za, zb) - slack, zs_hi = fmax(za, zb) + slack;
                const double den_lo = f_plus_n - zs_hi * f_minus_n, den_hi = f_plus_n - zs_lo * f_minus_n;
                const bool sane = two_nf > 0 && f_minus_n > 0 && den_lo > 1e-9 * f_plus_n && h.nz != 0 &&
                                  fabs(inz) < 1e300 && den_hi < 1e300;
                const double lo = two_nf * (1.0 - 1e-12), hi = two_nf * (1.0 + 1e-12);
                // rh: a pixel passes when zbuf >= zq; lh: when zbuf <= zq
                const bool none = rh ? lo > zlim * den_hi : hi < zlim * den_lo;
                const bool all = rh ? hi <= zhard * den_lo : lo >= zhard * den_hi;
                if (sane && !counters && none) q_reject = true;
                q_allpass = sane && all;
            }
            unsigned long long todo = __ballot(!q_reject);
            const unsigned long long allpass = __ballot(q_allpass);

            while (todo) {
                const int j = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                // The whole record is fetched up front with 16-byte broadcast reads (every lane the same
                // LDS address) behind ONE wait: a heavy tile walks 50-100 quads one after the other, and
                // a read issued only where its value is needed puts an LDS round trip on that serial
                // chain each time (measured: 0.5 us per quad against 0.2).  Which edges need the per-pixel
                // test is wavefront-uniform (lane j's verdict), so the skipped ones cost no arithmetic.
                const uint4 *rec = s_quad + j * QUAD_STAGE_U4;
                const uint4 p0 = rec[0], p1 = rec[1], hb = rec[2], p3 = rec[3];
                const uint4 ea0 = rec[4], eb0 = rec[5], ea1 = rec[6], eb1 = rec[7], ea2 = rec[8], eb2 = rec[9],
                            ea3 = rec[10], eb3 = rec[11];
                const int em = __builtin_amdgcn_readlane(emask, j);
                const int x0 = (int)(int16_t)(hb.x & 0xffffu), x1 = (int)(int16_t)(hb.x >> 16);
                const int y0 = (int)(int16_t)(hb.y & 0xffffu), y1 = (int)(int16_t)(hb.y >> 16);
                const int nv = (int)hb.z;
                const bool front = hb.w != 0;
                bool in = live & (px >= x0) & (px < x1) & (py >= y0) & (py < y1);
                auto d2 = [](uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); };
                auto inner = [&](const uint4 &a, const uint4 &b) {       // staged edges: inner side is > 0
                    const double ax = dpx - d2(a.x, a.y), ay = dpy - d2(a.z, a.w);
                    return ax * d2(b.z, b.w) - ay * d2(b.x, b.y) > 0;
                };
                if (em & 1) in = in & inner(ea0, eb0);
                if (em & 2) in = in & inner(ea1, eb1);
                if (em & 4) in = in & inner(ea2, eb2);
                if (em & 8) in = in & inner(ea3, eb3);
                if (em & 16) {                              // clipped polygons with 5+ vertices are rare
                    const QuadRec *q = quads + s_id[j];
                    for (int i = 4; i < nv; ++i) {
                        const double ax = dpx - q->e[i].sx, ay = dpy - q->e[i].sy;
                        const double cr = ax * q->e[i].ey - ay * q->e[i].ex;
                        in = in & (front ? cr > 0 : cr < 0);
                    }
                }
                const unsigned long long m = __ballot(in);
                if (!m) continue;
                qfrags += (unsigned int)__popcll(m);
                bool pass = true;
                if (!((allpass >> j) & 1)) {
                // Depth of the quad at the sample and the test against the z-buffer
                // (obj/triangular.py:351-360): zq = two_nf / (f_plus_n + (t / nz) * f_minus_n) with
                // t the plane's value at the sample.  The reference's value needs two IEEE divisions;
                // the DECISION needs none: multiplied through by nz, zbest - zq has the sign of
                // E / D with D = f_plus_n * nz + t * f_minus_n and E = zbest * D - two_nf * nz.
                // That settles it unless z-buffer and quad depth agree to nine digits or D is close
                // to its pole (or something is not finite); only then is the exactly rounded
                // expression evaluated.  t itself is computed exactly as the reference does.
                // Decisions stay bit-exact.
                const double q_nx = __hiloint2double((int)p0.y, (int)p0.x), q_ny = __hiloint2double((int)p0.w, (int)p0.z);
                const double nzq = __hiloint2double((int)p1.y, (int)p1.x), q_d = __hiloint2double((int)p1.w, (int)p1.z);
                const double t = (q_nx * dpx + q_ny * dpy) + q_d;
                const double a0 = d2(p3.x, p3.y), b0 = d2(p3.z, p3.w);          // f_plus_n * nz, two_nf * nz (staging)
                const double tf = t * f_minus_n;
                const double den = a0 + tf;
                const double prod = zbest * den;
                const double e = prod - b0;
                // an empty z-buffer entry (+-inf) beats or loses against every finite depth
                const bool zinf = fabs(zbest) == INFINITY;
                pass = zinf ? (rh ? zbest > 0 : zbest < 0) : (rh ? ((e > 0) == (den > 0)) : ((e < 0) == (den > 0)));
                const bool clear = zinf || fabs(e) > 1e-9 * (fabs(prod) + fabs(b0));
                const bool unsure = in && !(clear && fabs(den) > 2e-4 * (fabs(a0) + fabs(tf)));
                if (__ballot(unsure)) {
                    if (unsure) {
                        const double z = two_nf / (f_plus_n - (-t / nzq) * f_minus_n);     // linearize_z (obj/core.py:226-228)
                        pass = rh ? (zbest >= z) : (zbest <= z);
                    }
                }
                }
                pass = pass && in;
                qupd += (unsigned int)__popcll(__ballot(pass));
                sten += pass ? (front ? 1 : -1) : 0;
            }
        }
    }

    if (SPLIT && n_parts > 1) {
        // Leave this part's counts, then take a ticket.  Hand-off between workgroups on different CUs
        // (MI355X_MICROARCH.md, inter-workgroup visibility): plain stores, EVERY storing wave drains its
        // own stores (s_waitcnt vmcnt(0): a workgroup barrier does not wait for vector memory), the
        // barrier, ONE agent-scope release, then the relaxed ticket; the last arriver does ONE agent-scope
        // acquire (invalidates its CU's L1) behind a barrier, then plain loads.
        __shared__ uint32_t s_ticket;
        const TileArgs &ta = kernargs<TileKernArgs>().ta;
        int32_t *mine = ta.split_sten + ((size_t)entry * HEAVY_SPLIT + part) * TILE_PX;
        mine[lp] = sten;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_ticket = atomicAdd(&ta.split_arrive[entry], 1u);
        }
        __syncthreads();
        if (s_ticket != (uint32_t)(n_parts - 1)) return;  // not the last one: whoever is adds these counts and shades
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ta.split_arrive[entry] = 0;                   // every part has arrived: ready for the next frame
        }
        __syncthreads();
        for (int k = 0; k < n_parts; ++k)
            if (k != part) sten += ta.split_sten[((size_t)entry * HEAVY_SPLIT + k) * TILE_PX + lp];
    }
    const unsigned long long t_quads = __builtin_amdgcn_s_memrealtime();

    // ---- 4. deferred shading + finalise (kernels_shade.h; obj/core.py:640)
    const bool lit = (int16_t)sten == 0;                  // the reference's buffer is int16
    if (live) {
        const TileKernArgs &ka = kernargs<TileKernArgs>();
        const FrameConst &fc = ka.fc;
        const TileArgs &ta = ka.ta;
        const ShadeArgs &sh = ka.sh;
        const LightConst lc = light_const(fc);
        int px, py;
        my_pixel(px, py);
        float rgb[3] = { fc.background[0], fc.background[1], fc.background[2] };
        bool ready_u8 = false;
        if (best >= 0) {
            const TriRec t = sh.tris[best];
            ShadedFace sf;
            load_shaded_face(sh, fc.pos32 != 0, best, sf);
            const Material *mp = mat_lds ? reinterpret_cast<const Material *>(s_mat) + t.material
                                         : sh.materials + t.material;
            shade_pixel(lc, t, sf, *mp, px, py, lit, rgb);
        } else if ((fc.flags & MR_FRAME_SKYBOX) && sh.sky) {
            sky_color(fc, sh.sky, px, py, rgb);
        } else if (fc.background_u8 >> 24) {
            ready_u8 = true;       // the host already finalised the colour with NumPy itself (obj/core.py:600,640)
        }
        const size_t at_px = (size_t)py * fc.width + px;
        if (sh.frame && taps) { sh.frame[at_px * 3 + 0] = rgb[0]; sh.frame[at_px * 3 + 1] = rgb[1]; sh.frame[at_px * 3 + 2] = rgb[2]; }
        uint8_t *o = sh.out + ((size_t)out_row(fc, py, ltr) * fc.width + px) * 3;
        if (ready_u8) {
            o[0] = (uint8_t)fc.background_u8; o[1] = (uint8_t)(fc.background_u8 >> 8); o[2] = (uint8_t)(fc.background_u8 >> 16);
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) o[j] = gamma_u8(rgb[j], s_gamma);
        }
        if (ta.zbuf && taps) {                            // taps for the parity tests / per-face status / overlay
            ta.zbuf[at_px] = zbest;
            ta.winner[at_px] = best;
            ta.stencil[at_px] = sten;
        }
    }

    // ---- per-tile statistics and housekeeping
    if (counters) {
        const unsigned long long cov = __ballot(covered), litm = __ballot(covered && lit);
        if (lane == 0) {
            if (cov) atomicAdd(&s_cnt[3], (unsigned int)__popcll(cov));
            if (litm) atomicAdd(&s_cnt[4], (unsigned int)__popcll(litm));
            if (qfrags) atomicAdd(&s_cnt[1], qfrags);
            if (qupd) atomicAdd(&s_cnt[2], qupd);
        }
    }
    __syncthreads();
    // per-tile partial counts, summed by k_reduce_tile_stats (thousands of workgroups adding to
    // one cache line of counters would serialise at the memory side)
    const TileArgs &ta = kernargs<TileKernArgs>().ta;
    uint32_t *rec = ta.tile_stats + (size_t)tile * TILE_REC;
    if (tid < TILE_STATS) rec[tid] = s_cnt[tid];
    if (tid == 0) {                                       // list lengths; timing for mr_debug_read_tile_records
        rec[5] = n_small_raw; rec[6] = n_big_raw; rec[7] = n_quad_raw;
        rec[8] = (uint32_t)t_start;
        rec[9] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        rec[10] = (uint32_t)t_raster; rec[11] = (uint32_t)t_quads;
        if (n_small_raw > ta.cap[0]) { atomicOr(&ta.ctr->overflow, 1u); atomicMax(&ta.ctr->max_list[0], n_small_raw); }
        if (n_big_raw > ta.cap[1]) { atomicOr(&ta.ctr->overflow, 2u); atomicMax(&ta.ctr->max_list[1], n_big_raw); }
        if (n_quad_raw > ta.cap[2]) { atomicOr(&ta.ctr->overflow, 4u); atomicMax(&ta.ctr->max_list[2], n_quad_raw); }
    }
    if (tid < BIN_CLASSES) ta.bin_count[tid * n_tiles + tile] = 0;   // cursors zeroed for the next frame
    if (tid == 0) {                                       // what the slot's next frame should know about this tile
        int cls = tile_class(cost);
        // where tiles are split, class 1 holds those whose quad walk is worth sharing
        if (SPLIT) cls = (cost >= ta.split_cost && n_quad_raw >= ta.split_quads) ? 1 : cls < 2 ? 2 : cls;
        ta.tile_class[tile] = (uint8_t)cls;
    }
}

// Sums the per-tile partial counts and list lengths into the frame counters; grid-stride over
// tiles, one atomic per workgroup and counter; run only when the statistics are asked for.
__global__ void __launch_bounds__(256)
k_reduce_tile_stats(const uint32_t *__restrict__ tile_stats, int n_tiles, Counters *__restrict__ ctr)
{
    constexpr int NS = TILE_STATS + 2;                    // + triangle pairs, all pairs
    __shared__ unsigned long long part[NS][256 / WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + tid, gsz = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc[NS] = {};
    for (size_t t = gid; t < (size_t)n_tiles; t += gsz) {
        const uint32_t *rec = tile_stats + t * TILE_REC;
#pragma unroll
        for (int k = 0; k < TILE_STATS; ++k) acc[k] += rec[k];
        acc[TILE_STATS] += (unsigned long long)rec[5] + rec[6];
        acc[TILE_STATS + 1] += (unsigned long long)rec[5] + rec[6] + rec[7];
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        unsigned long long v = acc[k];
        for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) part[k][wv] = v;
    }
    __syncthreads();
    if (tid < NS) {
        unsigned long long v = 0;
        for (int w = 0; w < 256 / WAVE; ++w) v += part[tid][w];
        if (tid < TILE_STATS) {
            unsigned long long *dst = tid == 0 ? &ctr->frag_tri : tid == 1 ? &ctr->frag_quad
                                    : tid == 2 ? &ctr->stencil_updates : tid == 3 ? &ctr->covered_px : &ctr->lit_px;
            if (v) atomicAdd(dst, v);
        } else if (v) {
            atomicAdd(tid == TILE_STATS ? &ctr->tri_bin_total : &ctr->bin_total, (unsigned int)v);
        }
    }
}

}  // namespace mr
