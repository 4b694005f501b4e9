// kernels_bin.h -- screen-tile binning: which primitives touch which 16x16-pixel tile.
//
// Count -> scan -> fill, with the count folded into the kernels that create the primitives:
//   k_tri_setup      every face that goes on to the visibility kernel adds itself to the
//                    counts of the <= BIN_SMALL tiles it touches (bin_triangles); a face
//                    that spans more tiles is cut into work items of 64 tiles
//   k_quad_setup     every shadow quad is cut into work items of 64 tiles
//   k_bin_large_and_count   (kernels_geometry.h) count pass of the work items: one wavefront per
//                    item, one tile per lane
//   k_scan_bins      exclusive scan of the per-(class, tile) counts
//   k_bin_fill       the same enumeration again, now writing the item array: small triangles
//                    from the list of set-up faces, everything else from the work items
// Both passes decide with the same functions on the same records (tile_span, pair_class,
// quad_touches_tile), so the fill writes exactly the slots the count reserved.  A face's bins
// do not depend on k_tri_count's later verdict: a face that turns out to have no surviving
// fragment (CLIPPED) is still listed, and the visibility kernel finds no fragment for it.
//
// Bin layout: count/offset arrays hold BIN_CLASSES * n_tiles entries (small triangle pairs,
// big triangle pairs, quads); one scan lays all of them out in a single item array.
#pragma once

#include "rast_math.h"

namespace mr {

struct TileSpan { int tx0, tx1, ty0, ty1; };   // band-local tile coordinates, half-open

__device__ __forceinline__ bool tile_span(const FrameConst &fc, int x0, int x1, int y0, int y1, TileSpan &s)
{
    y0 = max(y0, fc.band_y0);
    y1 = min(y1, fc.band_y1);
    if (x0 >= x1 || y0 >= y1) return false;
    s.tx0 = x0 / TILE_W;            s.tx1 = (x1 - 1) / TILE_W + 1;
    s.ty0 = y0 / TILE_H - fc.tile_y0; s.ty1 = (y1 - 1) / TILE_H + 1 - fc.tile_y0;
    return true;
}

// Exact tile rejection for a shadow quad.  The inside test of a sample is, per edge,
// sign(rn(rn(ax*ey) - rn(ay*ex))) (obj/triangular.py:305-316).  rn(ax*ey) is monotone in the
// sample's x and rn(ay*ex) in its y, so over a tile the extreme of the rounded expression is
// attained at one of the four corner samples: if no corner is on the inner side of some
// edge, no sample of the tile is.  No margin is needed and no fragment can be lost.
__device__ __forceinline__ bool quad_touches_tile(const QuadRec &q, int tx, int ty)
{
    const double xa = (double)(tx * TILE_W), xb = (double)(tx * TILE_W + TILE_W - 1);
    const double ya = (double)(ty * TILE_H), yb = (double)(ty * TILE_H + TILE_H - 1);
    const bool front = q.is_front != 0;
    for (int i = 0; i < q.n; ++i) {
        const QuadEdge e = q.e[i];
        const double px0 = (xa - e.sx) * e.ey, px1 = (xb - e.sx) * e.ey;
        const double py0 = (ya - e.sy) * e.ex, py1 = (yb - e.sy) * e.ex;
        const double c00 = px0 - py0, c10 = px1 - py0, c01 = px0 - py1, c11 = px1 - py1;
        const bool any = front ? (c00 > 0 || c10 > 0 || c01 > 0 || c11 > 0)
                               : (c00 < 0 || c10 < 0 || c01 < 0 || c11 < 0);
        if (!any) return false;
    }
    return true;
}

struct BinArgs {
    const TriRec *tris;
    const uint32_t *valid_list;   // faces k_tri_setup passed on (fill pass)
    const QuadRec *quads;
    Counters *ctr;
    uint32_t quad_cap;
    uint32_t *bin_count;          // [BIN_CLASSES * n_tiles]
    const uint32_t *bin_offset;   // [BIN_CLASSES * n_tiles + 1]
    uint32_t *items;
    uint32_t item_cap;
    uint2 *work;                  // (primitive, chunk of 64 tiles); primitive = face, or WORK_QUAD | quad
    uint32_t work_cap;
    uint4 *quad_work;             // (tile, first item, count, -) of k_tile_quads
    uint32_t quad_work_cap;
};

constexpr int TILE_STATS = 5;     // per-tile partial counters written by k_tile_raster
constexpr int TILE_REC = 8;       // words per tile record: the counters, then start / end time (10 ns ticks) and list sizes
constexpr int QUAD_BATCH = 16;    // shadow quads per work item of k_tile_quads
constexpr int BIN_SMALL = 4;      // triangles touching <= this many tiles are binned by their own lane
constexpr uint32_t WORK_QUAD = 0x80000000u;

struct PrimBox { int x0, x1, y0, y1; };

// Bin class of a (primitive, tile) pair: 0 small triangle pair (walked by a few lanes), 1 big
// triangle pair (one pixel per lane), 2 shadow quad.  A triangle whose fragments need the
// per-fragment clip test always goes the per-pixel way: there its corner data is wavefront-
// uniform (scalar loads), whereas lanes that each test their own triangle would need 36 more
// vector registers for it, halving the workgroups a CU can hold for a path that only
// triangles on the frustum's border take.
__device__ __forceinline__ int pair_class(const FrameConst &fc, bool is_quad, bool clip, const PrimBox &pb, int tx, int ty)
{
    if (is_quad) return 2;
    if (clip) return 1;
    const int gx = tx * TILE_W, gy = (ty + fc.tile_y0) * TILE_H;
    const int w = min(pb.x1, gx + TILE_W) - max(pb.x0, gx);
    const int h = min(min(pb.y1, gy + TILE_H), fc.band_y1) - max(max(pb.y0, gy), fc.band_y0);
    return w * h > BIG_PAIR_PX ? 1 : 0;
}

template <bool FILL>
__device__ __forceinline__ void bin_emit(const FrameConst &fc, const BinArgs &a, int cls, uint32_t id, int tx, int ty)
{
    const uint32_t bin = (uint32_t)cls * (uint32_t)(fc.tiles_x * fc.tiles_y) + (uint32_t)ty * fc.tiles_x + tx;
    const uint32_t pos = atomicAdd(&a.bin_count[bin], 1u);
    if (FILL) {
        const uint32_t at = a.bin_offset[bin] + pos;
        if (at < a.item_cap) a.items[at] = id;
    }
}

// The same for a whole wavefront at once (every lane calls it, `want` says whether it has a
// pair to emit).  Neighbouring triangles of a mesh fall into the same tile, and a tile's
// counter is one L2 location: lanes with the same bin are combined into ONE atomic and share
// out the returned range by rank, instead of queueing up to 64 deep on that location.  The
// groups are found first (ALU only), then every group leader issues its atomic in the same
// instruction: one memory round trip however many different bins the wavefront touches.
template <bool FILL>
__device__ __forceinline__ void bin_emit_wave(const FrameConst &fc, const BinArgs &a, bool want, int cls,
                                              uint32_t id, int tx, int ty)
{
    const uint32_t bin = (uint32_t)cls * (uint32_t)(fc.tiles_x * fc.tiles_y) + (uint32_t)ty * fc.tiles_x + tx;
    const int lane = threadIdx.x & (WAVE - 1);
    unsigned long long todo = __ballot(want), mine = 0;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lbin = (uint32_t)__shfl((int)bin, leader);
        const unsigned long long same = __ballot(want && bin == lbin) & todo;
        if ((same >> lane) & 1ull) mine = same;
        todo &= ~same;
    }
    if (!want) mine = 0;
    const int my_leader = mine ? __ffsll((long long)mine) - 1 : lane;
    if (!FILL) {
        // the count pass does not need the old value: the atomic is issued without a return and the
        // wavefront moves on (with the return it waited out a memory round trip per slot)
        if (mine && lane == my_leader) atomicAdd(&a.bin_count[bin], (uint32_t)__popcll(mine));
        return;
    }
    uint32_t base = 0;
    if (mine && lane == my_leader) base = atomicAdd(&a.bin_count[bin], (uint32_t)__popcll(mine));
    base = (uint32_t)__shfl((int)base, my_leader);
    if (FILL && mine) {
        const uint32_t at = a.bin_offset[bin] + base + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
        if (at < a.item_cap) a.items[at] = id;
    }
}

// A wavefront reserves work items for its lanes' large primitives with a single atomic (prefix
// sum of the lanes' chunk counts), then every lane writes its own.  Every lane must call it.
__device__ __forceinline__ void push_work_items(const BinArgs &a, uint32_t prim, uint32_t chunks)
{
    const int lane = threadIdx.x & (WAVE - 1);
    uint32_t incl = chunks;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    const uint32_t wave_total = __shfl(incl, WAVE - 1);
    if (wave_total == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&a.ctr->n_work, wave_total);
    base = __shfl(base, 0) + incl - chunks;
    for (uint32_t c = 0; c < chunks; ++c) {
        if (base + c < a.work_cap) a.work[base + c] = make_uint2(prim, c);
        else atomicOr(&a.ctr->overflow, 2u);
    }
}

// One lane per triangle (valid: the lane has one; face, pixel box and TF_CLIP given).  Small
// spans are emitted one "slot" at a time so that the wavefront can combine lanes that hit the
// same bin; with FILL == false a larger span becomes work items (the fill pass reuses them).
// Every lane of the wavefront must call it.
template <bool FILL>
__device__ __forceinline__ void bin_triangles(const FrameConst &fc, const BinArgs &a, bool valid, uint32_t face,
                                              const PrimBox &pb, bool clip)
{
    TileSpan sp = { 0, 0, 0, 0 };
    valid = valid && tile_span(fc, pb.x0, pb.x1, pb.y0, pb.y1, sp);
    const int ntiles = valid ? (sp.tx1 - sp.tx0) * (sp.ty1 - sp.ty0) : 0;
    const bool small = valid && ntiles <= BIN_SMALL;
    if (__ballot(small)) {
        const int bw = max(sp.tx1 - sp.tx0, 1);
#pragma unroll
        for (int slot = 0; slot < BIN_SMALL; ++slot) {
            const bool want = small && slot < ntiles;
            if (!__ballot(want)) break;
            const int tx = sp.tx0 + slot % bw, ty = sp.ty0 + slot / bw;
            bin_emit_wave<FILL>(fc, a, want, want ? pair_class(fc, false, clip, pb, tx, ty) : 0, face, tx, ty);
        }
    }
    if (!FILL) push_work_items(a, face, (valid && ntiles > BIN_SMALL) ? (uint32_t)(ntiles + WAVE - 1) / WAVE : 0u);
}

// Work items of a shadow quad (count pass; called by the lane that owns the quad's record).
__device__ __forceinline__ uint32_t quad_chunks(const FrameConst &fc, int x0, int x1, int y0, int y1)
{
    TileSpan sp;
    if (!tile_span(fc, x0, x1, y0, y1, sp)) return 0u;
    return (uint32_t)((sp.tx1 - sp.tx0) * (sp.ty1 - sp.ty0) + WAVE - 1) / WAVE;
}

// Count (FILL == false) or fill pass over the work items: one wavefront per item, one tile per lane.
template <bool FILL>
__device__ __forceinline__ void bin_large_body(const FrameConst &fc, const BinArgs &a, uint32_t block, uint32_t n_blocks)
{
    const uint32_t n_work = min(a.ctr->n_work, a.work_cap);
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t waves = n_blocks * (blockDim.x / WAVE);
    for (uint32_t w = block * (blockDim.x / WAVE) + threadIdx.x / WAVE; w < n_work; w += waves) {
        const uint2 item = a.work[w];
        const bool is_quad = (item.x & WORK_QUAD) != 0;
        const uint32_t id = item.x & ~WORK_QUAD;
        PrimBox pb;
        bool clip = false;
        if (is_quad) {
            const QuadRec &q = a.quads[id];
            pb = { q.x0, q.x1, q.y0, q.y1 };
        } else {
            const TriRec &t = a.tris[id];
            pb = { t.x0, t.x1, t.y0, t.y1 };
            clip = (t.flags & TF_CLIP) != 0;
        }
        TileSpan sp;
        if (!tile_span(fc, pb.x0, pb.x1, pb.y0, pb.y1, sp)) continue;
        const int bw = sp.tx1 - sp.tx0, total = bw * (sp.ty1 - sp.ty0);
        const int j = (int)item.y * WAVE + lane;
        if (j >= total) continue;
        const int tx = sp.tx0 + j % bw, ty = sp.ty0 + j / bw;
        if (!is_quad || quad_touches_tile(a.quads[id], tx, ty + fc.tile_y0))
            bin_emit<FILL>(fc, a, pair_class(fc, is_quad, clip, pb, tx, ty), id, tx, ty);
    }
}

// Fill pass in one launch: workgroups [0, large_blocks) fill from the work items, the rest walk
// the list of set-up faces (small triangles only; the others were work items) and, thread u for
// tile u, cut each tile's shadow-quad list into work items of at most QUAD_BATCH quads for
// k_tile_quads, so that the tiles under a dense shadow volume are shared out over many workgroups.
__global__ void __launch_bounds__(256)
k_bin_fill(const FrameConst fc, const BinArgs a, uint32_t large_blocks)
{
    if (blockIdx.x < large_blocks) { bin_large_body<true>(fc, a, blockIdx.x, large_blocks); return; }
    const uint32_t u = (blockIdx.x - large_blocks) * blockDim.x + threadIdx.x;
    const bool valid = u < a.ctr->n_valid_tris;
    uint32_t face = 0;
    PrimBox pb = { 0, 0, 0, 0 };
    bool clip = false;
    if (valid) {
        face = a.valid_list[u];
        const TriRec &t = a.tris[face];
        pb = { t.x0, t.x1, t.y0, t.y1 };
        clip = (t.flags & TF_CLIP) != 0;
    }
    bin_triangles<true>(fc, a, valid, face, pb, clip);
    const uint32_t n_tiles = (uint32_t)(fc.tiles_x * fc.tiles_y);
    if (u < n_tiles) {
        const uint32_t first = a.bin_offset[2 * n_tiles + u], cnt = a.bin_offset[2 * n_tiles + u + 1] - first;
        if (cnt) {
            const uint32_t nb = (cnt + QUAD_BATCH - 1) / QUAD_BATCH;
            const uint32_t at = atomicAdd(&a.ctr->n_quad_work, nb);
            for (uint32_t b = 0; b < nb; ++b) {
                if (at + b < a.quad_work_cap)
                    a.quad_work[at + b] = make_uint4(u, first + b * QUAD_BATCH, min((uint32_t)QUAD_BATCH, cnt - b * QUAD_BATCH), 0u);
                else atomicOr(&a.ctr->overflow, 8u);
            }
        }
    }
}

// Exclusive scan of the BIN_CLASSES * n_tiles bin counts; zeroes the counts (the fill pass
// reuses them as cursors), records the totals and flags overflow of the item array.
// Single pass over many workgroups: each one scans 1024 counts (one uint4 per thread),
// publishes its total tagged with the frame's epoch, and reads the totals of all the
// workgroups before it in one go -- one lane per predecessor -- instead of waiting for a
// chained prefix.  A workgroup only ever waits for lower-numbered ones, which are dispatched
// first, so the wait always ends; the epoch tag means the slots never need clearing.
constexpr int SCAN_BLOCK = 256, SCAN_ITEMS = SCAN_BLOCK * 4;

__global__ void __launch_bounds__(SCAN_BLOCK)
k_scan_bins(uint32_t *__restrict__ bin_count, uint32_t *__restrict__ bin_offset, int n_tiles,
            uint32_t item_cap, Counters *__restrict__ ctr, unsigned long long *__restrict__ partials,
            uint32_t epoch)
{
    constexpr int NW = SCAN_BLOCK / WAVE;
    __shared__ uint32_t s_wave[NW], s_before;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int n = BIN_CLASSES * n_tiles, b = (int)blockIdx.x;
    const int i0 = b * SCAN_ITEMS + tid * 4;

    uint32_t v[4] = { 0u, 0u, 0u, 0u };
    if (i0 + 3 < n) {
        const uint4 q = *reinterpret_cast<const uint4 *>(bin_count + i0);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        for (int j = 0; j < 4; ++j) v[j] = i0 + j < n ? bin_count[i0 + j] : 0u;
    }
    const uint32_t mine = (v[0] + v[1]) + (v[2] + v[3]);
    uint32_t inc = mine;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t y = __shfl_up(inc, off);
        if (lane >= off) inc += y;
    }
    if (lane == WAVE - 1) s_wave[wv] = inc;
    __syncthreads();
    uint32_t before = 0, block_total = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const uint32_t s = s_wave[k];
        before += k < wv ? s : 0u;
        block_total += s;
    }
    if (tid == 0)
        __hip_atomic_store(&partials[b], ((unsigned long long)epoch << 32) | block_total, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (wv == 0) {
        uint32_t sum = 0;
        for (int k = lane; k < b; k += WAVE) {
            unsigned long long p;
            do {
                p = __hip_atomic_load(&partials[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            } while ((uint32_t)(p >> 32) != epoch);
            sum += (uint32_t)p;
        }
#pragma unroll
        for (int off = WAVE / 2; off; off >>= 1) sum += __shfl_xor(sum, off);
        if (lane == 0) s_before = sum;
    }
    __syncthreads();
    const uint32_t base = s_before + before + inc - mine;
    const uint32_t o[4] = { base, base + v[0], base + v[0] + v[1], base + v[0] + v[1] + v[2] };
    if (i0 + 3 < n) {
        *reinterpret_cast<uint4 *>(bin_offset + i0) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4 *>(bin_count + i0) = make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (int j = 0; j < 4; ++j)
            if (i0 + j < n) { bin_offset[i0 + j] = o[j]; bin_count[i0 + j] = 0u; }
    }
    // the triangle classes end where the quad class begins
    const int split = 2 * n_tiles;
    if (split >= i0 && split < i0 + 4 && split < n) ctr->tri_bin_total = o[split - i0];
    if (b == (int)gridDim.x - 1 && tid == 0) {
        const uint32_t total = s_before + block_total;
        bin_offset[n] = total;
        ctr->bin_total = total;
        if (n_tiles == 0 || split >= n) ctr->tri_bin_total = total;
        if (total > item_cap) atomicOr(&ctr->overflow, 1u);
    }
}

}  // namespace mr
