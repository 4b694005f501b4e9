// kernels_raster.h -- screen-tile binning and the tile-owner visibility kernel.
//
// Design (MI355X-first, not a port of the reference's per-face loops): the frame is cut into
// 8x8-pixel tiles, one 64-lane wavefront owns one tile and keeps its z / winner / stencil in
// registers while it walks the tile's primitive lists.  No atomics touch the frame buffers and
// every buffer is written exactly once, coalesced.  The reference's sequential semantics
// (obj/triangular.py:101-118: later face wins on z ties) are reproduced order-free by
// resolving ties on the face index.
//
//   k_bin_classify<FILL>  one lane per primitive (triangles, then shadow quads): primitives
//                         touching a few tiles are binned by their lane; large ones are cut into
//                         work items of 64 tiles
//   k_bin_large<FILL>     one wavefront per work item, one tile per lane
//   k_scan_bins           exclusive scan of the per-tile counts (single workgroup, coalesced)
//   k_tile_raster         coverage + clip + z for triangles, then the stencil count of the
//                         shadow quads against the final z (obj/triangular.py:72-118, 335-368)
//
// Bin layout: count/offset arrays hold 2 * n_tiles entries, triangles' bins first, then the
// quads' bins; one scan lays both out in a single item array.
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
    const uint32_t *valid_list;
    const uint8_t *status;
    const QuadRec *quads;
    Counters *ctr;
    uint32_t quad_cap;
    uint32_t *bin_count;          // [2 * n_tiles]
    const uint32_t *bin_offset;   // [2 * n_tiles + 1]
    uint32_t *items;
    uint32_t item_cap;
    uint2 *work;                  // (unified primitive index, chunk)
    uint32_t work_cap;
};

constexpr int TILE_STATS = 5;     // per-tile partial counters written by k_tile_raster
constexpr int BIN_SMALL = 4;      // primitives touching <= this many tiles are binned by their own lane

// unified primitive index -> record, tile span
__device__ __forceinline__ bool prim_span(const FrameConst &fc, const BinArgs &a, uint32_t u, uint32_t n_tris,
                                          uint32_t n_quads, bool &is_quad, uint32_t &id, TileSpan &sp)
{
    if (u < n_tris) {
        is_quad = false;
        id = a.valid_list[u];
        const TriRec &t = a.tris[id];
        return a.status[id] == FACE_OK && tile_span(fc, t.x0, t.x1, t.y0, t.y1, sp);
    }
    if (u < n_tris + n_quads) {
        is_quad = true;
        id = u - n_tris;
        const QuadRec &q = a.quads[id];
        return tile_span(fc, q.x0, q.x1, q.y0, q.y1, sp);
    }
    return false;
}

template <bool FILL>
__device__ __forceinline__ void bin_emit(const FrameConst &fc, const BinArgs &a, bool is_quad, uint32_t id,
                                         int tx, int ty)
{
    const uint32_t bin = (is_quad ? (uint32_t)(fc.tiles_x * fc.tiles_y) : 0u) + (uint32_t)ty * fc.tiles_x + tx;
    const uint32_t pos = atomicAdd(&a.bin_count[bin], 1u);
    if (FILL) {
        const uint32_t at = a.bin_offset[bin] + pos;
        if (at < a.item_cap) a.items[at] = id;
    }
}

template <bool FILL>
__global__ void __launch_bounds__(256)
k_bin_classify(const FrameConst fc, const BinArgs a)
{
    const uint32_t n_tris = a.ctr->n_valid_tris, n_quads = min(a.ctr->n_quads_drawn, a.quad_cap);
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    bool is_quad = false;
    uint32_t id = 0;
    TileSpan sp = { 0, 0, 0, 0 };
    const bool valid = prim_span(fc, a, u, n_tris, n_quads, is_quad, id, sp);
    const int ntiles = valid ? (sp.tx1 - sp.tx0) * (sp.ty1 - sp.ty0) : 0;
    if (valid && ntiles <= BIN_SMALL) {
        for (int ty = sp.ty0; ty < sp.ty1; ++ty)
            for (int tx = sp.tx0; tx < sp.tx1; ++tx)
                if (!is_quad || quad_touches_tile(a.quads[id], tx, ty + fc.tile_y0))
                    bin_emit<FILL>(fc, a, is_quad, id, tx, ty);
    }
    if (FILL) return;             // the work list of the count pass is reused by the fill pass
    // large primitives: one work item per 64 tiles of the span, written by the whole wavefront
    unsigned long long big = __ballot(valid && ntiles > BIN_SMALL);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t pu = __shfl(u, src);
        const int chunks = (__shfl(ntiles, src) + WAVE - 1) / WAVE;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&a.ctr->n_work, (uint32_t)chunks);
        base = __shfl(base, 0);
        for (int c = lane; c < chunks; c += WAVE) {
            if (base + c < a.work_cap) a.work[base + c] = make_uint2(pu, (uint32_t)c);
            else atomicOr(&a.ctr->overflow, 2u);
        }
    }
}

template <bool FILL>
__global__ void __launch_bounds__(256)
k_bin_large(const FrameConst fc, const BinArgs a)
{
    const uint32_t n_tris = a.ctr->n_valid_tris, n_quads = min(a.ctr->n_quads_drawn, a.quad_cap);
    const uint32_t n_work = min(a.ctr->n_work, a.work_cap);
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t waves = gridDim.x * (blockDim.x / WAVE);
    for (uint32_t w = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE; w < n_work; w += waves) {
        const uint2 item = a.work[w];
        bool is_quad;
        uint32_t id;
        TileSpan sp;
        if (!prim_span(fc, a, item.x, n_tris, n_quads, is_quad, id, sp)) continue;
        const int bw = sp.tx1 - sp.tx0, total = bw * (sp.ty1 - sp.ty0);
        const int j = (int)item.y * WAVE + lane;
        if (j >= total) continue;
        const int tx = sp.tx0 + j % bw, ty = sp.ty0 + j / bw;
        if (!is_quad || quad_touches_tile(a.quads[id], tx, ty + fc.tile_y0)) bin_emit<FILL>(fc, a, is_quad, id, tx, ty);
    }
}

// Exclusive scan of the 2 * n_tiles bin counts by one workgroup of 1024 threads, 8 coalesced
// rows of 1024 counts per round; zeroes the counts (the fill pass reuses them as cursors),
// records the per-class totals and flags overflow of the item array.
__global__ void __launch_bounds__(1024)
k_scan_bins(uint32_t *__restrict__ bin_count, uint32_t *__restrict__ bin_offset, int n_tiles,
            uint32_t item_cap, Counters *__restrict__ ctr)
{
    constexpr int ROWS = 8, NT = 1024, NW = NT / WAVE;
    __shared__ uint32_t wave_sum[ROWS][NW];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int n = 2 * n_tiles;
    uint32_t carry = 0;
    for (int base = 0; base < n; base += ROWS * NT) {
        uint32_t v[ROWS], inc[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int i = base + r * NT + tid;
            v[r] = i < n ? bin_count[i] : 0u;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            uint32_t x = v[r];
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                const uint32_t y = __shfl_up(x, off);
                if (lane >= off) x += y;
            }
            inc[r] = x;
            if (lane == WAVE - 1) wave_sum[r][wv] = x;
        }
        __syncthreads();
        uint32_t row_prefix = 0;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            uint32_t before = 0, row_total = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const uint32_t s = wave_sum[r][k];
                before += k < wv ? s : 0u;
                row_total += s;
            }
            const int i = base + r * NT + tid;
            if (i < n) {
                bin_offset[i] = carry + row_prefix + before + inc[r] - v[r];
                bin_count[i] = 0;
            }
            row_prefix += row_total;
        }
        carry += row_prefix;
        __syncthreads();
    }
    if (tid == 0) {
        bin_offset[n] = carry;
        const uint32_t tri_total = bin_offset[n_tiles];   // written above by this workgroup
        ctr->tri_bin_total = tri_total;
        ctr->quad_bin_total = carry - tri_total;
        if (carry > item_cap) atomicOr(&ctr->overflow, 1u);
    }
}

// ---- register staging: every lane fetches one primitive record of the tile's list, then the
// records are broadcast one at a time with v_readlane (uniform values land in SGPRs, with no
// memory latency in the inner loop and no LDS traffic).
__device__ __forceinline__ int bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float bcast(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// One wavefront per tile, one pixel per lane.
__global__ void __launch_bounds__(256)
k_tile_raster(const FrameConst fc, const TriRec *__restrict__ tris, const TriClip *__restrict__ clips,
              const QuadRec *__restrict__ quads, const uint32_t *__restrict__ bin_offset,
              const uint32_t *__restrict__ items, uint32_t item_cap, double *__restrict__ zbuf,
              int32_t *__restrict__ winner, int16_t *__restrict__ stencil, uint32_t *__restrict__ tile_stats)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int tile = blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
    const int n_tiles = fc.tiles_x * fc.tiles_y;
    if (tile >= n_tiles) return;
    const int tx = tile % fc.tiles_x, ty = tile / fc.tiles_x + fc.tile_y0;
    const int px = tx * TILE_W + (lane & (TILE_W - 1));
    const int py = ty * TILE_H + (lane / TILE_W);
    const bool live = px < fc.width && py >= fc.band_y0 && py < fc.band_y1;
    const double dpx = (double)px, dpy = (double)py;
    const bool rh = fc.system == 1;

    double zbest = rh ? INFINITY : -INFINITY;
    int best = -1;
    unsigned int frags = 0;

    // ---- triangles: coverage, per-fragment clip, depth (obj/triangular.py:72-118)
    // (an overflowing item array is truncated; the host then grows it and renders the frame again)
    const uint32_t tbeg = min(bin_offset[tile], item_cap), tend = min(bin_offset[tile + 1], item_cap);
    for (uint32_t base = tbeg; base < tend; base += WAVE) {
        const int n = (int)min((uint32_t)WAVE, tend - base);
        TriRec mine;
        {
            const uint32_t id = lane < n ? items[base + lane] : items[base];
            mine = tris[id];
        }
        for (int j = 0; j < n; ++j) {
            TriRec t;
            t.ax = bcast(mine.ax, j); t.ay = bcast(mine.ay, j);
            t.v0x = bcast(mine.v0x, j); t.v0y = bcast(mine.v0y, j);
            t.v1x = bcast(mine.v1x, j); t.v1y = bcast(mine.v1y, j);
            t.d00 = bcast(mine.d00, j); t.d01 = bcast(mine.d01, j);
            t.d11 = bcast(mine.d11, j); t.inv_den = bcast(mine.inv_den, j);
            const int bx = bcast((int)(uint16_t)mine.x0 | ((int)(uint16_t)mine.x1 << 16), j);
            const int by = bcast((int)(uint16_t)mine.y0 | ((int)(uint16_t)mine.y1 << 16), j);
            const uint32_t flags = (uint32_t)bcast((int)mine.flags, j);
            const int f = bcast(mine.face, j);
            const bool single = (flags & TF_SINGLE_BOX) != 0;
            bool in = live && px >= (bx & 0xffff) && px < (bx >> 16) && py >= (by & 0xffff) && py < (by >> 16);
            float u, v, w;
            tri_bary(t, dpx, dpy, single, u, v, w);
            in = in && u >= 0 && v >= 0 && w >= 0;
            const unsigned long long m = __ballot(in);
            if (!m) continue;
            frags += (unsigned int)__popcll(m);
            if (flags & TF_CLIP) {
                if (in) {
                    const TriClip &c = clips[f];
                    const double wc = rows_dot3(single, (double)u, (double)v, (double)w, c.dp[0], c.dp[1], c.dp[2]);
                    double p[3] = { ((double)u * c.dp[0]) / wc, ((double)v * c.dp[1]) / wc, ((double)w * c.dp[2]) / wc };
                    in = inside_clip(p, c.clip) && inside_clip(p, c.clipd);
                }
            }
            const double z = rows_dot3((flags & TF_SINGLE_Z) != 0, (double)u, (double)v, (double)w,
                                       bcast(mine.zl0, j), bcast(mine.zl1, j), bcast(mine.zl2, j));
            // sequential rule "zbuf >= z writes" == smallest z, and among equal z the latest face
            const bool closer = rh ? (z < zbest) : (z > zbest);
            if (in && (closer || (z == zbest && f > best))) { zbest = z; best = f; }
        }
    }

    // ---- shadow quads against the final z: stencil +-1 (obj/triangular.py:335-368)
    int sten = 0;
    unsigned int qfrags = 0, qupd = 0;
    if (fc.flags & MR_FRAME_SHADOWS) {
        const uint32_t qbeg = min(bin_offset[n_tiles + tile], item_cap), qend = min(bin_offset[n_tiles + tile + 1], item_cap);
        for (uint32_t base = qbeg; base < qend; base += WAVE) {
            const int n = (int)min((uint32_t)WAVE, qend - base);
            const uint32_t myid = lane < n ? items[base + lane] : items[base];
            const QuadRec *mq = quads + myid;
            const double m_nx = mq->nx, m_ny = mq->ny, m_nz = mq->nz, m_d = mq->d;
            const int m_bx = (int)(uint16_t)mq->x0 | ((int)(uint16_t)mq->x1 << 16);
            const int m_by = (int)(uint16_t)mq->y0 | ((int)(uint16_t)mq->y1 << 16);
            const int m_nf = mq->n | (mq->is_front ? 0x100 : 0);
            QuadEdge me[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) me[i] = mq->e[i];
            for (int j = 0; j < n; ++j) {
                const int bx = bcast(m_bx, j), by = bcast(m_by, j), nf = bcast(m_nf, j);
                const bool front = (nf & 0x100) != 0;
                const int nv = nf & 0xff;
                bool in = live && px >= (bx & 0xffff) && px < (bx >> 16) && py >= (by & 0xffff) && py < (by >> 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < 3 || nv > 3) {
                        const double ax = dpx - bcast(me[i].sx, j), ay = dpy - bcast(me[i].sy, j);
                        const double cr = ax * bcast(me[i].ey, j) - ay * bcast(me[i].ex, j);
                        in = in && (front ? cr > 0 : cr < 0);
                    }
                }
                if (nv > 4) {                       // clipped polygons with 5+ vertices are rare
                    const QuadRec *q = quads + bcast((int)myid, j);
                    for (int i = 4; i < nv; ++i) {
                        const double ax = dpx - q->e[i].sx, ay = dpy - q->e[i].sy;
                        const double cr = ax * q->e[i].ey - ay * q->e[i].ex;
                        in = in && (front ? cr > 0 : cr < 0);
                    }
                }
                const unsigned long long m = __ballot(in);
                if (!m) continue;
                qfrags += (unsigned int)__popcll(m);
                double z = -((bcast(m_nx, j) * dpx + bcast(m_ny, j) * dpy) + bcast(m_d, j)) / bcast(m_nz, j);
                z = linearize_z(fc, z);
                const bool pass = in && (rh ? (zbest >= z) : (zbest <= z));
                qupd += (unsigned int)__popcll(__ballot(pass));
                sten += pass ? (front ? 1 : -1) : 0;
            }
        }
    }

    if (live) {
        const size_t at = (size_t)py * fc.width + px;
        zbuf[at] = zbest;
        winner[at] = best;
        stencil[at] = (int16_t)sten;
    }
    const unsigned long long cov = __ballot(live && best >= 0);
    const unsigned long long litm = __ballot(live && best >= 0 && (int16_t)sten == 0);
    // per-tile partial counts, summed by k_reduce_tile_stats (32 400 wavefronts adding to one
    // cache line of counters would serialise at the memory side)
    if (lane == 0) {
        uint32_t *o = tile_stats + (size_t)tile * TILE_STATS;
        o[0] = frags; o[1] = qfrags; o[2] = qupd;
        o[3] = (uint32_t)__popcll(cov); o[4] = (uint32_t)__popcll(litm);
    }
}

// Sums the per-tile partial counts into the frame counters (one workgroup).
__global__ void __launch_bounds__(1024)
k_reduce_tile_stats(const uint32_t *__restrict__ tile_stats, int n_tiles, Counters *__restrict__ ctr)
{
    __shared__ unsigned long long part[TILE_STATS][1024 / WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    unsigned long long acc[TILE_STATS] = {};
    for (int t = tid; t < n_tiles; t += blockDim.x)
#pragma unroll
        for (int k = 0; k < TILE_STATS; ++k) acc[k] += tile_stats[(size_t)t * TILE_STATS + k];
#pragma unroll
    for (int k = 0; k < TILE_STATS; ++k) {
        unsigned long long v = acc[k];
        for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) part[k][wv] = v;
    }
    __syncthreads();
    if (tid < TILE_STATS) {
        unsigned long long v = 0;
        for (int w = 0; w < 1024 / WAVE; ++w) v += part[tid][w];
        unsigned long long *dst = tid == 0 ? &ctr->frag_tri : tid == 1 ? &ctr->frag_quad
                                : tid == 2 ? &ctr->stencil_updates : tid == 3 ? &ctr->covered_px : &ctr->lit_px;
        atomicAdd(dst, v);
    }
}

}  // namespace mr
