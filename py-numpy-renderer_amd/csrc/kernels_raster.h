// kernels_raster.h -- screen-tile binning and the tile-owner visibility kernel.
//
// Design (MI355X-first, not a port of the reference's per-face loops): the frame is cut into
// 8x8-pixel tiles, one 64-lane wavefront owns one tile and keeps its z / winner / stencil in
// registers while it walks the tile's primitive lists.  No atomics touch the frame buffers and
// every buffer is written exactly once, coalesced.  The reference's sequential semantics
// (obj/triangular.py:101-118: later face wins on z ties) are reproduced order-free by
// resolving ties on the face index.
//
//   k_bin<FILL,QUADS>  count / fill (primitive, tile) pairs
//   k_scan_bins        exclusive scan of the per-tile counts (single workgroup)
//   k_tile_raster      coverage + clip + z for triangles, then the stencil count of the
//                      shadow quads against the final z (obj/triangular.py:72-118, 335-368)
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
    for (int i = 0; i < q.n; ++i) {
        const double ax0 = xa - q.sx[i], ax1 = xb - q.sx[i], ay0 = ya - q.sy[i], ay1 = yb - q.sy[i];
        const double c00 = ax0 * q.ey[i] - ay0 * q.ex[i], c10 = ax1 * q.ey[i] - ay0 * q.ex[i];
        const double c01 = ax0 * q.ey[i] - ay1 * q.ex[i], c11 = ax1 * q.ey[i] - ay1 * q.ex[i];
        const bool any = q.is_front ? (c00 > 0 || c10 > 0 || c01 > 0 || c11 > 0)
                                    : (c00 < 0 || c10 < 0 || c01 < 0 || c11 < 0);
        if (!any) return false;
    }
    return true;
}

constexpr int BIN_SMALL = 4;     // primitives touching <= this many tiles are binned by their own lane

// One lane per primitive.  Lanes whose primitive touches a few tiles bin it themselves; the
// others are taken one at a time by the whole wavefront, lanes striding over the tile span
// (a floor triangle or a shadow quad spans thousands of tiles).
template <bool FILL, bool QUADS>
__global__ void __launch_bounds__(256)
k_bin(const FrameConst fc, const TriRec *__restrict__ tris, const uint32_t *__restrict__ valid_list,
      const uint8_t *__restrict__ status, const QuadRec *__restrict__ quads,
      const Counters *__restrict__ ctr_in, uint32_t quad_cap, uint32_t *__restrict__ bin_count,
      const uint32_t *__restrict__ bin_offset, uint32_t *__restrict__ items, uint32_t cap)
{
    const uint32_t n = QUADS ? min(ctr_in->n_quads_drawn, quad_cap) : ctr_in->n_valid_tris;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    uint32_t prim = 0;
    TileSpan sp = { 0, 0, 0, 0 };
    bool valid = i < n;
    if (valid) {
        if (QUADS) {
            prim = i;
            const QuadRec &q = quads[i];
            valid = tile_span(fc, q.x0, q.x1, q.y0, q.y1, sp);
        } else {
            prim = valid_list[i];
            const TriRec &t = tris[prim];
            valid = status[prim] == FACE_OK && tile_span(fc, t.x0, t.x1, t.y0, t.y1, sp);
        }
    }
    const int tw = sp.tx1 - sp.tx0, th = sp.ty1 - sp.ty0;
    const int ntiles = valid ? tw * th : 0;
    const bool small = valid && ntiles <= BIN_SMALL;

    auto emit = [&](int tx, int ty, uint32_t p) {
        const uint32_t tile = (uint32_t)ty * fc.tiles_x + tx;
        const uint32_t pos = atomicAdd(&bin_count[tile], 1u);
        if (FILL) {
            const uint32_t at = bin_offset[tile] + pos;
            if (at < cap) items[at] = p;
        }
    };

    if (small) {
        for (int ty = sp.ty0; ty < sp.ty1; ++ty)
            for (int tx = sp.tx0; tx < sp.tx1; ++tx)
                if (!QUADS || quad_touches_tile(quads[prim], tx, ty + fc.tile_y0)) emit(tx, ty, prim);
    }
    unsigned long long big = __ballot(valid && !small);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t p = __shfl(prim, src);
        const int bx0 = __shfl(sp.tx0, src), by0 = __shfl(sp.ty0, src);
        const int bw = __shfl(tw, src), total = __shfl(ntiles, src);
        for (int j = lane; j < total; j += WAVE) {
            const int tx = bx0 + j % bw, ty = by0 + j / bw;
            if (!QUADS || quad_touches_tile(quads[p], tx, ty + fc.tile_y0)) emit(tx, ty, p);
        }
    }
}

// Exclusive scan of the per-tile counts by one workgroup; also zeroes the counts so the fill
// pass can reuse them as cursors, records the total and flags overflow of the item buffer.
__global__ void __launch_bounds__(1024)
k_scan_bins(uint32_t *__restrict__ bin_count, uint32_t *__restrict__ bin_offset, int n_tiles,
            uint32_t cap, uint32_t *__restrict__ total_out, uint32_t overflow_bit, Counters *__restrict__ ctr)
{
    __shared__ uint32_t partial[1024];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int per = (n_tiles + nt - 1) / nt;
    const int beg = min(tid * per, n_tiles), end = min(beg + per, n_tiles);
    uint32_t sum = 0;
    for (int i = beg; i < end; ++i) sum += bin_count[i];
    partial[tid] = sum;
    __syncthreads();
    for (int off = 1; off < nt; off <<= 1) {
        uint32_t v = tid >= off ? partial[tid - off] : 0;
        __syncthreads();
        partial[tid] += v;
        __syncthreads();
    }
    uint32_t run = partial[tid] - sum;
    for (int i = beg; i < end; ++i) {
        uint32_t c = bin_count[i];
        bin_offset[i] = run;
        bin_count[i] = 0;
        run += c;
    }
    if (tid == nt - 1) {
        bin_offset[n_tiles] = partial[tid];
        *total_out = partial[tid];
        if (partial[tid] > cap) atomicOr(&ctr->overflow, overflow_bit);
    }
}

// One wavefront per tile, one pixel per lane.
__global__ void __launch_bounds__(256)
k_tile_raster(const FrameConst fc, const TriRec *__restrict__ tris, const TriClip *__restrict__ clips,
              const uint32_t *__restrict__ tri_offset, const uint32_t *__restrict__ tri_items,
              uint32_t tri_item_cap, const QuadRec *__restrict__ quads,
              const uint32_t *__restrict__ quad_offset, const uint32_t *__restrict__ quad_items,
              uint32_t quad_item_cap, double *__restrict__ zbuf,
              int32_t *__restrict__ winner, int16_t *__restrict__ stencil, Counters *__restrict__ ctr)
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
    // (an overflowing item list is truncated; the host then grows it and renders the frame again)
    const uint32_t tbeg = tri_offset[tile], tend = min(tri_offset[tile + 1], tri_item_cap);
    for (uint32_t k = tbeg; k < tend; ++k) {
        const int f = __builtin_amdgcn_readfirstlane((int)tri_items[k]);
        const TriRec &t = tris[f];
        const bool single = (t.flags & TF_SINGLE_BOX) != 0;
        bool in = live && px >= t.x0 && px < t.x1 && py >= t.y0 && py < t.y1;
        float u, v, w;
        tri_bary(t, dpx, dpy, single, u, v, w);
        in = in && u >= 0 && v >= 0 && w >= 0;
        const unsigned long long m = __ballot(in);
        if (!m) continue;
        frags += (unsigned int)__popcll(m);
        if (t.flags & TF_CLIP) {
            if (in) {
                double p[3];
                persp_bary(t, u, v, w, single, p);
                in = inside_clip(p, clips[f].clip) && inside_clip(p, clips[f].clipd);
            }
        }
        const double z = rows_dot3((t.flags & TF_SINGLE_Z) != 0, (double)u, (double)v, (double)w,
                                   t.zl0, t.zl1, t.zl2);
        // sequential rule "zbuf >= z writes" == smallest z, and among equal z the latest face
        const bool closer = rh ? (z < zbest) : (z > zbest);
        if (in && (closer || (z == zbest && f > best))) { zbest = z; best = f; }
    }

    // ---- shadow quads against the final z: stencil +-1 (obj/triangular.py:335-368)
    int sten = 0;
    unsigned int qfrags = 0, qupd = 0;
    if (fc.flags & MR_FRAME_SHADOWS) {
        const uint32_t qbeg = quad_offset[tile], qend = min(quad_offset[tile + 1], quad_item_cap);
        for (uint32_t k = qbeg; k < qend; ++k) {
            const int qi = __builtin_amdgcn_readfirstlane((int)quad_items[k]);
            const QuadRec &q = quads[qi];
            bool in = live && px >= q.x0 && px < q.x1 && py >= q.y0 && py < q.y1;
            const bool front = q.is_front != 0;
            for (int i = 0; i < q.n; ++i) {
                const double ax = dpx - q.sx[i], ay = dpy - q.sy[i];
                const double cr = ax * q.ey[i] - ay * q.ex[i];
                in = in && (front ? cr > 0 : cr < 0);
            }
            const unsigned long long m = __ballot(in);
            if (!m) continue;
            qfrags += (unsigned int)__popcll(m);
            double z = -((q.nx * dpx + q.ny * dpy) + q.d) / q.nz;
            z = linearize_z(fc, z);
            const bool pass = in && (rh ? (zbest >= z) : (zbest <= z));
            qupd += (unsigned int)__popcll(__ballot(pass));
            sten += pass ? (front ? 1 : -1) : 0;
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
    if (lane == 0) {
        if (frags) atomicAdd(&ctr->frag_tri, (unsigned long long)frags);
        if (qfrags) atomicAdd(&ctr->frag_quad, (unsigned long long)qfrags);
        if (qupd) atomicAdd(&ctr->stencil_updates, (unsigned long long)qupd);
        if (cov) atomicAdd(&ctr->covered_px, (unsigned long long)__popcll(cov));
        if (litm) atomicAdd(&ctr->lit_px, (unsigned long long)__popcll(litm));
    }
}

}  // namespace mr
