// kernels_bin.h -- screen-tile binning: which primitives touch which 16x16-pixel tile.
//
// One pass, no count / scan / fill: every (class, tile) has a list of fixed capacity in HBM
// (BinArgs::items[class], `cap[class]` entries per tile) and a cursor; a primitive is appended with
// one returning atomic on the cursor.  The capacities are per frame slot and grow on the host when
// a tile's list runs over (the frame is then rendered again, exactly like the other work lists);
// HBM is 288 GB, a 1080p frame's lists at the default capacities are 30 MB and only the filled
// part is ever touched.
//   k_setup      a face that touches <= BIN_SMALL tiles appends itself (bin_triangles); a face that
//                spans more tiles, and every shadow quad, is cut into work items of 64 tiles
//   k_bin_work   one wavefront per work item, one tile per lane
// Classes per tile: small triangle pairs, big triangle pairs, shadow quads (pair_class).
#pragma once

#include "rast_math.h"

namespace mr {

struct TileSpan { int tx0, tx1, ty0, ty1; };   // device-local tile coordinates, half-open

// frame tile row of local tile row l, and the first screen row of it
__device__ __forceinline__ int tile_row_frame(const FrameConst &fc, int l) { return fc.tile_y0 + l * fc.tile_step; }

// Local tile rows / tile columns a pixel box touches on this device.  A device owns the frame
// tile rows tile_y0 + l * tile_step (tile_step == 1: a band of rows; > 1: interleaved stripes),
// clamped to the screen rows [band_y0, band_y1).
__device__ __forceinline__ bool tile_span(const FrameConst &fc, int x0, int x1, int y0, int y1, TileSpan &s)
{
    y0 = max(y0, fc.band_y0);
    y1 = min(y1, fc.band_y1);
    if (x0 >= x1 || y0 >= y1) return false;
    s.tx0 = x0 / TILE_W;            s.tx1 = (x1 - 1) / TILE_W + 1;
    const int g0 = y0 / TILE_H - fc.tile_y0, g1 = (y1 - 1) / TILE_H - fc.tile_y0;    // frame tile rows relative to the first owned
    if (g1 < 0) return false;
    const int st = fc.tile_step;
    s.ty0 = g0 <= 0 ? 0 : (g0 + st - 1) / st;
    s.ty1 = min(g1 / st + 1, fc.tiles_y);
    return s.ty0 < s.ty1;
}

// Exact tile rejection for a shadow quad.  The inside test of a sample is, per edge,
// sign(rn(rn(ax*ey) - rn(ay*ex))) (obj/triangular.py:305-316).  rn(ax*ey) is monotone in the
// sample's x and rn(ay*ex) in its y, so over a tile the extreme of the rounded expression is
// attained at one of the four corner samples: if no corner is on the inner side of some
// edge, no sample of the tile is.  No margin is needed and no fragment can be lost.
// (tx, ty) are FRAME tile coordinates.
__device__ __forceinline__ bool quad_touches_tile(const QuadRec &q, int tx, int ty)
{
    const double xa = (double)(tx * TILE_W), xb = (double)(tx * TILE_W + TILE_W - 1);
    const double ya = (double)(ty * TILE_H), yb = (double)(ty * TILE_H + TILE_H - 1);
    const bool front = q.is_front != 0;
    auto outside = [&](const QuadEdge &e) {                 // no corner of the tile on the inner side of this edge
        const double px0 = (xa - e.sx) * e.ey, px1 = (xb - e.sx) * e.ey;
        const double py0 = (ya - e.sy) * e.ex, py1 = (yb - e.sy) * e.ex;
        const double c00 = px0 - py0, c10 = px1 - py0, c01 = px0 - py1, c11 = px1 - py1;
        return front ? !(c00 > 0 || c10 > 0 || c01 > 0 || c11 > 0) : !(c00 < 0 || c10 < 0 || c01 < 0 || c11 < 0);
    };
    // the first four edges are fetched together (nearly every quad has exactly four): one memory round
    // trip instead of one per edge on a path that is nothing but latency
    const int n = q.n;
    const QuadEdge e0 = q.e[0], e1 = q.e[1], e2 = q.e[2], e3 = q.e[3];
    const bool o0 = outside(e0), o1 = outside(e1), o2 = outside(e2), o3 = outside(e3);   // no short circuit: no branch per edge
    bool out = o0 || o1 || o2 || (n > 3 && o3);
    for (int i = 4; i < n && !out; ++i) out = outside(q.e[i]);
    return !out;
}

struct BinArgs {
    const TriRec *tris;
    const QuadRec *quads;
    Counters *ctr;
    uint32_t quad_cap;
    uint32_t *bin_count;          // [BIN_CLASSES * n_tiles] list cursors (= lengths once the frame's binning is done)
    uint32_t *items[BIN_CLASSES]; // [n_tiles * cap[c]] face index (classes 0, 1) / quad slot (class 2)
    uint32_t cap[BIN_CLASSES];
    uint2 *work;                  // (primitive, chunk of 64 tiles); primitive = face, or WORK_QUAD | quad
    uint32_t work_cap;
};

constexpr int TILE_STATS = 5;     // per-tile partial counters written by the tile kernel
constexpr int TILE_REC = 12;      // words per tile record: counters, list lengths, start / end time (10 ns ticks)
constexpr int QUAD_BATCH = 64;    // shadow quads staged in LDS per round of the tile kernel: one per lane of a wavefront
constexpr int BIN_SMALL = 4;      // triangles touching <= this many tiles are binned by their own lane
constexpr uint32_t WORK_QUAD = 0x80000000u;
constexpr int ORDER_CLASSES = 8;  // cost classes of the tile order (tile_class in kernels_tile.h); class bytes are 1..8, 0 = unknown
constexpr int ORDER_HEAD = 4;     // words before the order itself: [0] = tiles in class 1

struct PrimBox { int x0, x1, y0, y1; };

// Bin class of a (primitive, tile) pair: 0 small triangle pair (walked by a few lanes), 1 big
// triangle pair (one pixel per lane), 2 shadow quad.  A triangle whose fragments need the
// per-fragment clip test always goes the per-pixel way: there its corner data is wavefront-
// uniform (scalar loads), whereas lanes that each test their own triangle would need 36 more
// vector registers for it.  (tx, ty) are device-local tile coordinates.
__device__ __forceinline__ int pair_class(const FrameConst &fc, bool is_quad, bool clip, const PrimBox &pb, int tx, int ty)
{
    if (is_quad) return 2;
    if (clip) return 1;
    const int gx = tx * TILE_W, gy = tile_row_frame(fc, ty) * TILE_H;
    const int w = min(pb.x1, gx + TILE_W) - max(pb.x0, gx);
    const int h = min(min(pb.y1, gy + TILE_H), fc.band_y1) - max(max(pb.y0, gy), fc.band_y0);
    return w * h > BIG_PAIR_PX ? 1 : 0;
}

__device__ __forceinline__ void bin_store(const BinArgs &a, int cls, uint32_t tile, uint32_t pos, uint32_t id)
{
    // a list that runs over is truncated here; the tile kernel sees cursor > cap and reports it
    if (pos < a.cap[cls]) a.items[cls][(size_t)tile * a.cap[cls] + pos] = id;
}

__device__ __forceinline__ void bin_emit(const FrameConst &fc, const BinArgs &a, int cls, uint32_t id, int tx, int ty)
{
    const uint32_t tile = (uint32_t)ty * fc.tiles_x + tx;
    const uint32_t pos = atomicAdd(&a.bin_count[(uint32_t)cls * (uint32_t)(fc.tiles_x * fc.tiles_y) + tile], 1u);
    bin_store(a, cls, tile, pos, id);
}

// The same for a whole wavefront at once (every lane calls it, `want` says whether it has a
// pair to emit).  Neighbouring triangles of a mesh fall into the same tile, and a tile's
// cursor is one L2 location: lanes with the same bin are combined into ONE atomic and share
// out the returned range by rank, instead of queueing up to 64 deep on that location.  The
// groups are found first (ALU only), then every group leader issues its atomic in the same
// instruction: one memory round trip however many different bins the wavefront touches.
struct WaveEmit { unsigned long long mine; uint32_t base, tile; int leader, cls; };
// first half: the groups and the leaders' atomics (the returned value is not looked at here, so several
// of these can be in flight at once)
__device__ __forceinline__ WaveEmit bin_emit_wave_issue(const FrameConst &fc, const BinArgs &a, bool want, int cls, int tx, int ty)
{
    WaveEmit w;
    w.tile = (uint32_t)ty * fc.tiles_x + tx;
    w.cls = cls;
    const uint32_t bin = (uint32_t)cls * (uint32_t)(fc.tiles_x * fc.tiles_y) + w.tile;
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
    w.mine = mine;
    w.leader = mine ? __ffsll((long long)mine) - 1 : lane;
    w.base = 0;
    if (mine && lane == w.leader) w.base = atomicAdd(&a.bin_count[bin], (uint32_t)__popcll(mine));
    return w;
}
// second half: the group shares out the range its leader got
__device__ __forceinline__ void bin_emit_wave_store(const BinArgs &a, const WaveEmit &w, uint32_t id)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t base = (uint32_t)__shfl((int)w.base, w.leader);
    if (w.mine) bin_store(a, w.cls, w.tile, base + (uint32_t)__popcll(w.mine & ((1ull << lane) - 1ull)), id);
}
__device__ __forceinline__ void bin_emit_wave(const FrameConst &fc, const BinArgs &a, bool want, int cls,
                                              uint32_t id, int tx, int ty)
{
    const WaveEmit w = bin_emit_wave_issue(fc, a, want, cls, tx, ty);
    bin_emit_wave_store(a, w, id);
}

// A wavefront reserves work items for its lanes' large primitives with a single atomic (prefix
// sum of the lanes' chunk counts), then every lane writes its own.  Every lane must call it.
constexpr uint32_t WORK_NONE = 0xffffffffu;     // a reserved work item nobody needs after all (k_bin_work skips it)

// first half: the wavefront's reservation (one atomic; its result is not looked at here, so other requests can
// be in flight beside it).  Every lane must call it; returns false when no lane has anything.
struct WorkSlot { uint32_t base_raw, offset, shard; };
__device__ __forceinline__ bool reserve_work_items(const BinArgs &a, uint32_t chunks, WorkSlot &w)
{
    const int lane = threadIdx.x & (WAVE - 1);
    w.base_raw = 0; w.offset = 0;
    w.shard = (blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE) & (WORK_SHARDS - 1);     // neighbouring wavefronts, different cursors
    if (!__ballot(chunks != 0)) return false;
    uint32_t incl = chunks;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    const uint32_t wave_total = __shfl(incl, WAVE - 1);
    if (lane == 0) w.base_raw = atomicAdd(&a.ctr->work[w.shard].n, wave_total);
    w.offset = incl - chunks;
    return true;
}
// second half: every lane writes its own items (into its cursor's stretch of the list: work_cap / WORK_SHARDS items)
__device__ __forceinline__ void fill_work_items(const BinArgs &a, const WorkSlot &w, uint32_t prim, uint32_t chunks)
{
    const uint32_t per = a.work_cap / WORK_SHARDS;
    const uint32_t base = __shfl(w.base_raw, 0) + w.offset;
    for (uint32_t c = 0; c < chunks; ++c) {
        if (base + c < per) a.work[w.shard * per + base + c] = make_uint2(prim, c);
        else atomicOr(&a.ctr->overflow, 8u);
    }
}
__device__ __forceinline__ void push_work_items(const BinArgs &a, uint32_t prim, uint32_t chunks)
{
    WorkSlot w;
    if (reserve_work_items(a, chunks, w)) fill_work_items(a, w, prim, chunks);
}

// One lane per triangle (valid: the lane has one; face, pixel box and TF_CLIP given).  Small
// spans are emitted one "slot" at a time so that the wavefront can combine lanes that hit the
// same bin; a larger span becomes work items.  Every lane of the wavefront must call it.
__device__ __forceinline__ void bin_triangles(const FrameConst &fc, const BinArgs &a, bool valid, uint32_t face,
                                              const PrimBox &pb, bool clip)
{
    TileSpan sp = { 0, 0, 0, 0 };
    valid = valid && tile_span(fc, pb.x0, pb.x1, pb.y0, pb.y1, sp);
    const int ntiles = valid ? (sp.tx1 - sp.tx0) * (sp.ty1 - sp.ty0) : 0;
    const bool small = valid && ntiles <= BIN_SMALL;
    if (__ballot(small)) {
        // the (up to BIN_SMALL) cursors' atomics are all issued before the first returned value is used: a
        // device-scope returning atomic takes ~2 us, and one after the other they were the longest stretch
        // of a face's chain
        const int bw = max(sp.tx1 - sp.tx0, 1);
        WaveEmit w[BIN_SMALL];
        bool any[BIN_SMALL];
#pragma unroll
        for (int slot = 0; slot < BIN_SMALL; ++slot) {
            const bool want = small && slot < ntiles;
            any[slot] = __ballot(want) != 0;
            if (any[slot]) {
                const int tx = sp.tx0 + slot % bw, ty = sp.ty0 + slot / bw;
                w[slot] = bin_emit_wave_issue(fc, a, want, want ? pair_class(fc, false, clip, pb, tx, ty) : 0, tx, ty);
            }
        }
#pragma unroll
        for (int slot = 0; slot < BIN_SMALL; ++slot)
            if (any[slot]) bin_emit_wave_store(a, w[slot], face);
    }
    push_work_items(a, face, (valid && ntiles > BIN_SMALL) ? (uint32_t)(ntiles + WAVE - 1) / WAVE : 0u);
}

// Work items of a shadow quad (called by the lane that owns the quad's record).
__device__ __forceinline__ uint32_t quad_chunks(const FrameConst &fc, int x0, int x1, int y0, int y1)
{
    TileSpan sp;
    if (!tile_span(fc, x0, x1, y0, y1, sp)) return 0u;
    return (uint32_t)((sp.tx1 - sp.tx0) * (sp.ty1 - sp.ty0) + WAVE - 1) / WAVE;
}

// The work items: one wavefront per item, one tile per lane.
__device__ __forceinline__ void bin_work_body(const FrameConst &fc, const BinArgs &a, uint32_t block, uint32_t n_blocks)
{
    const int lane = threadIdx.x & (WAVE - 1);
    // the list is WORK_SHARDS stretches (reserve_work_items): lane s knows how many items stretch s holds and where
    // they start in the walk over all of them
    const uint32_t per = a.work_cap / WORK_SHARDS;
    const uint32_t mine = lane < WORK_SHARDS ? min(a.ctr->work[lane].n, per) : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < WORK_SHARDS; off <<= 1) {
        const uint32_t y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    const uint32_t n_work = __shfl(incl, WORK_SHARDS - 1);
    const uint32_t waves = n_blocks * (blockDim.x / WAVE);
    for (uint32_t w = block * (blockDim.x / WAVE) + threadIdx.x / WAVE; w < n_work; w += waves) {
        const int s = __ffsll((long long)__ballot(lane < WORK_SHARDS && incl > w)) - 1;      // the stretch item w lies in
        const uint32_t first = __shfl(incl - mine, s);
        const uint2 item = a.work[(uint32_t)s * per + (w - first)];
        if (item.x == WORK_NONE) continue;
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
        if (!is_quad || quad_touches_tile(a.quads[id], tx, tile_row_frame(fc, ty)))
            bin_emit(fc, a, pair_class(fc, is_quad, clip, pb, tx, ty), id, tx, ty);
    }
}

// The order the frame's tile kernel renders its tiles in: a counting sort of the class bytes the slot's
// previous frame left (kernels_tile.h, tile_class), heaviest class first.  ONE workgroup of 256, run beside
// the set-up of the faces.  Thread t owns a contiguous run of the tiles (whole 4-byte words of class bytes): it counts
// its tiles per class in two registers of 12-bit fields, the counts of all threads are scanned class by class through
// LDS (one wavefront per class, four threads' counts per lane), and a second walk over the same words sends every tile
// to  base of its class + tiles of that class in earlier threads + those met so far in this one.  Stable, a
// permutation by construction.  Any tile of unknown class (0: a first frame, a new grid) -> row-major order.
// (Round 2's version ranked with nine ballots per tile byte, a wavefront per quarter of the grid: 15 us for 8 160
// tiles, which for a mesh of a few thousand faces was the longest thing in the launch; this one takes a third.)
__device__ void order_tiles_block(const uint8_t *__restrict__ cls, uint32_t *__restrict__ order, int n_tiles)
{
    constexpr int NT = 256, NW = NT / WAVE, NC = ORDER_CLASSES + 1, FIELD = 12, LOW = 5;     // classes 0..4 in `lo`, 5..8 in `hi`
    __shared__ uint32_t s_cnt[NC][NT];
    __shared__ uint32_t s_tot[NC];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int per = ((n_tiles + NT - 1) / NT + 3) & ~3;       // tiles per thread, whole words
    const int t0 = tid * per, nwords = per >> 2;
    const uint32_t *__restrict__ words = reinterpret_cast<const uint32_t *>(cls);
    // the class bytes of word i of this thread's run; bytes past the last tile read as "no class"
    auto fetch = [&](int i) -> uint32_t {
        const int t = t0 + 4 * i;
        uint32_t v = t < n_tiles ? words[t >> 2] : 0xffffffffu;
        if (t < n_tiles && t + 4 > n_tiles) v |= 0xffffffffu << (8 * (n_tiles - t));
        return v;
    };
    auto bump = [&](unsigned long long &lo, unsigned long long &hi, uint32_t c) {
        lo += c < (uint32_t)LOW ? 1ull << (FIELD * c) : 0ull;
        hi += (c >= (uint32_t)LOW && c < (uint32_t)NC) ? 1ull << (FIELD * (c - LOW)) : 0ull;
    };
    auto field = [&](unsigned long long lo, unsigned long long hi, uint32_t c) -> uint32_t {
        return (uint32_t)((c < (uint32_t)LOW ? lo >> (FIELD * c) : hi >> (FIELD * (c - LOW))) & ((1u << FIELD) - 1u));
    };
    static_assert(LOW * FIELD <= 64 && (NC - LOW) * FIELD <= 64, "count fields");
    unsigned long long lo = 0, hi = 0;
    if (per < (1 << FIELD)) {                 // (a frame of more than a million tiles would need wider fields: row-major then)
#pragma unroll 1
        for (int i = 0; i < nwords; ++i) {
            const uint32_t w = fetch(i);
#pragma unroll
            for (int b = 0; b < 4; ++b) bump(lo, hi, (w >> (8 * b)) & 0xffu);
        }
    } else {
        lo = 1;                               // "a tile of unknown class"
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) s_cnt[k][tid] = field(lo, hi, (uint32_t)k);
    __syncthreads();
    // exclusive scan over the threads, one class per wavefront and turn: lane l owns threads 4l .. 4l + 3
    for (int k = wv; k < NC; k += NW) {
        const uint32_t a = s_cnt[k][4 * lane], b = s_cnt[k][4 * lane + 1], c = s_cnt[k][4 * lane + 2], d = s_cnt[k][4 * lane + 3];
        uint32_t incl = a + b + c + d;
        const uint32_t own = incl;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        const uint32_t before = incl - own;
        s_cnt[k][4 * lane] = before; s_cnt[k][4 * lane + 1] = before + a;
        s_cnt[k][4 * lane + 2] = before + a + b; s_cnt[k][4 * lane + 3] = before + a + b + c;
        if (lane == WAVE - 1) s_tot[k] = incl;
    }
    __syncthreads();
    const bool known = s_tot[0] == 0;
    uint32_t base[NC];
    base[0] = 0;
    {
        uint32_t at = 0;
#pragma unroll
        for (int k = 1; k < NC; ++k) { base[k] = at; at += s_tot[k]; }
    }
    if (tid == 0) order[0] = known ? s_tot[1] : 0u;
    // where this thread's next tile of each class goes
#pragma unroll
    for (int k = 1; k < NC; ++k) base[k] += s_cnt[k][tid];
    lo = hi = 0;
#pragma unroll 1
    for (int i = 0; i < nwords; ++i) {
        const uint32_t w = fetch(i);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t c = (w >> (8 * b)) & 0xffu;
            const uint32_t t = (uint32_t)(t0 + 4 * i + b);
            if (c <= (uint32_t)ORDER_CLASSES) {
                uint32_t at = t;
                if (known) {
                    uint32_t first = 0;
#pragma unroll
                    for (int k = 1; k < NC; ++k) first = c == (uint32_t)k ? base[k] : first;
                    at = first + field(lo, hi, c);
                }
                order[ORDER_HEAD + at] = t;
            }
            bump(lo, hi, c);
        }
    }
}

}  // namespace mr
