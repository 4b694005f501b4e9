// host_overlay.h -- the debug-camera frustum overlay's line lists, built on the host in plain C++.
//
// What frustums.py does with a few thousand NumPy calls per frame (obj/frustums.py:46-103, obj/line.py:6-16,
// obj/plane_intersection.py:59-86): clip the frustum's six faces against the viewing camera's planes, project
// them, walk their edges with the DDA, dash the hidden ones, and flatten everything into the statement lists
// k_overlay replays (kernels_overlay.h).  Every product here is the ascending-k fma chain that NumPy's BLAS
// uses for these shapes on the reference's stack (mr_host_matmul_chain), every other step
// the same IEEE operation NumPy performs element by element, so the lists are identical to the Python ones
// (tests/test_overlay.py compares them on random camera pairs).  The two matrix inverses of the recipe stay
// with NumPy (LAPACK): the caller passes the frustum's corners already un-projected.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace mr_host {

struct Vec4 { double v[4]; };

inline double chain4(const double *a, const double *b)
{
    double acc = a[0] * b[0];
    for (int k = 1; k < 4; ++k) acc = std::fma(a[k], b[k], acc);
    return acc;
}

// row vector times 4x4 (row-major) matrix, ascending-k chains
inline Vec4 row_times(const Vec4 &p, const double *m)
{
    Vec4 out;
    for (int j = 0; j < 4; ++j) {
        double acc = p.v[0] * m[j];
        for (int k = 1; k < 4; ++k) acc = std::fma(p.v[k], m[k * 4 + j], acc);
        out.v[j] = acc;
    }
    return out;
}

// Sutherland-Hodgman against each plane in turn (plane_intersection.clipping)
inline std::vector<Vec4> clip_polygon(std::vector<Vec4> poly, const double *planes, int n_planes)
{
    for (int pl = 0; pl < n_planes && !poly.empty(); ++pl) {
        const double *P = planes + pl * 4;
        const size_t n = poly.size();
        std::vector<double> dist(n);
        for (size_t i = 0; i < n; ++i) dist[i] = chain4(P, poly[i].v);
        std::vector<Vec4> kept;
        for (size_t i = 0; i < n; ++i) {
            const size_t j = (i + 1) % n;
            const bool cur_in = dist[i] >= 0, next_in = dist[j] >= 0;
            if (cur_in) kept.push_back(poly[i]);
            if (cur_in != next_in) {
                double dir[4];
                for (int k = 0; k < 4; ++k) dir[k] = poly[i].v[k] - poly[j].v[k];     // current - following
                const double den = chain4(P, dir);
                if (!(std::fabs(den) < 1e-10)) {
                    const double weight = -dist[j] / den;
                    if (0 <= weight && weight <= 1) {
                        Vec4 hit;
                        for (int k = 0; k < 4; ++k) hit.v[k] = poly[j].v[k] + weight * dir[k];
                        kept.push_back(hit);
                    }
                }
            }
        }
        poly.swap(kept);
    }
    return poly;
}

struct OverlayLists {
    std::vector<int32_t> seg_first, seg_count;
    std::vector<int32_t> target[5], next[5];
    std::vector<double> z;
    std::vector<int32_t> touched;
    std::vector<uint8_t> tile_mask;           // (want_tiles) the 16x16-pixel tiles that hold a target, row-major
};

// corners: the frustum's eight corners (8 x 4, already divided by w); faces: 6 x 4 corner indices
inline void build_overlay_lists(const double *corners, const int32_t *faces, const double *planes, const double *mvp,
                                const double *viewport, double near_, double far_, bool camera_inside,
                                int height, int width, int dash, OverlayLists &out, bool want_links = true,
                                bool want_tiles = false)
{
    const int tiles_x = (width + 15) / 16, tiles_y = (height + 15) / 16;
    if (want_tiles) out.tile_mask.assign((size_t)tiles_x * tiles_y, 0);
    const double near_far = 2 * near_ * far_;
    const double f_plus_n = far_ + near_, f_minus_n = far_ - near_;
    std::vector<int64_t> rows, cols;            // raw (unwrapped) indices of the kept points
    rows.reserve(16384); cols.reserve(16384);
    for (int f = 0; f < 6; ++f) {
        std::vector<Vec4> quad(4);
        for (int c = 0; c < 4; ++c)
            for (int k = 0; k < 4; ++k) quad[c].v[k] = corners[faces[f * 4 + c] * 4 + k];
        std::vector<Vec4> poly = clip_polygon(quad, planes, 6);
        if (poly.size() < 3) continue;
        for (auto &p : poly) {
            p = row_times(p, mvp);
            const double w = p.v[3];
            for (int k = 0; k < 4; ++k) p.v[k] = p.v[k] / w;
            p = row_times(p, viewport);
        }
        // facing: z of cross(b - a, c - a) over the xyz of the first three vertices
        const double e0x = poly[1].v[0] - poly[0].v[0], e0y = poly[1].v[1] - poly[0].v[1];
        const double e1x = poly[2].v[0] - poly[0].v[0], e1y = poly[2].v[1] - poly[0].v[1];
        const double facing = e0x * e1y - e0y * e1x;
        for (auto &p : poly) p.v[2] = near_far / (f_plus_n - p.v[2] * f_minus_n);
        const bool dashed = facing > 0 && !camera_inside;
        const size_t count = poly.size();
        for (size_t i = 0; i < count; ++i) {
            const Vec4 *start = &poly[i], *end = &poly[(i + 1) % count];
            double delta[4];
            for (int k = 0; k < 4; ++k) delta[k] = end->v[k] - start->v[k];
            if (delta[0] > 0) {                          // always walked towards decreasing x
                std::swap(start, end);
                for (int k = 0; k < 4; ++k) delta[k] = end->v[k] - start->v[k];
            }
            const double steps = std::max(std::fabs(delta[0]), std::fabs(delta[1]));
            long long n_pts;
            double inc[4] = { 0, 0, 0, 0 };
            if (steps == 0) n_pts = 1;
            else { n_pts = (long long)steps; for (int k = 0; k < 4; ++k) inc[k] = delta[k] / steps; }
            const int32_t first = (int32_t)out.z.size();
            for (long long s = 0; s < n_pts; ++s) {
                if (dashed && !((s / dash) & 1)) continue;
                // start + s * (delta / steps), element by element (steps == 0: the start point itself)
                const double x = steps == 0 ? start->v[0] : start->v[0] + (double)s * inc[0];
                const double y = steps == 0 ? start->v[1] : start->v[1] + (double)s * inc[1];
                const double zz = steps == 0 ? start->v[2] : start->v[2] + (double)s * inc[2];
                const int64_t col = (int64_t)(int32_t)x - 1, row = (int64_t)(int32_t)y - 1;
                if (!(row >= -height && row < height && col >= -width && col < width)) continue;
                rows.push_back(row); cols.push_back(col); out.z.push_back(zz);
            }
            const int32_t n_kept = (int32_t)out.z.size() - first;
            if (n_kept > 0) { out.seg_first.push_back(first); out.seg_count.push_back(n_kept); }
        }
    }
    const size_t n = out.z.size();
    for (int k = 0; k < 5; ++k) { out.target[k].assign(n, 0); out.next[k].assign(n, -1); }
    auto clampi = [](int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); };
    for (size_t i = 0; i < n; ++i) {
        const int64_t r = rows[i], c = cols[i];
        const int64_t wr = ((r % height) + height) % height, wc = ((c % width) + width) % width;   // the centre index wraps like Python's
        const int64_t rm = clampi(r - 1, 0, height - 1), rp = clampi(r + 1, 0, height - 1);           // neighbours: clipped from the RAW index
        const int64_t cm = clampi(c - 1, 0, width - 1), cp = clampi(c + 1, 0, width - 1);
        out.target[0][i] = (int32_t)(wr * width + wc);
        out.target[1][i] = (int32_t)(rm * width + wc);
        out.target[2][i] = (int32_t)(wr * width + cm);
        out.target[3][i] = (int32_t)(rp * width + wc);
        out.target[4][i] = (int32_t)(wr * width + cp);
        if (want_tiles) {
            uint8_t *m = out.tile_mask.data();
            m[(wr >> 4) * tiles_x + (wc >> 4)] = 1; m[(rm >> 4) * tiles_x + (wc >> 4)] = 1; m[(rp >> 4) * tiles_x + (wc >> 4)] = 1;
            m[(wr >> 4) * tiles_x + (cm >> 4)] = 1; m[(wr >> 4) * tiles_x + (cp >> 4)] = 1;
        }
    }
    if (!want_links) return;                    // (the device needs neither the links nor the list of touched pixels)
    // next point (later in the same segment) with the same target: walk the segment backwards with a small
    // open-addressing table of (target -> latest point seen)
    std::vector<int32_t> keys, vals;
    for (size_t s = 0; s < out.seg_first.size(); ++s) {
        const int32_t first = out.seg_first[s], cnt = out.seg_count[s];
        size_t cap = 16;
        while (cap < (size_t)cnt * 2) cap <<= 1;
        for (int k = 0; k < 5; ++k) {
            keys.assign(cap, -1);
            vals.resize(cap);
            const int32_t *t = out.target[k].data();
            for (int32_t i = first + cnt - 1; i >= first; --i) {
                size_t h = ((uint32_t)t[i] * 2654435761u) & (cap - 1);
                while (keys[h] != -1 && keys[h] != t[i]) h = (h + 1) & (cap - 1);
                out.next[k][(size_t)i] = keys[h] == t[i] ? vals[h] : -1;
                keys[h] = t[i]; vals[h] = i;
            }
        }
    }
    // every pixel any statement writes, once, ascending: a bitmap over the frame
    std::vector<uint64_t> bits(((size_t)height * width + 63) / 64, 0);
    for (int k = 0; k < 5; ++k)
        for (int32_t t : out.target[k]) bits[(size_t)t >> 6] |= 1ull << (t & 63);
    out.touched.clear();
    for (size_t w = 0; w < bits.size(); ++w)
        for (uint64_t m = bits[w]; m; m &= m - 1) out.touched.push_back((int32_t)(w * 64 + (size_t)__builtin_ctzll(m)));
}

// The overlay's targets as SLOTS of the list of touched pixels (for a frame assembled from several devices: the
// overlay is then replayed on a compact copy of the touched pixels' state, gathered from the devices that own
// them): slot_of[k][p] = slot of target k of point p, touched[slot] = pixel, slots in order of first appearance.
// `stamp` / `value` are frame-sized work arrays kept between calls (generation stamps instead of clears).
struct OverlaySlotWork { std::vector<uint32_t> stamp, value; uint32_t generation = 0; };
inline void build_overlay_slots(const int32_t *target /* (5, n_points) */, size_t n_points, size_t n_pixels, OverlaySlotWork &ws,
                                std::vector<int32_t> &slot_of, std::vector<int32_t> &touched)
{
    if (ws.stamp.size() < n_pixels) { ws.stamp.assign(n_pixels, 0u); ws.value.assign(n_pixels, 0u); ws.generation = 0; }
    if (ws.generation > 0xfffffff0u) { std::fill(ws.stamp.begin(), ws.stamp.end(), 0u); ws.generation = 0; }
    const uint32_t gen = ++ws.generation;
    slot_of.resize(5 * n_points);
    touched.clear();
    for (size_t i = 0; i < 5 * n_points; ++i) {
        const size_t pixel = (size_t)target[i];
        if (ws.stamp[pixel] != gen) { ws.stamp[pixel] = gen; ws.value[pixel] = (uint32_t)touched.size(); touched.push_back((int32_t)pixel); }
        slot_of[i] = (int32_t)ws.value[pixel];
    }
}

}  // namespace mr_host
