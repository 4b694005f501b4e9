// kernels_geometry.h -- per-vertex, per-face and per-silhouette-edge work of one frame.
//
//   k_vertex      obj/triangular.py:36-45  (once per unique vertex instead of per face corner);
//                 k_vertex_mfma does the same on the matrix cores (default), bit-identical
//   k_tri_setup   obj/triangular.py:47-78, obj/core.py:127-136, obj/transformation.py:12-43
//   k_tri_count   how many fragments of a face survive coverage + clip (decides NumPy's
//                 dot-vs-gemv rounding of z, and the CLIPPED status)
//   k_silhouette  obj/triangular.py:286-302 + obj/core.py:610-622 + obj/plane_intersection.py:59-86
//                 + obj/triangular.py:320-340 (extrusion, clip, projection, plane, box)
#pragma once

#include "kernels_bin.h"

namespace mr {

// unit normal of the world-space triangle in the vertices' own dtype (obj/core.py:127-130),
// dotted with light.position (obj/triangular.py:295)
__device__ __forceinline__ bool faces_light(const FrameConst &fc, const double *a, const double *b,
                                            const double *c, bool verts_f32)
{
    double n[3];
    if (verts_f32) {
        float e0[3], e1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            e0[j] = (float)b[j] - (float)a[j];
            e1[j] = (float)c[j] - (float)a[j];
        }
        float cr[3] = { e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2],
                        e0[0] * e1[1] - e0[1] * e1[0] };
        float l = sqrtf((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]);
        if (l == 0) l = 1;
#pragma unroll
        for (int j = 0; j < 3; ++j) n[j] = (double)(cr[j] / l);
    } else {
        double e0[3], e1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { e0[j] = b[j] - a[j]; e1[j] = c[j] - a[j]; }
        double cr[3] = { e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2],
                         e0[0] * e1[1] - e0[1] * e1[0] };
        normalize3(cr, n);
    }
    return chain3(n[0], n[1], n[2], fc.light_pos[0], fc.light_pos[1], fc.light_pos[2]) > 0;
}

// Light-facing flag of every face (input of the silhouette search), thread per face.  It only
// needs the static mesh and the light, so it rides along with the vertex transform as extra
// workgroups of the frame's first launch instead of lengthening k_tri_setup.
__device__ __forceinline__ void lit_body(const FrameConst &fc, const int32_t *__restrict__ faces,
                                         const uint8_t *__restrict__ face_flags, const double *__restrict__ verts,
                                         uint8_t *__restrict__ lit, uint32_t block)
{
    const int f = (int)(block * blockDim.x + threadIdx.x);
    if (f >= fc.n_faces) return;
    const int32_t *fcx = faces + (size_t)f * 12;
    lit[f] = faces_light(fc, verts + (size_t)fcx[0] * 4, verts + (size_t)fcx[4] * 4, verts + (size_t)fcx[8] * 4,
                         (face_flags[f] & FF_VERTS_F32) != 0) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
k_vertex(const FrameConst fc, const double *__restrict__ verts, VertexOut *__restrict__ out,
         VertexClip *__restrict__ out_clip, Counters *__restrict__ ctr, const int32_t *__restrict__ faces,
         const uint8_t *__restrict__ face_flags, uint8_t *__restrict__ lit, uint32_t vertex_blocks)
{
    if (blockIdx.x >= vertex_blocks) { lit_body(fc, faces, face_flags, verts, lit, blockIdx.x - vertex_blocks); return; }
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *ctr = Counters{};      // first kernel of the frame: every later one is stream-ordered after it
    if (i >= fc.n_vertices) return;
    double v[4] = { verts[i * 4 + 0], verts[i * 4 + 1], verts[i * 4 + 2], verts[i * 4 + 3] };
    VertexOut o;
    VertexClip oc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        oc.clip[j] = row_times_col(v, fc.mvp, j);
        oc.clipd[j] = row_times_col(v, fc.debug_mvp, j);
    }
    double depth = 1.0 / oc.clip[3];
    double ndc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ndc[j] = oc.clip[j] * depth;
    o.sx = row_times_col(ndc, fc.viewport, 0);
    o.sy = row_times_col(ndc, fc.viewport, 1);
    o.sz = row_times_col(ndc, fc.viewport, 2);
    o.depth = depth;
    o.zlin = linearize_z(fc, o.sz);
    // Strictly inside both cameras' clip volumes with a relative margin of 1e-12.  A fragment's
    // clip coordinates are a non-negative combination of the corners' (weights u*dp/wc, all
    // >= 0 when u,v,w >= 0 and every w > 0), evaluated with a few ulp (1e-16) of rounding, so
    // when all three corners carry this flag the strict test of obj/triangular.py:85-87 cannot
    // fail for any fragment of the face and need not be evaluated.
    const double k = 1.0 - 1e-12;
    const double wl = oc.clip[3] * k, wd = oc.clipd[3] * k;
    o.safe = (fabs(oc.clip[0]) < wl && fabs(oc.clip[1]) < wl && fabs(oc.clip[2]) < wl &&
              fabs(oc.clipd[0]) < wd && fabs(oc.clipd[1]) < wd && fabs(oc.clipd[2]) < wd) ? 1 : 0;
    o.pad = 0;
    out[i] = o;
    out_clip[i] = oc;
}

// The same vertex stage on the matrix cores.  The two products of obj/triangular.py:36-45
// (v @ [MVP | debug MVP] and ndc @ viewport) are dense 16x4 by 4x16 contractions per 16
// vertices, i.e. exactly one v_mfma_f64_16x16x4_f64 each.  Measured on MI355X
// (tools/micro/mfma_vertex_check.hip, 8.4 M outputs): that instruction accumulates k = 0..3 in
// order with one rounding per step, bit-identical to the ascending fma chain the reference's
// BLAS uses, so the result is the same VertexOut / VertexClip as k_vertex's, bit for bit.
// One wavefront = 16 vertices.  Operand layout (cdna_hip_programming.md section 3): A lane l holds
// A[row l%16][k l/16], B lane l holds B[k l/16][col l%16], D register i of lane l holds
// D[row (l>>4) + 4i][col l&15]; the D -> A re-layout between the two products goes through LDS.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
k_vertex_mfma(const FrameConst fc, const double *__restrict__ verts, VertexOut *__restrict__ out,
              VertexClip *__restrict__ out_clip, Counters *__restrict__ ctr, const int32_t *__restrict__ faces,
              const uint8_t *__restrict__ face_flags, uint8_t *__restrict__ lit, uint32_t vertex_blocks)
{
    if (blockIdx.x >= vertex_blocks) { lit_body(fc, faces, face_flags, verts, lit, blockIdx.x - vertex_blocks); return; }
    __shared__ double s_clip[4][16][8];     // per wavefront: [vertex][MVP x,y,z,w | debug x,y,z,w]
    __shared__ double s_scr[4][16][4];      // per wavefront: [vertex][screen x, y, z]
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    if (blockIdx.x == 0 && threadIdx.x == 0) *ctr = Counters{};   // first kernel of the frame
    const int base = (blockIdx.x * (blockDim.x / WAVE) + wv) * 16;
    const int row = lane & 15, k = lane >> 4;
    const bool have = base + row < fc.n_vertices;

    // clip = v @ [MVP | debug MVP]
    const double a = have ? verts[(size_t)(base + row) * 4 + k] : 0.0;
    const double b = row < 4 ? fc.mvp[k * 4 + row] : (row < 8 ? fc.debug_mvp[k * 4 + row - 4] : 0.0);
    mfma_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    if (row < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s_clip[wv][k + 4 * i][row] = acc[i];
    }
    __syncthreads();

    // ndc = clip * (1 / clip.w);  screen = ndc @ viewport
    const double depth = 1.0 / s_clip[wv][row][3];
    const double ndc = s_clip[wv][row][k] * depth;
    const double b2 = row < 4 ? fc.viewport[k * 4 + row] : 0.0;
    mfma_d4 acc2 = { 0.0, 0.0, 0.0, 0.0 };
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ndc, b2, acc2, 0, 0, 0);
    if (row < 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s_scr[wv][k + 4 * i][row] = acc2[i];
    }
    __syncthreads();

    if (lane < 16 && base + lane < fc.n_vertices) {
        const int v = lane;
        VertexOut o;
        VertexClip oc;
#pragma unroll
        for (int j = 0; j < 4; ++j) { oc.clip[j] = s_clip[wv][v][j]; oc.clipd[j] = s_clip[wv][v][4 + j]; }
        o.sx = s_scr[wv][v][0]; o.sy = s_scr[wv][v][1]; o.sz = s_scr[wv][v][2];
        o.depth = 1.0 / oc.clip[3];
        o.zlin = linearize_z(fc, o.sz);
        const double kk = 1.0 - 1e-12;                  // "safely inside" flag: see k_vertex
        const double wl = oc.clip[3] * kk, wd = oc.clipd[3] * kk;
        o.safe = (fabs(oc.clip[0]) < wl && fabs(oc.clip[1]) < wl && fabs(oc.clip[2]) < wl &&
                  fabs(oc.clipd[0]) < wd && fabs(oc.clipd[1]) < wd && fabs(oc.clipd[2]) < wd) ? 1 : 0;
        o.pad = 0;
        out[base + v] = o;
        out_clip[base + v] = oc;
    }
}

// Survivors of coverage + clip among the first `limit` samples of a triangle's pixel box, as
// seen by one lane walking the box sample by sample (stops at two).

__device__ __forceinline__ bool sample_survives(const FrameConst &fc, const TriRec &t, const TriClip *clips,
                                                int px, int py, bool &covered_in_band)
{
    const bool single = (t.flags & TF_SINGLE_BOX) != 0;
    float u, v, w;
    tri_bary(t, (double)px, (double)py, single, u, v, w);
    bool ok = u >= 0 && v >= 0 && w >= 0;
    covered_in_band = ok && py >= fc.band_y0 && py < fc.band_y1;
    if (ok && (t.flags & TF_CLIP)) {
        const TriClip &c = clips[t.face];
        double p[3];
        persp_bary(c.dp, u, v, w, single, p);
        ok = inside_clip(p, c.clip) && (fc.same_clip || inside_clip(p, c.clipd));
    }
    return ok;
}

__device__ __forceinline__ void count_finish(TriRec *tris, uint8_t *status, int f, uint32_t flags, int found)
{
    if (found == 0) {
        // The face was binned before this verdict (kernels_bin.h) and stays listed: the
        // visibility kernel finds no surviving fragment for it and counts its covered ones
        // (the fragment count is taken before the clip, obj/triangular.py:78).
        status[f] = FACE_CLIPPED;
    } else if (found == 1) {
        tris[f].flags = flags | TF_SINGLE_Z;
    }
}

constexpr int COUNT_SMALL_BOX = 32;   // pixel boxes up to this size are walked by a single lane

// One face: status, TriRec / TriClip, light-facing flag.  Returns bit 0 = the face goes on to
// the visibility kernel, bit 1 = its survivor count is left to k_tri_count; `covered` receives
// the fragments of a face settled as CLIPPED right here.
__device__ __forceinline__ int tri_setup_one(const FrameConst &fc, int f, const int32_t *__restrict__ faces,
                                             const uint8_t *__restrict__ face_flags,
                                             const VertexOut *__restrict__ vout, const VertexClip *__restrict__ vclip,
                                             TriRec *__restrict__ tris, TriClip *__restrict__ clips,
                                             uint8_t *__restrict__ status, unsigned int &covered, PrimBox &pb, bool &clip)
{
    const int32_t *fcx = faces + (size_t)f * 12;
    const int va = fcx[0], vb = fcx[4], vc = fcx[8];
    const uint8_t ff = face_flags[f];

    const VertexOut A = vout[va], B = vout[vb], C = vout[vc];

    // obj/triangular.py:47-48: z of the normalised screen-space normal
    if (fc.backface_culling) {
        double e0[3] = { B.sx - A.sx, B.sy - A.sy, B.sz - A.sz };
        double e1[3] = { C.sx - A.sx, C.sy - A.sy, C.sz - A.sz };
        double n[3] = { e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2],
                        e0[0] * e1[1] - e0[1] * e1[0] };
        // The test is on the z of the NORMALISED normal, n[2] / |n|.  With |n| and |n[2]| well inside
        // the exponent range the quotient can neither overflow nor underflow to -0, so it is negative
        // exactly when n[2] is: no IEEE square root and division then (~60 instructions per face).
        const double l2 = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2];
        bool cull;
        if (l2 > 1e-280 && l2 < 1e280 && fabs(n[2]) >= 1e-160) {
            cull = n[2] < 0;
        } else {
            double u[3];
            normalize3(n, u);
            cull = u[2] < 0;
        }
        if (cull) { status[f] = FACE_BACK_FACE_CULLING; return 0; }
    }

    TriRec t;
    double xs[3] = { A.sx, B.sx, C.sx }, ys[3] = { A.sy, B.sy, C.sy };
    int bx0, bx1, by0, by1;
    if (!bound_box(xs, ys, 3, fc.width, fc.height, bx0, bx1, by0, by1)) {
        status[f] = FACE_EMPTY_Z;
        return 0;
    }
    // a device that renders a band of rows drops the faces whose pixel box misses the band right here
    if (by1 <= fc.band_y0 || by0 >= fc.band_y1) { status[f] = FACE_CLIPPED; return 0; }
    t.x0 = (int16_t)bx0; t.x1 = (int16_t)bx1; t.y0 = (int16_t)by0; t.y1 = (int16_t)by1;
    pb = { bx0, bx1, by0, by1 };
    t.ax = A.sx; t.ay = A.sy;
    t.v0x = B.sx - A.sx; t.v0y = B.sy - A.sy;
    t.v1x = C.sx - A.sx; t.v1y = C.sy - A.sy;
    t.d00 = (float)chain2(t.v0x, t.v0y, t.v0x, t.v0y);
    t.d01 = (float)chain2(t.v0x, t.v0y, t.v1x, t.v1y);
    t.d11 = (float)chain2(t.v1x, t.v1y, t.v1x, t.v1y);
    float den = t.d00 * t.d11 - t.d01 * t.d01;
    if (den == 0) { status[f] = FACE_EMPTY_B; return 0; }
    t.inv_den = 1.0f / den;
    t.zl0 = A.zlin; t.zl1 = B.zlin; t.zl2 = C.zlin;
    t.pad[0] = t.pad[1] = 0;
    long long box = (long long)(bx1 - bx0) * (long long)(by1 - by0);
    const bool need_clip = (ff & FF_CLIP) && !(A.safe && B.safe && C.safe);
    clip = need_clip;
    t.flags = (need_clip ? TF_CLIP : 0u) | (box == 1 ? TF_SINGLE_BOX : 0u) | ((uint32_t)ff << 8);   // bits 8-15: face flags, for k_shade
    t.face = f;
    if (box <= 0) { status[f] = FACE_CLIPPED; return 0; }    // no sample inside the box
    status[f] = FACE_OK;

    // How many fragments survive coverage + clip (0 -> CLIPPED, 1 -> z is a dot, TF_SINGLE_Z)?
    // A small pixel box that needs no clip test is settled right here, by this lane, over the
    // WHOLE frame (not just this device's band); the rest is left to k_tri_count.
    const bool count_here = !need_clip && box <= COUNT_SMALL_BOX;
    if (count_here) {
        const int bw = bx1 - bx0;
        int found = 0;
        for (int idx = 0; idx < (int)box && found < 2; ++idx) {
            bool cov;
            found += sample_survives(fc, t, nullptr, bx0 + idx % bw, by0 + idx / bw, cov) ? 1 : 0;
            covered += cov ? 1u : 0u;
        }
        if (found == 0) {
            status[f] = FACE_CLIPPED;              // never reaches the visibility kernel: its fragments are counted here
            return 0;
        }
        covered = 0;
        if (found == 1) t.flags |= TF_SINGLE_Z;
    }
    tris[f] = t;
    TriClip &cl = clips[f];
    cl.dp[0] = A.depth; cl.dp[1] = B.depth; cl.dp[2] = C.depth;
    if (need_clip) {
        const VertexClip ca = vclip[va], cb = vclip[vb], cc = vclip[vc];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cl.clip[0][j] = ca.clip[j]; cl.clip[1][j] = cb.clip[j]; cl.clip[2][j] = cc.clip[j];
            cl.clipd[0][j] = ca.clipd[j]; cl.clipd[1][j] = cb.clipd[j]; cl.clipd[2][j] = cc.clipd[j];
        }
    }
    return count_here ? 1 : 3;
}

constexpr int SETUP_BLOCK = 512;

// The two face lists (valid_list: faces to bin; count_list: faces k_tri_count still has to
// settle) are appended to with ONE atomic per workgroup and list: the frame's counters share a
// cache line, and same-line atomics retire at only ~0.3 per ns on MI355X
// (tools/micro/atomic_bench.hip), so per-wavefront appends alone cost more than the set-up.
__device__ __forceinline__ void
tri_setup_block(const FrameConst &fc, const int32_t *__restrict__ faces, const uint8_t *__restrict__ face_flags,
                const VertexOut *__restrict__ vout, const VertexClip *__restrict__ vclip, TriRec *__restrict__ tris,
                TriClip *__restrict__ clips, uint8_t *__restrict__ status, uint32_t *__restrict__ valid_list,
                uint32_t *__restrict__ count_list, Counters *__restrict__ ctr, const BinArgs &bins, uint32_t block)
{
    constexpr int NW = SETUP_BLOCK / WAVE;
    __shared__ uint32_t s_valid[NW], s_count[NW], s_covered;
    const int f = block * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    if (threadIdx.x == 0) s_covered = 0;
    unsigned int covered = 0;
    PrimBox pb = { 0, 0, 0, 0 };
    bool clip = false;
    const int r = f < fc.n_faces ? tri_setup_one(fc, f, faces, face_flags, vout, vclip, tris, clips, status, covered, pb, clip)
                                 : 0;
    // count pass of the binning for the faces that go on (kernels_bin.h)
    bin_triangles<false>(fc, bins, (r & 1) != 0, (uint32_t)f, pb, clip);
    const unsigned long long bv = __ballot(r & 1), bc = __ballot(r & 2);
    if (lane == 0) { s_valid[wv] = (uint32_t)__popcll(bv); s_count[wv] = (uint32_t)__popcll(bc); }
    __syncthreads();
    if (covered) atomicAdd(&s_covered, covered);
    if (threadIdx.x == 0) {
        uint32_t nv = 0, nc = 0;
        for (int w = 0; w < NW; ++w) { nv += s_valid[w]; nc += s_count[w]; }
        uint32_t bvb = nv ? atomicAdd(&ctr->n_valid_tris, nv) : 0u;
        uint32_t bcb = nc ? atomicAdd(&ctr->n_count, nc) : 0u;
        for (int w = 0; w < NW; ++w) {
            const uint32_t a = s_valid[w], b = s_count[w];
            s_valid[w] = bvb; s_count[w] = bcb;
            bvb += a; bcb += b;
        }
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    if (r & 1) valid_list[s_valid[wv] + (uint32_t)__popcll(bv & below)] = (uint32_t)f;
    if (r & 2) count_list[s_count[wv] + (uint32_t)__popcll(bc & below)] = (uint32_t)f;
    if (threadIdx.x == 0 && s_covered) atomicAdd(&ctr->frag_tri, (unsigned long long)s_covered);
}

// Counts, per set-up triangle, the fragments that survive coverage + clip over the WHOLE frame
// (not just this device's band), stopping as soon as two are found:
//   0 -> the reference returns CLIPPED for the face; 1 -> z = bar @ zlin is a dot (TF_SINGLE_Z).
// Only the faces k_tri_setup could not settle itself arrive here (pixel boxes over 32 samples,
// or a per-fragment clip test).
__device__ __forceinline__ void
tri_count_body(const FrameConst &fc, const uint32_t *__restrict__ count_list, TriRec *__restrict__ tris,
               const TriClip *__restrict__ clips, uint8_t *__restrict__ status, Counters *__restrict__ ctr,
               uint32_t block, uint32_t n_blocks)
{
    // one wavefront per listed face, 64 samples per step, starting at the chunk that holds the
    // centroid (a well-shaped triangle is settled by that chunk alone)
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t n_count = ctr->n_count;
    const uint32_t waves = n_blocks * (blockDim.x / WAVE);
    for (uint32_t i = block * (blockDim.x / WAVE) + threadIdx.x / WAVE; i < n_count; i += waves) {
        const int fb = (int)count_list[i];
        const TriRec tb = tris[fb];
        const int w = tb.x1 - tb.x0;
        const long long n = (long long)w * (tb.y1 - tb.y0);
        const long long chunks = (n + WAVE - 1) / WAVE;
        int cx = (int)(tb.ax + (tb.v0x + tb.v1x) * (1.0 / 3.0)), cy = (int)(tb.ay + (tb.v0y + tb.v1y) * (1.0 / 3.0));
        cx = min(max(cx, (int)tb.x0), tb.x1 - 1); cy = min(max(cy, (int)tb.y0), tb.y1 - 1);
        const long long first = ((long long)(cy - tb.y0) * w + (cx - tb.x0)) / WAVE;
        int found = 0;
        for (long long c = 0; c < chunks && found < 2; ++c) {
            const long long idx = ((first + c) % chunks) * WAVE + lane;
            bool cov = false, ok = false;
            if (idx < n) ok = sample_survives(fc, tb, clips, tb.x0 + (int)(idx % w), tb.y0 + (int)(idx / w), cov);
            found += __popcll(__ballot(ok));
        }
        if (lane == 0) count_finish(tris, status, fb, tb.flags, found);
    }
}

// Per-face result of the reference's lit pass (obj/triangular.py:101-112 with a stencil
// buffer): a face that reached the depth stage is "rendered" when at least one of its fragments
// has z <= the final z-buffer value (i.e. equals it) where stencil == 0, else EMPTY_Z.  Runs
// after the visibility kernels, only when MR_FRAME_FACE_STATUS is set (obj/core.py:625-636
// prints the histogram).  Same work split as k_tri_count; stops at the first such fragment.
__device__ __forceinline__ bool sample_is_drawn(const FrameConst &fc, const TriRec &t, const TriClip *clips,
                                                const double *zbuf, const int32_t *stencil, int px, int py)
{
    bool cov;
    if (!sample_survives(fc, t, clips, px, py, cov)) return false;
    float u, v, w;
    tri_bary(t, (double)px, (double)py, (t.flags & TF_SINGLE_BOX) != 0, u, v, w);
    const double z = rows_dot3((t.flags & TF_SINGLE_Z) != 0, (double)u, (double)v, (double)w, t.zl0, t.zl1, t.zl2);
    const size_t at = (size_t)py * fc.width + px;
    const bool pass = fc.system == 1 ? (zbuf[at] >= z) : (zbuf[at] <= z);
    return pass && (int16_t)stencil[at] == 0;
}

__global__ void __launch_bounds__(256)
k_face_status(const FrameConst fc, const uint32_t *__restrict__ valid_list, const TriRec *__restrict__ tris,
              const TriClip *__restrict__ clips, const double *__restrict__ zbuf, const int32_t *__restrict__ stencil,
              uint8_t *__restrict__ status, const Counters *__restrict__ ctr)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    bool valid = i < ctr->n_valid_tris;
    const int f = valid ? (int)valid_list[i] : 0;
    valid = valid && status[f] == FACE_OK;
    TriRec t = {};
    if (valid) t = tris[f];
    const int bw = t.x1 - t.x0, bh = t.y1 - t.y0;
    const int total = valid ? bw * bh : 0;
    if (valid && total <= COUNT_SMALL_BOX) {
        bool drawn = false;
        for (int idx = 0; idx < total && !drawn; ++idx)
            drawn = sample_is_drawn(fc, t, clips, zbuf, stencil, t.x0 + idx % bw, t.y0 + idx / bw);
        if (!drawn) status[f] = FACE_EMPTY_Z;
    }
    unsigned long long big = __ballot(valid && total > COUNT_SMALL_BOX);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int fb = __shfl(f, src);
        const TriRec tb = tris[fb];
        const int w = tb.x1 - tb.x0;
        const long long n = (long long)w * (tb.y1 - tb.y0);
        bool any = false;
        for (long long base = 0; base < n && !any; base += WAVE) {
            const long long idx = base + lane;
            const bool d = idx < n && sample_is_drawn(fc, tb, clips, zbuf, stencil, tb.x0 + (int)(idx % w), tb.y0 + (int)(idx / w));
            any = __ballot(d) != 0;
        }
        if (lane == 0 && !any) status[fb] = FACE_EMPTY_Z;
    }
}

// plane . point >= 0 (obj/plane_intersection.py:39-40), a 1-D dot of length 4
__device__ __forceinline__ double plane_dot(const double *P, const double *q)
{
    return chain4(P[0], P[1], P[2], P[3], q[0], q[1], q[2], q[3]);
}

// One thread per unique undirected edge of the scene.  An edge is on the silhouette when an
// odd number of its incident light-facing faces toggled it; it keeps the orientation of the
// last such face in face order (set add/discard semantics of obj/triangular.py:294-302).
__device__ __forceinline__ void
silhouette_body(const FrameConst &fc, const uint32_t *__restrict__ edge_offset, const uint32_t *__restrict__ edge_inc,
                const int32_t *__restrict__ faces, const uint8_t *__restrict__ lit,
                int32_t *__restrict__ sil_edges, uint32_t quad_cap, Counters *__restrict__ ctr, uint32_t block)
{
    int e = (int)(block * blockDim.x + threadIdx.x);
    if (e >= fc.n_edges) return;
    uint32_t cnt = 0, last = 0;
    for (uint32_t k = edge_offset[e]; k < edge_offset[e + 1]; ++k) {
        uint32_t inc = edge_inc[k];
        if (lit[inc >> 2]) { ++cnt; last = inc; }
    }
    if (!(cnt & 1u)) return;
    const int f = (int)(last >> 2), k = (int)(last & 3u), k2 = (k + 1) % 3;
    const int ia = faces[(size_t)f * 12 + k * 4], ib = faces[(size_t)f * 12 + k2 * 4];

    uint32_t sslot = atomicAdd(&ctr->n_quads, 1u);
    if (sslot < quad_cap) {
        sil_edges[sslot * 3 + 0] = f;        // the host maps the face back to its model
        sil_edges[sslot * 3 + 1] = ia;
        sil_edges[sslot * 3 + 2] = ib;
    }
}

// The set-up of the faces and the silhouette search are independent (both read what the frame's
// first launch wrote: screen-space vertices, light-facing flags), so they share a launch:
// workgroups [0, setup_blocks) set faces up, the rest look for silhouette edges.  The leftover
// survivor counts ride with the count pass of the large primitives' work items further down
// the chain (nothing before the visibility kernel needs their verdict).  A nearly empty kernel
// costs ~10 us of latency on its own, and under several frames in flight each stream spends a
// third of its time between dependent kernels: every launch the chain loses is time gained.
__global__ void __launch_bounds__(SETUP_BLOCK)
k_tri_setup(const FrameConst fc, const int32_t *__restrict__ faces, const uint8_t *__restrict__ face_flags,
            const VertexOut *__restrict__ vout, const VertexClip *__restrict__ vclip, TriRec *__restrict__ tris,
            TriClip *__restrict__ clips, uint8_t *__restrict__ status, uint32_t *__restrict__ valid_list,
            uint32_t *__restrict__ count_list, Counters *__restrict__ ctr, const BinArgs bins, uint32_t setup_blocks,
            const uint32_t *__restrict__ edge_offset, const uint32_t *__restrict__ edge_inc,
            const uint8_t *__restrict__ lit, int32_t *__restrict__ sil_edges, uint32_t quad_cap)
{
    if (blockIdx.x < setup_blocks)
        tri_setup_block(fc, faces, face_flags, vout, vclip, tris, clips, status, valid_list, count_list, ctr, bins, blockIdx.x);
    else
        silhouette_body(fc, edge_offset, edge_inc, faces, lit, sil_edges, quad_cap, ctr, blockIdx.x - setup_blocks);
}

// Count pass of the work items (large faces, shadow quads) and the leftover survivor counts: the
// few counting workgroups come first so that their latency-bound work is in flight early.
__global__ void __launch_bounds__(256)
k_bin_large_and_count(const FrameConst fc, const BinArgs bins, const uint32_t *__restrict__ count_list,
                      TriRec *__restrict__ tris, const TriClip *__restrict__ clips, uint8_t *__restrict__ status,
                      Counters *__restrict__ ctr, uint32_t count_blocks)
{
    if (blockIdx.x < count_blocks)
        tri_count_body(fc, count_list, tris, clips, status, ctr, blockIdx.x, count_blocks);
    else
        bin_large_body<false>(fc, bins, blockIdx.x - count_blocks, gridDim.x - count_blocks);
}

// Shadow-quad set-up: extrusion away from the light, Sutherland-Hodgman clipping against the
// camera frustum, projection, plane equation and pixel box (obj/core.py:610-622,
// obj/plane_intersection.py:59-86, obj/triangular.py:320-340).
//
// Sixteen lanes work on one silhouette edge, ONE POLYGON VERTEX PER LANE (a quad clipped by six
// planes has at most ten).  A clipping step is then data-parallel: every lane tests its vertex
// against the plane, fetches its successor with a lane shuffle, emits itself and/or the
// intersection with the plane, and the emitted vertices are compacted (prefix sum of the emit
// counts, through a few hundred bytes of LDS) back to one per lane.  A plane that keeps every
// vertex is skipped (the walk would copy the polygon verbatim).  The arithmetic per vertex and
// per edge is exactly the sequential algorithm's, so the quads are bit-identical; a thread
// walking a scratch-resident polygon took 28 us for 1 133 quads, this takes a few.
constexpr int QS_LANES = 16;
static_assert(MAX_POLY <= QS_LANES, "one polygon vertex per lane");

__device__ __forceinline__ double shfl_d(double v, int src)
{
    return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src));
}

__global__ void __launch_bounds__(64)
k_quad_setup(const FrameConst fc, const int32_t *__restrict__ sil_edges, const double *__restrict__ verts,
             QuadRec *__restrict__ quads, uint32_t quad_cap, Counters *__restrict__ ctr, const BinArgs bins)
{
    __shared__ double s_poly[WAVE / QS_LANES][MAX_POLY + 4][4];
    const int lane = threadIdx.x & (WAVE - 1);
    const int grp = lane / QS_LANES, gl = lane % QS_LANES, g0 = grp * QS_LANES;
    const uint32_t n_sil = min(ctr->n_quads, quad_cap);
    constexpr uint32_t PER_BLOCK = WAVE / QS_LANES;
    // fixed grid striding over the silhouette edges (their number is only known on the device)
  for (uint32_t first = blockIdx.x * PER_BLOCK; first < n_sil; first += gridDim.x * PER_BLOCK) {
    const uint32_t s_idx = first + grp;
    const bool have = s_idx < n_sil;

    // ---- extrusion (obj/core.py:612-621): quad = (A, B, D, C); lanes 0..3 hold A, B, D, C
    double v[4] = { 0, 0, 0, 0 };
    int n = have ? 4 : 0;
    if (have && gl < 4) {
        const int ia = sil_edges[s_idx * 3 + 1], ib = sil_edges[s_idx * 3 + 2];
        const double *src = verts + (size_t)((gl == 0 || gl == 3) ? ia : ib) * 4;
        for (int j = 0; j < 4; ++j) v[j] = src[j];
        if (gl >= 2) {
            if (fc.light_type == MR_LIGHT_POINT) {
                double d[4] = { v[0] - fc.light_pos[0], v[1] - fc.light_pos[1], v[2] - fc.light_pos[2], v[3] - 1.0 };
                double l = sqrt(((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3]);
                if (l == 0) l = 1;
                for (int j = 0; j < 4; ++j) v[j] = v[j] + 1000 * (d[j] / l);
            } else {
                for (int j = 0; j < 3; ++j) v[j] = v[j] + fc.light_dir[j] * -1000;
                v[3] = v[3] + 1.0;
            }
        }
    }

    // ---- clipping, one plane at a time
    for (int pl = 0; pl < 6; ++pl) {
        const double *P = fc.planes + pl * 4;
        const bool mine = gl < n;
        const bool vis = mine && plane_dot(P, v) >= 0;
        const unsigned int gvis = (unsigned int)(__ballot(vis) >> g0) & 0xffffu;
        const bool all_in = gvis == ((1u << n) - 1u);
        // every group takes part in the shuffles below; groups with nothing to clip keep v
        const int nxt_lane = g0 + ((gl + 1 >= n) ? 0 : gl + 1);
        double w[4];
        for (int j = 0; j < 4; ++j) w[j] = shfl_d(v[j], nxt_lane);
        const bool nvis = (gvis >> ((gl + 1 >= n) ? 0 : gl + 1)) & 1u;
        bool emit_cur = mine && vis, emit_int = false;
        double ipt[4] = { 0, 0, 0, 0 };
        if (mine && vis != nvis) {
            // line_plane_intersection(next, current, plane) (obj/plane_intersection.py:24-36, 81)
            double dir[4];
            for (int j = 0; j < 4; ++j) dir[j] = v[j] - w[j];
            const double den = plane_dot(P, dir);
            if (!(fabs(den) < 1e-10)) {
                const double wgt = -plane_dot(P, w) / den;
                if (0 <= wgt && wgt <= 1) {
                    for (int j = 0; j < 4; ++j) ipt[j] = w[j] + wgt * dir[j];
                    emit_int = true;
                }
            }
        }
        const int cnt = (emit_cur ? 1 : 0) + (emit_int ? 1 : 0);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < QS_LANES; off <<= 1) {
            const int y = __shfl_up(incl, off);
            if (gl >= off) incl += y;
        }
        const int total = __shfl(incl, g0 + QS_LANES - 1);
        const bool clip_now = n > 0 && !all_in;
        if (clip_now) {
            int pos = incl - cnt;
            if (emit_cur && pos < MAX_POLY) { for (int j = 0; j < 4; ++j) s_poly[grp][pos][j] = v[j]; ++pos; }
            if (emit_int && pos < MAX_POLY) { for (int j = 0; j < 4; ++j) s_poly[grp][pos][j] = ipt[j]; }
        }
        __syncthreads();
        if (clip_now) {
            n = min(total, MAX_POLY);
            if (gl < n) for (int j = 0; j < 4; ++j) v[j] = s_poly[grp][gl][j];
        }
        __syncthreads();
    }
    const bool alive = n >= 3;                           // obj/triangular.py:322-323

    // ---- projection of the lane's vertex (obj/triangular.py:325-327)
    double sx = 0, sy = 0, sz = 0;
    if (alive && gl < n) {
        double c4[4], nd[4];
        for (int j = 0; j < 4; ++j) c4[j] = row_times_col(v, fc.mvp, j);
        for (int j = 0; j < 4; ++j) nd[j] = c4[j] / c4[3];
        sx = row_times_col(nd, fc.viewport, 0);
        sy = row_times_col(nd, fc.viewport, 1);
        sz = row_times_col(nd, fc.viewport, 2);
    }
    const int nxt_lane = g0 + ((gl + 1 >= n) ? 0 : gl + 1);
    const double nsx = shfl_d(sx, nxt_lane), nsy = shfl_d(sy, nxt_lane);

    // ---- pixel box over the group's vertices (obj/transformation.py:35-43)
    const bool used = alive && gl < n;
    double lo_x = used ? sx : INFINITY, hi_x = used ? sx : -INFINITY;
    double lo_y = used ? sy : INFINITY, hi_y = used ? sy : -INFINITY;
#pragma unroll
    for (int off = 1; off < QS_LANES; off <<= 1) {
        const double a0 = shfl_d(lo_x, lane ^ off), a1 = shfl_d(hi_x, lane ^ off);
        const double b0 = shfl_d(lo_y, lane ^ off), b1 = shfl_d(hi_y, lane ^ off);
        lo_x = a0 < lo_x ? a0 : lo_x; hi_x = a1 > hi_x ? a1 : hi_x;
        lo_y = b0 < lo_y ? b0 : lo_y; hi_y = b1 > hi_y ? b1 : hi_y;
    }
    // ---- plane through the first three vertices (obj/triangular.py:328-333)
    const double x0 = shfl_d(sx, g0), y0 = shfl_d(sy, g0), z0 = shfl_d(sz, g0);
    const double x1 = shfl_d(sx, g0 + 1), y1 = shfl_d(sy, g0 + 1), z1 = shfl_d(sz, g0 + 1);
    const double x2 = shfl_d(sx, g0 + 2), y2 = shfl_d(sy, g0 + 2), z2 = shfl_d(sz, g0 + 2);

    double xs[2] = { lo_x, hi_x }, ys[2] = { lo_y, hi_y };
    int bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
    const bool boxed = alive && bound_box(xs, ys, 2, fc.width, fc.height, bx0, bx1, by0, by1);
    uint32_t slot = 0;
    if (boxed && gl == 0) slot = atomicAdd(&ctr->n_quads_drawn, 1u);
    slot = (uint32_t)__shfl((int)slot, g0);
    // count pass of the binning: the quad as work items of 64 tiles (kernels_bin.h); all lanes take part
    push_work_items(bins, WORK_QUAD | slot,
                    (boxed && gl == 0 && slot < quad_cap) ? quad_chunks(fc, bx0, bx1, by0, by1) : 0u);
    if (!boxed) continue;
    if (slot >= quad_cap) { if (gl == 0) atomicOr(&ctr->overflow, 4u); continue; }

    QuadRec &q = quads[slot];
    if (gl < MAX_POLY) {
        QuadEdge e;
        e.sx = used ? sx : 0.0; e.sy = used ? sy : 0.0;
        e.ex = used ? nsx - sx : 0.0; e.ey = used ? nsy - sy : 0.0;
        q.e[gl] = e;
    }
    if (gl == 0) {
        const double ab[3] = { x0 - x1, y0 - y1, z0 - z1 }, ac[3] = { x0 - x2, y0 - y2, z0 - z2 };
        const double nx = ab[1] * ac[2] - ab[2] * ac[1];
        const double ny = ab[2] * ac[0] - ab[0] * ac[2];
        const double nz = ab[0] * ac[1] - ab[1] * ac[0];
        q.nx = nx; q.ny = ny; q.nz = nz;
        q.d = chain3(-x0, -y0, -z0, nx, ny, nz);
        q.is_front = nz < 0;
        q.n = n;
        q.edge = (int32_t)s_idx;
        q.x0 = (int16_t)bx0; q.x1 = (int16_t)bx1; q.y0 = (int16_t)by0; q.y1 = (int16_t)by1;
        q.pad[0] = q.pad[1] = q.pad[2] = 0;
    }
  }
}

}  // namespace mr
