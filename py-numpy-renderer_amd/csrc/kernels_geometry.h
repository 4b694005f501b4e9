// kernels_geometry.h -- per-face and per-silhouette-edge work of one frame: everything that has
// to exist before a tile can be rasterised, in TWO launches.
//
//   k_setup       face workgroups: vertex transform of the face's own corners
//                 (obj/triangular.py:36-45), cull, pixel box, barycentric constants
//                 (obj/triangular.py:47-78, obj/core.py:127-136, obj/transformation.py:12-43), the
//                 shading attributes of the face (TriAttr), and the face's own tile lists;
//                 edge workgroups: light-facing test from the static per-edge normals, silhouette
//                 (obj/triangular.py:286-302), then extrusion, clip, projection, plane and box of
//                 the shadow quad (obj/core.py:610-622, obj/plane_intersection.py:59-86,
//                 obj/triangular.py:320-340)
//   k_bin_work    the large primitives' tile lists (kernels_bin.h) and the survivor counts that
//                 need a whole wavefront (how many fragments of a face survive coverage + clip
//                 decides NumPy's dot-vs-gemv rounding of z, and the CLIPPED status)
//   k_vertex_mfma the vertex transform once per unique vertex on the matrix cores, as a separate
//                 launch in front of k_setup (MR_VERTEX_PATH=mfma; bit-identical)
//   k_face_normals / k_edge_normals   static per scene: run when the scene is committed
#pragma once

#include "kernels_bin.h"

namespace mr {

// Unit normal of the world-space triangle in the vertices' own dtype (obj/core.py:127-130); the
// light-facing test dots it with light.position (obj/triangular.py:295).  Static per face: computed
// when the scene is committed.
__device__ __forceinline__ void face_unit_normal(const double *a, const double *b, const double *c, bool verts_f32,
                                                 double n[3])
{
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
}

__global__ void __launch_bounds__(256)
k_face_normals(int n_faces, const int32_t *__restrict__ faces, const uint8_t *__restrict__ face_flags,
               const double *__restrict__ verts, double *__restrict__ face_n)
{
    const int f = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (f >= n_faces) return;
    const int32_t *fcx = faces + (size_t)f * 12;
    double n[3];
    face_unit_normal(verts + (size_t)fcx[0] * 4, verts + (size_t)fcx[4] * 4, verts + (size_t)fcx[8] * 4,
                     (face_flags[f] & FF_VERTS_F32) != 0, n);
    face_n[(size_t)f * 4 + 0] = n[0]; face_n[(size_t)f * 4 + 1] = n[1]; face_n[(size_t)f * 4 + 2] = n[2];
    face_n[(size_t)f * 4 + 3] = 0.0;
}

// Copies the incident faces' normals into the edge records (the host filled in the incidences).
__global__ void __launch_bounds__(256)
k_edge_normals(int n_edges, EdgeRec *__restrict__ edges, const double *__restrict__ face_n)
{
    const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (e >= n_edges) return;
    EdgeRec r = edges[e];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.n[i][j] = r.inc[i] != 0xffffffffu ? face_n[(size_t)(r.inc[i] >> 2) * 4 + j] : 0.0;
    edges[e] = r;
}

// EdgeRec[] -> EdgeRec32[] in place is not possible (overlap): into a second buffer, once per commit.
__global__ void __launch_bounds__(256)
k_edge_compact(int n_edges, const EdgeRec *__restrict__ edges, EdgeRec32 *__restrict__ out)
{
    const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (e >= n_edges) return;
    const EdgeRec r = edges[e];
    EdgeRec32 c;
    c.inc[0] = r.inc[0]; c.inc[1] = r.inc[1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c.n[i][j] = (float)r.n[i][j];
    out[e] = c;
}

// The static face records (rast_types.h, FacePosT / FaceAttr): one gather per scene, when it is committed.
template <class T>
__global__ void __launch_bounds__(256)
k_face_static(int n_faces, const int32_t *__restrict__ faces, const uint8_t *__restrict__ face_flags,
              const double *__restrict__ verts, const float *__restrict__ uv, const float *__restrict__ normals,
              FacePosT<T> *__restrict__ out_pos, FaceAttr *__restrict__ out_attr)
{
    const int f = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (f >= n_faces) return;
    const int32_t *row = faces + (size_t)f * 12;           // [vertex, uv, normal, material] per corner
    const uint8_t ff = face_flags[f];
    FacePosT<T> p;
    FaceAttr a;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) p.v[k][j] = (T)verts[(size_t)row[k * 4] * 4 + j];
        a.uv[k][0] = (ff & FF_HAS_UV) ? uv[(size_t)row[k * 4 + 1] * 3] : 0.f;
        a.uv[k][1] = (ff & FF_HAS_UV) ? uv[(size_t)row[k * 4 + 1] * 3 + 1] : 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) a.n[k][j] = (ff & FF_HAS_NORMALS) ? normals[(size_t)row[k * 4 + 2] * 3 + j] : 0.f;
    }
    p.material = row[3]; p.flags = ff; p.pad[0] = p.pad[1] = 0;
    a.pad = 0.f;
    out_pos[f] = p;
    out_attr[f] = a;
}

// The three corners of face f (and its material and flags) from the static records, as float64
__device__ __forceinline__ void load_face_pos(const void *face_pos, bool pos32, int f, double va[4], double vb[4], double vc[4],
                                              int32_t &material, uint8_t &ff)
{
    if (pos32) {
        const FacePos32 p = static_cast<const FacePos32 *>(face_pos)[f];
#pragma unroll
        for (int j = 0; j < 4; ++j) { va[j] = (double)p.v[0][j]; vb[j] = (double)p.v[1][j]; vc[j] = (double)p.v[2][j]; }
        material = p.material; ff = (uint8_t)p.flags;
    } else {
        const FacePos64 p = static_cast<const FacePos64 *>(face_pos)[f];
#pragma unroll
        for (int j = 0; j < 4; ++j) { va[j] = p.v[0][j]; vb[j] = p.v[1][j]; vc[j] = p.v[2][j]; }
        material = p.material; ff = (uint8_t)p.flags;
    }
}

// One face corner through obj/triangular.py:36-45: clip = v @ MVP (and @ debug MVP), depth =
// 1 / clip.w, ndc = clip * depth, screen = ndc @ viewport; plus linearize_z of the screen z.
struct CornerOut {
    double sx, sy, sz, depth, zlin;
    bool safe;
};

// v @ camera.MVP and v @ debug_camera.MVP of one corner
__device__ __forceinline__ void clip_coords(const FrameConst &fc, const double v[4], double clip[4], double clipd[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        clip[j] = row_times_col(v, fc.mvp, j);
        clipd[j] = fc.same_clip ? clip[j] : row_times_col(v, fc.debug_mvp, j);
    }
}

__device__ __forceinline__ void xform_vertex(const FrameConst &fc, const double v[4], CornerOut &o)
{
    // the clip-space coordinates are not kept: the few faces that need them for the per-fragment
    // clip test recompute them (clip_coords), which is cheaper than 48 live registers in every lane
    double clip[4], clipd[4];
    clip_coords(fc, v, clip, clipd);
    o.depth = 1.0 / clip[3];
    double ndc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ndc[j] = clip[j] * o.depth;
    o.sx = row_times_col(ndc, fc.viewport, 0);
    o.sy = row_times_col(ndc, fc.viewport, 1);
    o.sz = row_times_col(ndc, fc.viewport, 2);
    o.zlin = linearize_z(fc, o.sz);
    // Strictly inside both cameras' clip volumes with a relative margin of 1e-12.  A fragment's
    // clip coordinates are a non-negative combination of the corners' (weights u*dp/wc, all
    // >= 0 when u,v,w >= 0 and every w > 0), evaluated with a few ulp (1e-16) of rounding, so
    // when all three corners carry this flag the strict test of obj/triangular.py:85-87 cannot
    // fail for any fragment of the face and need not be evaluated.
    const double k = 1.0 - 1e-12;
    const double wl = clip[3] * k, wd = clipd[3] * k;
    o.safe = fabs(clip[0]) < wl && fabs(clip[1]) < wl && fabs(clip[2]) < wl &&
             fabs(clipd[0]) < wd && fabs(clipd[1]) < wd && fabs(clipd[2]) < wd;
}

// The vertex stage on the matrix cores, once per unique vertex.  The two products of
// obj/triangular.py:36-45 (v @ [MVP | debug MVP] and ndc @ viewport) are dense 16x4 by 4x16
// contractions per 16 vertices, i.e. exactly one v_mfma_f64_16x16x4_f64 each.  Measured on MI355X
// (tools/micro/mfma_vertex_check.hip, 8.4 M outputs): that instruction accumulates k = 0..3 in
// order with one rounding per step, bit-identical to the ascending fma chain the reference's
// BLAS uses, so the result is what xform_vertex computes, bit for bit.
// One wavefront = 16 vertices.  Operand layout (cdna_hip_programming.md section 3): A lane l holds
// A[row l%16][k l/16], B lane l holds B[k l/16][col l%16], D register i of lane l holds
// D[row (l>>4) + 4i][col l&15]; the D -> A re-layout between the two products goes through LDS.
// On CDNA4 the FP64 matrix rate equals the FP64 vector rate and half of the 16 output columns are
// padding here, so this is a separate, optional launch (MR_VERTEX_PATH=mfma): by default every
// face transforms its own three corners inside k_setup, which costs 3x the arithmetic and one
// dependent launch less (DESIGN.md has both measured).
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
k_vertex_mfma(const FrameConst fc, const double *__restrict__ verts, VertexOut *__restrict__ out,
              VertexClip *__restrict__ out_clip)
{
    __shared__ double s_clip[4][16][8];     // per wavefront: [vertex][MVP x,y,z,w | debug x,y,z,w]
    __shared__ double s_scr[4][16][4];      // per wavefront: [vertex][screen x, y, z]
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
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
        const double kk = 1.0 - 1e-12;                  // "safely inside" flag: see xform_vertex
        const double wl = oc.clip[3] * kk, wd = oc.clipd[3] * kk;
        o.safe = (fabs(oc.clip[0]) < wl && fabs(oc.clip[1]) < wl && fabs(oc.clip[2]) < wl &&
                  fabs(oc.clipd[0]) < wd && fabs(oc.clipd[1]) < wd && fabs(oc.clipd[2]) < wd) ? 1 : 0;
        o.pad = 0;
        out[base + v] = o;
        out_clip[base + v] = oc;
    }
}

// Survivors of coverage + clip among the samples of a triangle's pixel box, as seen by one lane.
// (what the sample tests read of the frame constants, as values: fetched once in front of a loop over samples)
struct SampleConst { int band_y0, band_y1, same_clip; };
__device__ __forceinline__ SampleConst sample_const(const FrameConst &fc) { return { fc.band_y0, fc.band_y1, fc.same_clip }; }

__device__ __forceinline__ bool sample_survives(const SampleConst &fc, const TriRec &t, const double dp[3],
                                                const TriClip *clips, int px, int py, bool &covered_in_band)
{
    const bool single = (t.flags & TF_SINGLE_BOX) != 0;
    float u, v, w;
    tri_bary(t, (double)px, (double)py, single, u, v, w);
    bool ok = u >= 0 && v >= 0 && w >= 0;
    covered_in_band = ok && py >= fc.band_y0 && py < fc.band_y1;
    if (ok && (t.flags & TF_CLIP)) {
        const TriClip &c = clips[t.face];
        double p[3];
        persp_bary(dp, u, v, w, single, p);
        ok = inside_clip(p, c.clip) && (fc.same_clip || inside_clip(p, c.clipd));
    }
    return ok;
}

__device__ __forceinline__ void count_finish(TriRec *tris, uint8_t *status, int f, uint32_t flags, int found)
{
    if (found == 0) {
        // The face was binned before this verdict (kernels_bin.h) and stays listed: the tile
        // kernel finds no surviving fragment for it and counts its covered ones (the fragment
        // count is taken before the clip, obj/triangular.py:78).
        status[f] = FACE_CLIPPED;
    } else if (found == 1) {
        tris[f].flags = flags | TF_SINGLE_Z;
    }
}

constexpr int COUNT_SMALL_BOX = 32;   // pixel boxes up to this size are walked by a single lane

struct SetupArgs {
    const int32_t *faces;            // (F, 3, 4) global indices [vertex, uv, normal, material]
    const uint8_t *face_flags;
    const double *verts;             // (V, 4) world space
    const float *uv, *normals;
    const VertexOut *vout;           // PRE_XFORM only: what k_vertex_mfma left
    const VertexClip *vclip;
    const void *face_pos;            // static per face: FacePos32[] / FacePos64[] (fc.pos32)
    TriRec *tris;
    TriClip *clips;
    uint8_t *status;
    uint32_t *count_list;            // faces whose survivor count needs a wavefront (k_bin_work)
    Counters *ctr;
    const uint8_t *tile_class;       // what the slot's previous frame left (kernels_tile.h, tile_class) ...
    uint32_t *order;                 // ... and the tile order made of it for this frame's tile kernel
    const ClusterRec *clusters;      // static: one per 64 consecutive faces (cluster_culled)
    const EdgeRec *edges;            // static unique-edge table (EdgeRec32[] when fc.edge_compact)
    const uint32_t *edge_inc;        // incidences beyond an edge's first two
    const double *face_n;            // static face normals (for those)
    int32_t *sil_edges;              // (quad_cap, 2): face, corner of each silhouette edge
    QuadRec *quads;
    uint32_t quad_cap;
};

// k_setup's arguments in one block, read phase by phase through kernargs<>() (rast_math.h): the three 4x4 matrices,
// the six planes and two dozen pointers do not fit the scalar registers at once, and read as plain arguments they
// were all fetched at the top and parked in vector-register lanes.
struct SetupKernArgs { FrameConst fc; SetupArgs sa; BinArgs bins; uint32_t face_blocks; uint32_t edge_spread; };
#define SETUP_ARGS() const SetupKernArgs &ka_ = kernargs<SetupKernArgs>(); const FrameConst &fc = ka_.fc; \
                     const SetupArgs &sa = ka_.sa; const BinArgs &bins = ka_.bins; (void)fc; (void)sa; (void)bins

struct CornerOut;
template <bool PRE_XFORM>
__device__ __forceinline__ int tri_setup_record(int f, int32_t material, uint8_t ff,
                                                const double va[4], const double vb[4], const double vc[4],
                                                const CornerOut &A, const CornerOut &B, const CornerOut &C,
                                                unsigned int &covered, PrimBox &pb, bool &clip);

// One face: status, TriRec / TriAttr / TriClip.  Returns bit 0 = the face goes on to the tile
// kernel, bit 1 = its survivor count is left to k_bin_work; `covered` receives the fragments of a
// face settled as CLIPPED right here.
template <bool PRE_XFORM>
__device__ __forceinline__ int tri_setup_one(int f, unsigned int &covered, PrimBox &pb, bool &clip)
{
    SETUP_ARGS();                                                // phase 1: the face's static record, transform, cull
    double va[4], vb[4], vc[4];
    int32_t material;
    uint8_t ff;
    load_face_pos(sa.face_pos, fc.pos32 != 0, f, va, vb, vc, material, ff);
    CornerOut A, B, C;
    if (PRE_XFORM) {
        const int32_t *row = sa.faces + (size_t)f * 12;        // (the matrix-core vertex path goes by vertex index)
        auto take = [&](int v, CornerOut &o) {
            const VertexOut q = sa.vout[v];
            o.sx = q.sx; o.sy = q.sy; o.sz = q.sz; o.depth = q.depth; o.zlin = q.zlin; o.safe = q.safe != 0;
        };
        take(row[0], A); take(row[4], B); take(row[8], C);
    } else {
        xform_vertex(fc, va, A); xform_vertex(fc, vb, B); xform_vertex(fc, vc, C);
    }
    uint8_t *status = sa.status;

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

    return tri_setup_record<PRE_XFORM>(f, material, ff, va, vb, vc, A, B, C, covered, pb, clip);
}

// second half of tri_setup_one: the faces that survive the cull (its own view of the arguments: phase 2)
template <bool PRE_XFORM>
__device__ __forceinline__ int tri_setup_record(int f, int32_t material, uint8_t ff,
                                                const double va[4], const double vb[4], const double vc[4],
                                                const CornerOut &A, const CornerOut &B, const CornerOut &C,
                                                unsigned int &covered, PrimBox &pb, bool &clip)
{
    SETUP_ARGS();
    uint8_t *status = sa.status;
    TriRec t;
    double xs[3] = { A.sx, B.sx, C.sx }, ys[3] = { A.sy, B.sy, C.sy };
    int bx0, bx1, by0, by1;
    if (!bound_box(xs, ys, 3, fc.width, fc.height, bx0, bx1, by0, by1)) {
        status[f] = FACE_EMPTY_Z;
        return 0;
    }
    // a device that renders a band of rows, or interleaved tile rows, drops the faces whose pixel box touches
    // none of its tiles right here, before any record is written (with N devices a dense mesh's set-up
    // records are then written once across the node, not N times)
    {
        TileSpan own;
        if (bx0 < bx1 && by0 < by1 && !tile_span(fc, bx0, bx1, by0, by1, own)) { status[f] = FACE_CLIPPED; return 0; }
    }
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
    t.material = material;
    t.pad = 0;
    t.dp[0] = A.depth; t.dp[1] = B.depth; t.dp[2] = C.depth; t.pad2 = 0.0;
    long long box = (long long)(bx1 - bx0) * (long long)(by1 - by0);
    const bool need_clip = (ff & FF_CLIP) && !(A.safe && B.safe && C.safe);
    clip = need_clip;
    t.flags = (need_clip ? TF_CLIP : 0u) | (box == 1 ? TF_SINGLE_BOX : 0u) | ((uint32_t)ff << 8);   // bits 8-15: face flags, for shading
    t.face = f;
    if (box <= 0) { status[f] = FACE_CLIPPED; return 0; }    // no sample inside the box
    status[f] = FACE_OK;

    // How many fragments survive coverage + clip (0 -> CLIPPED, 1 -> z is a dot, TF_SINGLE_Z)?
    // A small pixel box that needs no clip test is settled right here, by this lane, over the
    // WHOLE frame (not just this device's band); the rest is left to k_bin_work.
    const bool count_here = !need_clip && box <= COUNT_SMALL_BOX;
    const double dp[3] = { A.depth, B.depth, C.depth };
    if (count_here) {
        const int bw = bx1 - bx0;
        const SampleConst sc = sample_const(fc);
        int found = 0;
        for (int idx = 0; idx < (int)box && found < 2; ++idx) {
            bool cov;
            found += sample_survives(sc, t, dp, nullptr, bx0 + idx % bw, by0 + idx / bw, cov) ? 1 : 0;
            covered += cov ? 1u : 0u;
        }
        if (found == 0) {
            status[f] = FACE_CLIPPED;              // never reaches the tile kernel: its fragments are counted here
            return 0;
        }
        covered = 0;
        if (found == 1) t.flags |= TF_SINGLE_Z;
    }
    sa.tris[f] = t;        // (shading finds the rest of the face -- world corners, uv, normals -- in the static records)

    if (need_clip) {
        TriClip &cl = sa.clips[f];
        if (PRE_XFORM) {
            const int32_t *row = sa.faces + (size_t)f * 12;
            const VertexClip ca = sa.vclip[row[0]], cb = sa.vclip[row[4]], cc = sa.vclip[row[8]];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cl.clip[0][j] = ca.clip[j]; cl.clip[1][j] = cb.clip[j]; cl.clip[2][j] = cc.clip[j];
                cl.clipd[0][j] = ca.clipd[j]; cl.clipd[1][j] = cb.clipd[j]; cl.clipd[2][j] = cc.clipd[j];
            }
        } else {
            // (rare: the corners are read again here rather than kept in 24 registers of every lane since the cull)
            double wa[4], wb[4], wc[4];
            int32_t m2;
            uint8_t f2;
            load_face_pos(sa.face_pos, fc.pos32 != 0, f, wa, wb, wc, m2, f2);
            clip_coords(fc, wa, cl.clip[0], cl.clipd[0]);
            clip_coords(fc, wb, cl.clip[1], cl.clipd[1]);
            clip_coords(fc, wc, cl.clip[2], cl.clipd[2]);
        }
    }
    return count_here ? 1 : 3;
}

__device__ __forceinline__ double shfl_d(double v, int src)
{
    return __hiloint2double(__shfl(__double2hiint(v), src), __shfl(__double2loint(v), src));
}

// Does the per-face work of cluster `cid` (the 64 faces of one wavefront) come to nothing for every one of them?
// Wavefront-uniform; conservative against the per-face tests of tri_setup_one / tri_setup_record:
//   * all eight corners of the cluster's box are in front of the camera (clip w > 0: the projection is then monotone
//     along every edge of the box, so the faces' screen coordinates lie between the corners'), and either
//   * the corners' screen box, widened by two pixels, holds no sample of the frame or none on this device's rows --
//     every face's own pixel box is inside it, so bound_box / tile_span would have dropped each of them -- or
//   * the frame culls back faces and every normal of the cluster's cone points away from the camera even from the
//     corner of the box where it points away least, with a margin of 1e-4 of the distance (the per-face test decides on
//     the sign of a screen-space area evaluated in float64: its rounding is ten orders of magnitude below that).
// Faces it lets through are set up exactly as before; faces it stops would have been stopped one by one.
__device__ __forceinline__ bool cluster_culled(uint32_t cid)
{
    SETUP_ARGS();
    const int lane = threadIdx.x & (WAVE - 1);
    const ClusterRec c = sa.clusters[__builtin_amdgcn_readfirstlane(cid)];
    // corner (lane & 7) of the box through the frame's matrices
    const double v[4] = { (double)((lane & 1) ? c.hi[0] : c.lo[0]), (double)((lane & 2) ? c.hi[1] : c.lo[1]),
                          (double)((lane & 4) ? c.hi[2] : c.lo[2]), 1.0 };
    const double cx = row_times_col(v, fc.mvp, 0), cy = row_times_col(v, fc.mvp, 1), cz = row_times_col(v, fc.mvp, 2),
                 cw = row_times_col(v, fc.mvp, 3);
    double r = __builtin_amdgcn_rcp(cw);
    r = fma(fma(-cw, r, 1.0), r, r);
    const double nd[4] = { cx * r, cy * r, cz * r, 1.0 };
    double x_lo = row_times_col(nd, fc.viewport, 0), y_lo = row_times_col(nd, fc.viewport, 1), w_lo = cw;
    double x_hi = x_lo, y_hi = y_lo, w_hi = cw;
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        const double a = shfl_d(x_lo, lane ^ off), b = shfl_d(x_hi, lane ^ off), d = shfl_d(y_lo, lane ^ off),
                     e = shfl_d(y_hi, lane ^ off), f = shfl_d(w_lo, lane ^ off), g = shfl_d(w_hi, lane ^ off);
        x_lo = a < x_lo ? a : x_lo; x_hi = b > x_hi ? b : x_hi;
        y_lo = d < y_lo ? d : y_lo; y_hi = e > y_hi ? e : y_hi;
        w_lo = f < w_lo ? f : w_lo; w_hi = g > w_hi ? g : w_hi;
    }
    // (anything not finite fails these comparisons: no culling then)
    if (!(w_lo > 1e-6 && w_lo > 1e-9 * w_hi && w_hi < 1e300)) return false;
    if (!(fabs(x_lo) < 1e9 && fabs(x_hi) < 1e9 && fabs(y_lo) < 1e9 && fabs(y_hi) < 1e9)) return false;
    {
        const double xs[2] = { x_lo - 2.0, x_hi + 2.0 }, ys[2] = { y_lo - 2.0, y_hi + 2.0 };
        int bx0, bx1, by0, by1;
        if (!bound_box(xs, ys, 2, fc.width, fc.height, bx0, bx1, by0, by1)) return true;
        TileSpan own;
        if (!tile_span(fc, bx0, bx1, by0, by1, own)) return true;
    }
    if (!(fc.cluster_cull & CC_CONE) || !fc.backface_culling || !(c.cos_half >= 0.f)) return false;
    // the cone: s * n . (a - eye) > 0 for every face normal n and face point a of the cluster
    const double s = (fc.cluster_cull & CC_NEGATIVE) ? -1.0 : 1.0;
    const double ctr[3] = { 0.5 * ((double)c.lo[0] + (double)c.hi[0]), 0.5 * ((double)c.lo[1] + (double)c.hi[1]),
                            0.5 * ((double)c.lo[2] + (double)c.hi[2]) };
    const double hx = 0.5 * ((double)c.hi[0] - (double)c.lo[0]), hy = 0.5 * ((double)c.hi[1] - (double)c.lo[1]),
                 hz = 0.5 * ((double)c.hi[2] - (double)c.lo[2]);
    const double radius = sqrt(hx * hx + hy * hy + hz * hz) * (1.0 + 1e-6);
    const double d[3] = { s * (ctr[0] - fc.cull_eye[0]), s * (ctr[1] - fc.cull_eye[1]), s * (ctr[2] - fc.cull_eye[2]) };
    const double dist2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const double along = d[0] * (double)c.axis[0] + d[1] * (double)c.axis[1] + d[2] * (double)c.axis[2];      // |d| cos(theta)
    const double across2 = dist2 - along * along;
    const double across = across2 > 0 ? sqrt(across2) * (1.0 + 1e-6) : 0.0;                                    // |d| sin(theta)
    // least of n . d over the cone is |d| cos(theta + alpha); a face point is at most `radius` from the centre
    const double least = along * (double)c.cos_half - across * (double)c.sin_half - radius;
    return least > 1e-4 * sqrt(dist2) + 1e-9 && dist2 < 1e300;
}

constexpr int SETUP_BLOCK = 256;

// Face workgroup: one face per lane.  The list of faces whose survivor count needs a wavefront and
// the frame's count of set-up faces are appended to with ONE atomic per workgroup.
template <bool PRE_XFORM>
__device__ __forceinline__ void tri_setup_block(uint32_t block)
{
    constexpr int NW = SETUP_BLOCK / WAVE;
    __shared__ uint32_t s_valid[NW], s_count[NW], s_covered;
    const int f = block * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    if (threadIdx.x == 0) s_covered = 0;
    unsigned int covered = 0;
    PrimBox pb = { 0, 0, 0, 0 };
    bool clip = false;
    bool go;
    {
        const FrameConst &fc0 = kernargs<SetupKernArgs>().fc;
        go = f < fc0.n_faces;
        // (the cluster of a wavefront's faces: both are 64 consecutive faces)
        if (fc0.cluster_cull && (block * blockDim.x + wv * WAVE) < (uint32_t)fc0.n_faces &&
            cluster_culled(block * (blockDim.x / WAVE) + (uint32_t)wv)) {
            go = false;
            if (lane == 0 && (fc0.cluster_cull & CC_COUNT)) atomicAdd(&kernargs<SetupKernArgs>().sa.ctr->pad0[0], 1u);   // (MR_CLUSTER_CULL=count: mr_debug_clusters_culled)
        }
    }
    const int r = go ? tri_setup_one<PRE_XFORM>(f, covered, pb, clip) : 0;
    SETUP_ARGS();                                                // phase 3: the tile lists, the workgroup's epilogue
    // the face's own tile lists (kernels_bin.h)
    bin_triangles(fc, bins, (r & 1) != 0, (uint32_t)f, pb, clip);
    const unsigned long long bv = __ballot(r & 1), bc = __ballot(r & 2);
    if (lane == 0) { s_valid[wv] = (uint32_t)__popcll(bv); s_count[wv] = (uint32_t)__popcll(bc); }
    __syncthreads();
    if (covered) atomicAdd(&s_covered, covered);
    if (threadIdx.x == 0) {
        uint32_t nv = 0, nc = 0;
        for (int w = 0; w < NW; ++w) { nv += s_valid[w]; nc += s_count[w]; }
        if (nv) atomicAdd(&sa.ctr->n_valid_tris, nv);
        uint32_t bcb = nc ? atomicAdd(&sa.ctr->n_count, nc) : 0u;
        for (int w = 0; w < NW; ++w) {
            const uint32_t b = s_count[w];
            s_count[w] = bcb;
            bcb += b;
        }
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    if (r & 2) sa.count_list[s_count[wv] + (uint32_t)__popcll(bc & below)] = (uint32_t)f;
    if (threadIdx.x == 0 && s_covered) atomicAdd(&sa.ctr->frag_tri, (unsigned long long)s_covered);
}

// Counts, per listed face, the fragments that survive coverage + clip over the WHOLE frame
// (not just this device's band), stopping as soon as two are found:
//   0 -> the reference returns CLIPPED for the face; 1 -> z = bar @ zlin is a dot (TF_SINGLE_Z).
// Only the faces k_setup could not settle itself arrive here (pixel boxes over 32 samples,
// or a per-fragment clip test).
__device__ __forceinline__ void
tri_count_body(const FrameConst &fc, const uint32_t *__restrict__ count_list, TriRec *__restrict__ tris,
               const TriClip *__restrict__ clips, uint8_t *__restrict__ status,
               Counters *__restrict__ ctr, uint32_t block, uint32_t n_blocks)
{
    // one wavefront per listed face, 64 samples per step, starting at the chunk that holds the
    // centroid (a well-shaped triangle is settled by that chunk alone)
    const int lane = threadIdx.x & (WAVE - 1);
    const uint32_t n_count = ctr->n_count;
    const uint32_t waves = n_blocks * (blockDim.x / WAVE);
    const SampleConst sc = sample_const(fc);
    for (uint32_t i = block * (blockDim.x / WAVE) + threadIdx.x / WAVE; i < n_count; i += waves) {
        const int fb = (int)count_list[i];
        const TriRec tb = tris[fb];
        const double dp[3] = { tb.dp[0], tb.dp[1], tb.dp[2] };
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
            if (idx < n) ok = sample_survives(sc, tb, dp, clips, tb.x0 + (int)(idx % w), tb.y0 + (int)(idx / w), cov);
            found += __popcll(__ballot(ok));
        }
        if (lane == 0) count_finish(tris, status, fb, tb.flags, found);
    }
}

// Per-face result of the reference's lit pass (obj/triangular.py:101-112 with a stencil
// buffer): a face that reached the depth stage is "rendered" when at least one of its fragments
// has z <= the final z-buffer value (i.e. equals it) where stencil == 0, else EMPTY_Z.  Runs
// after the tile kernel, only when MR_FRAME_FACE_STATUS is set (obj/core.py:625-636 prints the
// histogram).  Thread per face; large boxes are walked by the whole wavefront.
__device__ __forceinline__ bool sample_is_drawn(const FrameConst &fc, const TriRec &t, const double dp[3],
                                                const TriClip *clips, const double *zbuf, const int32_t *stencil,
                                                int px, int py)
{
    bool cov;
    if (!sample_survives(sample_const(fc), t, dp, clips, px, py, cov)) return false;
    float u, v, w;
    tri_bary(t, (double)px, (double)py, (t.flags & TF_SINGLE_BOX) != 0, u, v, w);
    const double z = rows_dot3((t.flags & TF_SINGLE_Z) != 0, (double)u, (double)v, (double)w, t.zl0, t.zl1, t.zl2);
    const size_t at = (size_t)py * fc.width + px;
    const bool pass = fc.system == 1 ? (zbuf[at] >= z) : (zbuf[at] <= z);
    return pass && (int16_t)stencil[at] == 0;
}

__global__ void __launch_bounds__(256)
k_face_status(const FrameConst fc, const TriRec *__restrict__ tris,
              const TriClip *__restrict__ clips, const double *__restrict__ zbuf, const int32_t *__restrict__ stencil,
              uint8_t *__restrict__ status)
{
    const int f = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int lane = threadIdx.x & (WAVE - 1);
    const bool valid = f < fc.n_faces && status[f] == FACE_OK;
    TriRec t = {};
    double dp[3] = { 0, 0, 0 };
    if (valid) { t = tris[f]; dp[0] = t.dp[0]; dp[1] = t.dp[1]; dp[2] = t.dp[2]; }
    const int bw = t.x1 - t.x0, bh = t.y1 - t.y0;
    const int total = valid ? bw * bh : 0;
    if (valid && total <= COUNT_SMALL_BOX) {
        bool drawn = false;
        for (int idx = 0; idx < total && !drawn; ++idx)
            drawn = sample_is_drawn(fc, t, dp, clips, zbuf, stencil, t.x0 + idx % bw, t.y0 + idx / bw);
        if (!drawn) status[f] = FACE_EMPTY_Z;
    }
    unsigned long long big = __ballot(valid && total > COUNT_SMALL_BOX);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const int fb = __shfl(f, src);
        const TriRec tb = tris[fb];
        const double dpb[3] = { tb.dp[0], tb.dp[1], tb.dp[2] };
        const int w = tb.x1 - tb.x0;
        const long long n = (long long)w * (tb.y1 - tb.y0);
        bool any = false;
        for (long long base = 0; base < n && !any; base += WAVE) {
            const long long idx = base + lane;
            const bool d = idx < n && sample_is_drawn(fc, tb, dpb, clips, zbuf, stencil, tb.x0 + (int)(idx % w), tb.y0 + (int)(idx / w));
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


// Shadow-quad set-up: extrusion away from the light, Sutherland-Hodgman clipping against the
// camera frustum, projection, plane equation and pixel box (obj/core.py:610-622,
// obj/plane_intersection.py:59-86, obj/triangular.py:320-340).
//
// Sixteen lanes work on one silhouette edge, ONE POLYGON VERTEX PER LANE (a quad clipped by six
// planes has at most ten).  A clipping step is then data-parallel: every lane tests its vertex
// against the plane, fetches its successor with a lane shuffle, emits itself and/or the
// intersection with the plane, and the emitted vertices are compacted (prefix sum of the emit
// counts, through a few hundred bytes of LDS private to the wavefront) back to one per lane.  A
// plane that keeps every vertex is skipped (the walk would copy the polygon verbatim).  The
// arithmetic per vertex and per edge is exactly the sequential algorithm's, so the quads are
// bit-identical.  `have`: this 16-lane group has an edge (face, corner = sil_f, sil_k; list slot
// s_idx).  Every lane of the wavefront must call it; s_poly is the wavefront's scratch.
constexpr int QS_LANES = 16;
constexpr uint32_t EDGE_DENSE = 0xffu;       // SetupKernArgs::edge_spread: two edges per lane (see edge_block)
static_assert(MAX_POLY <= QS_LANES, "one polygon vertex per lane");

__device__ __forceinline__ void quad_setup_group(bool have, int sil_f, int sil_k, uint32_t s_base_raw, uint32_t s_rank,
                                                 double (*s_poly)[MAX_POLY + 4][4])
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int grp = lane / QS_LANES, gl = lane % QS_LANES, g0 = grp * QS_LANES;

    // ---- extrusion (obj/core.py:612-621): quad = (A, B, D, C); lanes 0..3 hold A, B, D, C
    double v[4] = { 0, 0, 0, 0 };
    int n = have ? 4 : 0;
    {
    SETUP_ARGS();                                                // phase: the edge's corners, extrusion
    if (have && gl < 4) {
        const int corner = (gl == 0 || gl == 3) ? sil_k : (sil_k + 1) % 3;       // A, B, D, C: the edge's first / second end
        if (fc.pos32) {
            const float *src = static_cast<const FacePos32 *>(sa.face_pos)[sil_f].v[corner];
            for (int j = 0; j < 4; ++j) v[j] = (double)src[j];
        } else {
            const double *src = static_cast<const FacePos64 *>(sa.face_pos)[sil_f].v[corner];
            for (int j = 0; j < 4; ++j) v[j] = src[j];
        }
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
    }

    // ---- clipping, one plane at a time
    {
    const FrameConst &fc = kernargs<SetupKernArgs>().fc;         // phase: the six planes
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
        // the scratch is private to this wavefront and the LDS serves a wavefront's accesses in
        // program order: a wave barrier (no code motion across it) is all the ordering it takes
        __builtin_amdgcn_wave_barrier();
        if (clip_now) {
            int pos = incl - cnt;
            if (emit_cur && pos < MAX_POLY) { for (int j = 0; j < 4; ++j) s_poly[grp][pos][j] = v[j]; ++pos; }
            if (emit_int && pos < MAX_POLY) { for (int j = 0; j < 4; ++j) s_poly[grp][pos][j] = ipt[j]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (clip_now) {
            n = min(total, MAX_POLY);
            if (gl < n) for (int j = 0; j < 4; ++j) v[j] = s_poly[grp][gl][j];
        }
    }
    }
    const bool alive = n >= 3;                           // obj/triangular.py:322-323
    SETUP_ARGS();                                                // phase: projection, box, record

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
    // The quad's record sits at its silhouette index (the wavefront's ONE atomic on n_quads, requested before the set-up
    // began, came back long ago): a slot of its own from a second counter cost a returning atomic per quad, and ~1 200
    // of those on one cache line within a few microseconds are served one after the other, 12 ns each
    // (tools/micro/atomic_same_addr.hip) -- up to 14 us at the end of every chain.  Its work items of 64 tiles
    // (kernels_bin.h; all lanes take part) take one more returning atomic per wavefront, on one of WORK_SHARDS cursors.
    const uint32_t s_idx = (uint32_t)__shfl((int)s_base_raw, 0) + s_rank;
    const uint32_t slot = s_idx;
    const uint32_t chunks = (boxed && gl == 0) ? quad_chunks(fc, bx0, bx1, by0, by1) : 0u;
    WorkSlot ws;
    const bool any_work = reserve_work_items(bins, chunks, ws);
    {
        const unsigned long long drawn = __ballot(boxed && gl == 0);           // (a statistic: nobody waits for it)
        if (lane == 0 && drawn) atomicAdd(&sa.ctr->n_quads_drawn, (uint32_t)__popcll(drawn));
    }
    if (any_work) fill_work_items(bins, ws, slot < sa.quad_cap ? (WORK_QUAD | slot) : WORK_NONE, chunks);
    if (!boxed) return;
    if (slot >= sa.quad_cap) { if (gl == 0) atomicOr(&sa.ctr->overflow, 16u); return; }

    QuadRec &q = sa.quads[slot];
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
        q.edge = (int32_t)s_idx;                  // (the silhouette list's atomic has had the whole set-up to come back)
        q.x0 = (int16_t)bx0; q.x1 = (int16_t)bx1; q.y0 = (int16_t)by0; q.y1 = (int16_t)by1;
        q.pad[0] = q.pad[1] = q.pad[2] = 0;
    }
}

// Edge workgroup: one unique undirected edge per lane.  An edge is on the silhouette when an odd
// number of its incident light-facing faces toggled it; it keeps the orientation of the last
// such face in face order (set add/discard semantics of obj/triangular.py:294-302).  The
// incident faces' normals sit in the edge record, so the test is one 64-byte load and two dots.
// The scene's edges are stored in a scrambled order (host, build_edge_table): the silhouette of a
// mesh runs along consecutive vertex indices, and without that a wavefront would find dozens
// of silhouette edges among its 64 and set their quads up four at a time while the rest of the
// device idles.
__device__ __forceinline__ void edge_block(uint32_t block)
{
    SETUP_ARGS();
    __shared__ double s_poly[SETUP_BLOCK / WAVE][WAVE / QS_LANES][MAX_POLY + 4][4];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    // How many edges a lane looks at.  A wavefront sets its silhouette edges up four at a time, one round of 10-15 us
    // after the other, so what matters is how many it FINDS:
    //   small scenes (edge_spread s > 0): a mesh of a few thousand edges of which one in ten is on the silhouette (c3:
    //     7 500 edges, 120 wavefronts with five or six each) spent two or three rounds -- only every (1 << s)-th lane
    //     takes an edge, and there are that many more wavefronts;
    //   large scenes (EDGE_DENSE): one edge in 250 is on the silhouette, the launch holds more wavefronts than fit at
    //     once (c4: 3 128 of faces + 4 688 of edges against 5 120 places) and the edge wavefronts that wait for a place
    //     start their chain late -- two edges per lane halve their number and still find five in one wavefront less
    //     than once per frame.
    const uint32_t spread = ka_.edge_spread;
    const bool dense = spread == EDGE_DENSE;
    const uint32_t slot_in_grid = block * blockDim.x + threadIdx.x;
    int e[2];
    bool want[2];
    if (dense) {
        e[0] = (int)(block * 2u * blockDim.x + threadIdx.x); e[1] = e[0] + (int)blockDim.x;
        want[0] = e[0] < fc.n_edges; want[1] = e[1] < fc.n_edges;
    } else {
        e[0] = (int)(slot_in_grid >> spread); e[1] = 0;
        want[0] = (slot_in_grid & ((1u << spread) - 1u)) == 0 && e[0] < fc.n_edges; want[1] = false;
    }
    bool sil[2] = { false, false };
    uint32_t last[2] = { 0, 0 };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!want[i]) continue;
        EdgeRec r;
        if (fc.edge_compact) {
            const EdgeRec32 c = reinterpret_cast<const EdgeRec32 *>(sa.edges)[e[i]];
            r.inc[0] = c.inc[0]; r.inc[1] = c.inc[1];
            r.extra_off = r.extra_cnt = 0;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 3; ++j) r.n[a][j] = (double)c.n[a][j];
        } else {
            r = sa.edges[e[i]];
        }
        uint32_t cnt = 0;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (r.inc[a] != 0xffffffffu &&
                chain3(r.n[a][0], r.n[a][1], r.n[a][2], fc.light_pos[0], fc.light_pos[1], fc.light_pos[2]) > 0) {
                ++cnt; last[i] = r.inc[a];
            }
        }
        for (uint32_t k = 0; k < r.extra_cnt; ++k) {
            const uint32_t inc = sa.edge_inc[r.extra_off + k];
            const double *fn = sa.face_n + (size_t)(inc >> 2) * 4;
            if (chain3(fn[0], fn[1], fn[2], fc.light_pos[0], fc.light_pos[1], fc.light_pos[2]) > 0) { ++cnt; last[i] = inc; }
        }
        sil[i] = (cnt & 1u) != 0;
    }
    unsigned long long todo0 = __ballot(sil[0]), todo1 = __ballot(sil[1]);
    if (!(todo0 | todo1)) return;
    // the wavefront's silhouette edges get consecutive list slots with one atomic; its answer is first needed
    // at the end of the quad set-up (the records' slots), so it is not waited for here
    const uint32_t found0 = (uint32_t)__popcll(todo0), found = found0 + (uint32_t)__popcll(todo1);
    uint32_t base_raw = 0;
    if (lane == 0) base_raw = atomicAdd(&sa.ctr->n_quads, found);
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t my_rank0 = (uint32_t)__popcll(todo0 & below), my_rank1 = found0 + (uint32_t)__popcll(todo1 & below);
    // quad set-up, four silhouette edges per round (one per 16-lane group): the lanes' first edges, then their second
    const int grp = lane / QS_LANES;
    while (todo0 | todo1) {
        int src = -1, set = 0;
        {
            const int c0 = (int)__popcll(todo0);
            unsigned long long t = grp < c0 ? todo0 : todo1;
            const int nth = grp < c0 ? grp : grp - c0;
            set = grp < c0 ? 0 : 1;
            for (int g = 0; g <= nth && t; ++g) {                   // this group takes the nth set bit
                const int b = __ffsll((long long)t) - 1;
                t &= t - 1;
                if (g == nth) src = b;
            }
        }
        for (int g = 0; g < WAVE / QS_LANES; ++g) {                  // the round's (up to) four leave the lists
            if (todo0) todo0 &= todo0 - 1;
            else if (todo1) todo1 &= todo1 - 1;
        }
        const bool have = src >= 0;
        const int from = have ? src : 0;
        const uint32_t l0 = (uint32_t)__shfl((int)last[0], from), l1 = (uint32_t)__shfl((int)last[1], from);
        const uint32_t r0 = (uint32_t)__shfl((int)my_rank0, from), r1 = (uint32_t)__shfl((int)my_rank1, from);
        const uint32_t ls = set ? l1 : l0, rank = set ? r1 : r0;
        quad_setup_group(have, (int)(ls >> 2), (int)(ls & 3u), base_raw, rank, s_poly[wv]);
    }
    const uint32_t base = (uint32_t)__shfl((int)base_raw, 0);
    const SetupArgs &sa2 = kernargs<SetupKernArgs>().sa;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t my_slot = base + (i ? my_rank1 : my_rank0);
        if (sil[i] && my_slot < sa2.quad_cap) {
            sa2.sil_edges[my_slot * 2 + 0] = (int32_t)(last[i] >> 2);      // the host maps the face back to its model
            sa2.sil_edges[my_slot * 2 + 1] = (int32_t)(last[i] & 3u);
        }
    }
}

// First launch of the frame: workgroup 0 puts the frame's tiles in order (a serial walk of some microseconds,
// hidden behind the others), workgroups [1, 1 + face_blocks) set faces up, the rest look at edges.
#ifndef MR_SETUP_WAVES
#define MR_SETUP_WAVES 5
#endif
template <bool PRE_XFORM>
__global__ void __launch_bounds__(SETUP_BLOCK, MR_SETUP_WAVES)
k_setup(const SetupKernArgs)            // read through kernargs<SetupKernArgs>(), phase by phase
{
    uint32_t face_blocks;
    {
        const SetupKernArgs &ka = kernargs<SetupKernArgs>();
        if (blockIdx.x == 0) { order_tiles_block(ka.sa.tile_class, ka.sa.order, ka.fc.tiles_x * ka.fc.tiles_y); return; }
        face_blocks = ka.face_blocks;
    }
    const uint32_t b = blockIdx.x - 1;
    if (b < face_blocks) tri_setup_block<PRE_XFORM>(b);
    else edge_block(b - face_blocks);
}

// Second launch: the leftover survivor counts (a few workgroups, first so that their
// latency-bound work is in flight early) and the tile lists of the large primitives.
__global__ void __launch_bounds__(256)
k_bin_work(const FrameConst fc, const BinArgs bins, const uint32_t *__restrict__ count_list,
           TriRec *__restrict__ tris, const TriClip *__restrict__ clips,
           uint8_t *__restrict__ status, Counters *__restrict__ ctr, uint32_t count_blocks)
{
    if (blockIdx.x < count_blocks)
        tri_count_body(fc, count_list, tris, clips, status, ctr, blockIdx.x, count_blocks);
    else
        bin_work_body(fc, bins, blockIdx.x - count_blocks, gridDim.x - count_blocks);
}

}  // namespace mr
