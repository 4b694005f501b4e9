// rast_types.h -- records shared by the host side of libmi355rast and its gfx950 kernels.
//
// Vocabulary follows the reference renderer: faces (triangles), silhouette edges, shadow
// quads, fragments, z / stencil / frame buffers.  A "tile" is the block of TILE_W x TILE_H
// pixels owned by one workgroup of the tile kernel, which keeps the tile's z-buffer, winner map
// and stencil counts on chip from the first triangle to the finished uint8 pixels.
#pragma once

#include <stdint.h>

namespace mr {

constexpr int TILE_W = 16;
constexpr int TILE_H = 16;
constexpr int TILE_PX = TILE_W * TILE_H;    // one thread per pixel in the tile kernel
constexpr int WAVE = 64;
// A (triangle, tile) pair whose pixel box inside the tile holds more samples than this is
// evaluated one pixel per thread ("big" pair); smaller ones a few lanes per triangle.
constexpr int BIG_PAIR_PX = 24;
constexpr int BIN_CLASSES = 3;              // small triangle pairs, big triangle pairs, shadow quads

constexpr int MAX_POLY = 12;   // a quad clipped by six planes has at most 4 + 6 vertices

// Status codes are the reference's Errors flag values (obj/triangular.py:15-20).
enum : uint8_t {
    FACE_OK = 0, FACE_BACK_FACE_CULLING = 1, FACE_WRONG_MIN_MAX = 2,
    FACE_EMPTY_B = 4, FACE_EMPTY_Z = 8, FACE_CLIPPED = 16
};

// per-face flags copied from the owning Model
enum : uint8_t { FF_CLIP = 1, FF_VERTS_F32 = 2, FF_HAS_NORMALS = 4, FF_HAS_UV = 8,
                 FF_NO_DEPTH = 16 };    // Model.depth_test == False: tested against z, never written to it

// TriRec.flags (bits 8-15 carry the face flags above)
enum : uint32_t {
    TF_CLIP = 1,          // Model.clip: per-fragment frustum test against both cameras
    TF_SINGLE_BOX = 2,    // pixel box holds exactly one sample  -> NumPy's (1,K)@(K,) is a dot
    TF_SINGLE_Z = 4       // exactly one fragment survives coverage + clip -> z is a dot
};

// Per-frame constants, passed to every kernel by value (kernarg segment, scalar loads).
struct FrameConst {
    int32_t width, height;
    int32_t system;              // +1 RH, -1 LH
    int32_t backface_culling;
    int32_t light_type;
    int32_t flags;
    int32_t band_y0, band_y1;    // screen rows [band_y0, band_y1) this device owns (y up, unflipped)
    int32_t tiles_x, tiles_y;    // tile grid this device owns: tiles_y local tile rows
    int32_t tile_y0, tile_step;  // local tile row l is frame tile row tile_y0 + l * tile_step
    int32_t out_tile_rows;       // > 0: striped output layout with this many tile rows per device (see out_row)
    int32_t n_vertices, n_faces, n_edges, n_materials;
    int32_t same_clip;           // debug_mvp == mvp bit for bit: the second clip test repeats the first
    int32_t pos32;               // the scene's static face records hold float32 corners (FacePos32[], else FacePos64[])
    int32_t edge_compact;        // the scene's edge table is EdgeRec32[] (else EdgeRec[])
    double mvp[16], viewport[16], debug_mvp[16];
    double planes[24];
    double two_nf, f_plus_n, f_minus_n;      // linearize_z constants (obj/core.py:226-228)
    double camera_pos[3];
    double light_pos[3], light_dir[3], light_color[3], light_ambient[3];
    double specular_strength, att_constant, att_linear, att_quadratic;
    double spot_edge0, spot_edge1;
    float background[3];
    uint32_t background_u8;      // finalised background r | g << 8 | b << 16 | 1 << 24 (0 = not given)
    int32_t sky_tri[12];         // skybox triangles' integer screen vertices [t][v][xy]
    int32_t sky_size;            // cubemap face size
    int32_t has_no_depth;        // some model has depth_test == False (the tile kernel then runs its second look at the big pairs)
    double sky_rays[18];         // their un-projected corner rays [t][v][xyz]
    // cluster culling (ClusterRec), CC_* bits: 0 off; CC_BOX the clusters' screen boxes against the screen and this
    // device's rows; CC_CONE also the back-face cone, a face being culled when  s * n . (a - cull_eye) > 0  with
    // s = -1 if CC_NEGATIVE else +1; CC_COUNT count the culled clusters (a diagnostic)
    int32_t cluster_cull;
    int32_t pad_cc;
    double cull_eye[3];          // the camera's centre of projection in world space (the null vector of MVP's x, y, w columns)
};

// Output of the optional stand-alone vertex kernel (k_vertex_mfma): everything
// obj/triangular.py:36-45 derives per face corner, once per unique vertex.
struct alignas(16) VertexOut {
    double sx, sy, sz, depth;    // screen x, y, z and 1/clip.w
    double zlin;                 // linearize_z(sz)
    int32_t safe;                // strictly inside both clip volumes with margin (see xform_vertex)
    int32_t pad;
};
static_assert(sizeof(VertexOut) == 48, "VertexOut layout");
struct alignas(16) VertexClip {
    double clip[4];              // v @ camera.MVP
    double clipd[4];             // v @ debug_camera.MVP
};

// Triangle set-up record walked by the tile kernel: the per-face constants of
// obj/transformation.py:12-32 and obj/triangular.py:96-97.  112 bytes = 7 x 16 so a lane can
// fetch a whole record with seven 16-byte loads.
struct alignas(16) TriRec {
    double ax, ay;               // screen position of corner a
    double v0x, v0y, v1x, v1y;   // b - a, c - a
    double zl0, zl1, zl2;        // linearised z of the corners
    float d00, d01, d11, inv_den;
    int16_t x0, x1, y0, y1;      // half-open pixel box
    uint32_t flags;
    int32_t face;                // global face index
    int32_t material;            // global material index of the face (first corner's group, obj/core.py:125)
    uint32_t pad;
    double dp[3];                // 1 / clip.w per corner (perspective-correct barycentrics, obj/core.py:155-160)
    double pad2;
};
static_assert(sizeof(TriRec) == 144, "TriRec layout");

// STATIC per face, built when the scene is committed (k_face_static): what the index rows point at, gathered once
// per scene instead of once per frame -- the three corners' world positions (what the set-up kernel transforms and
// shading interpolates), the face's material and flags, and its texture coordinates and vertex normals.  A face's
// set-up is then ONE contiguous record away (round 2: index row -> three vertex gathers, and for the faces that
// survive the cull three uv and three normal gathers more and a 176-byte attribute record written per frame:
// a chain of dependent trips to memory that was most of the set-up kernel's 25 us), and shading reads the same
// records instead of a per-frame copy.  Positions are float32 when every model's vertices are (FrameConst::pos32).
// STATIC per 64 consecutive faces (one wavefront of a face workgroup), built when the scene is committed: the faces'
// world-space bounding box (float32, rounded outwards) and the cone their unit normals lie in.  A wavefront of k_setup
// whose cluster is entirely off the screen, off this device's rows or -- when the frame culls back faces -- turned
// away from the camera ends before it has read a face (kernels_geometry.h, cluster_culled): conservative, so the faces
// that are set up, and their order-free results, are exactly those of the per-face tests.
struct alignas(16) ClusterRec {
    float lo[3], hi[3];
    float axis[3];               // cone axis (unit); cos_half < -1: no cone (a degenerate face, or normals more than 90 degrees apart)
    float cos_half, sin_half;    // every face normal n of the cluster has n . axis >= cos_half
    uint32_t pad[5];
};
static_assert(sizeof(ClusterRec) == 64, "ClusterRec layout");
constexpr int CLUSTER_FACES = 64;
constexpr int CC_BOX = 1, CC_CONE = 2, CC_NEGATIVE = 4, CC_COUNT = 8;

template <class T>
struct alignas(16) FacePosT {
    T v[3][4];                   // world-space corners (x, y, z, w)
    int32_t material;            // global material index (first corner's group, obj/core.py:125)
    uint32_t flags;              // FF_* of the owning model
    uint32_t pad[2];
};
typedef FacePosT<float> FacePos32;
typedef FacePosT<double> FacePos64;
static_assert(sizeof(FacePos32) == 64 && sizeof(FacePos64) == 112, "FacePos layout");
struct alignas(16) FaceAttr {
    float uv[3][2];
    float n[3][3];
    float pad;
};
static_assert(sizeof(FaceAttr) == 64, "FaceAttr layout");

// Both cameras' clip-space corners; written only for faces whose fragments need the clip test.
struct alignas(16) TriClip {
    double clip[3][4];
    double clipd[3][4];
};

// Shadow quad after extrusion, clipping and projection (obj/triangular.py:319-349).
struct alignas(16) QuadEdge { double sx, sy, ex, ey; };   // vertex i and the edge vector to vertex i+1
struct alignas(16) QuadRec {
    double nx, ny, nz, d;        // plane through the first three vertices
    int16_t x0, x1, y0, y1;      // half-open pixel box
    int32_t n;                   // vertex count (>= 3)
    int32_t is_front;
    int32_t edge;                // silhouette-list entry it came from
    uint32_t pad[3];
    QuadEdge e[MAX_POLY];
};
static_assert(sizeof(QuadRec) == 64 + 32 * MAX_POLY, "QuadRec layout");

// Static per unique undirected edge (built when the scene is committed): its first two incident
// (face, corner) pairs in face order with those faces' unit normals inline -- the light-facing
// test of obj/triangular.py:294-295 is normal . light.position > 0 and the normal does not
// depend on the frame -- and, for the rare edge with more incidences, a range of the spill array.
struct alignas(16) EdgeRec {
    uint32_t inc[2];             // face * 4 + corner, 0xffffffff = none
    uint32_t extra_off, extra_cnt;
    double n[2][3];
};
static_assert(sizeof(EdgeRec) == 64, "EdgeRec layout");
// The compact form, used when every model's vertices are float32 (then the normals ARE float32 values,
// core.py:127-130) and no edge has more than two incident faces (any manifold mesh): half the bytes of
// what is the largest stream of the set-up kernel.
struct alignas(16) EdgeRec32 {
    uint32_t inc[2];
    float n[2][3];
};
static_assert(sizeof(EdgeRec32) == 32, "EdgeRec32 layout");

struct Texture {
    const float *rgb;
    int32_t h, w;
};

struct Material {
    double kd[3];
    double ks255[3];
    double ns;
    int32_t tex_kd, tex_norm, tex_ks, norm_tangent;
    // the three maps' headers, copied in when the scene is committed (rgb == nullptr: no map), so
    // that a pixel reaches its texel in one dependent load after the material instead of two
    Texture map_kd, map_norm, map_ks;
};

// Device-side counters of one frame; copied back when somebody asks.  The words that many
// wavefronts add to during a frame sit on cache lines of their own: atomics to one line retire
// at ~0.3 per ns on MI355X whatever the address in it (tools/micro/atomic_bench.hip).
constexpr int WORK_SHARDS = 16;
struct alignas(128) Counters {
    unsigned long long frag_tri, frag_quad, covered_px, lit_px, stencil_updates;
    unsigned int n_valid_tris, tri_bin_total, bin_total;
    unsigned int overflow;       // bit0-2: a tile's small / big / quad list, bit3: work list, bit4: quad list
    unsigned int max_list[BIN_CLASSES];       // longest list seen by an overflowing tile, per class
    unsigned int pad0[15];
    unsigned int n_quads;        unsigned int pad1[31];     // silhouette edges
    unsigned int n_quads_drawn;  unsigned int pad2[31];     // quads that got a record
    unsigned int n_count;        unsigned int pad3[31];     // faces whose survivor count is left to k_bin_work
    // (large primitive, 64-tile chunk) work items: WORK_SHARDS cursors, one cache line each, every one with its own
    // stretch of the work list.  Returning atomics on ONE line are served one after the other, ~12 ns each on MI355X
    // (tools/micro/atomic_same_addr.hip): the thousand wavefronts that reserve items within a few microseconds of each
    // other queued for 12 us behind a single cursor.
    struct alignas(128) WorkCursor { unsigned int n; unsigned int pad[31]; } work[WORK_SHARDS];
};
static_assert(sizeof(Counters) == 128 * (4 + WORK_SHARDS), "Counters layout");

// What must outlive a frame's counters.  The counters are double-buffered by frame parity and a frame's tile
// kernel clears the block of the frame after it, i.e. the block of the frame BEFORE it: the overflow verdicts of a
// frame enqueued without host synchronisation would be gone two frames later.  Before clearing, the tile kernel
// folds them into this record (one per frame slot), which only the host resets, after it has acted on it.
struct alignas(64) Sticky {
    unsigned int overflow;                    // OR of Counters::overflow
    unsigned int max_list[BIN_CLASSES];       // maxima of the rest
    unsigned int n_work, n_quads, n_quads_drawn;
    unsigned int pad[9];
};
static_assert(sizeof(Sticky) == 64, "Sticky layout");

}  // namespace mr
