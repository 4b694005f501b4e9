// mi355rast.hip -- C ABI (include/mi355rast.h) and host-side frame orchestration.
//
// One process drives one GPU.  A scene keeps its static arrays (vertices, attributes, index
// arrays, textures, the unique-edge table with the incident faces' normals) resident in HBM; a
// frame is THREE kernels on one HIP stream, with no host synchronisation in between:
//
//   k_setup     faces: vertex transform, cull, set-up records, own tile lists
//               edges: silhouette search, shadow-quad extrusion / clip / projection
//   k_bin_work  tile lists of the large primitives (floor triangles, shadow quads)
//   k_tile      per 16x16 tile: coverage + z + winner, stencil count, shading, finalise -> uint8
//   (D2H of the uint8 rows for mr_render)
//
// Per-frame work buffers live in a "frame slot".  Every stream a caller renders on gets its own
// slot, so frames enqueued on different streams are independent and may overlap on the device.
//
// Built for gfx950 only, with -ffp-contract=off (see rast_math.h).
#include "../../include/mi355rast.h"
#include "host_overlay.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "rast_types.h"
#include "kernels_geometry.h"
#include "kernels_tile.h"
#include "kernels_overlay.h"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg)
{
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(MR_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

hipStream_t g_stream = nullptr;
bool g_initialised = false;

// growable device allocation
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

struct ModelInfo {
    int32_t vert_off, n_verts, uv_off, n_uv, normal_off, n_normals, face_off, n_faces, mat_off, n_mats;
};

struct EdgeKey {
    uint64_t key;      // (lo << 32) | hi of the two global vertex indices
    uint32_t inc;      // face * 4 + corner
};

// Event marks of one frame:
//   0 start | 1 after k_vertex_mfma (optional) | 2 after k_setup | 3 after k_bin_work | 4 after k_tile | 5 device->host copy
constexpr int EVENT_RING = 512, N_MARKS = 6;     // (bench.py marks one frame in 13: ~40 marked frames per stream to average over)

// Everything one in-flight frame writes.
struct FrameSlot {
    hipStream_t stream = nullptr;                 // the stream this slot serves
    DevBuf d_vout, d_vclip, d_tris, d_clips, d_status, d_count_list, d_quads, d_sil, d_counters;
    DevBuf d_bin_count, d_items[mr::BIN_CLASSES], d_work, d_tile_stats, d_hist, d_split;
    DevBuf d_z, d_winner, d_stencil, d_frame, d_out;
    // capacities this slot's buffers were last bound with (the scene holds the current ones)
    uint32_t bin_cap[mr::BIN_CLASSES] = { 0, 0, 0 }, work_cap = 0, quad_cap = 0;
    int bins_zeroed_for = 0;

    mr::Counters *h_counters = nullptr;           // pinned: the last frame's counters, then (h_sticky) the slot's sticky record
    mr::Sticky *h_sticky = nullptr;
    hipEvent_t ev_ring[EVENT_RING][N_MARKS] = {};
    uint8_t ev_marks[EVENT_RING] = {};            // 0: the frame recorded no events, 1: frame + tile kernel, 2: every stage
    hipEvent_t *ev = ev_ring[0];
    uint64_t frames_enqueued = 0;
    uint64_t last_serial = 0;                // the scene's frame serial when this slot last took a frame
    bool events_ok = false;

    // this slot's device copy of the scene's overlay lists: ONE buffer filled with one asynchronous copy from a
    // page-locked staging buffer on the slot's stream (behind the slot's earlier frames, in front of the next one),
    // and the overlay kernel's scratch
    struct OverlayCopy {
        DevBuf lists, scratch;
        void *staging = nullptr;
        size_t staging_cap = 0;
        size_t off[6] = {};                  // byte offsets of z, targets, segments, tile mask (+ slot ids, touched pixels) in `lists`
        bool with_slots = false;             // `lists` also holds the slot lists of a split frame
        size_t state_entries = 0;            // entries of win / any in `scratch` (zero between frames)
        uint64_t serial = 0;                 // the scene's ov_serial this copy holds
        hipEvent_t copied = nullptr;         // the last copy out of the staging buffer
    } ov;

    mr_frame_desc last_frame = {};
    int last_n_tiles = 0;
    bool last_ordered = false;               // the last frame's tile kernel followed the order buffer (else row-major)
    bool overlay_deferred = false;           // the last enqueue_frame left the overlay to finish_overlay
    bool have_frame = false, stats_reduced = false;
    bool last_copied = false;                // the last frame was followed by a timed device-to-host copy (mr_render, mr_render_async)

    void reset_caps() { bins_zeroed_for = 0; have_frame = false; }
    void release()
    {
        DevBuf *bufs[] = { &d_vout, &d_vclip, &d_count_list, &d_tris, &d_clips, &d_status, &d_quads, &d_sil, &d_counters,
                           &d_bin_count, &d_items[0], &d_items[1], &d_items[2], &d_work, &d_tile_stats, &d_hist, &d_split,
                           &d_z, &d_winner, &d_stencil, &d_frame, &d_out };
        for (DevBuf *b : bufs) b->release();
        ov.lists.release(); ov.scratch.release();
        if (ov.staging) (void)hipHostFree(ov.staging);
        if (ov.copied) (void)hipEventDestroy(ov.copied);
        ov.staging = nullptr; ov.staging_cap = 0; ov.serial = 0; ov.copied = nullptr;
        if (events_ok) {
            for (auto &set : ev_ring) for (auto &e : set) (void)hipEventDestroy(e);
            (void)hipHostFree(h_counters);
            events_ok = false;
        }
    }
    // the counters are double-buffered by frame parity: a frame's tile kernel clears the next frame's
    mr::Counters *ctr(uint64_t frame) const { return d_counters.as<mr::Counters>() + (frame & 1); }
    // overflow verdicts of the frames before the last one (rast_types.h, Sticky): behind the two counter blocks
    mr::Sticky *sticky() const { return reinterpret_cast<mr::Sticky *>(d_counters.as<mr::Counters>() + 2); }
};

constexpr int MAX_SLOTS = 32;

}  // namespace

struct mr_scene {
    // ---- host staging of the static scene (concatenated over models, indices made global)
    std::vector<double> verts;
    std::vector<float> uv, normals;
    std::vector<int32_t> faces;
    std::vector<uint8_t> face_flags;
    std::vector<mr::Material> materials;
    std::vector<mr::Texture> textures;       // device pointers
    std::vector<void *> texture_allocs;
    std::vector<ModelInfo> models;
    std::vector<int32_t> edge_ids;            // per face corner: raw vertex identity for silhouette edges (made unique per model)
    std::vector<int32_t> edge_raw;            // the same as the caller passed it (for mr_read_silhouette)
    std::vector<mr::EdgeRec> edges;           // unique undirected edges, scrambled order
    std::vector<uint32_t> edge_inc;           // incidences beyond an edge's first two
    bool dirty = true;

    // ---- device copies of the static scene
    DevBuf d_verts, d_uv, d_normals, d_faces, d_face_flags, d_materials, d_textures, d_edges, d_edge_inc, d_face_n;
    DevBuf d_edges32;                        // the compact edge table, when the scene allows it
    bool edge_compact = false;
    DevBuf d_face_pos, d_face_attr;          // static per face (rast_types.h, FacePosT / FaceAttr), built by commit()
    DevBuf d_clusters;                       // static per 64 faces (rast_types.h, ClusterRec), built by commit()
    bool pos32 = false;                      // d_face_pos holds FacePos32 (every model's vertices are float32)
    bool has_no_depth = false;               // some model has depth_test == False (what a frame asks once per scene, not once per frame)
    // debug-frustum overlay: the level lists (host_overlay.h, OverlayLevels) in ONE device buffer, filled with one
    // copy from a page-locked staging buffer on the library's stream, and the kernel's scratch
    // debug-frustum overlay: the lines' points as built on the host (five targets and a depth per point, segment by
    // segment); every frame slot keeps its own device copy (FrameSlot::ov), brought up to date when a frame of that
    // slot draws the overlay
    // mr_scene_set_overlay_cameras only leaves its arguments here; the lists are built when first needed (realize_overlay)
    // -- for mr_render / mr_render_async AFTER the frame's three kernels have been launched, so that the host walks the
    // lines while the device renders (the lists' only early use, the tile kernel's tap mask, is given up for that frame)
    struct OvPending {
        bool set = false;
        double corners[32], planes[24], mvp[16], viewport[16], near_ = 0, far_ = 0;
        int32_t inside = 0, height = 0, width = 0;
    } ov_pending;
    int32_t ov_height = 0, ov_width = 0;     // the frame the lists were built for
    int32_t ov_points = 0, ov_segments = 0;
    uint64_t ov_serial = 0;                  // bumped whenever the lists change
    std::vector<int32_t> ov_target;          // (5, n_points) pixel row * width + col of every target
    std::vector<double> ov_z;
    std::vector<int32_t> ov_seg;             // first point, number of points per segment
    std::vector<uint8_t> ov_tile_mask;       // the 16x16 tiles that hold a target
    // the same targets as slots of the list of touched pixels, for a frame assembled from several devices (built
    // when first asked for: build_overlay_slots)
    std::vector<int32_t> ov_slot_of, ov_touched;
    uint64_t ov_slots_serial = 0;            // the ov_serial the slot lists were built for
    mr_host::OverlaySlotWork ov_slot_work;
    DevBuf d_sky;                            // cubemap texels, uint8 (6, S, S, 3)
    DevBuf d_gamma;                          // GAMMA_LUT_SIZE float32 thresholds of the finalise step function
    int32_t sky_size = 0;

    // ---- lanes of mr_render_async: a stream of the library's own each, and what is in flight on it
    struct Lane { hipStream_t stream = nullptr; bool busy = false; } lanes[MR_ASYNC_LANES];

    // ---- frame slots, one per stream that has rendered this scene
    std::vector<std::unique_ptr<FrameSlot>> slots;
    FrameSlot *last = nullptr;               // slot of the most recently enqueued frame
    uint64_t frame_serial = 0;               // frames enqueued on any stream
    mr_stats stats = {};
    int n_silhouette = 0;
    // Capacities of the per-frame work lists, shared by all slots: what one frame learnt (a tile with
    // a longer list, more silhouette edges) holds for the frames rendered on other streams too.
    uint32_t bin_cap[mr::BIN_CLASSES] = { 512u, 128u, 256u };   // entries per tile and class
    uint32_t work_cap = 1u << 18, quad_cap = 0;
    void reset_caps() { bin_cap[0] = 512u; bin_cap[1] = 128u; bin_cap[2] = 256u; work_cap = 1u << 18; quad_cap = 0; }
};

namespace {

int ensure_init()
{
    if (g_initialised) return MR_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MR_E_DEVICE, "no HIP device visible: libmi355rast has no CPU fallback");
    HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_initialised = true;
    return MR_OK;
}

template <class T>
int upload(DevBuf &buf, const std::vector<T> &v, hipStream_t s)
{
    HIP_TRY(buf.ensure(std::max<size_t>(v.size() * sizeof(T), 16)));
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return MR_OK;
}

// Unique undirected edges with their incident (face, corner) pairs in face order: what the
// reference's per-model set of Edge objects (obj/triangular.py:286-302) reduces to.  An edge is
// identified by the RAW vertex ids of its corners (mr_model_desc.edge_ids), as in the reference.
// The table is stored in a scrambled order (sorted by a hash of the edge): silhouettes run along
// consecutive vertex indices, and a wavefront of k_setup that found dozens of silhouette edges
// among its 64 would set their quads up four at a time while the rest of the device idles.
void build_edge_table(mr_scene *sc)
{
    const size_t nf = sc->faces.size() / 12;
    std::vector<EdgeKey> keys;
    keys.reserve(nf * 3);
    for (size_t f = 0; f < nf; ++f) {
        const int32_t *id = &sc->edge_ids[f * 3];
        for (int k = 0; k < 3; ++k) {
            uint32_t a = (uint32_t)id[k], b = (uint32_t)id[(k + 1) % 3];
            uint32_t lo = std::min(a, b), hi = std::max(a, b);
            keys.push_back({ ((uint64_t)lo << 32) | hi, (uint32_t)(f * 4 + k) });
        }
    }
    auto scramble = [](uint64_t k) {                   // splitmix64 finaliser: a bijection
        k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull; k ^= k >> 27; k *= 0x94d049bb133111ebull; k ^= k >> 31;
        return k;
    };
    for (EdgeKey &e : keys) e.key = scramble(e.key);
    std::sort(keys.begin(), keys.end(), [](const EdgeKey &x, const EdgeKey &y) {
        return x.key != y.key ? x.key < y.key : x.inc < y.inc;
    });
    sc->edges.clear();
    sc->edge_inc.clear();
    for (size_t i = 0; i < keys.size();) {
        size_t j = i;
        while (j < keys.size() && keys[j].key == keys[i].key) ++j;
        mr::EdgeRec r;
        std::memset(&r, 0, sizeof r);
        r.inc[0] = keys[i].inc;
        r.inc[1] = j - i > 1 ? keys[i + 1].inc : 0xffffffffu;
        r.extra_off = (uint32_t)sc->edge_inc.size();
        r.extra_cnt = j - i > 2 ? (uint32_t)(j - i - 2) : 0u;
        for (size_t k = i + 2; k < j; ++k) sc->edge_inc.push_back(keys[k].inc);
        sc->edges.push_back(r);
        i = j;
    }
}

// Finalise is uint8(frame ** 0.8 * 255) in float32 (obj/core.py:640): a monotone step function
// of the colour with 255 steps.  GAMMA_LUT[k] is the smallest float32 in [0, 1] whose step is
// >= k, found by bisection over the bit patterns against the host's own powf, so that k_shade
// can place a colour with one approximate exp2/log2 and two table compares and still return
// exactly what powf would (k_shade's gamma_u8).
std::vector<float> gamma_thresholds()
{
    auto step = [](float x) { return (int)(uint8_t)(powf(x, 0.8f) * 255.0f); };
    std::vector<float> lut(mr::GAMMA_LUT_SIZE);
    lut[0] = 0.0f;
    for (int k = 1; k < 256; ++k) {
        uint32_t lo = 0, hi = 0x3f800000u;      // step(0) = 0 < k <= 255 = step(1)
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            float x;
            memcpy(&x, &mid, 4);
            if (step(x) >= k) hi = mid; else lo = mid;
        }
        memcpy(&lut[k], &hi, 4);
    }
    lut[256] = INFINITY;
    return lut;
}

// The static cluster records (rast_types.h, ClusterRec): bounding box and normal cone of every 64 consecutive faces.
std::vector<mr::ClusterRec> build_clusters(const mr_scene *sc)
{
    const size_t nf = sc->faces.size() / 12, nc = (nf + mr::CLUSTER_FACES - 1) / mr::CLUSTER_FACES;
    std::vector<mr::ClusterRec> out(nc);
    auto down = [](double x) { float f = (float)x; return (double)f > x ? std::nextafter(f, -INFINITY) : f; };
    auto up = [](double x) { float f = (float)x; return (double)f < x ? std::nextafter(f, INFINITY) : f; };
    for (size_t c = 0; c < nc; ++c) {
        mr::ClusterRec r;
        std::memset(&r, 0, sizeof r);
        double lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY }, sum[3] = { 0, 0, 0 };
        bool boxed = true, coned = true;
        const size_t f0 = c * mr::CLUSTER_FACES, f1 = std::min(nf, f0 + mr::CLUSTER_FACES);
        std::vector<std::array<double, 3>> normals;
        normals.reserve(f1 - f0);
        for (size_t f = f0; f < f1; ++f) {
            const double *v[3];
            for (int k = 0; k < 3; ++k) {
                v[k] = &sc->verts[(size_t)sc->faces[f * 12 + k * 4] * 4];
                if (!(v[k][3] == 1.0)) boxed = false;                     // (a homogeneous coordinate other than 1: no box)
                for (int j = 0; j < 3; ++j) { lo[j] = std::min(lo[j], v[k][j]); hi[j] = std::max(hi[j], v[k][j]); if (!std::isfinite(v[k][j])) boxed = false; }
            }
            const double a[3] = { v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2] };
            const double b[3] = { v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2] };
            double n[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
            const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (!(l > 0) || !std::isfinite(l)) { coned = false; continue; }
            for (int j = 0; j < 3; ++j) { n[j] /= l; sum[j] += n[j]; }
            normals.push_back({ n[0], n[1], n[2] });
        }
        if (boxed) {
            for (int j = 0; j < 3; ++j) { r.lo[j] = down(lo[j]); r.hi[j] = up(hi[j]); }
        } else {
            for (int j = 0; j < 3; ++j) { r.lo[j] = NAN; r.hi[j] = NAN; }      // never culled: every comparison fails
        }
        r.cos_half = -2.f; r.sin_half = 1.f;
        const double sl = std::sqrt(sum[0] * sum[0] + sum[1] * sum[1] + sum[2] * sum[2]);
        if (coned && sl > 1e-6 * (double)(f1 - f0)) {
            double least = 1.0;
            for (const auto &n : normals) least = std::min(least, (n[0] * sum[0] + n[1] * sum[1] + n[2] * sum[2]) / sl);
            least -= 1e-6;
            if (least > 0.05) {                                             // a cone wider than ~87 degrees never culls anything
                for (int j = 0; j < 3; ++j) r.axis[j] = (float)(sum[j] / sl);
                // the axis as stored (float32) is not the axis the dots were taken with: 1e-6 covers it
                r.cos_half = (float)(least - 1e-6);
                r.sin_half = (float)std::min(1.0, std::sqrt(std::max(0.0, 1.0 - (double)r.cos_half * (double)r.cos_half)) + 1e-6);
            }
        }
        out[c] = r;
    }
    return out;
}

int commit(mr_scene *sc)
{
    if (!sc->dirty) return MR_OK;
    HIP_TRY(hipDeviceSynchronize());          // no frame may still be reading the old arrays
    if (!sc->d_gamma.p) {
        const std::vector<float> lut = gamma_thresholds();
        HIP_TRY(sc->d_gamma.ensure(lut.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(sc->d_gamma.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    build_edge_table(sc);
    for (mr::Material &m : sc->materials) {
        const mr::Texture none = { nullptr, 0, 0 };
        auto header = [&](int32_t id) { return id >= 0 && id < (int32_t)sc->textures.size() ? sc->textures[id] : none; };
        m.map_kd = header(m.tex_kd); m.map_norm = header(m.tex_norm); m.map_ks = header(m.tex_ks);
    }
    int rc;
    if ((rc = upload(sc->d_verts, sc->verts, g_stream))) return rc;
    if ((rc = upload(sc->d_uv, sc->uv, g_stream))) return rc;
    if ((rc = upload(sc->d_normals, sc->normals, g_stream))) return rc;
    if ((rc = upload(sc->d_faces, sc->faces, g_stream))) return rc;
    if ((rc = upload(sc->d_face_flags, sc->face_flags, g_stream))) return rc;
    if ((rc = upload(sc->d_materials, sc->materials, g_stream))) return rc;
    if ((rc = upload(sc->d_textures, sc->textures, g_stream))) return rc;
    if ((rc = upload(sc->d_edges, sc->edges, g_stream))) return rc;
    if ((rc = upload(sc->d_edge_inc, sc->edge_inc, g_stream))) return rc;
    // static per scene: the faces' unit normals (light-facing test), copied into the edge records
    const int nf = (int)(sc->faces.size() / 12), ne = (int)sc->edges.size();
    HIP_TRY(sc->d_face_n.ensure(std::max<size_t>((size_t)nf * 4 * sizeof(double), 16)));
    if (nf > 0)
        hipLaunchKernelGGL(mr::k_face_normals, dim3((nf + 255) / 256), dim3(256), 0, g_stream, nf, sc->d_faces.as<int32_t>(),
                           sc->d_face_flags.as<uint8_t>(), sc->d_verts.as<double>(), sc->d_face_n.as<double>());
    if (ne > 0)
        hipLaunchKernelGGL(mr::k_edge_normals, dim3((ne + 255) / 256), dim3(256), 0, g_stream, ne, sc->d_edges.as<mr::EdgeRec>(),
                           sc->d_face_n.as<double>());
    // the static face records: float32 corners when every model's vertices are float32
    sc->pos32 = true;
    sc->has_no_depth = false;
    for (uint8_t ff : sc->face_flags) {
        if (!(ff & mr::FF_VERTS_F32)) sc->pos32 = false;
        if (ff & mr::FF_NO_DEPTH) sc->has_no_depth = true;
    }
    HIP_TRY(sc->d_face_pos.ensure(std::max<size_t>((size_t)nf * (sc->pos32 ? sizeof(mr::FacePos32) : sizeof(mr::FacePos64)), 16)));
    HIP_TRY(sc->d_face_attr.ensure(std::max<size_t>((size_t)nf * sizeof(mr::FaceAttr), 16)));
    {
        const std::vector<mr::ClusterRec> clusters = build_clusters(sc);
        HIP_TRY(sc->d_clusters.ensure(std::max<size_t>(clusters.size() * sizeof(mr::ClusterRec), 64)));
        if (!clusters.empty())
            HIP_TRY(hipMemcpyAsync(sc->d_clusters.p, clusters.data(), clusters.size() * sizeof(mr::ClusterRec), hipMemcpyHostToDevice, g_stream));
        HIP_TRY(hipStreamSynchronize(g_stream));            // (the vector goes out of scope)
    }
    if (nf > 0 && sc->pos32)
        hipLaunchKernelGGL(mr::k_face_static<float>, dim3((nf + 255) / 256), dim3(256), 0, g_stream, nf, sc->d_faces.as<int32_t>(),
                           sc->d_face_flags.as<uint8_t>(), sc->d_verts.as<double>(), sc->d_uv.as<float>(), sc->d_normals.as<float>(),
                           sc->d_face_pos.as<mr::FacePos32>(), sc->d_face_attr.as<mr::FaceAttr>());
    else if (nf > 0)
        hipLaunchKernelGGL(mr::k_face_static<double>, dim3((nf + 255) / 256), dim3(256), 0, g_stream, nf, sc->d_faces.as<int32_t>(),
                           sc->d_face_flags.as<uint8_t>(), sc->d_verts.as<double>(), sc->d_uv.as<float>(), sc->d_normals.as<float>(),
                           sc->d_face_pos.as<mr::FacePos64>(), sc->d_face_attr.as<mr::FaceAttr>());
    // compact edge records when every model's vertices are float32 and no edge has more than two faces
    sc->edge_compact = ne > 0 && sc->edge_inc.empty() && sc->pos32;
    if (sc->edge_compact) {
        HIP_TRY(sc->d_edges32.ensure((size_t)ne * sizeof(mr::EdgeRec32)));
        hipLaunchKernelGGL(mr::k_edge_compact, dim3((ne + 255) / 256), dim3(256), 0, g_stream, ne, sc->d_edges.as<mr::EdgeRec>(),
                           sc->d_edges32.as<mr::EdgeRec32>());
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g_stream));
    sc->dirty = false;
    return MR_OK;
}

FrameSlot *slot_for(mr_scene *sc, hipStream_t stream)
{
    for (auto &s : sc->slots)
        if (s->stream == stream) return s.get();
    if ((int)sc->slots.size() >= MAX_SLOTS) return nullptr;
    sc->slots.emplace_back(new (std::nothrow) FrameSlot());
    if (!sc->slots.back()) { sc->slots.pop_back(); return nullptr; }
    sc->slots.back()->stream = stream;
    return sc->slots.back().get();
}

int validate_frame(const mr_frame_desc *fr)
{
    if (!fr) return fail(MR_E_INVALID, "frame descriptor is NULL");
    if (fr->width <= 0 || fr->height <= 0 || fr->width > 32767 || fr->height > 32767)
        return fail(MR_E_INVALID, "resolution out of range");
    if (fr->system != 1 && fr->system != -1) return fail(MR_E_INVALID, "system must be +1 (RH) or -1 (LH)");
    if (fr->row_begin < 0 || fr->row_end > fr->height || fr->row_begin >= fr->row_end)
        return fail(MR_E_INVALID, "row band must satisfy 0 <= row_begin < row_end <= height");
    if (fr->light_type < 0 || fr->light_type > 2) return fail(MR_E_INVALID, "unknown light type");
    if (fr->stripe_count > 1) {
        if (fr->stripe_index < 0 || fr->stripe_index >= fr->stripe_count)
            return fail(MR_E_INVALID, "stripe_index must satisfy 0 <= stripe_index < stripe_count");
        if (fr->row_begin != 0 || fr->row_end != fr->height)
            return fail(MR_E_INVALID, "a striped frame spans all rows: row_begin / row_end must be 0 / height");
    } else if (fr->stripe_count < 0) {
        return fail(MR_E_INVALID, "stripe_count must not be negative");
    }
    return MR_OK;
}

mr::FrameConst make_const(const mr_scene *sc, const mr_frame_desc *fr)
{
    mr::FrameConst fc;
    std::memset(&fc, 0, sizeof fc);
    fc.width = fr->width; fc.height = fr->height; fc.system = fr->system;
    fc.backface_culling = fr->backface_culling; fc.light_type = fr->light_type; fc.flags = fr->flags;
    // output rows count from the top, the reference's buffers from the bottom (obj/core.py:640 flips)
    fc.band_y0 = fr->height - fr->row_end;
    fc.band_y1 = fr->height - fr->row_begin;
    fc.tiles_x = (fr->width + mr::TILE_W - 1) / mr::TILE_W;
    if (fr->stripe_count > 1) {
        // interleaved tile rows: this device owns frame tile rows stripe_index, stripe_index + N, ...
        const int rows = (fr->height + mr::TILE_H - 1) / mr::TILE_H;
        fc.tile_y0 = fr->stripe_index;
        fc.tile_step = fr->stripe_count;
        fc.tiles_y = rows > fr->stripe_index ? (rows - 1 - fr->stripe_index) / fr->stripe_count + 1 : 0;
        fc.out_tile_rows = (rows + fr->stripe_count - 1) / fr->stripe_count;
    } else {
        fc.tile_y0 = fc.band_y0 / mr::TILE_H;
        fc.tile_step = 1;
        fc.tiles_y = (fc.band_y1 - 1) / mr::TILE_H + 1 - fc.tile_y0;
        fc.out_tile_rows = 0;
    }
    fc.n_vertices = (int32_t)(sc->verts.size() / 4);
    fc.n_faces = (int32_t)(sc->faces.size() / 12);
    fc.n_edges = (int32_t)sc->edges.size();
    fc.n_materials = (int32_t)sc->materials.size();
    std::memcpy(fc.mvp, fr->mvp, sizeof fc.mvp);
    std::memcpy(fc.viewport, fr->viewport, sizeof fc.viewport);
    std::memcpy(fc.debug_mvp, fr->debug_mvp, sizeof fc.debug_mvp);
    std::memcpy(fc.planes, fr->frustum_planes, sizeof fc.planes);
    fc.two_nf = 2 * fr->z_near * fr->z_far;          // obj/core.py:228, evaluated left to right
    fc.f_plus_n = fr->z_far + fr->z_near;
    fc.f_minus_n = fr->z_far - fr->z_near;
    for (int j = 0; j < 3; ++j) {
        fc.camera_pos[j] = fr->camera_pos[j];
        fc.light_pos[j] = fr->light_pos[j]; fc.light_dir[j] = fr->light_dir[j];
        fc.light_color[j] = fr->light_color[j]; fc.light_ambient[j] = fr->light_ambient[j];
        fc.background[j] = fr->background[j];
    }
    fc.background_u8 = (uint32_t)fr->background_u8;
    std::memcpy(fc.sky_tri, fr->sky_tri, sizeof fc.sky_tri);
    std::memcpy(fc.sky_rays, fr->sky_rays, sizeof fc.sky_rays);
    fc.sky_size = sc->sky_size;
    fc.has_no_depth = sc->has_no_depth ? 1 : 0;     // (found at commit: a scan of the face flags here was 0.2 ms of host time per frame of a million faces)
    fc.same_clip = memcmp(fr->mvp, fr->debug_mvp, sizeof(fr->mvp)) == 0 ? 1 : 0;
    fc.edge_compact = sc->edge_compact ? 1 : 0;
    fc.pos32 = sc->pos32 ? 1 : 0;
    // cluster culling (kernels_geometry.h, cluster_culled).  Not when the caller wants per-face status or the fragment
    // counters: those see faces one by one.  The back-face cone needs the camera's centre of projection E and the sign
    // convention of obj/triangular.py:47-48 in world space: with e the null vector of MVP's (x, y, w) columns, e = ew (E, 1),
    // the screen-space area of a face whose corners are all in front of the camera has the sign of
    // det(viewport xy) * ew * n . (E - a)  for its world normal n = (b - a) x (c - a)  (Cauchy-Binet on the 3x4 by 4x3
    // product; checked against the per-face test on random cameras and triangles, tests/test_host_api.py).
    {
        const char *env = getenv("MR_CLUSTER_CULL");         // (looked up per frame: the tests switch it)
        // Measured on MI355X (round 3, A/B on one box): on a whole frame the test in front of every wavefront's first
        // load costs more than the 40 % of c4's face wavefronts it ends are worth -- the face half is not what the
        // launch waits for -- quoted regime c4 +2.5 %, c5 +0.8 %; on one rank's rows of a split frame, where most
        // clusters go, set-up -3 us (c5, a rank of eight).  So: on for partial frames, off for whole ones;
        // MR_CLUSTER_CULL=0 / 1 / box / count force it off / on / boxes only / on and counted.
        const bool partial = fr->row_begin != 0 || fr->row_end != fr->height || fr->stripe_count > 1;
        int mode = partial ? mr::CC_BOX | mr::CC_CONE : 0;
        if (env && !strcmp(env, "0")) mode = 0;
        if (env && (!strcmp(env, "1") || !strcmp(env, "count"))) mode = mr::CC_BOX | mr::CC_CONE;
        if (env && !strcmp(env, "box")) mode = mr::CC_BOX;
        if (fr->flags & (MR_FRAME_FACE_STATUS | MR_FRAME_COUNTERS)) mode = 0;
        if (mode & mr::CC_CONE) {
            const double *m = fr->mvp, *vp = fr->viewport;
            auto P = [&](int r, int c) { return m[r * 4 + (c == 2 ? 3 : c)]; };       // columns x, y, w
            double e[4];
            for (int i = 0; i < 4; ++i) {
                int r[3], k = 0;
                for (int j = 0; j < 4; ++j) if (j != i) r[k++] = j;
                const double det = P(r[0], 0) * (P(r[1], 1) * P(r[2], 2) - P(r[1], 2) * P(r[2], 1))
                                 - P(r[0], 1) * (P(r[1], 0) * P(r[2], 2) - P(r[1], 2) * P(r[2], 0))
                                 + P(r[0], 2) * (P(r[1], 0) * P(r[2], 1) - P(r[1], 1) * P(r[2], 0));
                e[i] = (i & 1) ? -det : det;
            }
            const double det_v = vp[0] * vp[5] - vp[1] * vp[4];
            const double big = std::max(std::max(fabs(e[0]), fabs(e[1])), std::max(fabs(e[2]), fabs(e[3])));
            const bool ok = std::isfinite(big) && big > 0 && fabs(e[3]) > 1e-9 * big && std::isfinite(det_v) && det_v != 0 &&
                            vp[8] == 0 && vp[9] == 0;          // (an orthographic camera has no centre: the boxes only)
            if (ok) {
                for (int j = 0; j < 3; ++j) fc.cull_eye[j] = e[j] / e[3];
                if (det_v * e[3] < 0) mode |= mr::CC_NEGATIVE;
            } else {
                mode &= ~mr::CC_CONE;
            }
        }
        if (env && !strcmp(env, "count") && mode) mode |= mr::CC_COUNT;
        fc.cluster_cull = mode;
    }
    fc.specular_strength = fr->specular_strength;
    fc.att_constant = fr->att_constant; fc.att_linear = fr->att_linear; fc.att_quadratic = fr->att_quadratic;
    fc.spot_edge0 = fr->spot_edge0; fc.spot_edge1 = fr->spot_edge1;
    return fc;
}

inline unsigned blocks_for(long long n, int per_block) { return (unsigned)std::max<long long>(1, (n + per_block - 1) / per_block); }

// bytes of the uint8 output of a frame: the band's rows, or the striped layout's blocks
size_t out_bytes(const mr_frame_desc *fr)
{
    if (fr->stripe_count > 1) {
        const int rows = (fr->height + mr::TILE_H - 1) / mr::TILE_H;
        return (size_t)((rows + fr->stripe_count - 1) / fr->stripe_count) * mr::TILE_H * fr->width * 3;
    }
    return (size_t)(fr->row_end - fr->row_begin) * fr->width * 3;
}

// Builds the overlay's lists from the cameras mr_scene_set_overlay_cameras left (host_overlay.h: clipping, projection,
// DDA, dashes, index wrapping -- obj/frustums.py:61-103, obj/line.py:6-16), if that has not happened yet.
void realize_overlay(mr_scene *sc)
{
    mr_scene::OvPending &p = sc->ov_pending;
    if (!p.set) return;
    p.set = false;
    static const int32_t faces[24] = { 2, 4, 5, 3,  0, 1, 7, 6,  0, 2, 3, 1,  5, 4, 6, 7,  3, 5, 7, 1,  4, 2, 0, 6 };
    static thread_local mr_host::OverlayLists lists;            // (its vectors keep their capacity from call to call)
    lists.seg_first.clear(); lists.seg_count.clear(); lists.z.clear();
    mr_host::build_overlay_lists(p.corners, faces, p.planes, p.mvp, p.viewport, p.near_, p.far_, p.inside != 0, p.height, p.width, 13,
                                 lists, false, true);
    sc->ov_points = sc->ov_segments = 0;
    sc->ov_serial += 1;
    if (lists.z.empty()) return;
    const size_t np = lists.z.size();
    sc->ov_target.resize((size_t)mr::OVERLAY_TARGETS * np);
    for (int k = 0; k < mr::OVERLAY_TARGETS; ++k) std::copy(lists.target[k].begin(), lists.target[k].end(), sc->ov_target.begin() + (size_t)k * np);
    sc->ov_z.assign(lists.z.begin(), lists.z.end());
    sc->ov_seg.resize(2 * lists.seg_first.size());
    for (size_t i = 0; i < lists.seg_first.size(); ++i) { sc->ov_seg[2 * i] = lists.seg_first[i]; sc->ov_seg[2 * i + 1] = lists.seg_count[i]; }
    sc->ov_tile_mask.swap(lists.tile_mask);
    sc->ov_height = p.height; sc->ov_width = p.width;
    sc->ov_points = (int32_t)np; sc->ov_segments = (int32_t)lists.seg_first.size();
}

void fill_overlay_args(const mr_scene *sc, const FrameSlot *fs, mr::OverlayArgs &oa)
{
    const char *base = static_cast<const char *>(fs->ov.lists.p);
    oa.z = reinterpret_cast<const double *>(base + fs->ov.off[0]);
    oa.idx = reinterpret_cast<const int32_t *>(base + fs->ov.off[1]);
    oa.seg = reinterpret_cast<const int32_t *>(base + fs->ov.off[2]);
    oa.n_points = sc->ov_points; oa.n_segments = sc->ov_segments;
    char *scratch = static_cast<char *>(fs->ov.scratch.p);
    oa.win = reinterpret_cast<uint32_t *>(scratch);
    oa.any = oa.win + fs->ov.state_entries;
    oa.keep = reinterpret_cast<uint8_t *>(oa.any + fs->ov.state_entries);
    oa.pixel_of = nullptr;
}

// Brings the slot's device copy of the overlay lists up to date: packed into the slot's page-locked staging buffer
// and copied with ONE asynchronous copy on the slot's stream (behind the slot's earlier frames, which read the old
// lists, and in front of the frame that needs the new ones).  The staging buffer is rewritten only after the copy
// that last read it has completed (an event; mr_render and mr_render_wait have drained the stream long before).
void ensure_overlay_slots(mr_scene *sc)
{
    if (sc->ov_slots_serial == sc->ov_serial) return;
    mr_host::build_overlay_slots(sc->ov_target.data(), (size_t)sc->ov_points, (size_t)sc->ov_height * sc->ov_width, sc->ov_slot_work,
                                 sc->ov_slot_of, sc->ov_touched);
    sc->ov_slots_serial = sc->ov_serial;
}

int sync_slot_overlay(mr_scene *sc, FrameSlot *fs, bool with_slots = false)
{
    if (sc->ov_points == 0 || (fs->ov.serial == sc->ov_serial && (fs->ov.with_slots || !with_slots))) return MR_OK;
    if (with_slots) ensure_overlay_slots(sc);
    const int n_src = with_slots ? 6 : 4;
    const void *src[6] = { sc->ov_z.data(), sc->ov_target.data(), sc->ov_seg.data(), sc->ov_tile_mask.data(),
                           sc->ov_slot_of.data(), sc->ov_touched.data() };
    const size_t bytes[6] = { sc->ov_z.size() * 8, sc->ov_target.size() * 4, sc->ov_seg.size() * 4, sc->ov_tile_mask.size(),
                              with_slots ? sc->ov_slot_of.size() * 4 : 0, with_slots ? sc->ov_touched.size() * 4 : 0 };
    size_t total = 0;
    for (int i = 0; i < 6; ++i) { fs->ov.off[i] = total; total += (bytes[i] + 15) & ~(size_t)15; }
    // scratch: win and any (one word per pixel of the frame each, zero between segments and frames)
    const size_t entries = (size_t)sc->ov_height * sc->ov_width;
    const size_t scratch = entries * 8 + (size_t)sc->ov_points + 16;
    if (total > fs->ov.staging_cap || total > fs->ov.lists.cap || scratch > fs->ov.scratch.cap)
        HIP_TRY(hipStreamSynchronize(fs->stream));          // (growing frees the old buffers: nothing may still use them)
    if (total > fs->ov.staging_cap) {
        if (fs->ov.staging) (void)hipHostFree(fs->ov.staging);
        fs->ov.staging = nullptr; fs->ov.staging_cap = 0;
        HIP_TRY(hipHostMalloc(&fs->ov.staging, total + total / 2, hipHostMallocDefault));
        fs->ov.staging_cap = total + total / 2;
    }
    if (!fs->ov.copied) HIP_TRY(hipEventCreateWithFlags(&fs->ov.copied, hipEventDisableTiming));
    else HIP_TRY(hipEventSynchronize(fs->ov.copied));
    for (int i = 0; i < n_src; ++i) std::memcpy(static_cast<char *>(fs->ov.staging) + fs->ov.off[i], src[i], bytes[i]);
    HIP_TRY(fs->ov.lists.ensure(total));
    {
        const void *had = fs->ov.scratch.p;
        const size_t had_entries = fs->ov.state_entries;
        HIP_TRY(fs->ov.scratch.ensure(scratch));
        if (fs->ov.scratch.p != had || had_entries != entries)      // the kernel leaves win / any zeroed; a new layout starts so
            HIP_TRY(hipMemsetAsync(fs->ov.scratch.p, 0, fs->ov.scratch.cap, fs->stream));
        fs->ov.state_entries = entries;
    }
    HIP_TRY(hipMemcpyAsync(fs->ov.lists.p, fs->ov.staging, total, hipMemcpyHostToDevice, fs->stream));
    HIP_TRY(hipEventRecord(fs->ov.copied, fs->stream));
    fs->ov.serial = sc->ov_serial;
    fs->ov.with_slots = with_slots;
    return MR_OK;
}

// Enqueues one frame on the slot's stream.  d_out receives the uint8 rows.
// The overlay kernel on the slot's own z-buffer and float frame (so the debug taps show them after the overlay, like
// upstream's), finalising the touched pixels into d_out.
void launch_overlay(mr_scene *sc, FrameSlot *fs, uint8_t *d_out, int width, int height, int system, hipStream_t stream)
{
    mr::OverlayArgs oa;
    fill_overlay_args(sc, fs, oa);
    oa.st_z = fs->d_z.as<double>(); oa.st_f = fs->d_frame.as<float>(); oa.out = d_out;
    oa.out_width = width; oa.out_height = height;
    oa.gamma_lut = sc->d_gamma.as<float>();
    hipLaunchKernelGGL(mr::k_overlay, dim3(1), dim3(mr::OVERLAY_BLOCK), 0, stream, oa, (double)system);
}

// Second half of a frame whose overlay enqueue_frame left for later (may_defer_overlay): the device is busy with the
// frame's three kernels, the host builds the lines' lists meanwhile, then the upload and the overlay kernel follow on the
// frame's stream.
int finish_overlay(mr_scene *sc, FrameSlot *fs, uint8_t *d_out)
{
    if (!fs->overlay_deferred) return MR_OK;
    fs->overlay_deferred = false;
    realize_overlay(sc);
    if (sc->ov_points == 0) return MR_OK;
    const mr_frame_desc &fr = fs->last_frame;
    if (sc->ov_width != fr.width || sc->ov_height != fr.height) return fail(MR_E_INVALID, "overlay lists were built for a frame of another size");
    int rc = sync_slot_overlay(sc, fs, false);
    if (rc) return rc;
    launch_overlay(sc, fs, d_out, fr.width, fr.height, fr.system, fs->stream);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

int enqueue_frame(mr_scene *sc, FrameSlot *fs, const mr_frame_desc *fr, uint8_t *d_out, bool may_defer_overlay = false)
{
    using namespace mr;
    int rc = commit(sc);
    if (rc) return rc;
    hipStream_t stream = fs->stream;
    FrameConst fc = make_const(sc, fr);
    if (fc.flags & MR_FRAME_FACE_STATUS) fc.flags |= MR_FRAME_KEEP_BUFFERS;
    // (the lists of cameras left by mr_scene_set_overlay_cameras: built now -- or, for a whole frame of a caller that
    // finishes it with finish_overlay, after the frame's kernels have been launched)
    const bool partial = fr->row_begin != 0 || fr->row_end != fr->height || fr->stripe_count > 1;
    const bool deferred = may_defer_overlay && (fc.flags & MR_FRAME_OVERLAY) && sc->ov_pending.set && !partial &&
                          sc->ov_pending.width == fc.width && sc->ov_pending.height == fc.height;
    if ((fc.flags & MR_FRAME_OVERLAY) && !deferred) realize_overlay(sc);
    fs->overlay_deferred = deferred;
    const bool overlay = (fc.flags & MR_FRAME_OVERLAY) && sc->ov_points > 0 && !deferred;
    // did the caller ask for the z / stencil / winner / float-frame taps?  (The overlay needs z and colour too, but
    // only at the pixels its lines touch: then only the tiles that hold such a pixel write them, ov_off[7].)
    const bool taps_asked = (fc.flags & (MR_FRAME_KEEP_BUFFERS | MR_FRAME_KEEP_FLOAT)) != 0;
    // a device that owns only part of the frame (a rank of a multi-GPU split) cannot replay the overlay: its lines test
    // z at pixels other devices own.  It appends the state of the touched pixels it owns to its rows instead
    // (k_overlay_export), and the overlay is replayed on the assembled frame (mr_overlay_apply).
    if (fc.flags & MR_FRAME_OVERLAY) {
        if (partial && fr->stripe_count <= 1 && (fr->height % (fr->row_end - fr->row_begin) || fr->row_begin % (fr->row_end - fr->row_begin)))
            return fail(MR_E_INVALID, "overlay on a row band: the bands of the split must be equal");
        if (!deferred && sc->ov_points > 0 && (sc->ov_width != fc.width || sc->ov_height != fc.height))
            return fail(MR_E_INVALID, "overlay lists were built for a frame of another size");
        if (overlay && (rc = sync_slot_overlay(sc, fs, partial))) return rc;
        fc.flags |= MR_FRAME_KEEP_BUFFERS | MR_FRAME_KEEP_FLOAT;
    }
    const size_t npx = (size_t)fc.width * fc.height;
    const int n_tiles = fc.tiles_x * fc.tiles_y;
    const bool shadows = (fc.flags & MR_FRAME_SHADOWS) != 0;
    const bool keep = (fc.flags & MR_FRAME_KEEP_BUFFERS) != 0;
    const size_t nF = (size_t)std::max(fc.n_faces, 1), nV = (size_t)std::max(fc.n_vertices, 1);
    // MR_VERTEX_PATH=mfma: vertex transform once per unique vertex on the matrix cores, as a launch of
    // its own in front of k_setup (same bits; for A/B timing and the MFMA counters)
    static const bool vertex_mfma = [] { const char *e = getenv("MR_VERTEX_PATH"); return e && !strcmp(e, "mfma"); }();

    if (sc->quad_cap == 0) sc->quad_cap = (uint32_t)std::min<size_t>(std::max(fc.n_edges, 1), 1u << 17);
    fs->quad_cap = sc->quad_cap; fs->work_cap = sc->work_cap;
    for (int c = 0; c < BIN_CLASSES; ++c) fs->bin_cap[c] = sc->bin_cap[c];

    if (vertex_mfma) {
        HIP_TRY(fs->d_vout.ensure(nV * sizeof(VertexOut)));
        HIP_TRY(fs->d_vclip.ensure(nV * sizeof(VertexClip)));
    }
    HIP_TRY(fs->d_count_list.ensure(nF * sizeof(uint32_t)));
    HIP_TRY(fs->d_tris.ensure(nF * sizeof(TriRec)));
    HIP_TRY(fs->d_clips.ensure(nF * sizeof(TriClip)));
    HIP_TRY(fs->d_status.ensure(nF));
    HIP_TRY(fs->d_quads.ensure((size_t)fs->quad_cap * sizeof(QuadRec)));
    HIP_TRY(fs->d_sil.ensure((size_t)fs->quad_cap * 2 * sizeof(int32_t)));
    {
        const void *had = fs->d_counters.p;
        HIP_TRY(fs->d_counters.ensure(2 * sizeof(Counters) + sizeof(Sticky)));
        if (fs->d_counters.p != had) HIP_TRY(hipMemsetAsync(fs->d_counters.p, 0, fs->d_counters.cap, stream));
    }
    HIP_TRY(fs->d_bin_count.ensure((size_t)(BIN_CLASSES * n_tiles + 1) * 4));
    for (int c = 0; c < BIN_CLASSES; ++c) {
        const size_t bytes = (size_t)std::max(n_tiles, 1) * fs->bin_cap[c] * 4;
        if (bytes > ((size_t)48 << 30))
            return fail(MR_E_OVERFLOW, "more primitives in one 16x16 tile than the tile lists are allowed to grow to (48 GB per class)");
        HIP_TRY(fs->d_items[c].ensure(bytes));
    }
    HIP_TRY(fs->d_work.ensure((size_t)fs->work_cap * sizeof(uint2)));
    HIP_TRY(fs->d_tile_stats.ensure((size_t)std::max(n_tiles, 1) * TILE_REC * 4));
    if (keep) {
        HIP_TRY(fs->d_z.ensure(npx * sizeof(double)));
        HIP_TRY(fs->d_winner.ensure(npx * sizeof(int32_t)));
        HIP_TRY(fs->d_stencil.ensure(npx * sizeof(int32_t)));
    }
    if (fc.flags & MR_FRAME_KEEP_FLOAT) HIP_TRY(fs->d_frame.ensure(npx * 3 * sizeof(float)));
    if (!fs->events_ok) {
        for (auto &set : fs->ev_ring) for (auto &e : set) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipHostMalloc((void **)&fs->h_counters, sizeof(Counters) + sizeof(Sticky), hipHostMallocDefault));
        fs->h_sticky = reinterpret_cast<Sticky *>(fs->h_counters + 1);
        fs->events_ok = true;
    }
    fs->ev = fs->ev_ring[fs->frames_enqueued % EVENT_RING];

    Counters *ctr = fs->ctr(fs->frames_enqueued), *next_ctr = fs->ctr(fs->frames_enqueued + 1);
    // MR_FRAME_LIGHT_TIMING keeps only the marks around the frame and the tile kernel, MR_FRAME_NO_TIMING none
    const bool timing = !(fc.flags & MR_FRAME_NO_TIMING);
    const bool all_marks = timing && !(fc.flags & MR_FRAME_LIGHT_TIMING);
    fs->ev_marks[fs->frames_enqueued % EVENT_RING] = timing ? (all_marks ? 2 : 1) : 0;
    if (timing) HIP_TRY(hipEventRecord(fs->ev[0], stream));
    // The list cursors are left zeroed by k_tile and the frame counters are cleared by the previous
    // frame's k_tile, so a steady-state frame issues no memset; only a new tile grid needs one.
    // tile order: ORDER_HEAD + n_tiles words (written by k_bin_work), then the tiles' class bytes (k_tile, for the next frame)
    const size_t order_bytes = (((size_t)ORDER_HEAD + (size_t)std::max(n_tiles, 1)) * sizeof(uint32_t) + 15) & ~(size_t)15;
    HIP_TRY(fs->d_hist.ensure(order_bytes + (size_t)std::max(n_tiles, 1) + 16));     // class bytes: 16-byte aligned, padded
    {
        // split tiles: HEAVY0_MAX arrival counters (zero between frames), then the parts' stencil counts
        const void *had = fs->d_split.p;
        HIP_TRY(fs->d_split.ensure((size_t)HEAVY0_MAX * 4 + (size_t)HEAVY0_MAX * HEAVY_SPLIT * TILE_PX * 4));
        if (fs->d_split.p != had) HIP_TRY(hipMemsetAsync(fs->d_split.p, 0, (size_t)HEAVY0_MAX * 4, stream));
    }
    if (fs->bins_zeroed_for != BIN_CLASSES * n_tiles + 1) {
        HIP_TRY(hipMemsetAsync(fs->d_bin_count.p, 0, fs->d_bin_count.cap, stream));
        HIP_TRY(hipMemsetAsync(fs->d_hist.p, 0, fs->d_hist.cap, stream));       // no history for a new tile grid
        fs->bins_zeroed_for = BIN_CLASSES * n_tiles + 1;
    }
    uint32_t *order = fs->d_hist.as<uint32_t>();
    uint8_t *tile_class = reinterpret_cast<uint8_t *>(fs->d_hist.p) + order_bytes;

    BinArgs ba;
    ba.tris = fs->d_tris.as<TriRec>(); ba.quads = fs->d_quads.as<QuadRec>();
    ba.ctr = ctr; ba.quad_cap = fs->quad_cap;
    ba.bin_count = fs->d_bin_count.as<uint32_t>();
    for (int c = 0; c < BIN_CLASSES; ++c) { ba.items[c] = fs->d_items[c].as<uint32_t>(); ba.cap[c] = fs->bin_cap[c]; }
    ba.work = fs->d_work.as<uint2>(); ba.work_cap = fs->work_cap;

    SetupArgs sa;
    sa.faces = sc->d_faces.as<int32_t>(); sa.face_flags = sc->d_face_flags.as<uint8_t>();
    sa.verts = sc->d_verts.as<double>(); sa.uv = sc->d_uv.as<float>(); sa.normals = sc->d_normals.as<float>();
    sa.vout = fs->d_vout.as<VertexOut>(); sa.vclip = fs->d_vclip.as<VertexClip>();
    sa.face_pos = sc->d_face_pos.p;
    sa.clusters = sc->d_clusters.as<mr::ClusterRec>();
    sa.tris = fs->d_tris.as<TriRec>(); sa.clips = fs->d_clips.as<TriClip>();
    sa.status = fs->d_status.as<uint8_t>(); sa.count_list = fs->d_count_list.as<uint32_t>(); sa.ctr = ctr;
    sa.edges = sc->edge_compact ? reinterpret_cast<const EdgeRec *>(sc->d_edges32.p) : sc->d_edges.as<EdgeRec>(); sa.edge_inc = sc->d_edge_inc.as<uint32_t>(); sa.face_n = sc->d_face_n.as<double>();
    sa.tile_class = tile_class; sa.order = order;
    sa.sil_edges = fs->d_sil.as<int32_t>(); sa.quads = fs->d_quads.as<QuadRec>(); sa.quad_cap = fs->quad_cap;

    // ---- 1. set-up: faces and (with shadows) edges, one launch
    if (vertex_mfma && fc.n_vertices > 0)
        hipLaunchKernelGGL(k_vertex_mfma, dim3(blocks_for(fc.n_vertices, 64)), dim3(256), 0, stream, fc,
                           sc->d_verts.as<double>(), fs->d_vout.as<VertexOut>(), fs->d_vclip.as<VertexClip>());
    if (all_marks) HIP_TRY(hipEventRecord(fs->ev[1], stream));
    {
        const unsigned face_blocks = fc.n_faces > 0 ? blocks_for(fc.n_faces, SETUP_BLOCK) : 0u;
        // small meshes: one edge per 2 / 4 lanes, so that a wavefront rarely finds more silhouette edges than one round
        // of its quad set-up takes (kernels_geometry.h, edge_block)
        static const int spread_env = getenv("MR_EDGE_SPREAD") ? atoi(getenv("MR_EDGE_SPREAD")) : -2;     // -1: dense
        const bool dense = spread_env == -1 || (spread_env == -2 && fc.n_edges > (1 << 17));
        const unsigned spread = dense ? EDGE_DENSE : spread_env >= 0 ? (unsigned)std::min(spread_env, 4)
                              : fc.n_edges <= (1 << 15) ? 2u : 1u;
        const unsigned edge_blocks = !(shadows && fc.n_edges > 0) ? 0u
                                   : dense ? blocks_for(fc.n_edges, 2 * SETUP_BLOCK) : blocks_for((long long)fc.n_edges << spread, SETUP_BLOCK);
        SetupKernArgs ska;
        ska.fc = fc; ska.sa = sa; ska.bins = ba; ska.face_blocks = face_blocks; ska.edge_spread = spread;
        if (vertex_mfma)
            hipLaunchKernelGGL(k_setup<true>, dim3(1 + face_blocks + edge_blocks), dim3(SETUP_BLOCK), 0, stream, ska);
        else
            hipLaunchKernelGGL(k_setup<false>, dim3(1 + face_blocks + edge_blocks), dim3(SETUP_BLOCK), 0, stream, ska);
    }
    if (all_marks) HIP_TRY(hipEventRecord(fs->ev[2], stream));

    // ---- 2. tile lists of the large primitives + leftover survivor counts (one wavefront per face,
    // grid-stride: a mesh of large faces lists most of them)
    {
        static const unsigned work_blocks_env = [] { const char *e = getenv("MR_WORK_BLOCKS"); return e ? (unsigned)atoi(e) : 0u; }();
        // enough wavefronts for every one to be resident at once (five per SIMD): an item is three dependent trips to
        // memory and a returning atomic, so the kernel lasts as many of those chains as a wavefront has items (c4: 512
        // workgroups 12.7 us, 1 280 11.5; c5: 25.0 -> 19.0; quoted regime c4 -1.5 %)
        const unsigned work_blocks = work_blocks_env ? work_blocks_env : 1280;
        static const unsigned count_blocks_env = [] { const char *e = getenv("MR_COUNT_BLOCKS"); return e ? (unsigned)atoi(e) : 512u; }();
        const unsigned count_blocks = fc.n_faces > 0 ? std::min(count_blocks_env, blocks_for((long long)fc.n_faces * WAVE, 256)) : 0u;
        hipLaunchKernelGGL(k_bin_work, dim3(count_blocks + work_blocks), dim3(256), 0, stream, fc, ba,
                           fs->d_count_list.as<uint32_t>(), fs->d_tris.as<TriRec>(),
                           fs->d_clips.as<TriClip>(), fs->d_status.as<uint8_t>(), ctr, count_blocks);
    }
    if (timing) HIP_TRY(hipEventRecord(fs->ev[3], stream));

    // ---- 3. tiles: coverage, z, stencil, shading, finalise
    TileArgs ta;
    ta.clips = fs->d_clips.as<TriClip>(); ta.quads = fs->d_quads.as<QuadRec>();
    ta.bin_count = fs->d_bin_count.as<uint32_t>();
    for (int c = 0; c < BIN_CLASSES; ++c) { ta.items[c] = fs->d_items[c].as<uint32_t>(); ta.cap[c] = fs->bin_cap[c]; }
    ta.tap_mask = (overlay && !taps_asked) ? reinterpret_cast<const uint8_t *>(static_cast<const char *>(fs->ov.lists.p) + fs->ov.off[3])
                                           : nullptr;
    ta.zbuf = keep ? fs->d_z.as<double>() : nullptr;
    ta.winner = keep ? fs->d_winner.as<int32_t>() : nullptr;
    ta.stencil = keep ? fs->d_stencil.as<int32_t>() : nullptr;
    ta.tile_stats = fs->d_tile_stats.as<uint32_t>();
    ta.ctr = ctr; ta.next_ctr = next_ctr; ta.sticky = fs->sticky();
    // Heaviest-first order shortens the critical path of a frame that has the device to itself.  When the
    // scene is being rendered from several streams at once (frames in flight), the next frame's work fills
    // the tail anyway and bunching the heavy tiles at the front only makes them compete: measured on MI355X
    // with three streams, row-major is 4 % (c4) to 28 % (c2) faster per frame, and 10 % slower for a lone
    // frame.  So: ordered when no other stream took one of the scene's last frames.
    // MR_TILE_ORDER=rowmajor | heaviest forces either (for the ablation in DESIGN.md).
    static const int order_mode = [] {
        const char *e = getenv("MR_TILE_ORDER");
        return !e ? 0 : !strcmp(e, "rowmajor") ? 1 : !strcmp(e, "heaviest") ? 2 : 0;
    }();
    sc->frame_serial += 1;
    bool alone = true;
    for (auto &s : sc->slots)
        if (s.get() != fs && s->have_frame && sc->frame_serial - s->last_serial <= 8) alone = false;
    fs->last_serial = sc->frame_serial;
    const bool ordered = order_mode == 2 || (order_mode == 0 && alone);
    ta.order = ordered ? order : nullptr; ta.tile_class = tile_class;
    fs->last_ordered = ordered;
    // a device whose tiles all fit on the chip at once (a rank of a multi-GPU split) shares out the quads of
    // its heaviest tiles: there the launch lasts as long as the slowest tile (see HEAVY_SPLIT)
    ta.split_arrive = fs->d_split.as<uint32_t>();
    ta.split_sten = fs->d_split.as<int32_t>() + HEAVY0_MAX;
    ShadeArgs sh;
    sh.tris = fs->d_tris.as<TriRec>(); sh.face_pos = sc->d_face_pos.p; sh.face_attr = sc->d_face_attr.as<FaceAttr>();
    sh.materials = sc->d_materials.as<Material>();
    sh.sky = sc->sky_size > 0 ? sc->d_sky.as<uint8_t>() : nullptr;
    sh.gamma_lut = sc->d_gamma.as<float>();
    sh.frame = (fc.flags & MR_FRAME_KEEP_FLOAT) ? fs->d_frame.as<float>() : nullptr;
    sh.out = d_out;
    TileKernArgs tka;
    tka.fc = fc; tka.ta = ta; tka.sh = sh;
    // which tiles are shared out over HEAVY_SPLIT workgroups next frame: on a device whose tiles are all resident
    // at once every tile with a quad walk worth sharing; on a whole frame only the handful that outlast
    // everything else (the launch then ends with them).  MR_TILE_SPLIT=0 | 1 forces it off / on for every grid.
    static const int split_mode = [] { const char *e = getenv("MR_TILE_SPLIT"); return !e ? -1 : atoi(e); }();
    const bool small_grid = n_tiles <= 2048;
    static const unsigned split_cost_big = [] { const char *e = getenv("MR_SPLIT_COST"); return e ? (unsigned)atoi(e) : 400u; }();
    static const unsigned split_quads_big = [] { const char *e = getenv("MR_SPLIT_QUADS"); return e ? (unsigned)atoi(e) : 48u; }();
    tka.ta.split_cost = small_grid ? 350u : split_cost_big;
    tka.ta.split_quads = small_grid ? 32u : split_quads_big;
    static const unsigned split_max_big = [] { const char *e = getenv("MR_SPLIT_MAX"); return e ? (unsigned)atoi(e) : 64u; }();
    tka.ta.split_max = std::min<unsigned>(small_grid ? (unsigned)HEAVY0_MAX : split_max_big, (unsigned)HEAVY0_MAX);
    const bool split = split_mode < 0 ? (small_grid || ordered) : split_mode != 0;
    if (n_tiles > 0 && split)
        hipLaunchKernelGGL(k_tile<true>, dim3((unsigned)(n_tiles + SPLIT_FRONT)), dim3(TILE_PX), 0, stream, tka);
    else if (n_tiles > 0)
        hipLaunchKernelGGL(k_tile<false>, dim3((unsigned)n_tiles), dim3(TILE_PX), 0, stream, tka);
    else          // nothing to draw on this device (a stripe beyond the frame): still hand the counters on
        HIP_TRY(hipMemsetAsync(next_ctr, 0, sizeof(Counters), stream));
    if (timing) HIP_TRY(hipEventRecord(fs->ev[4], stream));
    if ((fc.flags & MR_FRAME_FACE_STATUS) && fc.n_faces > 0)
        hipLaunchKernelGGL(k_face_status, dim3(blocks_for(fc.n_faces, 256)), dim3(256), 0, stream, fc,
                           fs->d_tris.as<TriRec>(), fs->d_clips.as<TriClip>(),
                           fs->d_z.as<double>(), fs->d_stencil.as<int32_t>(), fs->d_status.as<uint8_t>());
    if (overlay && partial) {
        const int world = fr->stripe_count > 1 ? fr->stripe_count : fr->height / (fr->row_end - fr->row_begin);
        const int rank = fr->stripe_count > 1 ? fr->stripe_index : fr->row_begin / (fr->row_end - fr->row_begin);
        const int n_slots = (int)sc->ov_touched.size();
        OverlayState *state = reinterpret_cast<OverlayState *>(d_out + ((out_bytes(fr) + 15) & ~(size_t)15));
        hipLaunchKernelGGL(k_overlay_export, dim3(blocks_for(n_slots, 256)), dim3(256), 0, stream,
                           reinterpret_cast<const int32_t *>(static_cast<const char *>(fs->ov.lists.p) + fs->ov.off[5]), n_slots,
                           fs->d_z.as<double>(), fs->d_frame.as<float>(), fc.width, fc.height, world, fr->stripe_count > 1 ? 1 : 0, rank, state);
    } else if (overlay) {                   // after the lit pass' per-face verdicts, as in obj/core.py:624-638
        launch_overlay(sc, fs, d_out, fc.width, fc.height, fc.system, stream);
    }
    HIP_TRY(hipGetLastError());
    fs->last_frame = *fr;
    fs->last_frame.flags = fc.flags;
    fs->last_n_tiles = n_tiles;
    fs->have_frame = true;
    fs->stats_reduced = false;
    fs->last_copied = false;
    fs->frames_enqueued += 1;
    sc->last = fs;
    return MR_OK;
}

// The fragment / pixel counts of a frame are left as per-tile partials by the tile kernel; they
// are summed and fetched only when somebody asks (mr_render, mr_get_stats).
int fetch_counters(mr_scene *sc, FrameSlot *fs, bool reduce)
{
    using namespace mr;
    (void)sc;
    Counters *ctr = fs->ctr(fs->frames_enqueued - 1);
    // mr_render without a stats pointer only needs the overflow flags: no reduction launch
    if (reduce && !fs->stats_reduced && fs->last_n_tiles > 0) {
        hipLaunchKernelGGL(k_reduce_tile_stats, dim3(256), dim3(256), 0, fs->stream, fs->d_tile_stats.as<uint32_t>(),
                           fs->last_n_tiles, ctr);
        fs->stats_reduced = true;
    }
    HIP_TRY(hipMemcpyAsync(fs->h_counters, ctr, sizeof(Counters), hipMemcpyDeviceToHost, fs->stream));
    HIP_TRY(hipMemcpyAsync(fs->h_sticky, fs->sticky(), sizeof(Sticky), hipMemcpyDeviceToHost, fs->stream));
    return MR_OK;
}

// After the stream has drained: turn counters + events into mr_stats; grow work lists on overflow.
int collect(mr_scene *sc, FrameSlot *fs, bool with_copy)
{
    const mr::Counters &c = *fs->h_counters;
    mr_stats &s = sc->stats;
    if (fs->last_frame.flags & MR_FRAME_COUNTERS) {
        s.frag_tri = (int64_t)c.frag_tri; s.frag_quad = (int64_t)c.frag_quad;
        s.covered_px = (int64_t)c.covered_px; s.lit_px = (int64_t)c.lit_px;
        s.stencil_updates = (int64_t)c.stencil_updates;
    } else {                                  // not counted: the frame was rendered without MR_FRAME_COUNTERS
        s.frag_tri = s.frag_quad = s.covered_px = s.lit_px = s.stencil_updates = -1;
    }
    s.n_faces = (int64_t)(sc->faces.size() / 12); s.n_faces_setup = c.n_valid_tris;
    s.n_quads = c.n_quads; s.n_quads_drawn = c.n_quads_drawn;
    s.tri_bin_entries = c.tri_bin_total; s.quad_bin_entries = c.bin_total - c.tri_bin_total;
    sc->n_silhouette = (int)c.n_quads;
    float ms = 0;
    const bool timed = !(fs->last_frame.flags & MR_FRAME_NO_TIMING);
    auto span = [&](int a, int b) { ms = 0; if (timed) (void)hipEventElapsedTime(&ms, fs->ev[a], fs->ev[b]); return ms; };
    const bool light = (fs->last_frame.flags & MR_FRAME_LIGHT_TIMING) != 0;
    s.gpu_ms_setup = light ? 0.f : span(0, 2); s.gpu_ms_binning = light ? span(0, 3) : span(2, 3);
    s.gpu_ms_tile = span(3, 4);
    s.gpu_ms_copy = with_copy && timed ? span(4, 5) : 0.f;
    s.gpu_ms_total = span(0, with_copy ? 5 : 4);
    // the verdicts of the last frame and of every frame of this slot since the host last looked (Sticky)
    mr::Sticky &st = *fs->h_sticky;
    const uint32_t overflow = c.overflow | st.overflow;
    uint32_t longest_stretch = 0;             // the work list is WORK_SHARDS stretches: the fullest one decides what it needs
    for (const auto &w : c.work) longest_stretch = std::max(longest_stretch, w.n);
    const uint32_t n_work = std::max(longest_stretch * (uint32_t)mr::WORK_SHARDS, st.n_work), n_quads = std::max(c.n_quads, st.n_quads);
    const uint32_t n_quads_drawn = std::max(c.n_quads_drawn, st.n_quads_drawn);
    bool grown = false;
    if (overflow) {
        for (int cls = 0; cls < mr::BIN_CLASSES; ++cls)
            if (overflow & (1u << cls)) {
                const uint32_t longest = std::max(c.max_list[cls], st.max_list[cls]);
                uint32_t want = std::max(longest + longest / 2, fs->bin_cap[cls] * 2);
                uint32_t cap = 64;
                while (cap < want) cap <<= 1;
                sc->bin_cap[cls] = std::max(sc->bin_cap[cls], cap);
            }
        if (overflow & 8u) sc->work_cap = std::max(sc->work_cap, n_work + n_work / 2 + 1024);
        if (overflow & 16u) sc->quad_cap = std::max(sc->quad_cap, std::max(n_quads_drawn + n_quads_drawn / 2 + 64, fs->quad_cap * 2));
        grown = true;
    }
    if (n_quads > fs->quad_cap) { sc->quad_cap = std::max(sc->quad_cap, n_quads + n_quads / 2 + 64); grown = true; }
    if (grown) {
        // acted on: the device's record and the last frame's block start clean (stream order puts this in front of
        // the slot's next frame, whose tile kernel would fold that block into the record again)
        (void)hipMemsetAsync(fs->ctr(fs->frames_enqueued - 1), 0, sizeof(mr::Counters), fs->stream);
        (void)hipMemsetAsync(fs->sticky(), 0, sizeof(mr::Sticky), fs->stream);
        st = mr::Sticky{};
    }
    return grown ? MR_E_OVERFLOW : MR_OK;
}

template <class T>
int read_back(const DevBuf &buf, T *out, size_t count, const char *what)
{
    if (!out) return fail(MR_E_INVALID, "NULL argument");
    if (!buf.p) return fail(MR_E_INVALID, std::string(what) + ": nothing rendered yet");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, buf.p, count * sizeof(T), hipMemcpyDeviceToHost));
    return MR_OK;
}

FrameSlot *last_slot(mr_scene *sc)
{
    if (!sc || !sc->last || !sc->last->have_frame) { fail(MR_E_INVALID, "nothing rendered yet"); return nullptr; }
    return sc->last;
}

}  // namespace

// ============================================================================ C ABI

static thread_local mr_host::OverlayLists g_overlay_lists;

extern "C" {

int mr_abi_version(void) { return MR_ABI_VERSION; }

int mr_abi_struct_size(int which)
{
    switch (which) {
    case 0: return (int)sizeof(mr_frame_desc);
    case 1: return (int)sizeof(mr_material);
    case 2: return (int)sizeof(mr_model_desc);
    case 3: return (int)sizeof(mr_stats);
    case 4: return (int)sizeof(mr_overlay_desc);
    default: return -1;
    }
}

int mr_device_available(void)
{
    int n = 0;
    return (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? 1 : 0;
}

int mr_init(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(MR_E_DEVICE, "no HIP device visible: libmi355rast has no CPU fallback");
    if (device >= 0) {
        if (device >= n) return fail(MR_E_INVALID, "device index out of range");
        HIP_TRY(hipSetDevice(device));
    }
    return ensure_init();
}

const char *mr_last_error(void) { return g_error.c_str(); }

mr_scene *mr_scene_create(void)
{
    mr_scene *sc = new (std::nothrow) mr_scene();
    if (!sc) fail(MR_E_DEVICE, "out of host memory");
    return sc;
}

int mr_scene_clear(mr_scene *sc)
{
    if (!sc) return fail(MR_E_INVALID, "scene is NULL");
    if (g_initialised) (void)hipDeviceSynchronize();
    for (void *p : sc->texture_allocs) (void)hipFree(p);
    sc->texture_allocs.clear(); sc->textures.clear();
    sc->verts.clear(); sc->uv.clear(); sc->normals.clear(); sc->faces.clear(); sc->face_flags.clear();
    sc->materials.clear(); sc->models.clear(); sc->edges.clear(); sc->edge_inc.clear();
    sc->edge_ids.clear(); sc->edge_raw.clear();
    sc->dirty = true;
    sc->last = nullptr;
    sc->reset_caps();
    for (auto &fs : sc->slots) fs->reset_caps();
    return MR_OK;
}

int mr_scene_set_list_capacities(mr_scene *sc, uint32_t small_pairs, uint32_t big_pairs, uint32_t quads, uint32_t work)
{
    if (!sc) return fail(MR_E_INVALID, "scene is NULL");
    if (g_initialised) (void)hipDeviceSynchronize();
    if (small_pairs) sc->bin_cap[0] = small_pairs;
    if (big_pairs) sc->bin_cap[1] = big_pairs;
    if (quads) sc->bin_cap[2] = quads;
    if (work) sc->work_cap = std::max(work, (uint32_t)mr::WORK_SHARDS);
    return MR_OK;
}

void mr_scene_destroy(mr_scene *sc)
{
    if (!sc) return;
    mr_scene_clear(sc);
    DevBuf *bufs[] = { &sc->d_verts, &sc->d_uv, &sc->d_normals, &sc->d_faces, &sc->d_face_flags, &sc->d_materials,
                       &sc->d_textures, &sc->d_edges, &sc->d_edges32, &sc->d_edge_inc, &sc->d_face_n, &sc->d_sky, &sc->d_gamma,
                       &sc->d_face_pos, &sc->d_face_attr, &sc->d_clusters,
                       };
    for (DevBuf *b : bufs) b->release();
    for (auto &fs : sc->slots) fs->release();
    for (auto &ln : sc->lanes) if (ln.stream) (void)hipStreamDestroy(ln.stream);
    delete sc;
}

int mr_scene_add_texture(mr_scene *sc, const float *rgb, int32_t h, int32_t w)
{
    if (!sc || !rgb) return fail(MR_E_INVALID, "NULL argument");
    if (h <= 0 || w <= 0) return fail(MR_E_INVALID, "texture size must be positive");
    int rc = ensure_init();
    if (rc) return rc;
    void *d = nullptr;
    const size_t bytes = (size_t)h * w * 3 * sizeof(float);
    HIP_TRY(hipMalloc(&d, bytes));
    hipError_t e = hipMemcpy(d, rgb, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return fail(MR_E_DEVICE, hipGetErrorString(e)); }
    sc->texture_allocs.push_back(d);
    sc->textures.push_back({ static_cast<const float *>(d), h, w });
    sc->dirty = true;
    return (int)sc->textures.size() - 1;
}

int mr_scene_set_skybox(mr_scene *sc, const uint8_t *texels, int32_t size)
{
    if (!sc) return fail(MR_E_INVALID, "scene is NULL");
    if (size < 0 || size > 16384 || (size > 0 && !texels)) return fail(MR_E_INVALID, "bad cubemap");
    int rc = ensure_init();
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    sc->sky_size = 0;
    if (size == 0) return MR_OK;
    const size_t bytes = (size_t)6 * size * size * 3;
    HIP_TRY(sc->d_sky.ensure(bytes));
    HIP_TRY(hipMemcpy(sc->d_sky.p, texels, bytes, hipMemcpyHostToDevice));
    sc->sky_size = size;
    return MR_OK;
}

int mr_scene_set_overlay(mr_scene *sc, const mr_overlay_desc *ov)
{
    if (!sc) return fail(MR_E_INVALID, "scene is NULL");
    int rc = ensure_init();
    if (rc) return rc;
    sc->ov_pending.set = false;                          // explicit lists supersede cameras left earlier
    sc->ov_points = sc->ov_segments = 0;                 // (frames already enqueued keep the lists of their slot's copy)
    sc->ov_serial += 1;
    if (!ov || ov->n_points <= 0 || ov->n_segments <= 0) return MR_OK;
    if (!ov->seg_first || !ov->seg_count || !ov->target || !ov->z)
        return fail(MR_E_INVALID, "overlay description has NULL arrays");
    if (ov->width <= 0 || ov->height <= 0 || ov->width > 32767 || ov->height > 32767)
        return fail(MR_E_INVALID, "overlay description: bad frame size");
    const int np = ov->n_points;
    // validate before the kernel trusts them: segments inside the point range, one after the other; targets pixels
    // of the frame the description names
    long long expect = 0;
    for (int s = 0; s < ov->n_segments; ++s) {
        const long long first = ov->seg_first[s], count = ov->seg_count[s];
        if (first != expect || count <= 0 || first + count > np)
            return fail(MR_E_INVALID, "overlay segments must follow each other and cover the point array");
        if (count > mr::OVERLAY_MAX_SEGMENT) return fail(MR_E_INVALID, "overlay segment longer than a frame is wide or high");
        expect = first + count;
    }
    if (expect != np) return fail(MR_E_INVALID, "overlay segments must follow each other and cover the point array");
    int32_t max_target = -1;
    for (size_t i = 0; i < (size_t)mr::OVERLAY_TARGETS * np; ++i) {
        if (ov->target[i] < 0) return fail(MR_E_INVALID, "negative overlay target");
        max_target = std::max(max_target, ov->target[i]);
    }
    if ((long long)max_target >= (long long)ov->width * ov->height)
        return fail(MR_E_INVALID, "overlay description: targets outside the height x width it names");
    sc->ov_target.assign(ov->target, ov->target + (size_t)mr::OVERLAY_TARGETS * np);
    sc->ov_z.assign(ov->z, ov->z + np);
    sc->ov_seg.resize((size_t)2 * ov->n_segments);
    for (int i = 0; i < ov->n_segments; ++i) { sc->ov_seg[2 * i] = ov->seg_first[i]; sc->ov_seg[2 * i + 1] = ov->seg_count[i]; }
    const int tiles_x = (ov->width + mr::TILE_W - 1) / mr::TILE_W, tiles_y = (ov->height + mr::TILE_H - 1) / mr::TILE_H;
    sc->ov_tile_mask.assign((size_t)tiles_x * tiles_y, 0);
    for (int32_t t : sc->ov_target) sc->ov_tile_mask[(size_t)(t / ov->width / mr::TILE_H) * tiles_x + (size_t)(t % ov->width / mr::TILE_W)] = 1;
    sc->ov_height = ov->height; sc->ov_width = ov->width;
    sc->ov_points = np; sc->ov_segments = ov->n_segments;
    return MR_OK;
}

int mr_scene_set_overlay_cameras(mr_scene *sc, const double *corners, const double *planes, const double *mvp,
                                 const double *viewport, double near_, double far_, int32_t camera_inside,
                                 int32_t height, int32_t width)
{
    if (!sc) return fail(MR_E_INVALID, "scene is NULL");
    if (!corners || !planes || !mvp || !viewport || height <= 0 || width <= 0 || height > 32767 || width > 32767)
        return fail(MR_E_INVALID, "mr_scene_set_overlay_cameras: bad argument");
    int rc = ensure_init();
    if (rc) return rc;
    mr_scene::OvPending &p = sc->ov_pending;
    std::memcpy(p.corners, corners, sizeof p.corners); std::memcpy(p.planes, planes, sizeof p.planes);
    std::memcpy(p.mvp, mvp, sizeof p.mvp); std::memcpy(p.viewport, viewport, sizeof p.viewport);
    p.near_ = near_; p.far_ = far_; p.inside = camera_inside; p.height = height; p.width = width;
    p.set = true;
    return MR_OK;
}

int mr_scene_add_model(mr_scene *sc, const mr_model_desc *m)
{
    if (!sc || !m) return fail(MR_E_INVALID, "NULL argument");
    if (!m->vertices || !m->faces || !m->materials || m->n_vertices <= 0 || m->n_faces < 0 || m->n_materials <= 0)
        return fail(MR_E_INVALID, "model needs vertices, faces and at least one material");
    const int n_tex = (int)sc->textures.size();
    for (int i = 0; i < m->n_materials; ++i) {
        const mr_material &mm = m->materials[i];
        if (mm.tex_kd >= n_tex || mm.tex_norm >= n_tex || mm.tex_ks >= n_tex)
            return fail(MR_E_INVALID, "material refers to a texture id that was never added");
        if ((mm.tex_kd >= 0 || mm.tex_norm >= 0 || mm.tex_ks >= 0) && !m->uv)
            return fail(MR_E_INVALID, "textured material on a model without uv coordinates");
        if (mm.tex_norm >= 0 && mm.norm_tangent && !m->normals)
            return fail(MR_E_INVALID, "tangent-space normal map on a model without vertex normals");
    }
    ModelInfo mi;
    mi.vert_off = (int32_t)(sc->verts.size() / 4); mi.n_verts = m->n_vertices;
    mi.uv_off = (int32_t)(sc->uv.size() / 3); mi.n_uv = m->uv ? m->n_uv : 0;
    mi.normal_off = (int32_t)(sc->normals.size() / 3); mi.n_normals = m->normals ? m->n_normals : 0;
    mi.face_off = (int32_t)(sc->faces.size() / 12); mi.n_faces = m->n_faces;
    mi.mat_off = (int32_t)sc->materials.size(); mi.n_mats = m->n_materials;
    // validate indices before touching the scene: the kernels trust them
    for (int64_t i = 0; i < (int64_t)m->n_faces * 3; ++i) {
        const int32_t *c = m->faces + i * 4;
        if (c[0] < 0 || c[0] >= m->n_vertices) return fail(MR_E_INVALID, "vertex index out of range");
        if (m->uv && (c[1] < 0 || c[1] >= m->n_uv)) return fail(MR_E_INVALID, "uv index out of range");
        if (m->normals && (c[2] < 0 || c[2] >= m->n_normals)) return fail(MR_E_INVALID, "normal index out of range");
        if (c[3] < 0 || c[3] >= m->n_materials) return fail(MR_E_INVALID, "material index out of range");
        if (m->edge_ids && (m->edge_ids[i] < -m->n_vertices || m->edge_ids[i] >= m->n_vertices))
            return fail(MR_E_INVALID, "edge id out of range");
    }
    if (g_initialised) (void)hipDeviceSynchronize();     // frames in flight still use the old scene
    sc->verts.insert(sc->verts.end(), m->vertices, m->vertices + (size_t)m->n_vertices * 4);
    if (m->uv) sc->uv.insert(sc->uv.end(), m->uv, m->uv + (size_t)m->n_uv * 3);
    if (m->normals) sc->normals.insert(sc->normals.end(), m->normals, m->normals + (size_t)m->n_normals * 3);
    for (int i = 0; i < m->n_materials; ++i) {
        const mr_material &mm = m->materials[i];
        mr::Material d;
        for (int j = 0; j < 3; ++j) { d.kd[j] = mm.kd[j]; d.ks255[j] = mm.ks255[j]; }
        d.ns = mm.ns; d.tex_kd = mm.tex_kd; d.tex_norm = mm.tex_norm; d.tex_ks = mm.tex_ks;
        d.norm_tangent = mm.norm_tangent;
        sc->materials.push_back(d);
    }
    const uint8_t ff = (uint8_t)((m->clip ? mr::FF_CLIP : 0) | (m->vertices_are_f32 ? mr::FF_VERTS_F32 : 0) |
                                 (m->normals ? mr::FF_HAS_NORMALS : 0) | (m->uv ? mr::FF_HAS_UV : 0) |
                                 (m->depth_test ? 0 : mr::FF_NO_DEPTH));
    sc->faces.reserve(sc->faces.size() + (size_t)m->n_faces * 12);
    for (int64_t i = 0; i < (int64_t)m->n_faces * 3; ++i) {
        const int32_t *c = m->faces + i * 4;
        sc->faces.push_back(c[0] + mi.vert_off);
        sc->faces.push_back(m->uv ? c[1] + mi.uv_off : 0);
        sc->faces.push_back(m->normals ? c[2] + mi.normal_off : 0);
        sc->faces.push_back(c[3] + mi.mat_off);
    }
    sc->face_flags.insert(sc->face_flags.end(), (size_t)m->n_faces, ff);
    // silhouette edges are matched on the corners' RAW vertex ids (see mr_model_desc.edge_ids): raw values
    // lie in [-n_vertices, n_vertices); shifted by n_vertices + 2 * vert_off they are unique per model
    for (int64_t i = 0; i < (int64_t)m->n_faces * 3; ++i) {
        const int32_t raw = m->edge_ids ? m->edge_ids[i] : m->faces[i * 4];
        sc->edge_raw.push_back(raw);
        sc->edge_ids.push_back(raw + m->n_vertices + 2 * mi.vert_off);
    }
    sc->models.push_back(mi);
    sc->dirty = true;
    sc->last = nullptr;
    sc->quad_cap = 0;
    for (auto &fs : sc->slots) fs->reset_caps();
    return (int)sc->models.size() - 1;
}

int mr_render(mr_scene *sc, const mr_frame_desc *fr, uint8_t *out_rgb, mr_stats *stats)
{
    if (!sc || !out_rgb) return fail(MR_E_INVALID, "NULL argument");
    int rc = validate_frame(fr);
    if (rc) return rc;
    if ((rc = ensure_init())) return rc;
    FrameSlot *fs = slot_for(sc, g_stream);
    if (!fs) return fail(MR_E_DEVICE, "out of frame slots");
    if ((fr->flags & MR_FRAME_OVERLAY) && (fr->row_begin != 0 || fr->row_end != fr->height || fr->stripe_count > 1))
        return fail(MR_E_INVALID, "the overlay of a frame split over devices is drawn on the assembled whole frame: render the "
                                  "part with mr_render_device (which appends the touched pixels' state), then mr_overlay_apply");
    mr_frame_desc counted = *fr;
    if (stats) counted.flags |= MR_FRAME_COUNTERS;         // whoever asks for the counters gets them
    fr = &counted;
    const size_t band_bytes = out_bytes(fr);
    for (int attempt = 0; attempt < 6; ++attempt) {
        HIP_TRY(fs->d_out.ensure(band_bytes));
        if ((rc = enqueue_frame(sc, fs, fr, fs->d_out.as<uint8_t>(), true))) return rc;
        if ((rc = finish_overlay(sc, fs, fs->d_out.as<uint8_t>()))) return rc;      // (the host's share of the overlay, beside the device's kernels)
        if ((rc = fetch_counters(sc, fs, stats != nullptr))) return rc;
        HIP_TRY(hipMemcpyAsync(out_rgb, fs->d_out.p, band_bytes, hipMemcpyDeviceToHost, g_stream));
        if (!(fr->flags & MR_FRAME_NO_TIMING)) { HIP_TRY(hipEventRecord(fs->ev[5], g_stream)); fs->last_copied = true; }
        HIP_TRY(hipStreamSynchronize(g_stream));
        rc = collect(sc, fs, true);
        if (rc == MR_OK) {
            if (stats) *stats = sc->stats;
            return MR_OK;
        }
        if (rc != MR_E_OVERFLOW) return rc;       // work lists were grown: render the frame again
    }
    return fail(MR_E_OVERFLOW, "work lists kept overflowing");
}

int mr_render_async(mr_scene *sc, const mr_frame_desc *fr, uint8_t *out_rgb, int32_t lane)
{
    if (!sc || !out_rgb) return fail(MR_E_INVALID, "NULL argument");
    if (lane < 0 || lane >= MR_ASYNC_LANES) return fail(MR_E_INVALID, "lane out of range");
    int rc = validate_frame(fr);
    if (rc) return rc;
    if ((rc = ensure_init())) return rc;
    if ((fr->flags & MR_FRAME_OVERLAY) && (fr->row_begin != 0 || fr->row_end != fr->height || fr->stripe_count > 1))
        return fail(MR_E_INVALID, "the overlay of a frame split over devices is drawn on the assembled whole frame (mr_overlay_apply)");
    mr_scene::Lane &ln = sc->lanes[lane];
    if (ln.busy) return fail(MR_E_INVALID, "this lane still has a frame in flight: mr_render_wait first");
    if (!ln.stream) HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
    FrameSlot *fs = slot_for(sc, ln.stream);
    if (!fs) return fail(MR_E_DEVICE, "out of frame slots");
    const size_t band_bytes = out_bytes(fr);
    HIP_TRY(fs->d_out.ensure(band_bytes));
    if ((rc = enqueue_frame(sc, fs, fr, fs->d_out.as<uint8_t>(), true))) return rc;
    if ((rc = finish_overlay(sc, fs, fs->d_out.as<uint8_t>()))) return rc;
    if ((rc = fetch_counters(sc, fs, (fr->flags & MR_FRAME_COUNTERS) != 0))) return rc;
    HIP_TRY(hipMemcpyAsync(out_rgb, fs->d_out.p, band_bytes, hipMemcpyDeviceToHost, ln.stream));
    if (!(fr->flags & MR_FRAME_NO_TIMING)) { HIP_TRY(hipEventRecord(fs->ev[5], ln.stream)); fs->last_copied = true; }
    ln.busy = true;
    return MR_OK;
}

int mr_render_wait(mr_scene *sc, int32_t lane, mr_stats *stats)
{
    if (!sc) return fail(MR_E_INVALID, "scene is NULL");
    if (lane < 0 || lane >= MR_ASYNC_LANES) return fail(MR_E_INVALID, "lane out of range");
    mr_scene::Lane &ln = sc->lanes[lane];
    if (!ln.busy) return fail(MR_E_INVALID, "no frame in flight on this lane");
    HIP_TRY(hipStreamSynchronize(ln.stream));
    ln.busy = false;
    FrameSlot *fs = slot_for(sc, ln.stream);
    if (!fs) return fail(MR_E_DEVICE, "lane without a frame slot");
    const int rc = collect(sc, fs, true);
    if (stats) *stats = sc->stats;
    if (rc == MR_E_OVERFLOW) return fail(MR_E_OVERFLOW, "the frame overflowed a work list (now grown): render it again");
    return rc;
}

int64_t mr_overlay_state_bytes(mr_scene *sc)
{
    if (!sc) return fail(MR_E_INVALID, "scene is NULL");
    realize_overlay(sc);
    if (sc->ov_points == 0) return 0;
    ensure_overlay_slots(sc);
    return (int64_t)sc->ov_touched.size() * mr::OVERLAY_STATE_BYTES;
}

int mr_overlay_apply(mr_scene *sc, const void *d_parts, int64_t part_stride, int64_t state_offset, int32_t world, int32_t striped,
                     int32_t system, void *d_frame, void *stream_)
{
    using namespace mr;
    if (!sc || !d_parts || !d_frame) return fail(MR_E_INVALID, "NULL argument");
    if (world < 1 || part_stride <= 0 || state_offset < 0 || (system != 1 && system != -1)) return fail(MR_E_INVALID, "mr_overlay_apply: bad argument");
    realize_overlay(sc);
    if (sc->ov_points == 0) return MR_OK;
    if (!striped && sc->ov_height % world) return fail(MR_E_INVALID, "mr_overlay_apply: the bands of the split must be equal");
    int rc = ensure_init();
    if (rc) return rc;
    hipStream_t stream = stream_ ? (hipStream_t)stream_ : g_stream;
    FrameSlot *fs = slot_for(sc, stream);
    if (!fs) return fail(MR_E_DEVICE, "out of frame slots");
    if ((rc = sync_slot_overlay(sc, fs, true))) return rc;
    const int n_slots = (int)sc->ov_touched.size();
    // the compact state and the bidding words, per slot of the list of touched pixels (the words zero between frames)
    {
        const size_t need = (size_t)n_slots * (8 + 12 + 4 + 4) + (size_t)sc->ov_points + 64;
        const void *had = fs->d_vclip.p;           // (a buffer this path owns: the matrix-core vertex path does not run on assembled frames)
        HIP_TRY(fs->d_vclip.ensure(need));
        if (fs->d_vclip.p != had) HIP_TRY(hipMemsetAsync(fs->d_vclip.p, 0, fs->d_vclip.cap, stream));
    }
    char *scratch = static_cast<char *>(fs->d_vclip.p);
    OverlayArgs oa;
    const char *base = static_cast<const char *>(fs->ov.lists.p);
    oa.z = reinterpret_cast<const double *>(base + fs->ov.off[0]);
    oa.idx = reinterpret_cast<const int32_t *>(base + fs->ov.off[4]);
    oa.seg = reinterpret_cast<const int32_t *>(base + fs->ov.off[2]);
    oa.pixel_of = reinterpret_cast<const int32_t *>(base + fs->ov.off[5]);
    oa.n_points = sc->ov_points; oa.n_segments = sc->ov_segments;
    oa.st_z = reinterpret_cast<double *>(scratch);
    oa.st_f = reinterpret_cast<float *>(scratch + (size_t)n_slots * 8);
    oa.win = reinterpret_cast<uint32_t *>(scratch + (size_t)n_slots * 20);
    oa.any = oa.win + n_slots;
    oa.keep = reinterpret_cast<uint8_t *>(oa.any + n_slots);
    oa.out = static_cast<uint8_t *>(d_frame); oa.out_width = sc->ov_width; oa.out_height = sc->ov_height;
    oa.gamma_lut = sc->d_gamma.as<float>();
    hipLaunchKernelGGL(k_overlay_import, dim3(blocks_for(n_slots, 256)), dim3(256), 0, stream, oa.pixel_of, n_slots,
                       static_cast<const char *>(d_parts), (size_t)part_stride, (size_t)state_offset, sc->ov_width, sc->ov_height,
                       world, striped ? 1 : 0, oa.st_z, oa.st_f);
    hipLaunchKernelGGL(k_overlay, dim3(1), dim3(OVERLAY_BLOCK), 0, stream, oa, (double)system);
    HIP_TRY(hipGetLastError());
    return MR_OK;
}

void *mr_host_alloc(uint64_t bytes)
{
    if (ensure_init() != MR_OK) return nullptr;
    void *p = nullptr;
    if (hipHostMalloc(&p, (size_t)std::max<uint64_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) {
        fail(MR_E_DEVICE, "hipHostMalloc failed");
        return nullptr;
    }
    return p;
}

// out[i][j] = a[i][0] * b[0][j], then fma(a[i][k], b[k][j], .) for k = 1 .. K-1: the order NumPy's BLAS uses for
// the reference's tiny host-side products, spelled out so that it is the same on any host.
// Host arithmetic for the Python mirror's per-frame constants; no device involved.
void mr_host_matmul_chain(const double *a, const double *b, double *out, int32_t m, int32_t k, int32_t p)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < p; ++j) {
            double acc = a[(size_t)i * k] * b[j];
            for (int x = 1; x < k; ++x) acc = std::fma(a[(size_t)i * k + x], b[(size_t)x * p + j], acc);
            out[(size_t)i * p + j] = acc;
        }
}

// Host arithmetic: everything Scene.render() derives from a camera that moved, in one call (obj/core.py:383-405,
// obj/transformation.py:57-110, obj/plane_intersection.py:43-56, as the Python mirror spells them: _look_at_axes with
// scalar arithmetic, products as ascending fma chains): look-at = translate @ rotate, MVP = look-at @ projection, and
// the six normalised frustum planes of the MVP.  `eye`, `center`, `up` are the ARGUMENTS of look_at_rotate_* (the
// reference passes the camera's centre as eye and its position as centre), `position` the camera's position (the
// translation), `lh` the handedness of the rotation.  Bit-identical to the NumPy path (tests/test_host_api.py).
void mr_host_camera_constants(const double *eye, const double *center, const double *up, const double *position,
                              const double *projection, int32_t lh, double *lookat, double *mvp, double *planes)
{
    auto unit3 = [](const double v[3], double o[3]) {
        double l = std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        if (l == 0) l = 1.0;
        for (int j = 0; j < 3; ++j) o[j] = v[j] / l;
    };
    auto cross3 = [](const double a[3], const double b[3], double o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    };
    const double d[3] = { center[0] - eye[0], center[1] - eye[1], center[2] - eye[2] };
    double forward[3], right[3], c[3], new_up[3];
    unit3(d, forward);
    cross3(up, forward, c);
    unit3(c, right);
    cross3(forward, right, new_up);
    double rot[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1 }, tr[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    for (int r = 0; r < 3; ++r) { rot[r * 4 + 0] = right[r]; rot[r * 4 + 1] = new_up[r]; rot[r * 4 + 2] = lh ? -forward[r] : forward[r]; }
    for (int j = 0; j < 3; ++j) tr[12 + j] = -position[j];
    mr_host_matmul_chain(tr, rot, lookat, 4, 4, 4);
    mr_host_matmul_chain(lookat, projection, mvp, 4, 4, 4);
    for (int axis = 0; axis < 3; ++axis)
        for (int sgn = 0; sgn < 2; ++sgn) {
            double pl[4];
            for (int r = 0; r < 4; ++r) pl[r] = sgn ? mvp[r * 4 + 3] - mvp[r * 4 + axis] : mvp[r * 4 + 3] + mvp[r * 4 + axis];
            double acc = pl[0] * pl[0];
            for (int r = 1; r < 4; ++r) acc = std::fma(pl[r], pl[r], acc);
            const double n = std::sqrt(acc);
            for (int r = 0; r < 4; ++r) planes[(2 * axis + sgn) * 4 + r] = pl[r] / n;
        }
}

// The debug-frustum overlay's statement lists on the host (host_overlay.h): build into a per-thread buffer, report
// the sizes, then copy out into arrays of those sizes.
int mr_host_overlay_build(const double *corners, const double *planes, const double *mvp, const double *viewport,
                          double near_, double far_, int32_t camera_inside, int32_t height, int32_t width,
                          int32_t *n_segments, int32_t *n_points, int32_t *n_touched)
{
    if (!corners || !planes || !mvp || !viewport || !n_segments || !n_points || !n_touched || height <= 0 || width <= 0)
        return fail(MR_E_INVALID, "mr_host_overlay_build: bad argument");
    static const int32_t faces[24] = { 2, 4, 5, 3,  0, 1, 7, 6,  0, 2, 3, 1,  5, 4, 6, 7,  3, 5, 7, 1,  4, 2, 0, 6 };
    g_overlay_lists = mr_host::OverlayLists();
    mr_host::build_overlay_lists(corners, faces, planes, mvp, viewport, near_, far_, camera_inside != 0, height, width, 13,
                                 g_overlay_lists);
    *n_segments = (int32_t)g_overlay_lists.seg_first.size();
    *n_points = (int32_t)g_overlay_lists.z.size();
    *n_touched = (int32_t)g_overlay_lists.touched.size();
    return MR_OK;
}

int mr_host_overlay_fetch(int32_t *seg_first, int32_t *seg_count, int32_t *target, int32_t *next, double *z, int32_t *touched)
{
    const mr_host::OverlayLists &o = g_overlay_lists;
    const size_t n = o.z.size();
    if (seg_first) std::copy(o.seg_first.begin(), o.seg_first.end(), seg_first);
    if (seg_count) std::copy(o.seg_count.begin(), o.seg_count.end(), seg_count);
    for (int k = 0; k < 5; ++k) {
        if (target) std::copy(o.target[k].begin(), o.target[k].end(), target + (size_t)k * n);
        if (next) std::copy(o.next[k].begin(), o.next[k].end(), next + (size_t)k * n);
    }
    if (z) std::copy(o.z.begin(), o.z.end(), z);
    if (touched) std::copy(o.touched.begin(), o.touched.end(), touched);
    return MR_OK;
}

void mr_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int mr_render_device(mr_scene *sc, const mr_frame_desc *fr, void *d_out_rgb, void *stream)
{
    if (!sc || !d_out_rgb) return fail(MR_E_INVALID, "NULL argument");
    int rc = validate_frame(fr);
    if (rc) return rc;
    if ((rc = ensure_init())) return rc;
    FrameSlot *fs = slot_for(sc, stream ? (hipStream_t)stream : g_stream);
    if (!fs) return fail(MR_E_INVALID, "a scene can be rendered from at most 32 different streams");
    return enqueue_frame(sc, fs, fr, static_cast<uint8_t *>(d_out_rgb));
}

int mr_get_stats(mr_scene *sc, mr_stats *stats)
{
    if (!sc || !stats) return fail(MR_E_INVALID, "NULL argument");
    FrameSlot *fs = last_slot(sc);
    if (!fs) return MR_E_INVALID;
    HIP_TRY(hipDeviceSynchronize());
    // a frame on any stream may have overflowed its lists: grow them all before reporting
    bool overflowed = false;
    for (auto &s : sc->slots) {
        if (!s->have_frame) continue;
        int rc = fetch_counters(sc, s.get(), true);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (collect(sc, s.get(), s->last_copied) == MR_E_OVERFLOW) overflowed = true;
    }
    (void)collect(sc, fs, fs->last_copied);     // report the most recent frame
    *stats = sc->stats;
    if (overflowed)
        return fail(MR_E_OVERFLOW, "a frame overflowed a work list (now grown): render it again");
    return MR_OK;
}

namespace {
// adds the marked frames of one slot (most recent first, at most `limit`) to acc; returns how many
uint64_t add_slot_times(const FrameSlot &s, uint64_t limit, double acc[MR_N_KERNEL_TIMES])
{
    const uint64_t have = std::min<uint64_t>(s.frames_enqueued, EVENT_RING);
    uint64_t taken = 0;
    for (uint64_t i = 0; i < have && taken < limit; ++i) {
        const uint64_t slot = (s.frames_enqueued - 1 - i) % EVENT_RING;
        const int marks = s.ev_marks[slot];
        if (!marks) continue;
        const hipEvent_t *ev = s.ev_ring[slot];
        auto span = [&](int a, int b) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[a], ev[b]); return (double)ms; };
        if (marks == 1) {
            acc[2] += span(0, 3);
        } else {
            acc[0] += span(0, 1); acc[1] += span(1, 2); acc[2] += span(2, 3);
        }
        acc[3] += span(3, 4);
        acc[4] += span(0, 4);
        ++taken;
    }
    return taken;
}
}  // namespace

int mr_get_kernel_times(mr_scene *sc, int n_frames, float *out_ms, int cap)
{
    if (!sc || !out_ms || cap < MR_N_KERNEL_TIMES) return fail(MR_E_INVALID, "need room for MR_N_KERNEL_TIMES floats");
    if (!last_slot(sc)) return MR_E_INVALID;
    HIP_TRY(hipDeviceSynchronize());
    double acc[MR_N_KERNEL_TIMES] = {};
    uint64_t used = 0;
    // only the slots that render the same kind of frame as the most recent one (same flags apart from
    // the timing bits, same rows) are averaged: a whole-frame mr_render on the library's stream must
    // not be mixed into the statistics of band frames enqueued on the caller's streams
    const mr_frame_desc &ref = sc->last->last_frame;
    constexpr int timing_bits = MR_FRAME_NO_TIMING | MR_FRAME_LIGHT_TIMING;
    auto same_kind = [&](const FrameSlot &s) {
        return s.have_frame && (s.last_frame.flags & ~timing_bits) == (ref.flags & ~timing_bits) &&
               s.last_frame.row_begin == ref.row_begin && s.last_frame.row_end == ref.row_end &&
               s.last_frame.stripe_count == ref.stripe_count && s.last_frame.stripe_index == ref.stripe_index;
    };
    int active = 0;
    for (auto &s : sc->slots) active += same_kind(*s) ? 1 : 0;
    const uint64_t per_slot = std::max<uint64_t>(1, ((uint64_t)std::max(n_frames, 1) + active - 1) / std::max(active, 1));
    for (auto &s : sc->slots)
        if (same_kind(*s)) used += add_slot_times(*s, per_slot, acc);
    for (int k = 0; k < MR_N_KERNEL_TIMES; ++k) out_ms[k] = used ? (float)(acc[k] / (double)used) : 0.f;
    return (int)used;
}

int mr_get_stream_kernel_times(mr_scene *sc, void *stream, int n_frames, float *out_ms, int cap)
{
    if (!sc || !out_ms || cap < MR_N_KERNEL_TIMES) return fail(MR_E_INVALID, "need room for MR_N_KERNEL_TIMES floats");
    const hipStream_t want = stream ? (hipStream_t)stream : g_stream;
    HIP_TRY(hipDeviceSynchronize());
    double acc[MR_N_KERNEL_TIMES] = {};
    uint64_t used = 0;
    for (auto &s : sc->slots)
        if (s->stream == want && s->have_frame) used += add_slot_times(*s, (uint64_t)std::max(n_frames, 1), acc);
    for (int k = 0; k < MR_N_KERNEL_TIMES; ++k) out_ms[k] = used ? (float)(acc[k] / (double)used) : 0.f;
    return (int)used;
}

int mr_read_z(mr_scene *sc, double *out)
{
    FrameSlot *fs = last_slot(sc);
    if (!fs) return MR_E_INVALID;
    if (!(fs->last_frame.flags & MR_FRAME_KEEP_BUFFERS))
        return fail(MR_E_INVALID, "the last frame was rendered without MR_FRAME_KEEP_BUFFERS");
    return read_back(fs->d_z, out, (size_t)fs->last_frame.width * fs->last_frame.height, "z");
}

int mr_read_stencil(mr_scene *sc, int16_t *out)
{
    FrameSlot *fs = last_slot(sc);
    if (!fs) return MR_E_INVALID;
    if (!(fs->last_frame.flags & MR_FRAME_KEEP_BUFFERS))
        return fail(MR_E_INVALID, "the last frame was rendered without MR_FRAME_KEEP_BUFFERS");
    const size_t n = (size_t)fs->last_frame.width * fs->last_frame.height;
    std::vector<int32_t> wide(n);           // the device accumulates in 32 bits; the reference's buffer is int16
    int rc = read_back(fs->d_stencil, wide.data(), n, "stencil");
    if (rc) return rc;
    if (!out) return fail(MR_E_INVALID, "NULL argument");
    for (size_t i = 0; i < n; ++i) out[i] = (int16_t)wide[i];
    return MR_OK;
}

int mr_read_winner(mr_scene *sc, int32_t *out)
{
    FrameSlot *fs = last_slot(sc);
    if (!fs) return MR_E_INVALID;
    if (!(fs->last_frame.flags & MR_FRAME_KEEP_BUFFERS))
        return fail(MR_E_INVALID, "the last frame was rendered without MR_FRAME_KEEP_BUFFERS");
    return read_back(fs->d_winner, out, (size_t)fs->last_frame.width * fs->last_frame.height, "winner");
}

int mr_read_frame_f32(mr_scene *sc, float *out)
{
    FrameSlot *fs = last_slot(sc);
    if (!fs) return MR_E_INVALID;
    if (!(fs->last_frame.flags & MR_FRAME_KEEP_FLOAT))
        return fail(MR_E_INVALID, "the last frame was rendered without MR_FRAME_KEEP_FLOAT");
    return read_back(fs->d_frame, out, (size_t)fs->last_frame.width * fs->last_frame.height * 3, "frame");
}

int mr_read_face_status(mr_scene *sc, uint8_t *out)
{
    FrameSlot *fs = last_slot(sc);
    if (!fs) return MR_E_INVALID;
    if (!(fs->last_frame.flags & MR_FRAME_FACE_STATUS))
        return fail(MR_E_INVALID, "the last frame was rendered without MR_FRAME_FACE_STATUS");
    if (fs->last_frame.row_begin != 0 || fs->last_frame.row_end != fs->last_frame.height || fs->last_frame.stripe_count > 1)
        return fail(MR_E_INVALID, "per-face status needs the whole frame on one device (no row band, no stripes)");
    return read_back(fs->d_status, out, sc->faces.size() / 12, "face status");
}

int mr_debug_read_tile_records(mr_scene *sc, uint32_t *out, int32_t cap_tiles)
{
    FrameSlot *fs = last_slot(sc);
    if (!fs) return MR_E_INVALID;
    if (!out) return fail(MR_E_INVALID, "NULL argument");
    const int n = std::min(fs->last_n_tiles, cap_tiles);
    HIP_TRY(hipDeviceSynchronize());
    static_assert(mr::TILE_REC == MR_TILE_RECORD_WORDS, "tile record size is part of the ABI");
    if (n > 0) HIP_TRY(hipMemcpy(out, fs->d_tile_stats.p, (size_t)n * mr::TILE_REC * 4, hipMemcpyDeviceToHost));
    return fs->last_n_tiles;
}

int mr_debug_clusters_culled(mr_scene *sc)
{
    FrameSlot *fs = last_slot(sc);
    if (!fs) return MR_E_INVALID;
    HIP_TRY(hipDeviceSynchronize());
    int rc = fetch_counters(sc, fs, false);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(fs->stream));
    return (int)fs->h_counters->pad0[0];
}

int mr_debug_read_tile_order(mr_scene *sc, uint32_t *out, int32_t cap_tiles)
{
    FrameSlot *fs = last_slot(sc);
    if (!fs) return MR_E_INVALID;
    if (!out) return fail(MR_E_INVALID, "NULL argument");
    const int n = std::min(fs->last_n_tiles, cap_tiles);
    HIP_TRY(hipDeviceSynchronize());
    if (n > 0 && fs->last_ordered)
        HIP_TRY(hipMemcpy(out, fs->d_hist.as<uint32_t>() + mr::ORDER_HEAD, (size_t)n * 4, hipMemcpyDeviceToHost));
    else
        for (int i = 0; i < n; ++i) out[i] = (uint32_t)i;     // frames rendered from several streams keep row-major order
    return fs->last_n_tiles;
}

int mr_read_silhouette(mr_scene *sc, int32_t *out, int32_t cap)
{
    FrameSlot *fs = last_slot(sc);
    if (!fs) return MR_E_INVALID;
    const int n = sc->n_silhouette;
    const int take = std::min(std::min(n, cap), (int)fs->quad_cap);
    if (take > 0 && out) {
        std::vector<int32_t> raw((size_t)take * 2);
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(raw.data(), fs->d_sil.p, raw.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int i = 0; i < take; ++i) {
            const int face = raw[i * 2], k = raw[i * 2 + 1];
            int model = 0;
            while (model + 1 < (int)sc->models.size() && face >= sc->models[model + 1].face_off) ++model;
            out[i * 3 + 0] = model;                              // entries of model.silhouette carry the raw ids
            out[i * 3 + 1] = sc->edge_raw[(size_t)face * 3 + k];
            out[i * 3 + 2] = sc->edge_raw[(size_t)face * 3 + (k + 1) % 3];
        }
    }
    return n;
}

}  // extern "C"
