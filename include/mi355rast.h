/*
 * mi355rast.h -- C ABI of libmi355rast.so, the MI355X (gfx950) rasteriser behind
 * py_numpy_renderer_amd.Scene.render().
 *
 * The reference (Denizantip/py-numpy-renderer) has no FFI: its boundary for this path is
 * the Python call Scene.render() -> uint8 (H, W, 3) (obj/core.py:587-640).  Each entry
 * point below names the reference code it stands in for; a host in any language binds
 * these symbols (INTEGRATION.md shows the ctypes stub).
 *
 * Conventions: plain pointers and sizes only; every function returns MR_OK (0) or a
 * negative MR_E_* code and never throws; mr_last_error() describes the last failure on the
 * calling thread.  The caller owns every host pointer it passes (they may be freed as soon
 * as the call returns); the library owns all device memory.  A scene is used from one
 * thread at a time.  Matrices are row-major float64 in the reference's row-vector
 * convention (clip = v @ MVP).
 */
#ifndef MI355RAST_H
#define MI355RAST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MR_ABI_VERSION 3

enum {
    MR_OK = 0,
    MR_E_INVALID = -1,      /* bad argument / inconsistent sizes */
    MR_E_DEVICE = -2,       /* HIP runtime error (no GPU, out of memory, launch failure) */
    MR_E_UNSUPPORTED = -3,  /* feature of the reference that this build does not implement */
    MR_E_OVERFLOW = -4      /* internal work list could not be grown */
};

/* light kinds: obj/lightning.py:4-7 */
enum { MR_LIGHT_DIRECTIONAL = 0, MR_LIGHT_POINT = 1, MR_LIGHT_SPOT = 2 };

/* per-face result of the lit pass, numerically the reference's Errors flag (obj/triangular.py:15-20) */
enum {
    MR_FACE_RENDERED = 0, MR_FACE_BACK_FACE_CULLING = 1, MR_FACE_WRONG_MIN_MAX = 2,
    MR_FACE_EMPTY_B = 4, MR_FACE_EMPTY_Z = 8, MR_FACE_CLIPPED = 16
};

/* mr_frame_desc.flags */
enum {
    MR_FRAME_SHADOWS = 1,     /* run the shadow-volume stencil pass (obj/core.py:610-622) */
    MR_FRAME_KEEP_FLOAT = 2,  /* also keep the float32 frame (needed by mr_read_frame_f32) */
    MR_FRAME_FACE_STATUS = 4, /* also compute the per-face status histogram (obj/core.py:625-636) */
    MR_FRAME_LIGHT_TIMING = 8, /* record only the event marks around the frame and the tile kernel */
    MR_FRAME_SKYBOX = 16,     /* fill the background from the cubemap of mr_scene_set_skybox (obj/core.py:595-596) */
    MR_FRAME_COUNTERS = 32,   /* keep the reference-equivalent fragment counters of mr_stats, and the reference's
                                 stencil values at pixels no triangle covers (mr_read_stencil).  Without it only
                                 the frame is the contract and the library skips work that cannot change it
                                 (shadow quads that cannot pass the depth test anywhere in a strip of pixels);
                                 mr_render sets it by itself when it is handed a stats pointer. */
    MR_FRAME_NO_TIMING = 128, /* record no HIP events at all for this frame (each one costs a few microseconds between
                                 two kernels): mr_stats.gpu_ms_* read 0 and mr_get_kernel_times skips the frame */
    MR_FRAME_OVERLAY = 256,   /* replay the statement lists of mr_scene_set_overlay (the debug-camera frustum of
                                 obj/core.py:638) on the frame after the tile kernel; implies MR_FRAME_KEEP_BUFFERS
                                 and MR_FRAME_KEEP_FLOAT.  On part of a frame (row band / stripes;
                                 mr_render_device only) the overlay is not drawn: the state it needs is appended to the
                                 rows for mr_overlay_apply, see there */
    MR_FRAME_KEEP_BUFFERS = 64 /* also write the reference's working buffers (z_buffer, stencil_buffer, winner
                                 face per pixel; obj/core.py:588-591) to device memory for mr_read_z /
                                 mr_read_stencil / mr_read_winner.  Without it they only ever exist on chip,
                                 tile by tile.  MR_FRAME_FACE_STATUS implies it. */
};

typedef struct mr_scene mr_scene;

/* Per-frame constants: what Scene.render() derives from camera / debug_camera / light /
 * resolution before its loops (obj/core.py:588-600, 394-429; obj/transformation.py:123-136;
 * obj/plane_intersection.py:43-56).  The host computes them in float64. */
typedef struct mr_frame_desc {
    int32_t width, height;          /* Scene.resolution = (height, width) */
    int32_t system;                 /* SYSTEM.RH = +1, SYSTEM.LH = -1 (obj/constants.py:29-31) */
    int32_t backface_culling;       /* Camera.backface_culling */
    int32_t light_type;             /* MR_LIGHT_* */
    int32_t flags;                  /* MR_FRAME_* */
    int32_t row_begin, row_end;     /* output rows [row_begin, row_end) this device renders;
                                       0, height for the whole frame (screen-tile split, contiguous bands) */
    int32_t stripe_count, stripe_index;  /* screen-tile split, interleaved: with stripe_count = N > 1 this device
                                       renders the tile rows (16 screen rows each, counted from the BOTTOM of the
                                       frame like the reference's buffers) t with t mod N == stripe_index;
                                       row_begin / row_end must then be 0 / height.  The output buffer holds
                                       ceil(ceil(height / 16) / N) blocks of 16 x width x 3 bytes, this device's
                                       highest tile row first, rows inside a block top-down (multigpu.py
                                       un-permutes after the all-gather).  0 or 1 = off */
    double mvp[16];                 /* camera.MVP */
    double viewport[16];            /* camera.viewport */
    double debug_mvp[16];           /* debug_camera.MVP (obj/triangular.py:39) */
    double frustum_planes[24];      /* camera.frustum_planes: left,right,bottom,top,near,far */
    double z_near, z_far;           /* camera.near / camera.far */
    double camera_pos[3];
    double light_pos[3], light_dir[3], light_color[3], light_ambient[3];
    double specular_strength;
    double att_constant, att_linear, att_quadratic;
    double spot_edge0, spot_edge1;  /* cos(20 deg), cos(10 deg) (obj/triangular.py:158-159) */
    float background[3];            /* obj/core.py:597-600 */
    uint32_t background_u8;         /* the same colour after obj/core.py:640's finalise, computed by the host:
                                       r | g << 8 | b << 16 | 1 << 24; 0 = let the device compute it */
    /* MR_FRAME_SKYBOX only (obj/cube_map.py:83-101): the two screen-filling triangles' vertices
     * truncated to int, [triangle][vertex][x, y], and their un-projected corner rays
     * face @ inv(view_without_translation @ projection) / w, [triangle][vertex][x, y, z] */
    int32_t sky_tri[12];
    double sky_rays[18];
} mr_frame_desc;

/* One material group of a model (obj/materials.py:47-55; obj/core.py:125). */
typedef struct mr_material {
    double kd[3];                   /* Material.Kd */
    double ks255[3];                /* Material.Ks * 255, evaluated by the host in Ks's dtype (obj/core.py:152) */
    double ns;                      /* Material.Ns */
    int32_t tex_kd, tex_norm, tex_ks;  /* ids from mr_scene_add_texture, or -1 */
    int32_t norm_tangent;           /* normal map is tangent-space (dtype.metadata['tangent']) */
} mr_material;

/* One Model (obj/core.py:231-251): arrays exactly as Model.load_model leaves them, with
 * indices already made non-negative. */
typedef struct mr_model_desc {
    const double *vertices;         /* (n_vertices, 4) Model.vertices widened to float64 */
    const float *uv;                /* (n_uv, 3) Model.uv, or NULL */
    const float *normals;           /* (n_normals, 3) Model.normals, or NULL */
    const int32_t *faces;           /* (n_faces, 3, 4) Model._faces: [vertex, uv, normal, material] per corner */
    const mr_material *materials;   /* (n_materials) indexed by the material column */
    const int32_t *edge_ids;        /* (n_faces, 3) the vertex column of Model._faces exactly as the loader left it
                                       (negative = relative index, obj/core.py:313), or NULL when it equals the
                                       vertex indices in `faces`.  The reference's silhouette set hashes these raw
                                       values (obj/triangular.py:286-302): an edge written once as (3, 2) and once
                                       as (-7, -8) does not cancel although both name the same two vertices */
    int32_t n_vertices, n_uv, n_normals, n_faces, n_materials;
    int32_t vertices_are_f32;       /* Model.vertices.dtype == float32: edge vectors and the silhouette
                                       normal are then formed in float32 like NumPy does */
    int32_t clip;                   /* Model.clip */
    int32_t depth_test;             /* Model.depth_test; 0: the model's fragments are tested against the z-buffer but
                                       never written to it (obj/triangular.py:117) */
} mr_model_desc;

/* Counters of the last mr_render call (BASELINE.md fragment definition). */
typedef struct mr_stats {
    int64_t frag_tri;               /* (triangle, pixel) pairs with u,v,w >= 0 (obj/triangular.py:78) */
    int64_t frag_quad;              /* (shadow quad, pixel) pairs inside the quad (obj/triangular.py:347) */
    int64_t covered_px;             /* pixels with a z-buffer winner */
    int64_t lit_px;                 /* covered pixels with stencil == 0 */
    int64_t stencil_updates;        /* quad fragments that passed the z test */
    int64_t n_faces, n_faces_setup; /* faces in the scene / faces that produced a pixel box */
    int64_t n_quads, n_quads_drawn; /* silhouette edges / quads that reached rasterisation */
    int64_t tri_bin_entries, quad_bin_entries;   /* (primitive, tile) pairs */
    float gpu_ms_total;             /* device time of the whole frame (HIP events) */
    float gpu_ms_setup;             /* k_setup: vertex transform, face set-up, silhouettes, shadow-quad set-up, own tile lists */
    float gpu_ms_binning;           /* k_bin_work: tile lists of the large primitives, leftover survivor counts */
    float gpu_ms_tile;              /* k_tile: coverage, z, stencil, shading, finalise */
    float gpu_ms_copy;              /* device -> host copy of the uint8 frame */
} mr_stats;

/* Selects the HIP device for the calling process (one process per GPU) and creates the
 * library's stream.  device < 0 keeps the current device. */
int mr_init(int device);

/* 1 when a HIP device is visible, else 0.  Never fails. */
int mr_device_available(void);

int mr_abi_version(void);

/* sizeof() of the ABI structs as this build sees them (0 mr_frame_desc, 1 mr_material,
 * 2 mr_model_desc, 3 mr_stats, 4 mr_overlay_desc; -1 otherwise), so a binding can verify its own layout. */
int mr_abi_struct_size(int which);

/* Scene() -- obj/core.py:563-582.  Returns NULL on failure. */
mr_scene *mr_scene_create(void);
void mr_scene_destroy(mr_scene *scene);

/* Model.textures.register / parse_mtl texture load (obj/core.py:90-105, 334-342): uploads a
 * float32 (h, w, 3) image exactly as the reference stores it.  Returns the texture id (>= 0)
 * or a negative error. */
int mr_scene_add_texture(mr_scene *scene, const float *rgb, int32_t h, int32_t w);

/* Scene(skymap=CubeMap(...)) -- obj/cube_map.py:22-34: the six cubemap faces as one uint8
 * (6, size, size, 3) stack in the reference's face order and orientation (CubeMap.textures * 255).
 * size = 0 removes the skybox.  Used by frames that set MR_FRAME_SKYBOX. */
int mr_scene_set_skybox(mr_scene *scene, const uint8_t *texels, int32_t size);

/* Scene.add_model (obj/core.py:584-585).  Returns the model index (>= 0) or a negative error. */
int mr_scene_add_model(mr_scene *scene, const mr_model_desc *model);

/* Drops all models and textures (keeps device allocations for reuse). */
int mr_scene_clear(mr_scene *scene);

/* Tuning / test hook: capacities of the per-frame work lists -- entries per 16x16 tile for the small
 * triangle pairs, big triangle pairs and shadow quads, and entries of the large primitives' work list;
 * 0 keeps the current value.  The lists grow by themselves (a frame that overflowed one is reported by
 * mr_render / mr_get_stats as MR_E_OVERFLOW internally and rendered again); this only sets where they start. */
int mr_scene_set_list_capacities(mr_scene *scene, uint32_t small_pairs, uint32_t big_pairs, uint32_t quads, uint32_t work);

/* The debug-camera frustum overlay (obj/core.py:638; obj/frustums.py:46-103; obj/line.py:6-16) as flat
 * statement lists.  The host walks the frustum's edges (float64 DDA) segment by segment; every kept point
 * of a segment performs the reference's writes in the reference's order: centre (z and red), then for step
 * -1 and +1: z into the row neighbour, z into the column neighbour, half blend into the row neighbour, half
 * blend into the column neighbour.  target is (5, n_points) row-major: per target set (centre, row-1, col-1,
 * row+1, col+1) the linear pixel index row * width + col (row = screen y, not flipped) in a frame of height x
 * width; segments follow each other in drawing order and cover the point array.  The device replays the segments
 * in order (csrc/kernels_overlay.h).  Pointers are read during the call only.  NULL (or n_points == 0) removes the overlay.  Frames that set
 * MR_FRAME_OVERLAY replay it; the frame must have the size named here. */
typedef struct mr_overlay_desc {
    int32_t n_segments, n_points, height, width;
    const int32_t *seg_first, *seg_count;   /* (n_segments) first point and number of points of each segment */
    const int32_t *target;                  /* (5, n_points) */
    const double *z;                        /* (n_points) linearised depth of each point */
} mr_overlay_desc;
int mr_scene_set_overlay(mr_scene *scene, const mr_overlay_desc *overlay);

/* The same overlay straight from the two cameras, everything on the host side of the library in one call
 * (replaces obj/frustums.py:61-103 + obj/line.py:6-16: clipping, projection, DDA, dashes, index wrapping; the
 * upload is one asynchronous copy in front of the overlay kernel of the next frame that draws it; the lists are built
 * when first needed -- by mr_render / mr_render_async after the frame's kernels have been launched): the frustum's eight corners
 * (8 x 4, already divided by w: CUBE @ inv(debug MVP), obj/frustums.py:52-53), the viewing camera's six planes
 * (6 x 4), its MVP and viewport (4 x 4, row vectors), near / far, whether the viewing camera sits inside the
 * frustum (obj/frustums.py:57-60), and the frame's size. */
int mr_scene_set_overlay_cameras(mr_scene *scene, const double *corners, const double *planes, const double *mvp,
                                 const double *viewport, double near_plane, double far_plane, int32_t camera_inside,
                                 int32_t height, int32_t width);

/* Scene.render() -- obj/core.py:587-640: depth/ambient pass, shadow-volume stencil pass, lit
 * pass and finalise (flip, **0.8, *255, uint8) on the GPU.  out_rgb receives
 * (row_end - row_begin) x width x 3 bytes, row 0 = top row of the band.  stats may be NULL. */
int mr_render(mr_scene *scene, const mr_frame_desc *frame, uint8_t *out_rgb, mr_stats *stats);

/* The overlay on a frame split over several devices.  The lines test z at pixels other devices own, so a device that
 * renders only part of the frame (row_begin / row_end of equal bands, or stripe_count > 1) with MR_FRAME_OVERLAY does not
 * draw the overlay; it APPENDS the state (z, float colour: 24 bytes) of every touched pixel it owns to its rows, at
 * offset round_up(bytes of its rows, 16) of d_out_rgb, mr_overlay_state_bytes() bytes in all (the same on every device:
 * one entry per touched pixel, the owner fills its own).  After the ONE all-gather of rows + state, every device calls
 * mr_overlay_apply: d_parts holds the `world` parts, part_stride bytes apart, the state at state_offset in each;
 * `striped` says how the rows were split (interleaved tile rows, else equal bands); d_frame is the assembled uint8
 * frame (height x width x 3, row 0 = top), whose touched pixels are rewritten.  The result is the frame one device
 * renders with the overlay on.  Work is enqueued on `stream` (NULL = the library's own). */
int64_t mr_overlay_state_bytes(mr_scene *scene);
int mr_overlay_apply(mr_scene *scene, const void *d_parts, int64_t part_stride, int64_t state_offset, int32_t world,
                     int32_t striped, int32_t system, void *d_frame, void *stream);

/* mr_render in two halves, for frames of a sequence: mr_render_async enqueues the frame and the device-to-host copy
 * of its uint8 rows into out_rgb on one of the scene's MR_ASYNC_LANES lanes (a stream and a set of work buffers
 * each) and returns at once; mr_render_wait blocks until that lane's frame is in out_rgb.  With two lanes the copy
 * of frame i (6 MB at 1080p: longer than the frame's kernels) runs beside the kernels of frame i + 1 and beside the
 * host's preparation of frame i + 2.  out_rgb should be page-locked (mr_host_alloc) for the copy to be
 * asynchronous at all.  A frame that overflowed a work list is reported by mr_render_wait as MR_E_OVERFLOW (the
 * lists have been grown): render it again.  stats may be NULL. */
#define MR_ASYNC_LANES 4
int mr_render_async(mr_scene *scene, const mr_frame_desc *frame, uint8_t *out_rgb, int32_t lane);
int mr_render_wait(mr_scene *scene, int32_t lane, mr_stats *stats);

/* Page-locked host memory for mr_render's out_rgb (hipHostMalloc / hipHostFree): the device-to-host copy
 * of the frame then runs at the full PCIe rate instead of being staged through the runtime's own buffers.
 * Optional: mr_render accepts any host pointer.  mr_host_alloc returns NULL on failure. */
void *mr_host_alloc(uint64_t bytes);
void mr_host_free(void *p);

/* Host arithmetic, no device involved: out (m x p) = a (m x k) @ b (k x p) in float64, every element
 * rn(a[i][0] * b[0][j]) followed by fma steps in ascending k -- the order the reference's NumPy / OpenBLAS stack
 * uses for its 4x4 products (obj/core.py:383-405, obj/transformation.py), which the per-frame constants must
 * reproduce bit for bit.  The Python mirror builds its matrices, planes and overlay polygons with it. */
void mr_host_matmul_chain(const double *a, const double *b, double *out, int32_t m, int32_t k, int32_t p);

/* Host arithmetic too: what Scene.render() derives from a camera that moved, in one call -- look-at = translate @
 * rotate, MVP = look-at @ projection (obj/core.py:383-405, obj/transformation.py:57-110) and the six normalised
 * frustum planes of the MVP (obj/plane_intersection.py:43-56), with the operation order of the Python mirror (scalar
 * look-at axes, products as ascending fma chains).  eye / center / up are the arguments of look_at_rotate_lh|rh (the
 * reference passes the camera's centre as eye and its position as centre), position the camera's position,
 * projection the 4 x 4 projection matrix (row vectors), lh != 0 the left-handed rotation.  Outputs: lookat (4 x 4),
 * mvp (4 x 4), planes (6 x 4: left, right, bottom, top, near, far). */
void mr_host_camera_constants(const double *eye, const double *center, const double *up, const double *position,
                              const double *projection, int32_t lh, double *lookat, double *mvp, double *planes);

/* Host arithmetic too: the statement lists of the debug-camera frustum overlay (what mr_scene_set_overlay takes),
 * from the frustum's eight corners (8 x 4, already divided by w: CUBE @ inv(debug MVP), obj/frustums.py:52-53), the
 * viewing camera's six planes (6 x 4), its MVP and viewport (4 x 4, row vectors), near / far, and whether the
 * viewing camera sits inside the frustum (obj/frustums.py:57-60).  Replaces obj/frustums.py:61-103 +
 * obj/line.py:6-16 on the host: clipping, projection, DDA, dashes, index wrapping, same-target links.
 * _build leaves the lists in a per-thread buffer and reports their sizes; _fetch copies them into the caller's
 * arrays (target / next: 5 x n_points, row-major; any pointer may be NULL). */
int mr_host_overlay_build(const double *corners, const double *planes, const double *mvp, const double *viewport,
                          double near_plane, double far_plane, int32_t camera_inside, int32_t height, int32_t width,
                          int32_t *n_segments, int32_t *n_points, int32_t *n_touched);
int mr_host_overlay_fetch(int32_t *seg_first, int32_t *seg_count, int32_t *target, int32_t *next, double *z,
                          int32_t *touched);

/* Same, but leaves the uint8 band in device memory at d_out_rgb (a device pointer owned by
 * the caller, e.g. a torch tensor's data_ptr) and does not synchronise the host: work is
 * enqueued on `stream` (a hipStream_t; NULL = the library's own stream).  Used by the
 * multi-GPU path, which all-gathers the bands with RCCL. */
int mr_render_device(mr_scene *scene, const mr_frame_desc *frame, void *d_out_rgb, void *stream);

/* Counters / timings of the last mr_render on this scene. */
int mr_get_stats(mr_scene *scene, mr_stats *stats);

/* Average device time in milliseconds (HIP events on the stream the kernels ran on) of each
 * stage over the most recent frames that carry event marks (at most n_frames of them; 64 frames per
 * stream are remembered, frames rendered with MR_FRAME_NO_TIMING are skipped).
 *   [0] k_vertex_mfma (0 unless MR_VERTEX_PATH=mfma)   [1] k_setup   [2] k_bin_work   [3] k_tile
 *   [4] whole frame (start -> after k_tile)
 * Frames rendered with MR_FRAME_LIGHT_TIMING report [0..1] as 0 and [2] as the span from the start of
 * the frame to the start of k_tile.
 * Synchronises the device.  Returns the number of frames averaged or a negative error. */
#define MR_N_KERNEL_TIMES 5
int mr_get_kernel_times(mr_scene *scene, int n_frames, float *out_ms, int cap);
/* The same over the frames enqueued on ONE stream (as passed to mr_render_device; NULL = the library's own). */
int mr_get_stream_kernel_times(mr_scene *scene, void *stream, int n_frames, float *out_ms, int cap);

/* Debug taps for parity tests: the reference's working buffers after the last render
 * (obj/core.py:588-591).  Row = screen y (not flipped), as in the reference. */
/* z, stencil and winner need a frame rendered with MR_FRAME_KEEP_BUFFERS. */
int mr_read_z(mr_scene *scene, double *out_hw);            /* z_buffer, float64 (H, W) */
int mr_read_stencil(mr_scene *scene, int16_t *out_hw);     /* stencil_buffer, int16 (H, W) */
int mr_read_winner(mr_scene *scene, int32_t *out_hw);      /* face that owns each pixel, -1 = none */
int mr_read_frame_f32(mr_scene *scene, float *out_hw3);    /* float frame before finalise (needs MR_FRAME_KEEP_FLOAT) */
int mr_read_face_status(mr_scene *scene, uint8_t *out_faces);  /* MR_FACE_* per face (needs MR_FRAME_FACE_STATUS) */
/* Silhouette edges of the last frame as (model, a, b) triples, oriented like the entries of
 * the reference's model.silhouette (obj/triangular.py:294-302).  Returns the number of edges
 * (may exceed cap; only cap are written) or a negative error. */
int mr_read_silhouette(mr_scene *scene, int32_t *out_triples, int32_t cap);

/* Diagnostics: the tile kernel's per-tile records of the last frame, MR_TILE_RECORD_WORDS uint32 per tile:
 * [0..4] triangle fragments, quad fragments, stencil updates, covered px, lit px (MR_FRAME_COUNTERS);
 * [5..7] lengths of the tile's lists: small triangle pairs, big triangle pairs, shadow quads;
 * [8],[9] start and end of the tile's workgroup in 10 ns ticks (low 32 bits); [10],[11] the ticks at
 * which its coverage / z phase and its shadow-quad phase ended.
 * Returns the number of tiles (tiles are 16 x 16 px, row-major over the rendered rows) or a negative error. */
#define MR_TILE_RECORD_WORDS 12
int mr_debug_read_tile_records(mr_scene *scene, uint32_t *out, int32_t cap_tiles);

/* Diagnostic: how many 64-face clusters the last frame's set-up kernel dropped whole (off the screen, off this device's
 * rows, or -- when the frame culls back faces -- turned away from the camera) before reading a face of them.  Counted
 * only when the environment has MR_CLUSTER_CULL=count; 0 otherwise.  The faces that ARE set up, and the frame, do not
 * depend on it.  By default clusters are culled on partial frames (a row band, stripes), where most of them go, and not
 * on whole frames, where the test costs more than it saves; MR_CLUSTER_CULL=0 / 1 / box force it off / on /
 * on with the screen and row test only. */
int mr_debug_clusters_culled(mr_scene *scene);

/* Diagnostics: the order in which the most recent frame's tile kernel took its tiles (entry b = the tile
 * of workgroup b): heaviest first by the estimate the slot's previous frame left, row-major for a first
 * frame or a new tile grid.  Always a permutation of 0 .. n_tiles-1.  Returns the number of tiles. */
int mr_debug_read_tile_order(mr_scene *scene, uint32_t *out, int32_t cap_tiles);

/* Human-readable description of the last error on this thread ("" if none). */
const char *mr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355RAST_H */
