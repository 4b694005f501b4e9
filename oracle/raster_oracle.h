/*
 * raster_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded restatement of the reference renderer's hot path
 * (Denizantip/py-numpy-renderer, obj/core.py:587-640 Scene.render and the functions it
 * drives in obj/triangular.py, obj/transformation.py, obj/plane_intersection.py).
 * It exists to check the HIP path (tests/, __graft_entry__.smoke()) and to be timed as the
 * CPU baseline (bench.py cpu_baseline, kind "port").  Nothing under py-numpy-renderer_amd/
 * may include, link or call it.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks this restatement against
 * frames / z-buffers / stencil buffers / silhouettes captured from the reference itself
 * (tests/golden/make_golden.py, run in the build container where /root/reference exists).
 */
#ifndef RASTER_ORACLE_H
#define RASTER_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_LIGHT_DIRECTIONAL = 0, ORC_LIGHT_POINT = 1, ORC_LIGHT_SPOT = 2 };

/* per-face status, numerically equal to the reference's Errors flag (triangular.py:15-20) */
enum {
    ORC_FACE_RENDERED = 0,
    ORC_FACE_BACK_FACE_CULLING = 1,
    ORC_FACE_WRONG_MIN_MAX = 2,
    ORC_FACE_EMPTY_B = 4,
    ORC_FACE_EMPTY_Z = 8,
    ORC_FACE_CLIPPED = 16
};

enum { ORC_FLAG_SHADOWS = 1, ORC_FLAG_OVERLAY = 2 };

typedef struct orc_frame {
    int32_t width, height;
    int32_t system;            /* +1 right-handed, -1 left-handed (constants.py:29-31) */
    int32_t backface_culling;
    int32_t light_type;
    int32_t flags;
    double mvp[16];            /* camera.MVP, row-major, clip = v @ MVP (core.py:419-421) */
    double viewport[16];       /* camera.viewport (transformation.py:123-136) */
    double debug_mvp[16];      /* debug_camera.MVP (triangular.py:39) */
    double planes[24];         /* camera.frustum_planes, 6 x (a,b,c,d) (plane_intersection.py:43-56) */
    double z_near, z_far;
    double camera_pos[3];
    double light_pos[3], light_dir[3], light_color[3], light_ambient[3];
    double specular_strength;
    double att_constant, att_linear, att_quadratic;
    double spot_edge0, spot_edge1;   /* cos(20 deg), cos(10 deg) (triangular.py:158-159) */
    float background[3];
    /* cubemap skybox (cube_map.py:83-101); sky_texels == NULL -> plain background colour */
    int32_t sky_size;
    const uint8_t *sky_texels;   /* (6, size, size, 3) uint8, CubeMap.textures * 255 */
    int32_t sky_tri[12];         /* the two triangles' screen vertices truncated to int, [t][v][xy] */
    double sky_rays[18];         /* their un-projected corner rays / w, [t][v][xyz] */
    /* Screen-tile split (what one rank of the multi-GPU path owns; all 0 = the whole frame): output
     * rows [own_row_begin, own_row_end) and, with own_stripe_count = N > 1, only the tile rows (16 screen
     * rows, counted from the bottom) t with t mod N == own_stripe_index.  Fragments on rows the rank
     * does not own are dropped; the buffers keep their initial values there. */
    int32_t own_row_begin, own_row_end, own_stripe_count, own_stripe_index;
} orc_frame;

typedef struct orc_texture {
    const float *rgb;          /* (h, w, 3) float32, row 0 first */
    int32_t h, w;
} orc_texture;

typedef struct orc_material {
    double kd[3];              /* material.Kd */
    double ks255[3];           /* material.Ks * 255 evaluated by the host in Ks's own dtype (core.py:152) */
    double ns;                 /* material.Ns */
    int32_t tex_kd, tex_norm, tex_ks;   /* texture index or -1 */
    int32_t norm_tangent;      /* normal map is tangent-space */
} orc_material;

typedef struct orc_model {
    const double *verts;       /* (n_verts, 4) world-space, already widened to f64 */
    const float *uv;           /* (n_uv, 3) or NULL */
    const float *normals;      /* (n_normals, 3) or NULL */
    const int32_t *faces;      /* (n_faces, 3, 4): per corner [vi, ti, ni, material] all >= 0 */
    const orc_material *materials;
    int32_t n_verts, n_uv, n_normals, n_faces, n_materials;
    int32_t verts_f32;         /* vertices were float32: edge vectors / silhouette normal use f32 arithmetic */
    int32_t clip;              /* Model.clip */
    int32_t depth_test;        /* Model.depth_test */
    /* (n_faces, 3) vertex column of Model._faces exactly as the loader left it (negative = relative
     * index, core.py:313), or NULL when it equals the vertex indices above.  The reference's Edge objects
     * hash these raw values (triangular.py:286-302), so an edge written once as (3, 2) and once as
     * (-7, -8) does not cancel although both name the same two vertices. */
    const int32_t *edge_ids;
} orc_model;

typedef struct orc_stats {
    int64_t frag_tri_pass1;    /* (triangle,pixel) pairs with u,v,w >= 0, pass 1 */
    int64_t frag_tri_pass2;
    int64_t frag_quad;         /* (shadow quad,pixel) pairs inside the quad */
    int64_t shaded_pass1;      /* fragments that passed the z test in pass 1 */
    int64_t shaded_pass2;
    int64_t bbox_px_tri;       /* sum of triangle bounding-box pixels, one pass */
    int64_t bbox_px_quad;
    int64_t n_quads;           /* silhouette edges */
    int64_t n_quads_drawn;     /* quads surviving the clip with >= 3 vertices and a box */
    int64_t stencil_updates;   /* quad fragments passing the z test */
} orc_stats;

typedef struct orc_outputs {
    float *frame;              /* (h, w, 3) float frame before finalisation, row = screen y; may be NULL */
    double *z;                 /* (h, w) */
    int16_t *stencil;          /* (h, w) */
    int32_t *winner;           /* (h, w) global face index of the last pass-1 writer, -1 = none; may be NULL */
    uint8_t *out;              /* (h, w, 3) final frame, row 0 = top; may be NULL */
    uint8_t *face_status;      /* concatenated per-face pass-2 status; may be NULL */
    int32_t *silhouette;       /* (cap, 3): model, a, b oriented; may be NULL */
    int32_t silhouette_cap;
    orc_stats stats;
} orc_outputs;

/* Renders one frame exactly like a first Scene.render() call on freshly loaded models
 * (the debug-frustum overlay and the cubemap skybox are separate entry points).
 * Returns 0, or a negative value on allocation failure / invalid input. */
int orc_render(const orc_frame *frame, const orc_model *models, int32_t n_models,
               const orc_texture *textures, int32_t n_textures, orc_outputs *out);

/* float frame -> uint8 (core.py:640); rows flipped. */
void orc_finalise(const float *frame, int32_t h, int32_t w, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
