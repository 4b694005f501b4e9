/*
 * raster_oracle.c -- TEST INFRASTRUCTURE (see raster_oracle.h).
 *
 * Sequential restatement of the reference's three render loops.  Every floating-point
 * expression is written in the operation order the reference's NumPy/OpenBLAS stack
 * evaluates it in: products that go through BLAS are explicit fma()
 * chains, NumPy element-wise expressions are separate rounded operations.  Build with
 * -ffp-contract=off so the compiler adds no fusion of its own.
 *
 * Reference lines are cited as file:line relative to the upstream repository's obj/ dir.
 */
#include "raster_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ BLAS-order products */

/* GEMM (M,K)@(K,P), P >= 2, and 1-D dot: ascending chain, first term a rounded product. */
static inline double chain2(double a0, double a1, double b0, double b1)
{
    return fma(a1, b1, a0 * b0);
}
static inline double chain3(double a0, double a1, double a2, double b0, double b1, double b2)
{
    return fma(a2, b2, fma(a1, b1, a0 * b0));
}
static inline double chain4(const double *a, const double *b, int bs)
{
    double acc = a[0] * b[0];
    acc = fma(a[1], b[bs], acc);
    acc = fma(a[2], b[2 * bs], acc);
    return fma(a[3], b[3 * bs], acc);
}
/* GEMV (N,2)@(2,): fma(a0,b0, rn(a1*b1)) */
static inline double gemv2(double a0, double a1, double b0, double b1)
{
    return fma(a0, b0, a1 * b1);
}
/* GEMV (N,3)@(3,) and (N,3)@(3,1): fma(a2,b2, fma(a0,b0, rn(a1*b1))) */
static inline double gemv3(double a0, double a1, double a2, double b0, double b1, double b2)
{
    return fma(a2, b2, fma(a0, b0, a1 * b1));
}
/* (a*b).sum(axis=1), K = 3: three rounded products, added left to right */
static inline double sum3(const double *a, const double *b)
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
/* transformation.py:46-49 normalize() on a float64 3-vector */
static inline void normalize3(const double *a, double *o)
{
    double l = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    if (l == 0) l = 1;
    o[0] = a[0] / l; o[1] = a[1] / l; o[2] = a[2] / l;
}

/* ------------------------------------------------------------------ per-frame constants */

typedef struct {
    const orc_frame *f;
    int W, H;
    double two_nf, f_plus_n, f_minus_n;
} ctx_t;

/* Does the rank this frame is rendered for own screen row py (orc_frame.own_*)? */
static inline int owned_row(const ctx_t *c, int py)
{
    const orc_frame *f = c->f;
    if (f->own_row_end > f->own_row_begin) {
        const int out_row = c->H - 1 - py;
        if (out_row < f->own_row_begin || out_row >= f->own_row_end) return 0;
    }
    if (f->own_stripe_count > 1 && (py / 16) % f->own_stripe_count != f->own_stripe_index) return 0;
    return 1;
}


/* core.py:226-228 Face.linearize_z; the same expression appears in triangular.py:352-354 */
static inline double linearize(const ctx_t *c, double d)
{
    return c->two_nf / (c->f_plus_n - d * c->f_minus_n);
}

/* transformation.py:35-43 bound_box -> half-open integer pixel box; 0 when empty */
static int bound_box(const double *xs, const double *ys, int n, int W, int H, int box[4])
{
    double lo_x = xs[0], hi_x = xs[0], lo_y = ys[0], hi_y = ys[0];
    for (int i = 1; i < n; ++i) {
        if (xs[i] < lo_x) lo_x = xs[i];
        if (xs[i] > hi_x) hi_x = xs[i];
        if (ys[i] < lo_y) lo_y = ys[i];
        if (ys[i] > hi_y) hi_y = ys[i];
    }
    if (lo_x < 0) lo_x = 0;
    if (hi_x > W) hi_x = W;
    if (lo_y < 0) lo_y = 0;
    if (hi_y > H) hi_y = H;
    if (lo_x > hi_x || lo_y > hi_y) return 0;
    box[0] = (int)ceil(lo_x); box[1] = (int)ceil(hi_x);
    box[2] = (int)ceil(lo_y); box[3] = (int)ceil(hi_y);
    return 1;
}

/* ------------------------------------------------------------------ triangle set-up */

typedef struct {
    const orc_model *m;
    const orc_material *mat;
    int vi[3], ti[3], ni[3];
    double world[3][4];     /* face.world_vertices */
    double clip[3][4];      /* world @ camera.MVP      (triangular.py:40) */
    double clipd[3][4];     /* world @ debug_camera.MVP (triangular.py:39) */
    double scr[3][4];       /* screen x, y, z and w := 1/clip.w (triangular.py:42-45) */
    double zlin[3];         /* linearised screen z (triangular.py:96) */
    double v0[2], v1[2];    /* barycentric edge vectors (transformation.py:16-17) */
    float d00, d01, d11, inv_den;
    int box[4];
} tri_t;

/* returns the reference's early-out status, 0 when the triangle has a pixel box */
static int tri_setup(const ctx_t *c, const orc_model *m, int f, tri_t *t)
{
    const orc_frame *fr = c->f;
    const int32_t *fc = m->faces + (size_t)f * 12;
    t->m = m;
    for (int k = 0; k < 3; ++k) {
        t->vi[k] = fc[k * 4 + 0]; t->ti[k] = fc[k * 4 + 1]; t->ni[k] = fc[k * 4 + 2];
        memcpy(t->world[k], m->verts + (size_t)t->vi[k] * 4, 4 * sizeof(double));
    }
    int mi = fc[3];
    t->mat = m->materials + ((mi >= 0 && mi < m->n_materials) ? mi : 0);

    for (int k = 0; k < 3; ++k) {
        for (int j = 0; j < 4; ++j) {
            t->clip[k][j] = chain4(t->world[k], fr->mvp + j, 4);
            t->clipd[k][j] = chain4(t->world[k], fr->debug_mvp + j, 4);
        }
        double depth = 1.0 / t->clip[k][3];
        double ndc[4];
        for (int j = 0; j < 4; ++j) ndc[j] = t->clip[k][j] * depth;
        for (int j = 0; j < 4; ++j) t->scr[k][j] = chain4(ndc, fr->viewport + j, 4);
        t->scr[k][3] = depth;
    }

    /* triangular.py:47-48 / core.py:132-136: sign of the normalised screen-space normal's z */
    if (fr->backface_culling) {
        double e0[3], e1[3], n[3], u[3];
        for (int j = 0; j < 3; ++j) {
            e0[j] = t->scr[1][j] - t->scr[0][j];
            e1[j] = t->scr[2][j] - t->scr[0][j];
        }
        n[0] = e0[1] * e1[2] - e0[2] * e1[1];
        n[1] = e0[2] * e1[0] - e0[0] * e1[2];
        n[2] = e0[0] * e1[1] - e0[1] * e1[0];
        normalize3(n, u);
        if (u[2] < 0) return ORC_FACE_BACK_FACE_CULLING;
    }

    double xs[3] = { t->scr[0][0], t->scr[1][0], t->scr[2][0] };
    double ys[3] = { t->scr[0][1], t->scr[1][1], t->scr[2][1] };
    if (!bound_box(xs, ys, 3, c->W, c->H, t->box)) return ORC_FACE_EMPTY_Z;   /* triangular.py:69-70 */

    /* transformation.py:16-28 */
    t->v0[0] = t->scr[1][0] - t->scr[0][0]; t->v0[1] = t->scr[1][1] - t->scr[0][1];
    t->v1[0] = t->scr[2][0] - t->scr[0][0]; t->v1[1] = t->scr[2][1] - t->scr[0][1];
    t->d00 = (float)chain2(t->v0[0], t->v0[1], t->v0[0], t->v0[1]);
    t->d01 = (float)chain2(t->v0[0], t->v0[1], t->v1[0], t->v1[1]);
    t->d11 = (float)chain2(t->v1[0], t->v1[1], t->v1[0], t->v1[1]);
    float den = t->d00 * t->d11 - t->d01 * t->d01;
    if (den == 0) return ORC_FACE_EMPTY_B;
    t->inv_den = 1.0f / den;

    for (int k = 0; k < 3; ++k) t->zlin[k] = linearize(c, t->scr[k][2]);
    return 0;
}

/* float32 barycentrics of the integer sample (px,py); transformation.py:18-31.
 * `single`: the face's pixel box holds one sample, so NumPy evaluates (1,2)@(2,) as a dot. */
static inline void tri_bary(const tri_t *t, int px, int py, int single, float b[3])
{
    double rx = (double)px - t->scr[0][0], ry = (double)py - t->scr[0][1];
    float d20 = (float)(single ? chain2(rx, ry, t->v0[0], t->v0[1]) : gemv2(rx, ry, t->v0[0], t->v0[1]));
    float d21 = (float)(single ? chain2(rx, ry, t->v1[0], t->v1[1]) : gemv2(rx, ry, t->v1[0], t->v1[1]));
    float v = (t->d11 * d20 - t->d01 * d21) * t->inv_den;
    float w = (t->d00 * d21 - t->d01 * d20) * t->inv_den;
    b[0] = 1.0f - v - w; b[1] = v; b[2] = w;
}

/* (N,3)@(3,) as NumPy evaluates it: a BLAS gemv for N > 1, a dot (ascending chain) for N == 1 */
static inline double rows_dot3(int single, double a0, double a1, double a2, double b0, double b1, double b2)
{
    return single ? chain3(a0, a1, a2, b0, b1, b2) : gemv3(a0, a1, a2, b0, b1, b2);
}

/* core.py:155-160 Face.screen_perspective; `single` = the array it is applied to has one row */
static inline void persp_bary(const tri_t *t, const float b[3], int single, double p[3])
{
    double wc = rows_dot3(single, b[0], b[1], b[2], t->scr[0][3], t->scr[1][3], t->scr[2][3]);
    for (int k = 0; k < 3; ++k) p[k] = ((double)b[k] * t->scr[k][3]) / wc;
}

/* triangular.py:83-87: strict -w < x,y,z < w in one camera's clip space */
static inline int inside_clip(const double p[3], const double cs[3][4])
{
    double q[4];
    for (int j = 0; j < 4; ++j) q[j] = chain3(p[0], p[1], p[2], cs[0][j], cs[1][j], cs[2][j]);
    return (-q[3] < q[0]) && (q[0] < q[3]) && (-q[3] < q[1]) && (q[1] < q[3]) &&
           (-q[3] < q[2]) && (q[2] < q[3]);
}

/* ------------------------------------------------------------------ shading */

/* core.py:138-143 Face.get_UV -> (row, col) with Python negative-index wrap */
static inline void tex_index(const tri_t *t, const double p[3], int single, int h, int w, int *row, int *col)
{
    const float *uv = t->m->uv;
    double tu = rows_dot3(single, p[0], p[1], p[2], uv[t->ti[0] * 3], uv[t->ti[1] * 3], uv[t->ti[2] * 3]);
    double tv = rows_dot3(single, p[0], p[1], p[2], uv[t->ti[0] * 3 + 1], uv[t->ti[1] * 3 + 1], uv[t->ti[2] * 3 + 1]);
    double cu = tu > 1.0 ? 1.0 : tu;
    double rv = 1.0 - tv;
    if (rv > 1.0) rv = 1.0;
    int ci = (int)(cu * (double)(w - 1));
    int ri = (int)(rv * (double)(h - 1));
    if (ci < 0) ci += w;
    if (ri < 0) ri += h;
    if (ci < 0) ci = 0;
    if (ri < 0) ri = 0;
    *row = ri; *col = ci;
}

static inline const float *texel(const orc_texture *tx, const tri_t *t, const double p[3], int single)
{
    int r, c;
    tex_index(t, p, single, tx->h, tx->w, &r, &c);
    return tx->rgb + ((size_t)r * tx->w + c) * 3;
}

/* 3x3 inverse by LU with partial pivoting (what np.linalg.inv's LAPACK gesv does) */
static int inv3(const double a[3][3], double inv[3][3])
{
    double lu[3][3];
    int perm[3] = { 0, 1, 2 };
    memcpy(lu, a, sizeof lu);
    for (int col = 0; col < 3; ++col) {
        int piv = col;
        double best = fabs(lu[col][col]);
        for (int r = col + 1; r < 3; ++r)
            if (fabs(lu[r][col]) > best) { best = fabs(lu[r][col]); piv = r; }
        if (best == 0) return 0;
        if (piv != col) {
            for (int j = 0; j < 3; ++j) { double tmp = lu[col][j]; lu[col][j] = lu[piv][j]; lu[piv][j] = tmp; }
            int ti = perm[col]; perm[col] = perm[piv]; perm[piv] = ti;
        }
        double r = 1.0 / lu[col][col];
        for (int i = col + 1; i < 3; ++i) {
            lu[i][col] *= r;
            for (int j = col + 1; j < 3; ++j) lu[i][j] = fma(-lu[i][col], lu[col][j], lu[i][j]);
        }
    }
    for (int j = 0; j < 3; ++j) {
        double y[3];
        for (int i = 0; i < 3; ++i) {
            double s = (perm[i] == j) ? 1.0 : 0.0;
            for (int k = 0; k < i; ++k) s = fma(-lu[i][k], y[k], s);
            y[i] = s;
        }
        for (int i = 2; i >= 0; --i) {
            double s = y[i];
            for (int k = i + 1; k < 3; ++k) s = fma(-lu[i][k], inv[k][j], s);
            inv[i][j] = s / lu[i][i];
        }
    }
    return 1;
}

/* core.py:127-130 unit normal of the world-space triangle, in the vertices' own dtype */
static void face_normal_world(const tri_t *t, double n[3])
{
    if (t->m->verts_f32) {
        float a[3], b[3], c[3], e0[3], e1[3], cr[3];
        for (int j = 0; j < 3; ++j) {
            a[j] = (float)t->world[0][j]; b[j] = (float)t->world[1][j]; c[j] = (float)t->world[2][j];
            e0[j] = b[j] - a[j]; e1[j] = c[j] - a[j];
        }
        cr[0] = e0[1] * e1[2] - e0[2] * e1[1];
        cr[1] = e0[2] * e1[0] - e0[0] * e1[2];
        cr[2] = e0[0] * e1[1] - e0[1] * e1[0];
        float l = sqrtf((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]);
        if (l == 0) l = 1;
        for (int j = 0; j < 3; ++j) n[j] = (double)(cr[j] / l);
    } else {
        double e0[3], e1[3], cr[3];
        for (int j = 0; j < 3; ++j) {
            e0[j] = t->world[1][j] - t->world[0][j];
            e1[j] = t->world[2][j] - t->world[0][j];
        }
        cr[0] = e0[1] * e1[2] - e0[2] * e1[1];
        cr[1] = e0[2] * e1[0] - e0[0] * e1[2];
        cr[2] = e0[0] * e1[1] - e0[1] * e1[0];
        normalize3(cr, n);
    }
}

/* core.py:175-224 Face.get_normals / tangent_ */
static void fragment_normal(const ctx_t *c, const orc_texture *textures, const tri_t *t,
                            const double p[3], int single, double out[3])
{
    (void)c;
    const orc_model *m = t->m;
    double raw[3];
    double interp[3] = { 0, 0, 0 };
    if (m->normals) {
        const float *n0 = m->normals + (size_t)t->ni[0] * 3;
        const float *n1 = m->normals + (size_t)t->ni[1] * 3;
        const float *n2 = m->normals + (size_t)t->ni[2] * 3;
        for (int j = 0; j < 3; ++j) interp[j] = chain3(p[0], p[1], p[2], n0[j], n1[j], n2[j]);
    }
    if (t->mat->tex_norm >= 0) {
        const float *tx = texel(&textures[t->mat->tex_norm], t, p, single);
        if (t->mat->norm_tangent) {
            double n[3];
            normalize3(interp, n);
            double A[3][3], AI[3][3];
            for (int j = 0; j < 3; ++j) {
                if (m->verts_f32) {
                    A[0][j] = (double)((float)t->world[1][j] - (float)t->world[0][j]);
                    A[1][j] = (double)((float)t->world[2][j] - (float)t->world[0][j]);
                } else {
                    A[0][j] = t->world[1][j] - t->world[0][j];
                    A[1][j] = t->world[2][j] - t->world[0][j];
                }
                A[2][j] = n[j];
            }
            if (!inv3(A, AI)) { double nan = NAN; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) AI[i][j] = nan; }
            const float *uv = m->uv;
            float u0 = uv[t->ti[0] * 3], u1 = uv[t->ti[1] * 3], u2 = uv[t->ti[2] * 3];
            float w0 = uv[t->ti[0] * 3 + 1], w1 = uv[t->ti[1] * 3 + 1], w2 = uv[t->ti[2] * 3 + 1];
            double du[3] = { (double)(u1 - u0), (double)(u2 - u0), 0.0 };
            double dv[3] = { (double)(w1 - w0), (double)(w2 - w0), 0.0 };
            double ti_[3], tj_[3], T[3], B[3];
            for (int r = 0; r < 3; ++r) {
                ti_[r] = chain3(AI[r][0], AI[r][1], AI[r][2], du[0], du[1], du[2]);
                tj_[r] = chain3(AI[r][0], AI[r][1], AI[r][2], dv[0], dv[1], dv[2]);
            }
            normalize3(ti_, T);
            normalize3(tj_, B);
            for (int r = 0; r < 3; ++r)
                raw[r] = chain3(T[r], B[r], n[r], (double)tx[0], (double)tx[1], (double)tx[2]);
        } else {
            /* object-space map: the texel IS the normal, and normalize() (transformation.py:46-49)
             * then runs on the float32 texels themselves: float32 squares, sum, sqrt and quotient */
            float l = sqrtf((tx[0] * tx[0] + tx[1] * tx[1]) + tx[2] * tx[2]);
            if (l == 0) l = 1;
            for (int j = 0; j < 3; ++j) out[j] = (double)(tx[j] / l);
            return;
        }
    } else if (m->normals) {
        raw[0] = interp[0]; raw[1] = interp[1]; raw[2] = interp[2];
    } else {
        double fn[3];
        face_normal_world(t, fn);
        for (int j = 0; j < 3; ++j) raw[j] = chain3(p[0], p[1], p[2], fn[j], fn[j], fn[j]);
    }
    normalize3(raw, out);
}

/* NumPy's scalar-exponent fast paths for ndarray ** scalar, else pow() */
static inline double np_power(double x, double e)
{
    if (e == 2.0) return x * x;
    if (e == 1.0) return x;
    if (e == 0.5) return sqrt(x);
    if (e == 0.0) return 1.0;
    if (e == -1.0) return 1.0 / x;
    return pow(x, e);
}

/* triangular.py:135-171 general_shading for one fragment; writes the float frame */
static void shade(const ctx_t *c, const orc_texture *textures, const tri_t *t,
                  const float b[3], int single, int first_pass, float *dst)
{
    const orc_frame *fr = c->f;
    double p[3];
    persp_bary(t, b, single, p);

    double color[3];
    if (t->mat->tex_kd >= 0) {
        const float *tx = texel(&textures[t->mat->tex_kd], t, p, single);
        color[0] = tx[0]; color[1] = tx[1]; color[2] = tx[2];
    } else {
        color[0] = t->mat->kd[0]; color[1] = t->mat->kd[1]; color[2] = t->mat->kd[2];
    }
    double pos[3], dl[3];
    for (int j = 0; j < 3; ++j) {
        pos[j] = chain3(p[0], p[1], p[2], t->world[0][j], t->world[1][j], t->world[2][j]);
        dl[j] = fr->light_pos[j] - pos[j];
    }
    /* core.py:517-524 Light.attenuation */
    double dist = sqrt((dl[0] * dl[0] + dl[1] * dl[1]) + dl[2] * dl[2]);
    double att = 1.0 / (fr->att_constant + dist * (fr->att_linear + fr->att_quadratic * dist));

    if (first_pass) {
        for (int j = 0; j < 3; ++j) {
            double v = (att * fr->light_ambient[j]) * color[j];
            v = v < 0.05 ? 0.05 : (v > 1.0 ? 1.0 : v);
            dst[j] = (float)v;
        }
        return;
    }

    double N[3], L[3], V[3], Hh[3], tmp[3];
    fragment_normal(c, textures, t, p, single, N);
    if (fr->light_type == ORC_LIGHT_DIRECTIONAL) {
        L[0] = fr->light_dir[0]; L[1] = fr->light_dir[1]; L[2] = fr->light_dir[2];
    } else {
        normalize3(dl, L);
    }
    for (int j = 0; j < 3; ++j) tmp[j] = fr->camera_pos[j] - pos[j];
    normalize3(tmp, V);
    if (fr->light_type == ORC_LIGHT_SPOT) {
        double x = (sum3(fr->light_dir, L) - fr->spot_edge0) / (fr->spot_edge1 - fr->spot_edge0);
        x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
        double in_light = x * x * (3 - 2 * x);
        for (int j = 0; j < 3; ++j) color[j] = color[j] * in_light;
    }
    double spec_light[3];
    if (t->mat->tex_ks >= 0) {
        const float *tx = texel(&textures[t->mat->tex_ks], t, p, single);
        float s = tx[0] * 255.0f;                       /* core.py:149, float32 product */
        spec_light[0] = spec_light[1] = spec_light[2] = (double)s;
    } else {
        spec_light[0] = t->mat->ks255[0]; spec_light[1] = t->mat->ks255[1]; spec_light[2] = t->mat->ks255[2];
    }
    for (int j = 0; j < 3; ++j) tmp[j] = L[j] + V[j];
    normalize3(tmp, Hh);
    double nh = sum3(N, Hh);
    if (nh < 0) nh = 0;
    double refl = np_power(nh, t->mat->ns);
    double nl = sum3(N, L);
    for (int j = 0; j < 3; ++j) {
        double spec = ((fr->light_color[j] * refl) * fr->specular_strength) * spec_light[j];
        double diff = nl * fr->light_color[j];
        double v = (att * color[j]) * ((fr->light_ambient[j] + diff) + spec);
        v = v < 0.05 ? 0.05 : (v > 1.0 ? 1.0 : v);
        dst[j] = (float)v;
    }
}

/* ------------------------------------------------------------------ triangular.py:29-132 rasterize */

typedef struct { int32_t px, py; float b[3]; double z; } frag_t;
typedef struct { frag_t *v; size_t cap; } frag_buf;

static frag_t *frag_reserve(frag_buf *fb, size_t n)
{
    if (n > fb->cap) {
        size_t cap = fb->cap ? fb->cap : 1024;
        while (cap < n) cap *= 2;
        frag_t *nv = (frag_t *)realloc(fb->v, cap * sizeof(frag_t));
        if (!nv) return NULL;
        fb->v = nv; fb->cap = cap;
    }
    return fb->v;
}

/* The reference works on whole-face arrays, and NumPy picks a different summation order for
 * an (N,3)@(3,) product when N == 1 (a dot) than when N > 1 (a gemv).  The array lengths
 * that matter are therefore tracked: the pixel box (coverage / clip stage), the survivors of
 * coverage + clip (depth stage) and the fragments that pass the depth/stencil test (shading). */
static int rasterize(const ctx_t *c, const orc_texture *textures, const orc_model *m, int f,
                     int gid, int second_pass, orc_outputs *o, frag_buf *fb)
{
    tri_t t;
    int64_t bbox = 0;
    int st = tri_setup(c, m, f, &t);
    if (st == 0 || st == ORC_FACE_EMPTY_B) {
        bbox = (int64_t)(t.box[1] - t.box[0]) * (int64_t)(t.box[3] - t.box[2]);
        if (bbox < 0) bbox = 0;
        if (!second_pass) o->stats.bbox_px_tri += bbox;
    }
    if (st) return st;
    const int W = c->W;
    const int rh = c->f->system == 1;
    const int single_box = bbox == 1;
    if (!frag_reserve(fb, (size_t)bbox + 1)) return ORC_FACE_WRONG_MIN_MAX;
    frag_t *fr = fb->v;

    /* coverage + per-fragment clip over the pixel box (triangular.py:72-91) */
    size_t n = 0;
    int64_t inside = 0;
    for (int px = t.box[0]; px < t.box[1]; ++px) {
        for (int py = t.box[2]; py < t.box[3]; ++py) {
            float b[3];
            tri_bary(&t, px, py, single_box, b);
            if (!(b[0] >= 0 && b[1] >= 0 && b[2] >= 0)) continue;
            ++inside;
            if (m->clip) {
                double p[3];
                persp_bary(&t, b, single_box, p);
                if (!inside_clip(p, t.clip) || !inside_clip(p, t.clipd)) continue;
            }
            fr[n].px = px; fr[n].py = py;
            fr[n].b[0] = b[0]; fr[n].b[1] = b[1]; fr[n].b[2] = b[2];
            ++n;
        }
    }
    if (second_pass) o->stats.frag_tri_pass2 += inside; else o->stats.frag_tri_pass1 += inside;
    if (!n) return ORC_FACE_CLIPPED;

    /* depth (triangular.py:96-112) */
    const int single_z = n == 1;
    size_t zpass = 0, keep = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!owned_row(c, fr[i].py)) continue;
        double z = rows_dot3(single_z, fr[i].b[0], fr[i].b[1], fr[i].b[2], t.zlin[0], t.zlin[1], t.zlin[2]);
        size_t at = (size_t)fr[i].py * W + fr[i].px;
        int pass = rh ? (o->z[at] >= z) : (o->z[at] <= z);
        if (!pass) continue;
        ++zpass;
        if (second_pass && o->stencil[at] != 0) continue;
        fr[keep] = fr[i];
        fr[keep].z = z;
        ++keep;
    }
    if (!zpass || !keep) return ORC_FACE_EMPTY_Z;
    if (second_pass) o->stats.shaded_pass2 += (int64_t)keep; else o->stats.shaded_pass1 += (int64_t)keep;

    /* z write (triangular.py:117-118) and shading (triangular.py:127) */
    const int single_shade = keep == 1;
    for (size_t i = 0; i < keep; ++i) {
        size_t at = (size_t)fr[i].py * W + fr[i].px;
        if (!second_pass && m->depth_test) o->z[at] = fr[i].z;
        if (!second_pass && o->winner) o->winner[at] = gid;      /* who wrote the pixel last in pass 1, z-writing or not */
        if (o->frame) shade(c, textures, &t, fr[i].b, single_shade, !second_pass, o->frame + at * 3);
    }
    return 0;
}

/* ------------------------------------------------------------------ silhouette (triangular.py:286-302) */

typedef struct { int32_t a, b; int32_t used; int32_t present; int32_t order; } edge_slot;
typedef struct { edge_slot *slots; size_t cap; } edge_set;

static int edge_set_init(edge_set *s, size_t n_faces)
{
    size_t want = n_faces * 6 + 16, cap = 16;
    while (cap < want) cap <<= 1;
    s->slots = (edge_slot *)calloc(cap, sizeof(edge_slot));
    s->cap = cap;
    return s->slots != NULL;
}

static void edge_toggle(edge_set *s, int32_t a, int32_t b, int32_t order)
{
    int32_t lo = a < b ? a : b, hi = a < b ? b : a;
    size_t h = ((size_t)(uint32_t)lo * 0x9E3779B1u) ^ ((size_t)(uint32_t)hi * 0x85EBCA77u);
    h &= s->cap - 1;
    for (;;) {
        edge_slot *e = &s->slots[h];
        if (!e->used) { e->used = 1; e->a = a; e->b = b; e->present = 1; e->order = order; return; }
        int32_t elo = e->a < e->b ? e->a : e->b, ehi = e->a < e->b ? e->b : e->a;
        if (elo == lo && ehi == hi) {
            if (e->present) e->present = 0;
            else { e->present = 1; e->a = a; e->b = b; e->order = order; }
            return;
        }
        h = (h + 1) & (s->cap - 1);
    }
}

/* ------------------------------------------------------------------ shadow quads */

/* plane_intersection.py:59-86 clipping (Sutherland-Hodgman against the six planes, in order) */
static int clip_polygon(const double planes[24], double poly[16][4], int n)
{
    double tmp[16][4];
    for (int pl = 0; pl < 6; ++pl) {
        const double *P = planes + pl * 4;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const double *cur = poly[i];
            const double *nxt = poly[(i + 1) % n];
            int cv = chain4(P, cur, 1) >= 0;
            int nv = chain4(P, nxt, 1) >= 0;
            if (cv) { memcpy(tmp[m], cur, 4 * sizeof(double)); ++m; }
            if (cv ^ nv) {
                /* line_plane_intersection(next, current, plane), plane_intersection.py:24-36 */
                double dir[4];
                for (int j = 0; j < 4; ++j) dir[j] = cur[j] - nxt[j];
                double den = chain4(P, dir, 1);
                if (!(fabs(den) < 1e-10)) {
                    double wgt = -chain4(P, nxt, 1) / den;
                    if (0 <= wgt && wgt <= 1) {
                        for (int j = 0; j < 4; ++j) tmp[m][j] = nxt[j] + wgt * dir[j];
                        ++m;
                    }
                }
            }
            if (m > 14) break;
        }
        n = m;
        memcpy(poly, tmp, sizeof(double) * 4 * (size_t)n);
        if (n == 0) break;
    }
    return n;
}

/* core.py:610-622 extrusion + triangular.py:319-368 resterize_quadrangle */
static void shadow_quad(const ctx_t *c, const orc_model *m, int32_t ea, int32_t eb, orc_outputs *o)
{
    const orc_frame *fr = c->f;
    const double *A = m->verts + (size_t)ea * 4, *B = m->verts + (size_t)eb * 4;
    double C[4], D[4];
    if (fr->light_type == ORC_LIGHT_POINT) {
        const double *src[2] = { A, B };
        double *dst[2] = { C, D };
        for (int s = 0; s < 2; ++s) {
            double d[4], l;
            for (int j = 0; j < 3; ++j) d[j] = src[s][j] - fr->light_pos[j];
            d[3] = src[s][3] - 1.0;
            l = sqrt(((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3]);
            if (l == 0) l = 1;
            for (int j = 0; j < 4; ++j) dst[s][j] = src[s][j] + 1000 * (d[j] / l);
        }
    } else {
        for (int j = 0; j < 3; ++j) {
            double off = fr->light_dir[j] * -1000;
            C[j] = A[j] + off; D[j] = B[j] + off;
        }
        C[3] = A[3] + 1.0; D[3] = B[3] + 1.0;
    }
    double poly[16][4];
    memcpy(poly[0], A, sizeof C); memcpy(poly[1], B, sizeof C);
    memcpy(poly[2], D, sizeof C); memcpy(poly[3], C, sizeof C);
    int n = clip_polygon(fr->planes, poly, 4);
    if (n < 3) return;

    double sx[16], sy[16], sz[16];
    for (int i = 0; i < n; ++i) {
        double q[4], nd[4];
        for (int j = 0; j < 4; ++j) q[j] = chain4(poly[i], fr->mvp + j, 4);
        for (int j = 0; j < 4; ++j) nd[j] = q[j] / q[3];
        sx[i] = chain4(nd, fr->viewport + 0, 4);
        sy[i] = chain4(nd, fr->viewport + 1, 4);
        sz[i] = chain4(nd, fr->viewport + 2, 4);
    }
    double ab[3] = { sx[0] - sx[1], sy[0] - sy[1], sz[0] - sz[1] };
    double ac[3] = { sx[0] - sx[2], sy[0] - sy[2], sz[0] - sz[2] };
    double nx = ab[1] * ac[2] - ab[2] * ac[1];
    double ny = ab[2] * ac[0] - ab[0] * ac[2];
    double nz = ab[0] * ac[1] - ab[1] * ac[0];
    int is_front = nz < 0;
    double Dp = chain3(-sx[0], -sy[0], -sz[0], nx, ny, nz);

    int box[4];
    if (!bound_box(sx, sy, n, c->W, c->H, box)) return;
    o->stats.n_quads_drawn += 1;
    int64_t bbox = (int64_t)(box[1] - box[0]) * (int64_t)(box[3] - box[2]);
    if (bbox > 0) o->stats.bbox_px_quad += bbox;
    const int rh = fr->system == 1;
    for (int px = box[0]; px < box[1]; ++px) {
        for (int py = box[2]; py < box[3]; ++py) {
            int in = 1;
            for (int i = 0; i < n && in; ++i) {
                int k = (i + 1) % n;
                double ax = (double)px - sx[i], ay = (double)py - sy[i];
                double bx = sx[k] - sx[i], by = sy[k] - sy[i];
                double cr = ax * by - ay * bx;
                in = is_front ? (cr > 0) : (cr < 0);
            }
            if (!in || !owned_row(c, py)) continue;
            o->stats.frag_quad += 1;
            double z = -((nx * (double)px + ny * (double)py) + Dp) / nz;
            z = linearize(c, z);
            size_t at = (size_t)py * c->W + px;
            int pass = rh ? (o->z[at] >= z) : (o->z[at] <= z);
            if (!pass) continue;
            o->stats.stencil_updates += 1;
            o->stencil[at] = (int16_t)(o->stencil[at] + (is_front ? 1 : -1));
        }
    }
}

/* ------------------------------------------------------------------ finalise (core.py:640) */

void orc_finalise(const float *frame, int32_t h, int32_t w, uint8_t *out)
{
    for (int r = 0; r < h; ++r) {
        const float *src = frame + (size_t)(h - 1 - r) * w * 3;
        uint8_t *dst = out + (size_t)r * w * 3;
        for (int i = 0; i < w * 3; ++i) dst[i] = (uint8_t)(powf(src[i], 0.8f) * 255.0f);
    }
}

/* ------------------------------------------------------------------ skybox (cube_map.py:63-101) */

/* fill_frame_from_skybox: two screen-covering triangles with integer vertices; the dots of
 * barycentric() are therefore exact integers, converted to float32 (transformation.py:19-23). */
static void fill_skybox(const orc_frame *fr, float *frame)
{
    const int W = fr->width, H = fr->height;
    const long long S = fr->sky_size;
    for (int t = 0; t < 2; ++t) {
        const int32_t *tv = fr->sky_tri + t * 6;
        const long long ax = tv[0], ay = tv[1];
        const long long v0x = tv[2] - ax, v0y = tv[3] - ay, v1x = tv[4] - ax, v1y = tv[5] - ay;
        const float d00 = (float)(v0x * v0x + v0y * v0y), d01 = (float)(v0x * v1x + v0y * v1y);
        const float d11 = (float)(v1x * v1x + v1y * v1y);
        const float den = d00 * d11 - d01 * d01;
        if (den == 0) continue;
        const float inv = 1.0f / den;
        const double *r = fr->sky_rays + t * 9;
        for (int px = 0; px < W; ++px) {
            for (int py = 0; py < H; ++py) {
                const long long v2x = px - ax, v2y = py - ay;
                const float d20 = (float)(v2x * v0x + v2y * v0y), d21 = (float)(v2x * v1x + v2y * v1y);
                const float v = (d11 * d20 - d01 * d21) * inv;
                const float w = (d00 * d21 - d01 * d20) * inv;
                const float u = 1.0f - v - w;
                if (!(u >= 0 && v >= 0 && w >= 0)) continue;
                double ray[3];
                for (int j = 0; j < 3; ++j) ray[j] = chain3(u, v, w, r[j], r[3 + j], r[6 + j]);
                /* CubeMap.__getitem__ (cube_map.py:63-80) */
                int major = 0;
                if (fabs(ray[1]) > fabs(ray[major])) major = 1;
                if (fabs(ray[2]) > fabs(ray[major])) major = 2;
                const double amp = ray[major];
                const double c0 = ray[major == 0 ? 1 : 0], c1 = ray[major == 2 ? 1 : 2];
                const double n0 = (c0 / amp + 1) / 2, n1 = (c1 / amp + 1) / 2;
                const int side = (amp < 0 ? 1 : 0) + 2 * major;
                long long i0 = (long long)(n0 * (double)S - 1), i1 = (long long)(n1 * (double)S - 1);
                if (i0 < 0) i0 += S;
                if (i1 < 0) i1 += S;
                if (i0 < 0) i0 = 0;
                if (i0 >= S) i0 = S - 1;
                if (i1 < 0) i1 = 0;
                if (i1 >= S) i1 = S - 1;
                const uint8_t *tx = fr->sky_texels + (((size_t)side * S + i0) * S + i1) * 3;
                float *dst = frame + ((size_t)py * W + px) * 3;
                for (int j = 0; j < 3; ++j) dst[j] = (float)((double)tx[j] / 255.0);
            }
        }
    }
}

/* ------------------------------------------------------------------ Scene.render (core.py:587-640) */

int orc_render(const orc_frame *frame, const orc_model *models, int32_t n_models,
               const orc_texture *textures, int32_t n_textures, orc_outputs *o)
{
    (void)n_textures;
    if (!frame || !o || !o->z || !o->stencil) return -1;
    ctx_t c;
    c.f = frame; c.W = frame->width; c.H = frame->height;
    c.two_nf = 2 * frame->z_near * frame->z_far;
    c.f_plus_n = frame->z_far + frame->z_near;
    c.f_minus_n = frame->z_far - frame->z_near;
    const size_t npx = (size_t)c.W * c.H;
    memset(&o->stats, 0, sizeof o->stats);

    for (size_t i = 0; i < npx; ++i) {
        o->z[i] = frame->system == 1 ? INFINITY : -INFINITY;
        o->stencil[i] = 0;
        if (o->winner) o->winner[i] = -1;
        if (o->frame) {
            o->frame[i * 3 + 0] = frame->background[0];
            o->frame[i * 3 + 1] = frame->background[1];
            o->frame[i * 3 + 2] = frame->background[2];
        }
    }

    frag_buf fb = { NULL, 0 };
    if (o->frame && frame->sky_texels && frame->sky_size > 0) fill_skybox(frame, o->frame);

    edge_set *sets = (edge_set *)calloc((size_t)n_models, sizeof(edge_set));
    if (!sets) return -2;
    const int shadows = (frame->flags & ORC_FLAG_SHADOWS) != 0;

    /* pass 1: silhouette toggling + ambient/depth (core.py:603-606) */
    int gid = 0;
    for (int mi = 0; mi < n_models; ++mi) {
        const orc_model *m = &models[mi];
        if (!edge_set_init(&sets[mi], (size_t)m->n_faces)) return -2;
        for (int f = 0; f < m->n_faces; ++f, ++gid) {
            if (shadows) {
                tri_t t;
                const int32_t *fc = m->faces + (size_t)f * 12;
                t.m = m;
                for (int k = 0; k < 3; ++k) {
                    t.vi[k] = fc[k * 4];
                    memcpy(t.world[k], m->verts + (size_t)t.vi[k] * 4, 4 * sizeof(double));
                }
                double n[3];
                face_normal_world(&t, n);
                if (chain3(n[0], n[1], n[2], frame->light_pos[0], frame->light_pos[1], frame->light_pos[2]) > 0)
                    for (int k = 0; k < 3; ++k) {
                        const int k2 = (k + 1) % 3;
                        const int32_t ra = m->edge_ids ? m->edge_ids[(size_t)f * 3 + k] : t.vi[k];
                        const int32_t rb = m->edge_ids ? m->edge_ids[(size_t)f * 3 + k2] : t.vi[k2];
                        edge_toggle(&sets[mi], ra, rb, f * 3 + k);
                    }
            }
            rasterize(&c, textures, m, f, gid, 0, o, &fb);
        }
    }

    /* stencil pass (core.py:610-622) */
    int n_sil = 0;
    for (int mi = 0; mi < n_models; ++mi) {
        const orc_model *m = &models[mi];
        for (size_t s = 0; s < sets[mi].cap; ++s) {
            const edge_slot *e = &sets[mi].slots[s];
            if (!e->used || !e->present) continue;
            if (o->silhouette && n_sil < o->silhouette_cap) {
                o->silhouette[n_sil * 3 + 0] = mi;
                o->silhouette[n_sil * 3 + 1] = e->a;
                o->silhouette[n_sil * 3 + 2] = e->b;
            }
            ++n_sil;
            /* model.vertices[e] (core.py:611): NumPy wraps negative indices */
            shadow_quad(&c, m, e->a < 0 ? e->a + m->n_verts : e->a, e->b < 0 ? e->b + m->n_verts : e->b, o);
        }
        free(sets[mi].slots);
    }
    free(sets);
    o->stats.n_quads = n_sil;

    /* pass 2: lit (core.py:624-636) */
    gid = 0;
    for (int mi = 0; mi < n_models; ++mi) {
        const orc_model *m = &models[mi];
        for (int f = 0; f < m->n_faces; ++f, ++gid) {
            int st = rasterize(&c, textures, m, f, gid, 1, o, &fb);
            if (o->face_status) o->face_status[gid] = (uint8_t)st;
        }
    }

    free(fb.v);
    if (o->out && o->frame) orc_finalise(o->frame, c.H, c.W, o->out);
    return 0;
}
