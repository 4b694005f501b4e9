// kernels_shade.h -- deferred shading and finalise of one pixel (device functions of the tile kernel).
//
// The reference shades every z-passing fragment twice (ambient pass, then lit pass where
// stencil == 0; obj/triangular.py:135-171) and converts the float frame at the end
// (obj/core.py:640).  Because pass 2 re-tests against the final z buffer, the pixel's colour is
// a function of (winner face, stencil == 0) only, so it is computed once here:
//   stencil == 0 -> lit colour of the winner, else its ambient colour, no winner -> background;
// then flipped, raised to 0.8, scaled to uint8.
#pragma once

#include "rast_math.h"

namespace mr {

struct ShadeArgs {
    const TriRec *tris;
    const void *face_pos;      // static per face: FacePos32[] / FacePos64[] (FrameConst::pos32)
    const FaceAttr *face_attr; // static per face: uv, vertex normals
    const Material *materials;
    const uint8_t *sky;        // cubemap texels (6, S, S, 3) or null
    const float *gamma_lut;    // GAMMA_LUT_SIZE thresholds of the finalise step function
    float *frame;              // optional float frame (row = screen y), may be null
    uint8_t *out;              // this device's rows of the final frame (see out_row)
};

// Face.get_UV (obj/core.py:138-143): nearest texel, truncation, Python negative-index wrap.
// (tu, tv) = perspective barycentrics @ uv, computed once per pixel for all of the face's maps.
__device__ __forceinline__ const float *texel(const Texture &tx, double tu, double tv)
{
    const double cu = tu > 1.0 ? 1.0 : tu;
    double rv = 1.0 - tv;
    rv = rv > 1.0 ? 1.0 : rv;
    int col = (int)(cu * (double)(tx.w - 1));
    int row = (int)(rv * (double)(tx.h - 1));
    if (col < 0) col += tx.w;
    if (row < 0) row += tx.h;
    col = min(max(col, 0), tx.w - 1);
    row = min(max(row, 0), tx.h - 1);
    return tx.rgb + ((size_t)row * tx.w + col) * 3;
}

// ---- colour-path arithmetic.  Coverage, z and texel indices above are bit-exact; the colour
// that follows only has to land within +-1 uint8 of the reference (it already differs from it
// in pow()), so its divisions and square roots use the hardware seeds (v_rcp_f64 / v_rsq_f64,
// good to ~2^-24) refined by ONE Newton step to ~1e-14 -- seven decimal orders below the
// float32 the colour is stored in -- instead of the ~35-instruction IEEE expansions: a lit
// pixel does some 30 divisions and 7 square roots.
__device__ __forceinline__ double c_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double c_rsqrt(double x)
{
    const double r = __builtin_amdgcn_rsq(x);
    return fma(0.5 * r, fma(-x * r, r, 1.0), r);
}
// normalize() of obj/transformation.py:46-49 (zero vectors stay zero)
__device__ __forceinline__ void c_normalize3(const double a[3], double o[3])
{
    const double l2 = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
    const double r = l2 > 0 ? c_rsqrt(l2) : 1.0;
    o[0] = a[0] * r; o[1] = a[1] * r; o[2] = a[2] * r;
}

// ndarray ** scalar: NumPy's scalar-exponent fast paths; small whole exponents (the default
// Ns = 64) by repeated squaring; else pow()
__device__ __forceinline__ double np_power(double x, double e)
{
    if (e == 2.0) return x * x;
    if (e == 1.0) return x;
    if (e == 0.5) return x > 0 ? x * c_rsqrt(x) : 0.0;
    if (e == 0.0) return 1.0;
    if (e == -1.0) return c_rcp(x);
    if (e > 0 && e <= 1024.0 && e == floor(e)) {
        double r = 1.0, b = x;
        for (int k = (int)e; k; k >>= 1) { if (k & 1) r *= b; b *= b; }
        return r;
    }
    return pow(x, e);
}

// Skybox colour of one background pixel (obj/cube_map.py:63-101).  Two screen-covering
// triangles with INTEGER vertices: the barycentric dots are exact integers rounded to float32,
// the later triangle overwrites the earlier on their shared diagonal, and the pixels neither
// covers (a few along the edges, because the vertices were truncated) stay black.
__device__ __forceinline__ void sky_color(const FrameConst &fc, const uint8_t *sky, int px, int py, float rgb[3])
{
    rgb[0] = rgb[1] = rgb[2] = 0.0f;
    const long long S = fc.sky_size;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int32_t *tv = fc.sky_tri + t * 6;
        const long long ax = tv[0], ay = tv[1];
        const long long v0x = tv[2] - ax, v0y = tv[3] - ay, v1x = tv[4] - ax, v1y = tv[5] - ay;
        const long long v2x = px - ax, v2y = py - ay;
        const float d00 = (float)(v0x * v0x + v0y * v0y), d01 = (float)(v0x * v1x + v0y * v1y);
        const float d11 = (float)(v1x * v1x + v1y * v1y);
        const float d20 = (float)(v2x * v0x + v2y * v0y), d21 = (float)(v2x * v1x + v2y * v1y);
        const float den = d00 * d11 - d01 * d01;
        if (den == 0) continue;
        const float inv = 1.0f / den;
        const float v = (d11 * d20 - d01 * d21) * inv;
        const float w = (d00 * d21 - d01 * d20) * inv;
        const float u = 1.0f - v - w;
        if (!(u >= 0 && v >= 0 && w >= 0)) continue;
        const double *r = fc.sky_rays + t * 9;
        double ray[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) ray[j] = chain3((double)u, (double)v, (double)w, r[j], r[3 + j], r[6 + j]);
        // CubeMap.__getitem__: major axis (first maximum of |.|), the other two components / it
        int major = 0;
        double amp = ray[0];
        if (fabs(ray[1]) > fabs(amp)) { major = 1; amp = ray[1]; }
        if (fabs(ray[2]) > fabs(amp)) { major = 2; amp = ray[2]; }
        const double c0 = major == 0 ? ray[1] : ray[0], c1 = major == 2 ? ray[1] : ray[2];
        const double n0 = (c0 / amp + 1) / 2, n1 = (c1 / amp + 1) / 2;
        const int side = (amp < 0 ? 1 : 0) + 2 * major;
        long long i0 = (long long)(n0 * (double)S - 1), i1 = (long long)(n1 * (double)S - 1);
        if (i0 < 0) i0 += S;
        if (i1 < 0) i1 += S;
        i0 = i0 < 0 ? 0 : (i0 >= S ? S - 1 : i0);
        i1 = i1 < 0 ? 0 : (i1 >= S ? S - 1 : i1);
        const uint8_t *tx = sky + (((size_t)side * S + i0) * S + i1) * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) rgb[j] = (float)((double)tx[j] / 255.0);
    }
}

constexpr int GAMMA_LUT_SIZE = 257;

// uint8(x ** 0.8 * 255) of obj/core.py:640.  For 0 <= x <= 1 the step is located with the
// hardware log2 / exp2 (a few ulp, i.e. never off by more than one step) and then settled
// against the two neighbouring thresholds, which were derived from powf itself: same result
// as the ~60-instruction powf, in about ten.  Anything else (NaN, negative, > 1: only a
// caller-supplied background can be) takes powf.
__device__ __forceinline__ uint8_t gamma_u8(float x, const float *lut)
{
    if (!(x >= 0.0f && x <= 1.0f)) return (uint8_t)(powf(x, 0.8f) * 255.0f);
    const float approx = __builtin_amdgcn_exp2f(0.8f * __builtin_amdgcn_logf(x));
    int k = min(max((int)(approx * 255.0f), 0), 255);
    k += x >= lut[k + 1] ? 1 : 0;
    k -= x < lut[k] ? 1 : 0;
    return (uint8_t)k;
}

// What shading reads of the frame constants, as values: fetched in one go where the shading phase begins (a few
// wide scalar loads) instead of one scalar load and one wait at every first use along the way.
struct LightConst {
    double camera_pos[3], light_pos[3], light_dir[3], light_color[3], light_ambient[3];
    double specular_strength, att_constant, att_linear, att_quadratic, spot_edge0, spot_edge1;
    int32_t light_type;
};
__device__ __forceinline__ LightConst light_const(const FrameConst &fc)
{
    LightConst lc;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        lc.camera_pos[j] = fc.camera_pos[j]; lc.light_pos[j] = fc.light_pos[j]; lc.light_dir[j] = fc.light_dir[j];
        lc.light_color[j] = fc.light_color[j]; lc.light_ambient[j] = fc.light_ambient[j];
    }
    lc.specular_strength = fc.specular_strength;
    lc.att_constant = fc.att_constant; lc.att_linear = fc.att_linear; lc.att_quadratic = fc.att_quadratic;
    lc.spot_edge0 = fc.spot_edge0; lc.spot_edge1 = fc.spot_edge1;
    lc.light_type = fc.light_type;
    return lc;
}

__device__ __forceinline__ double clip01(double v) { return v < 0.05 ? 0.05 : (v > 1.0 ? 1.0 : v); }

// Colour of one covered pixel: the reference's two shading passes collapsed: the
// pixel shows its winner face, lit (obj/triangular.py:149-171) where the stencil count is zero,
// ambient only (obj/triangular.py:135-147) where it is not.  `mat` may live in LDS.
// what shading reads of a face: its set-up record of this frame (TriRec) and its static records
struct ShadedFace {
    double world[3][3];        // world xyz per corner
    const FaceAttr *attr;
};
__device__ __forceinline__ void load_shaded_face(const ShadeArgs &sh, bool pos32, int face, ShadedFace &sf)
{
    if (pos32) {
        const FacePos32 &p = static_cast<const FacePos32 *>(sh.face_pos)[face];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j) sf.world[k][j] = (double)p.v[k][j];
    } else {
        const FacePos64 &p = static_cast<const FacePos64 *>(sh.face_pos)[face];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j) sf.world[k][j] = p.v[k][j];
    }
    sf.attr = sh.face_attr + face;
}

__device__ __forceinline__ void shade_pixel(const LightConst &fc, const TriRec &t, const ShadedFace &sf, const Material &mat,
                                            int px, int py, bool lit, float rgb[3])
{
    const FaceAttr &at = *sf.attr;
    const uint8_t ff = (uint8_t)(t.flags >> 8);
    float u, v, w;
    tri_bary(t, (double)px, (double)py, (t.flags & TF_SINGLE_BOX) != 0, u, v, w);
    double p[3];
    persp_bary(t.dp, u, v, w, false, p);
    const double tu = gemv3(p[0], p[1], p[2], (double)at.uv[0][0], (double)at.uv[1][0], (double)at.uv[2][0]);
    const double tv = gemv3(p[0], p[1], p[2], (double)at.uv[0][1], (double)at.uv[1][1], (double)at.uv[2][1]);

    // ---- Face.get_normals / tangent_ (obj/core.py:175-224), FIRST: it is the part with the most values in flight
    // (two edge vectors, two cross products, the uv deltas: ~50 registers of float64), and with the colour, the
    // position and the light vector already waiting in registers beside it the pixel did not fit the 80 registers
    // of six wavefronts per SIMD.  Only lit pixels need the normal.
    double raw[3] = { 0, 0, 0 }, interp[3] = { 0, 0, 0 };
    bool raw_is_unit = false;
    const double *wa = sf.world[0], *wb = sf.world[1], *wc = sf.world[2];
    if (lit) {
    const bool has_n = (ff & FF_HAS_NORMALS) != 0;
    if (has_n) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            interp[j] = chain3(p[0], p[1], p[2], (double)at.n[0][j], (double)at.n[1][j], (double)at.n[2][j]);
    }
    if (mat.map_norm.rgb) {
        const float *tx = texel(mat.map_norm, tu, tv);
        if (mat.norm_tangent) {
            // tangent_ (obj/core.py:191-224): T = normalize(AI @ (du, 0)), B = normalize(AI @ (dv, 0)) with
            // AI = inv([b - a; c - a; n]).  The third component of (du, 0) and (dv, 0) is zero, so only
            // the first two columns of the inverse are used: (e2 x n) / det and (n x e1) / det, and the
            // normalisation cancels |det|: no LU, no division, just two cross products and the sign of
            // the determinant.  Agrees with the LAPACK inverse to cond(A) * 1e-16, seven orders inside
            // the float32 the colour is stored in.  A singular A gives NaN (upstream's inv raises).
            double n[3], e1[3], e2[3];
            c_normalize3(interp, n);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (ff & FF_VERTS_F32) {
                    e1[j] = (double)((float)wb[j] - (float)wa[j]);
                    e2[j] = (double)((float)wc[j] - (float)wa[j]);
                } else {
                    e1[j] = wb[j] - wa[j];
                    e2[j] = wc[j] - wa[j];
                }
            }
            const double c0[3] = { e2[1] * n[2] - e2[2] * n[1], e2[2] * n[0] - e2[0] * n[2], e2[0] * n[1] - e2[1] * n[0] };
            const double c1[3] = { n[1] * e1[2] - n[2] * e1[1], n[2] * e1[0] - n[0] * e1[2], n[0] * e1[1] - n[1] * e1[0] };
            const double det = (e1[0] * c0[0] + e1[1] * c0[1]) + e1[2] * c0[2];
            const double sgn = det > 0 ? 1.0 : (det < 0 ? -1.0 : NAN);
            const double du[2] = { (double)(at.uv[1][0] - at.uv[0][0]) * sgn, (double)(at.uv[2][0] - at.uv[0][0]) * sgn };
            const double dv[2] = { (double)(at.uv[1][1] - at.uv[0][1]) * sgn, (double)(at.uv[2][1] - at.uv[0][1]) * sgn };
            double ti_[3], tj_[3], T[3], Bt[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                ti_[r] = fma(c1[r], du[1], c0[r] * du[0]);
                tj_[r] = fma(c1[r], dv[1], c0[r] * dv[0]);
            }
            c_normalize3(ti_, T);
            c_normalize3(tj_, Bt);
#pragma unroll
            for (int r = 0; r < 3; ++r)
                raw[r] = chain3(T[r], Bt[r], n[r], (double)tx[0], (double)tx[1], (double)tx[2]);
        } else {
            // object-space map: the texel is the normal and normalize() runs on the float32
            // texels themselves (float32 squares, sum, sqrt, quotient; obj/transformation.py:46-49)
            float l = sqrtf((tx[0] * tx[0] + tx[1] * tx[1]) + tx[2] * tx[2]);
            if (l == 0) l = 1;
            raw[0] = (double)(tx[0] / l); raw[1] = (double)(tx[1] / l); raw[2] = (double)(tx[2] / l);
            raw_is_unit = true;
        }
    } else if (has_n) {
        raw[0] = interp[0]; raw[1] = interp[1]; raw[2] = interp[2];
    } else {
        // face normal (obj/core.py:127-130, 187), in the vertices' dtype
        double fn[3];
        if (ff & FF_VERTS_F32) {
            float e0[3], e1[3];
            for (int j = 0; j < 3; ++j) { e0[j] = (float)wb[j] - (float)wa[j]; e1[j] = (float)wc[j] - (float)wa[j]; }
            float cr[3] = { e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2],
                            e0[0] * e1[1] - e0[1] * e1[0] };
            float l = sqrtf((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]);
            if (l == 0) l = 1;
            for (int j = 0; j < 3; ++j) fn[j] = (double)(cr[j] / l);
        } else {
            double e0[3], e1[3];
            for (int j = 0; j < 3; ++j) { e0[j] = wb[j] - wa[j]; e1[j] = wc[j] - wa[j]; }
            double cr[3] = { e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2],
                             e0[0] * e1[1] - e0[1] * e1[0] };
            c_normalize3(cr, fn);
        }
        for (int j = 0; j < 3; ++j) raw[j] = chain3(p[0], p[1], p[2], fn[j], fn[j], fn[j]);
    }
    }
    double color[3];
    if (mat.map_kd.rgb) {
        const float *tx = texel(mat.map_kd, tu, tv);
        color[0] = tx[0]; color[1] = tx[1]; color[2] = tx[2];
    } else {
        color[0] = mat.kd[0]; color[1] = mat.kd[1]; color[2] = mat.kd[2];
    }
    double pos[3], dl[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        pos[j] = chain3(p[0], p[1], p[2], wa[j], wb[j], wc[j]);
        dl[j] = fc.light_pos[j] - pos[j];
    }
    // Light.attenuation (obj/core.py:517-524)
    const double dl2 = (dl[0] * dl[0] + dl[1] * dl[1]) + dl[2] * dl[2];
    const double dist = dl2 > 0 ? dl2 * c_rsqrt(dl2) : 0.0;
    const double att = c_rcp(fc.att_constant + dist * (fc.att_linear + fc.att_quadratic * dist));

    if (!lit) {
#pragma unroll
        for (int j = 0; j < 3; ++j) rgb[j] = (float)clip01((att * fc.light_ambient[j]) * color[j]);
        return;
    }
    double N[3], L[3], V[3], Hh[3], tmp[3];
    if (raw_is_unit) { N[0] = raw[0]; N[1] = raw[1]; N[2] = raw[2]; }
    else c_normalize3(raw, N);

    // ---- Blinn-Phong (obj/triangular.py:151-171)
    if (fc.light_type == MR_LIGHT_DIRECTIONAL) {
        L[0] = fc.light_dir[0]; L[1] = fc.light_dir[1]; L[2] = fc.light_dir[2];
    } else {
        c_normalize3(dl, L);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) tmp[j] = fc.camera_pos[j] - pos[j];
    c_normalize3(tmp, V);
    if (fc.light_type == MR_LIGHT_SPOT) {
        double x = (sum3(fc.light_dir, L) - fc.spot_edge0) * c_rcp(fc.spot_edge1 - fc.spot_edge0);
        x = x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x);
        const double in_light = x * x * (3 - 2 * x);
#pragma unroll
        for (int j = 0; j < 3; ++j) color[j] = color[j] * in_light;
    }
    double spec_light[3];
    if (mat.map_ks.rgb) {
        const float *tx = texel(mat.map_ks, tu, tv);
        const float s = tx[0] * 255.0f;               // float32 product (obj/core.py:149)
        spec_light[0] = spec_light[1] = spec_light[2] = (double)s;
    } else {
        spec_light[0] = mat.ks255[0]; spec_light[1] = mat.ks255[1]; spec_light[2] = mat.ks255[2];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) tmp[j] = L[j] + V[j];
    c_normalize3(tmp, Hh);
    double nh = sum3(N, Hh);
    nh = nh < 0 ? 0 : nh;
    const double refl = np_power(nh, mat.ns);
    const double nl = sum3(N, L);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double spec = ((fc.light_color[j] * refl) * fc.specular_strength) * spec_light[j];
        const double diff = nl * fc.light_color[j];
        rgb[j] = (float)clip01((att * color[j]) * ((fc.light_ambient[j] + diff) + spec));
    }
}

}  // namespace mr
