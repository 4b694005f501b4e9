// rast_math.h -- device arithmetic in the exact operation order of the reference.
//
// The reference evaluates its products through NumPy/OpenBLAS, which fixes a rounding order
// per call shape.  Coverage, clip decisions and z must be bit-exact,
// so every such product is written out here as an explicit fma chain; element-wise NumPy
// expressions stay separate rounded operations.  The translation unit is compiled with
// -ffp-contract=off: nothing is fused unless it is spelled fma() below.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/mi355rast.h"
#include "rast_types.h"

namespace mr {

// The kernel's arguments as ONE PHASE of the kernel sees them: a reference into the kernarg segment through a
// pointer the compiler cannot see through, so the scalar loads of what a phase uses are issued in that phase and
// their registers are dead after it.  Read as plain by-value arguments, the ~1 KB of frame constants and 25
// pointers were all fetched at the top of the kernel and kept for its whole length: 106 SGPRs, and beyond those
// the compiler parked them in lanes of two VGPRs -- ~110 v_writelane at the head of every wavefront and up to
// 390 v_readlane along it, a fifth of the vector instructions k_tile issued (rocprofv3 SQ_INSTS_VALU).
template <class T>
__device__ __forceinline__ const T &kernargs()
{
    typedef const __attribute__((address_space(4))) char *kernarg_ptr;
    kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const T *)(const char *)p;
}

// 1-D dot and GEMM (M,K)@(K,P>=2): ascending k, first term a rounded product.
__device__ __forceinline__ double chain2(double a0, double a1, double b0, double b1)
{
    return fma(a1, b1, a0 * b0);
}
__device__ __forceinline__ double chain3(double a0, double a1, double a2, double b0, double b1, double b2)
{
    return fma(a2, b2, fma(a1, b1, a0 * b0));
}
__device__ __forceinline__ double chain4(double a0, double a1, double a2, double a3,
                                         double b0, double b1, double b2, double b3)
{
    return fma(a3, b3, fma(a2, b2, fma(a1, b1, a0 * b0)));
}
// row @ column j of a row-major 4x4
__device__ __forceinline__ double row_times_col(const double v[4], const double *m, int j)
{
    return chain4(v[0], v[1], v[2], v[3], m[j], m[4 + j], m[8 + j], m[12 + j]);
}
// GEMV (N,2)@(2,), N > 1
__device__ __forceinline__ double gemv2(double a0, double a1, double b0, double b1)
{
    return fma(a0, b0, a1 * b1);
}
// GEMV (N,3)@(3,), N > 1
__device__ __forceinline__ double gemv3(double a0, double a1, double a2, double b0, double b1, double b2)
{
    return fma(a2, b2, fma(a0, b0, a1 * b1));
}
// (N,K)@(K,) as NumPy dispatches it: one row -> dot (ascending chain), else gemv
__device__ __forceinline__ double rows_dot2(bool single, double a0, double a1, double b0, double b1)
{
    return single ? chain2(a0, a1, b0, b1) : gemv2(a0, a1, b0, b1);
}
__device__ __forceinline__ double rows_dot3(bool single, double a0, double a1, double a2,
                                            double b0, double b1, double b2)
{
    return single ? chain3(a0, a1, a2, b0, b1, b2) : gemv3(a0, a1, a2, b0, b1, b2);
}
// (a*b).sum(axis=1), K = 3: no fusion, left to right
__device__ __forceinline__ double sum3(const double a[3], const double b[3])
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
// normalize() of obj/transformation.py:46-49 for one float64 3-vector
__device__ __forceinline__ void normalize3(const double a[3], double o[3])
{
    double l = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    if (l == 0) l = 1;
    o[0] = a[0] / l; o[1] = a[1] / l; o[2] = a[2] / l;
}

// 1/x to ~1e-16 relative (v_rcp_f64 seed, ~2^-23, plus two Newton steps); NOT correctly rounded:
// only for colour and for pre-tests whose close calls are re-done exactly.
__device__ __forceinline__ double approx_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

__device__ __forceinline__ double linearize_z(const FrameConst &fc, double d)
{
    return fc.two_nf / (fc.f_plus_n - d * fc.f_minus_n);
}

// float32 barycentrics of an integer sample (obj/transformation.py:18-31).
// dpx, dpy are the sample coordinates already widened to double.
__device__ __forceinline__ void tri_bary(const TriRec &t, double dpx, double dpy, bool single,
                                         float &u, float &v, float &w)
{
    double rx = dpx - t.ax, ry = dpy - t.ay;
    float d20 = (float)rows_dot2(single, rx, ry, t.v0x, t.v0y);
    float d21 = (float)rows_dot2(single, rx, ry, t.v1x, t.v1y);
    v = (t.d11 * d20 - t.d01 * d21) * t.inv_den;
    w = (t.d00 * d21 - t.d01 * d20) * t.inv_den;
    u = 1.0f - v - w;
}

// a0/b, a1/b, a2/b with ONE reciprocal, bit-identical to IEEE-754 division.  The compiler's
// division is v_div_scale x2, v_rcp_f64, two Newton steps, a residual correction (v_div_fmas)
// and v_div_fixup, ~30 instructions each; the scale and fix-up steps only act on operands near
// the ends of the exponent range or on specials, and the reciprocal and its refinement depend
// on the denominator alone.  With every exponent well inside the range this shares them and
// keeps only the quotient + residual step per numerator (14 instructions instead of 90 for the
// three); anything else (zero denominators, denormals, infinities, NaN, huge ratios) takes the
// compiler's division.  Returns whether the shared path was taken.  Checked against "/" on
// MI355X over 1e9 operand sets, 0 mismatches (tools/micro/div_check.hip).
__device__ __forceinline__ bool div3(double a0, double a1, double a2, double b, double q[3])
{
    auto expo = [](double x) { return (unsigned int)(__double2hiint(x) >> 20) & 0x7ffu; };
    auto mid = [&](double x) { const unsigned int e = expo(x); return e > 0x200u && e < 0x5ffu; };
    const bool fast = mid(b) && (mid(a0) || a0 == 0) && (mid(a1) || a1 == 0) && (mid(a2) || a2 == 0);
    if (fast) {
        double r = __builtin_amdgcn_rcp(b);
        r = fma(fma(-b, r, 1.0), r, r);
        r = fma(fma(-b, r, 1.0), r, r);
        const double t0 = a0 * r, t1 = a1 * r, t2 = a2 * r;
        q[0] = fma(fma(-b, t0, a0), r, t0);
        q[1] = fma(fma(-b, t1, a1), r, t1);
        q[2] = fma(fma(-b, t2, a2), r, t2);
    } else {
        q[0] = a0 / b; q[1] = a1 / b; q[2] = a2 / b;
    }
    return fast;
}

// Face.screen_perspective (obj/core.py:155-160)
__device__ __forceinline__ void persp_bary(const double dp[3], float u, float v, float w, bool single,
                                           double p[3])
{
    double wc = rows_dot3(single, (double)u, (double)v, (double)w, dp[0], dp[1], dp[2]);
    div3((double)u * dp[0], (double)v * dp[1], (double)w * dp[2], wc, p);
}

// strict -w < x,y,z < w in one camera's clip space (obj/triangular.py:83-87)
__device__ __forceinline__ bool inside_clip(const double p[3], const double cs[3][4])
{
    double q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = chain3(p[0], p[1], p[2], cs[0][j], cs[1][j], cs[2][j]);
    return (-q[3] < q[0]) && (q[0] < q[3]) && (-q[3] < q[1]) && (q[1] < q[3]) &&
           (-q[3] < q[2]) && (q[2] < q[3]);
}

// transformation.py:35-43 bound_box for n points; false when empty.
__device__ __forceinline__ bool bound_box(const double *xs, const double *ys, int n, int W, int H,
                                          int &x0, int &x1, int &y0, int &y1)
{
    double lo_x = xs[0], hi_x = xs[0], lo_y = ys[0], hi_y = ys[0];
    for (int i = 1; i < n; ++i) {
        lo_x = xs[i] < lo_x ? xs[i] : lo_x;  hi_x = xs[i] > hi_x ? xs[i] : hi_x;
        lo_y = ys[i] < lo_y ? ys[i] : lo_y;  hi_y = ys[i] > hi_y ? ys[i] : hi_y;
    }
    lo_x = lo_x < 0 ? 0 : lo_x;  hi_x = hi_x > W ? (double)W : hi_x;
    lo_y = lo_y < 0 ? 0 : lo_y;  hi_y = hi_y > H ? (double)H : hi_y;
    if (lo_x > hi_x || lo_y > hi_y) return false;
    // the clamps keep the integer box inside the frame even for non-finite input
    x0 = min(max((int)ceil(lo_x), 0), W);  x1 = min(max((int)ceil(hi_x), 0), W);
    y0 = min(max((int)ceil(lo_y), 0), H);  y1 = min(max((int)ceil(hi_y), 0), H);
    return true;
}

}  // namespace mr
