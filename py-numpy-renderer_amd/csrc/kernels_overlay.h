// kernels_overlay.h -- the debug-camera frustum overlay (obj/core.py:638, obj/frustums.py:46-103) on
// the device's own z-buffer and float frame.
//
// The reference draws the frustum's edges segment by segment; every segment is a handful of NumPy
// fancy assignments over the segment's points (test against z, write z and red, then z writes and
// half blends into the four neighbours), so a segment sees what the earlier ones left in the
// z-buffer.  The host walks the lines (float64 DDA, a few thousand points) and hands over flat
// statement lists (mr_overlay_desc); ONE workgroup replays them here, statement by statement with a
// barrier in between: within a statement all right-hand sides are read before anything is written,
// and where several points of a statement hit the same pixel the last one wins (`next` links).
// It runs once per frame after the tile kernel, over a few thousand points; what it must be is
// exact (the z-buffer it leaves is compared bit for bit with the reference's), not fast.
#pragma once

#include "kernels_shade.h"

namespace mr {

constexpr int OVERLAY_TARGETS = 5;       // centre, row-1, col-1, row+1, col+1
constexpr int OVERLAY_BLOCK = 1024;

struct OverlayArgs {
    const int32_t *seg_first, *seg_count;
    const int32_t *target;               // [OVERLAY_TARGETS][n_points] linear pixel index row * W + col
    const int32_t *next;                 // [OVERLAY_TARGETS][n_points] next point of the segment with the same target, or -1
    const double *z;
    const int32_t *touched;              // every pixel any statement writes, once
    int32_t n_segments, n_points, n_touched;
    uint8_t *keep;                       // scratch [n_points]: the point passed the z test
    float *blend;                        // scratch [3 * n_points]: a blend statement's right-hand sides
    double *zbuf;
    float *frame;
    uint8_t *out;
    const float *gamma_lut;
};

__global__ void __launch_bounds__(OVERLAY_BLOCK)
k_overlay(const FrameConst fc, const OverlayArgs a)
{
    __shared__ float s_gamma[GAMMA_LUT_SIZE];
    const int tid = threadIdx.x;
    if (tid < GAMMA_LUT_SIZE) s_gamma[tid] = a.gamma_lut[tid];
    const double sign = (double)fc.system;
    const int np = a.n_points;
    // does a later kept point of the segment write the same target in this statement?
    auto shadowed = [&](int k, int p) {
        for (int j = a.next[k * np + p]; j >= 0; j = a.next[k * np + j])
            if (a.keep[j]) return true;
        return false;
    };
    for (int s = 0; s < a.n_segments; ++s) {
        const int first = a.seg_first[s], count = a.seg_count[s];
        // keep = (z_buffer[row, col] - z) * sign >= 0
        for (int i = tid; i < count; i += OVERLAY_BLOCK) {
            const int p = first + i;
            a.keep[p] = ((a.zbuf[a.target[p]] - a.z[p]) * sign >= 0) ? 1 : 0;
        }
        __syncthreads();
        // z_buffer[row, col] = z;  frame[row, col] = red
        for (int i = tid; i < count; i += OVERLAY_BLOCK) {
            const int p = first + i;
            if (a.keep[p] && !shadowed(0, p)) {
                const int t = a.target[p];
                a.zbuf[t] = a.z[p];
                a.frame[(size_t)t * 3 + 0] = 1.0f; a.frame[(size_t)t * 3 + 1] = 0.0f; a.frame[(size_t)t * 3 + 2] = 0.0f;
            }
        }
        __syncthreads();
        for (int step = 0; step < 4; step += 2) {
            // z into the row neighbour, then into the column neighbour
            for (int k = 1 + step; k <= 2 + step; ++k) {
                for (int i = tid; i < count; i += OVERLAY_BLOCK) {
                    const int p = first + i;
                    if (a.keep[p] && !shadowed(k, p)) a.zbuf[a.target[k * np + p]] = a.z[p];
                }
                __syncthreads();
            }
            // frame[nb] = frame[nb] * 0.5 + red / 2: float32 product, float64 sum, stored as float32
            for (int k = 1 + step; k <= 2 + step; ++k) {
                for (int i = tid; i < count; i += OVERLAY_BLOCK) {
                    const int p = first + i;
                    if (a.keep[p]) {
                        const float *f = a.frame + (size_t)a.target[k * np + p] * 3;
                        a.blend[(size_t)p * 3 + 0] = (float)((double)(f[0] * 0.5f) + 0.5);
                        a.blend[(size_t)p * 3 + 1] = (float)((double)(f[1] * 0.5f) + 0.0);
                        a.blend[(size_t)p * 3 + 2] = (float)((double)(f[2] * 0.5f) + 0.0);
                    }
                }
                __syncthreads();
                for (int i = tid; i < count; i += OVERLAY_BLOCK) {
                    const int p = first + i;
                    if (a.keep[p] && !shadowed(k, p)) {
                        float *f = a.frame + (size_t)a.target[k * np + p] * 3;
                        f[0] = a.blend[(size_t)p * 3 + 0]; f[1] = a.blend[(size_t)p * 3 + 1]; f[2] = a.blend[(size_t)p * 3 + 2];
                    }
                }
                __syncthreads();
            }
        }
    }
    // finalise the pixels the lines touched (obj/core.py:640): flip rows, ** 0.8, * 255, truncate
    for (int i = tid; i < a.n_touched; i += OVERLAY_BLOCK) {
        const int t = a.touched[i];
        const int py = t / fc.width, px = t - py * fc.width;
        uint8_t *o = a.out + ((size_t)(fc.height - 1 - py) * fc.width + px) * 3;
        const float *f = a.frame + (size_t)t * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) o[j] = gamma_u8(f[j], s_gamma);
    }
}

}  // namespace mr
