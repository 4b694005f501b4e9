// kernels_overlay.h -- the debug-camera frustum overlay (obj/core.py:638, obj/frustums.py:46-103) on the
// device's own z-buffer and float frame.
//
// The reference draws the frustum's edges segment by segment; every segment is a handful of NumPy fancy
// assignments over the segment's points (test against z, write z and red, then z writes and half blends into
// the four neighbours), so a segment sees what the earlier ones left in the z-buffer.  The host walks the
// lines (float64 DDA, a few thousand points) and hands over, per point, its five targets (centre, row-1,
// col-1, row+1, col+1) and its depth.  ONE workgroup replays the segments in order, three phases each:
//   A  keep[p] = the segment's z test at point p's centre;
//   B  every kept (target set k, point p) bids for its target X: atomicMax(win[X], code(k, p)), atomicOr(any[X], 1 << k);
//   C  the bidder whose code won -- exactly one per touched pixel -- applies the segment to X:
//        z[X]     <- its own z: it is the LAST kept (k, p) in statement order (k ascending, then p), which is the
//                    write NumPy leaves standing;
//        frame[X] <- red if some kept point has X as its centre (k = 0), then one half blend for every k = 1..4
//                    that has a kept point targeting X (a statement reads before it writes, so duplicates
//                    inside one statement all store the same value);
//      finalises the pixel (uint8), and clears win[X] / any[X] for the next segment.
// Round 2 replayed the fourteen statements of a segment one by one, a barrier and a trip to memory each: ~500
// phases, 206 us at 1080p.  What it must be is exact: the z-buffer it leaves is compared bit for bit with the
// reference's.  The state arrays are the frame's own buffers (index = pixel) or, when the frame was assembled from
// several devices, a compact copy of the touched pixels' state (index = slot, pixel_of[slot] the pixel).
#pragma once

#include "kernels_shade.h"

namespace mr {

constexpr int OVERLAY_TARGETS = 5;       // centre, row-1, col-1, row+1, col+1
constexpr int OVERLAY_BLOCK = 1024;

struct OverlayArgs {
    const int32_t *idx;                  // [5][n_points] state index of every target
    const double *z;                     // [n_points]
    const int32_t *seg;                  // [2 * n_segments] first point, number of points
    int32_t n_points, n_segments;
    uint8_t *keep;                       // scratch [n_points] (segments too long for the LDS table)
    uint32_t *win, *any;                 // scratch [state entries], zero between segments and frames (ditto)
    double *st_z;                        // state: z ...
    float *st_f;                         // ... and float colour (3 per entry)
    const int32_t *pixel_of;             // null: a state index is the pixel row * W + col (row = screen y); else slot -> pixel
    uint8_t *out;                        // the uint8 frame (row 0 = top), out_width x out_height
    int32_t out_width, out_height;
    const float *gamma_lut;
};

// one half blend of obj/frustums.py:100-103: float32 product, float64 sum, stored as float32
__device__ __forceinline__ void overlay_blend(float f[3])
{
    f[0] = (float)((double)(f[0] * 0.5f) + 0.5);
    f[1] = (float)((double)(f[1] * 0.5f) + 0.0);
    f[2] = (float)((double)(f[2] * 0.5f) + 0.0);
}

constexpr int OVERLAY_MAX_SEGMENT = 32768;      // points of one segment (a frame is at most 32767 pixels wide or high)
// Segments of up to OVERLAY_LDS_POINTS points bid in LDS: a wavefront's global atomic instruction leaves a CU every
// ~50 ns (MI355X_MICROARCH.md), and one workgroup bidding for the 10 000 targets of a 1 920-point line through
// global memory spent 15-20 us per segment on that alone (measured: 160 us per 1080p frame).  An open-addressing
// table of the segment's touched pixels -- at most three per point: a line's neighbours overlap -- in the workgroup's
// LDS takes the bids instead; longer segments (frames beyond 4K) keep the words in global memory.
constexpr int OVERLAY_LDS_POINTS = 4096, OVERLAY_TABLE = 16384;
static_assert(3 * OVERLAY_LDS_POINTS <= OVERLAY_TABLE * 3 / 4 + OVERLAY_LDS_POINTS, "table load");

__global__ void __launch_bounds__(OVERLAY_BLOCK)
k_overlay(const OverlayArgs a, const double sign)
{
    __shared__ float s_gamma[GAMMA_LUT_SIZE];
    __shared__ uint8_t s_keep[OVERLAY_LDS_POINTS];
    __shared__ uint32_t s_key[OVERLAY_TABLE], s_win[OVERLAY_TABLE], s_any[OVERLAY_TABLE / 4];     // any: a byte per entry
    const int tid = threadIdx.x;
    if (tid < GAMMA_LUT_SIZE) s_gamma[tid] = a.gamma_lut[tid];
    for (int i = tid; i < OVERLAY_TABLE; i += OVERLAY_BLOCK) { s_key[i] = 0; s_win[i] = 0; if (i < OVERLAY_TABLE / 4) s_any[i] = 0; }
    __syncthreads();
    const int np = a.n_points;
    auto hash = [](uint32_t x) { return (x * 2654435761u) >> (32 - 14); };
    static_assert(OVERLAY_TABLE == 1 << 14, "hash width");
    // applies the segment to pixel x as its winning bidder: z, colour, uint8 (obj/core.py:640: flip rows, ** 0.8,
    // * 255, truncate; a later segment may finalise the pixel again)
    auto apply = [&](int x, double zp, uint32_t any) {
        a.st_z[x] = zp;
        float f[3] = { a.st_f[3 * (size_t)x], a.st_f[3 * (size_t)x + 1], a.st_f[3 * (size_t)x + 2] };
        if (any & 1u) { f[0] = 1.0f; f[1] = 0.0f; f[2] = 0.0f; }
#pragma unroll
        for (int kk = 1; kk < OVERLAY_TARGETS; ++kk)
            if (any & (1u << kk)) overlay_blend(f);
        a.st_f[3 * (size_t)x] = f[0]; a.st_f[3 * (size_t)x + 1] = f[1]; a.st_f[3 * (size_t)x + 2] = f[2];
        const int t = a.pixel_of ? a.pixel_of[x] : x;
        const int py = t / a.out_width, px = t - py * a.out_width;
        uint8_t *o = a.out + ((size_t)(a.out_height - 1 - py) * a.out_width + px) * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) o[j] = gamma_u8(f[j], s_gamma);
    };
    // One POINT per thread and step, its five targets side by side: the loads of a step are independent of each other.
    for (int s = 0; s < a.n_segments; ++s) {
        const int first = a.seg[2 * s], count = min(a.seg[2 * s + 1], OVERLAY_MAX_SEGMENT);
        if (count <= OVERLAY_LDS_POINTS) {
            // A: keep = (z_buffer[row, col] - z) * sign >= 0; B: the kept (k, p) enter their targets in the table and bid
            for (int q = tid; q < count; q += OVERLAY_BLOCK) {
                const int p = first + q;
                int X[OVERLAY_TARGETS];
#pragma unroll
                for (int k = 0; k < OVERLAY_TARGETS; ++k) X[k] = a.idx[k * np + p];
                const bool keep = (a.st_z[X[0]] - a.z[p]) * sign >= 0;
                s_keep[q] = keep ? 1 : 0;
                if (keep) {
#pragma unroll
                    for (int k = 0; k < OVERLAY_TARGETS; ++k) {
                        const uint32_t key = (uint32_t)X[k] + 1u;
                        uint32_t h = hash(key);
                        for (;;) {
                            const uint32_t seen = atomicCAS(&s_key[h], 0u, key);
                            if (seen == 0u || seen == key) break;
                            h = (h + 1u) & (OVERLAY_TABLE - 1);
                        }
                        atomicMax(&s_win[h], ((uint32_t)k << 26 | (uint32_t)q) + 1u);
                        atomicOr(&s_any[h >> 2], (1u << k) << (8 * (h & 3u)));
                    }
                }
            }
            __syncthreads();
            // C1: who won which entry (nothing is cleared yet: a probe must not run into a hole)
            constexpr int PER = OVERLAY_LDS_POINTS / OVERLAY_BLOCK;
            uint32_t won[PER];                  // per step of this thread: bit k = won target k; entries in h[]
            uint16_t hh[PER][OVERLAY_TARGETS];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int q = tid + j * OVERLAY_BLOCK;
                won[j] = 0;
                if (q < count && s_keep[q]) {
                    const int p = first + q;
#pragma unroll
                    for (int k = 0; k < OVERLAY_TARGETS; ++k) {
                        const uint32_t key = (uint32_t)a.idx[k * np + p] + 1u;
                        uint32_t h = hash(key);
                        while (s_key[h] != key) h = (h + 1u) & (OVERLAY_TABLE - 1);
                        hh[j][k] = (uint16_t)h;
                        if (s_win[h] == ((uint32_t)k << 26 | (uint32_t)q) + 1u) won[j] |= 1u << k;
                    }
                }
            }
            __syncthreads();
            // C2: the winning bidder of every touched pixel applies the segment to it and leaves its entry empty
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (!won[j]) continue;
                const int q = tid + j * OVERLAY_BLOCK, p = first + q;
                const double zp = a.z[p];
#pragma unroll
                for (int k = 0; k < OVERLAY_TARGETS; ++k) {
                    if (!(won[j] & (1u << k))) continue;
                    const uint32_t h = hh[j][k];
                    const uint32_t any = (s_any[h >> 2] >> (8 * (h & 3u))) & 0xffu;
                    apply((int)(s_key[h] - 1u), zp, any);
                    s_key[h] = 0; s_win[h] = 0;
                    atomicAnd(&s_any[h >> 2], ~(0xffu << (8 * (h & 3u))));
                }
            }
            __syncthreads();
            continue;
        }
        // ---- long segments: the same with the words in global memory (win / any, one per state entry)
        for (int q = tid; q < count; q += OVERLAY_BLOCK) {
            const int p = first + q;
            int X[OVERLAY_TARGETS];
#pragma unroll
            for (int k = 0; k < OVERLAY_TARGETS; ++k) X[k] = a.idx[k * np + p];
            const bool keep = (a.st_z[X[0]] - a.z[p]) * sign >= 0;
            a.keep[p] = keep ? 1 : 0;
            if (keep) {
#pragma unroll
                for (int k = 0; k < OVERLAY_TARGETS; ++k) {
                    atomicMax(&a.win[X[k]], ((uint32_t)k << 26 | (uint32_t)q) + 1u);
                    atomicOr(&a.any[X[k]], 1u << k);
                }
            }
        }
        __syncthreads();
        // (the bids were made by atomics, which live in L2: they are read back past this CU's L1, which may still
        // hold the words of an earlier segment)
        for (int q = tid; q < count; q += OVERLAY_BLOCK) {
            const int p = first + q;
            if (!a.keep[p]) continue;
            int X[OVERLAY_TARGETS];
            uint32_t w[OVERLAY_TARGETS];
#pragma unroll
            for (int k = 0; k < OVERLAY_TARGETS; ++k) X[k] = a.idx[k * np + p];
#pragma unroll
            for (int k = 0; k < OVERLAY_TARGETS; ++k) w[k] = __hip_atomic_load(&a.win[X[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double zp = a.z[p];
#pragma unroll
            for (int k = 0; k < OVERLAY_TARGETS; ++k) {
                if (w[k] != ((uint32_t)k << 26 | (uint32_t)q) + 1u) continue;
                const int x = X[k];
                apply(x, zp, __hip_atomic_load(&a.any[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                __hip_atomic_store(&a.win[x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&a.any[x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
}

// ---- a frame split over several devices (multigpu.py): the lines' z tests read the z-buffer at pixels other devices
// own, so the overlay is replayed AFTER the all-gather, on the touched pixels' state, which every device appends to
// the rows it contributes: OVERLAY_STATE_BYTES per slot of the list of touched pixels, filled in by the device that
// owns the pixel (k_overlay_export), picked out of the owner's part by k_overlay_import.
constexpr int OVERLAY_STATE_BYTES = 24;          // z float64, colour 3 x float32, 4 bytes of padding
struct OverlayState { double z; float f[3]; uint32_t pad; };
static_assert(sizeof(OverlayState) == OVERLAY_STATE_BYTES, "OverlayState layout");

// which device of the split owns screen row py (y up): contiguous bands of output rows (which count from the top), or
// interleaved tile rows (include/mi355rast.h, stripe_count)
__device__ __forceinline__ int overlay_owner(int py, int height, int world, int striped)
{
    return striped ? (py / TILE_H) % world : (height - 1 - py) / (height / world);
}

__global__ void __launch_bounds__(256)
k_overlay_export(const int32_t *__restrict__ touched, int n_slots, const double *__restrict__ zbuf, const float *__restrict__ frame,
                 int width, int height, int world, int striped, int rank, OverlayState *__restrict__ state)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_slots) return;
    const int t = touched[i];
    if (overlay_owner(t / width, height, world, striped) != rank) return;
    OverlayState s;
    s.z = zbuf[t];
    s.f[0] = frame[(size_t)t * 3]; s.f[1] = frame[(size_t)t * 3 + 1]; s.f[2] = frame[(size_t)t * 3 + 2];
    s.pad = 0;
    state[i] = s;
}

__global__ void __launch_bounds__(256)
k_overlay_import(const int32_t *__restrict__ touched, int n_slots, const char *__restrict__ parts, size_t part_stride,
                 size_t state_offset, int width, int height, int world, int striped, double *__restrict__ st_z,
                 float *__restrict__ st_f)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n_slots) return;
    const int owner = overlay_owner(touched[i] / width, height, world, striped);
    const OverlayState s = reinterpret_cast<const OverlayState *>(parts + (size_t)owner * part_stride + state_offset)[i];
    st_z[i] = s.z;
    st_f[3 * i] = s.f[0]; st_f[3 * i + 1] = s.f[1]; st_f[3 * i + 2] = s.f[2];
}

}  // namespace mr
