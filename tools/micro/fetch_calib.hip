// fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports for the access shapes of this renderer's kernels.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/fetch_calib tools/micro/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- tools/micro/fetch_calib
//   (tools/prof_calib.sh does both and prints FETCH_SIZE against the known byte counts)
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE (= TCC_EA0_RDREQ x 64 B) for wide coalesced streaming
// reads only: there it reports half the bytes.  The tile kernel does not stream: a pixel gathers a
// 112-byte TriRec, a 176-byte TriAttr and 12-byte texels at data-dependent addresses.  Each kernel
// below reads a known number of bytes in one of those shapes from a table larger than L2 + the
// Infinity Cache, every record exactly once (a multiplicative permutation of the record index) and
// records spaced so that no two share a 128-byte line; the truth is then N x record bytes of
// payload in N x ceil(record / 64) 64-byte sectors (or x ceil(record / 128) 128-byte lines).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr unsigned LOG_N = 21;                 // 2 M records
constexpr unsigned N = 1u << LOG_N;
constexpr unsigned STRIDE = 256;               // bytes between records: 512 MB table
constexpr unsigned MULT = 2654435761u | 1u;    // odd: i -> i * MULT mod 2^LOG_N is a bijection

__device__ __forceinline__ unsigned perm(unsigned i) { return (i * MULT) & (N - 1); }

// coalesced streaming read, 16 bytes per lane (the calibrated case)
__global__ void k_stream16(const uint4 *__restrict__ t, unsigned *__restrict__ out, unsigned n16)
{
    unsigned acc = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) {
        const uint4 v = t[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// one record per lane at a permuted position, read with DWORDS 4-byte loads (texels: 3 floats)
template <int DWORDS>
__global__ void k_gather_dwords(const unsigned char *__restrict__ t, unsigned *__restrict__ out)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned *p = reinterpret_cast<const unsigned *>(t + (size_t)perm(i) * STRIDE);
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < DWORDS; ++k) acc ^= p[k];
    if (acc == 0x12345678u) out[0] = acc;
}

// one record per lane, read with U4 16-byte loads (TriRec: 7, TriAttr: 11, quad header + edges: 12)
template <int U4>
__global__ void k_gather_u4(const unsigned char *__restrict__ t, unsigned *__restrict__ out)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 *p = reinterpret_cast<const uint4 *>(t + (size_t)perm(i) * STRIDE);
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < U4; ++k) { const uint4 v = p[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

// the same record for the four lanes of a group (small pairs: SMALL_LANES lanes per triangle) and for a whole
// wavefront (neighbouring pixels with the same winner)
template <int U4, int SHARE>
__global__ void k_gather_shared(const unsigned char *__restrict__ t, unsigned *__restrict__ out)
{
    const unsigned i = (blockIdx.x * blockDim.x + threadIdx.x) / SHARE;
    const uint4 *p = reinterpret_cast<const uint4 *>(t + (size_t)perm(i) * STRIDE);
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < U4; ++k) { const uint4 v = p[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    unsigned char *t = nullptr;
    unsigned *out = nullptr;
    const size_t bytes = (size_t)N * STRIDE;
    CHECK(hipMalloc(&t, bytes));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(t, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const dim3 block(256), grid(N / 256);
    for (int rep = 0; rep < 2; ++rep) {        // the second launch of each is the one to read (first touches page tables)
        hipLaunchKernelGGL(k_stream16, dim3(4096), block, 0, 0, reinterpret_cast<const uint4 *>(t), out, (unsigned)(bytes / 16));
        hipLaunchKernelGGL(k_gather_dwords<3>, grid, block, 0, 0, t, out);
        hipLaunchKernelGGL(k_gather_u4<7>, grid, block, 0, 0, t, out);
        hipLaunchKernelGGL(k_gather_u4<11>, grid, block, 0, 0, t, out);
        hipLaunchKernelGGL(k_gather_u4<12>, grid, block, 0, 0, t, out);
        hipLaunchKernelGGL((k_gather_shared<7, 4>), grid, block, 0, 0, t, out);
        hipLaunchKernelGGL((k_gather_shared<11, 64>), grid, block, 0, 0, t, out);
        CHECK(hipDeviceSynchronize());
    }
    printf("records %u, stride %u B, table %.0f MB\n", N, STRIDE, bytes / 1e6);
    return 0;
}
