// Micro-benchmark: throughput of scattered 64-bit atomicMin / 32-bit atomicAdd / LDS atomics on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// mode 0: random address per lane; mode 1: lanes of a wave hit 64 consecutive u64 (coalesced);
// mode 2: each thread hits 9 consecutive px of a random row segment (small-triangle pattern)
__global__ void k_min64(unsigned long long *buf, uint32_t n_px, int mode, int per_thread)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) {
        uint32_t a;
        if (mode == 0) a = hash(t * 977 + i) % n_px;
        else if (mode == 1) a = (hash((t / 64) * 31 + i) % (n_px / 64)) * 64 + (t & 63);
        else a = (hash(t) % (n_px - 16)) + i;
        atomicMin(&buf[a], (unsigned long long)hash(t + i) << 20);
    }
}
__global__ void k_add32(int *buf, uint32_t n_px, int mode, int per_thread)
{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) {
        uint32_t a;
        if (mode == 0) a = hash(t * 977 + i) % n_px;
        else if (mode == 1) a = (hash((t / 64) * 31 + i) % (n_px / 64)) * 64 + (t & 63);
        else a = (hash(t) % (n_px - 16)) + i;
        atomicAdd(&buf[a], 1);
    }
}
__global__ void k_lds_min64(unsigned long long *out, int per_thread)
{
    __shared__ unsigned long long z[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) z[i] = ~0ull;
    __syncthreads();
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) atomicMin(&z[(hash(t) + i) & 1023], (unsigned long long)hash(t + i) << 20);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = z[blockIdx.x & 1023];
}

int main()
{
    const uint32_t n_px = 1920 * 1080;
    unsigned long long *z; int *s;
    CK(hipMalloc(&z, n_px * 8)); CK(hipMalloc(&s, n_px * 4));
    CK(hipMemset(z, 0xff, n_px * 8)); CK(hipMemset(s, 0, n_px * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 3; ++mode) {
        for (int kind = 0; kind < 2; ++kind) {
            const int threads = mode == 2 ? 100000 : 1 << 20, per = mode == 2 ? 9 : 1;
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(a));
                if (kind == 0) hipLaunchKernelGGL(k_min64, dim3((threads + 255) / 256), dim3(256), 0, 0, z, n_px, mode, per);
                else hipLaunchKernelGGL(k_add32, dim3((threads + 255) / 256), dim3(256), 0, 0, s, n_px, mode, per);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
            }
            printf("mode %d %s: %d atomics in %.1f us -> %.1f G atomics/s\n", mode, kind ? "add32" : "min64",
                   threads * per, best * 1e3, threads * per / (best * 1e-3) / 1e9);
        }
    }
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_lds_min64, dim3(2048), dim3(256), 0, 0, z, 64);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("LDS min64: %d atomics in %.1f us -> %.1f G atomics/s\n", 2048 * 256 * 64, best * 1e3, 2048.0 * 256 * 64 / (best * 1e-3) / 1e9);
    return 0;
}
