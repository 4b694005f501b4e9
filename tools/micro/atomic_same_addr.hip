// Micro-benchmark (MI355X): how long do N wavefronts' returning atomicAdds take when they all hit ONE address (a
// frame counter), one cache line (neighbouring counters), or a line each?  One atomic per wavefront, lane 0, like the
// kernels' wave-aggregated counters.   hipcc -O3 --offload-arch=gfx950 -o atomic_same_addr atomic_same_addr.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k(uint32_t *buf, uint32_t *sink, int mode, int per_wave)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    uint32_t acc = 0;
    if ((threadIdx.x & 63) == 0) {
        for (int i = 0; i < per_wave; ++i) {
            uint32_t *p = mode == 0 ? buf : mode == 1 ? buf + ((wave + i) & 15) : buf + (size_t)((wave * 7 + i) & 0xffff) * 32;
            acc += atomicAdd(p, 1u);          // returning
        }
        if (acc == 0xffffffffu) sink[0] = acc;
    }
}

int main()
{
    uint32_t *buf, *sink;
    CK(hipMalloc(&buf, (size_t)65536 * 128 + 4096));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, (size_t)65536 * 128 + 4096));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char *names[3] = { "one address", "one 64-byte line (16 counters)", "a line per wavefront" };
    for (int waves : { 1024, 4096, 16384 })
        for (int per_wave : { 1, 3 })
            for (int mode = 0; mode < 3; ++mode) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipEventRecord(a));
                    hipLaunchKernelGGL(k, dim3(waves / 4), dim3(256), 0, 0, buf, sink, mode, per_wave);
                    CK(hipEventRecord(b));
                    CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    best = ms < best ? ms : best;
                }
                printf("%6d wavefronts x %d returning atomicAdd, %-32s %8.1f us  (%.1f ns per atomic)\n", waves, per_wave,
                       names[mode], best * 1e3, best * 1e6 / ((double)waves * per_wave));
            }
    return 0;
}
