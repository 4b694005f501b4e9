// dispatch_bench.hip -- how fast does an MI355X start workgroups / wavefronts?
// A kernel that does (almost) nothing is launched over a 1920x1080 frame's worth of threads
// with different workgroup sizes, and with a little VGPR / LDS footprint.
//   hipcc -O3 --offload-arch=gfx950 -o dispatch_bench dispatch_bench.hip && ./dispatch_bench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_BYTES>
__global__ void k_empty(int *out, int never)
{
    __shared__ int s[LDS_BYTES / 4 + 1];
    if (LDS_BYTES) { s[threadIdx.x % (LDS_BYTES / 4 + 1)] = threadIdx.x; __syncthreads(); }
    if (never == 12345 + (int)threadIdx.x) out[blockIdx.x] = LDS_BYTES ? s[0] : 1;
}

template <int LDS_BYTES>
static void run(const char *what, int block, long threads, int *d)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int grid = (int)(threads / block);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_empty<LDS_BYTES>, dim3(grid), dim3(block), 0, 0, d, 0);
    (void)hipEventRecord(a, 0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_empty<LDS_BYTES>, dim3(grid), dim3(block), 0, 0, d, 0);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1000.0 / reps;
    printf("%-18s block %4d grid %6d : %7.2f us/launch  %7.1f WG/us  %8.1f waves/us\n", what, block, grid, us,
           grid / us, grid * (block / 64.0) / us);
}

int main()
{
    int *d;
    (void)hipMalloc(&d, 1 << 24);
    const long threads = 8160L * 256;
    for (int block : { 64, 128, 256, 512, 1024 }) run<0>("no LDS", block, threads, d);
    for (int block : { 256, 1024 }) run<2048>("2 KB LDS+barrier", block, threads, d);
    for (int block : { 256, 1024 }) run<32768>("32 KB LDS+barrier", block, threads, d);
    run<0>("no LDS x4 threads", 256, threads * 4, d);
    return 0;
}
