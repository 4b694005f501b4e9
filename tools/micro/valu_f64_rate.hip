// Micro-benchmark: what a vector instruction of the rasteriser's mix costs a SIMD on MI355X.
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/valu_f64_rate tools/micro/valu_f64_rate.hip && tools/micro/valu_f64_rate
// For each instruction class an unrolled block of INDEPENDENT instructions (8 accumulators) is timed with
// s_memtime inside the kernel, with 1 .. 8 wavefronts resident per SIMD (a 256-thread workgroup per CU puts one
// wavefront on each SIMD; LDS ballast makes exactly k workgroups fit a CU).  Reported: cycles of SIMD time per
// wave-instruction = elapsed cycles / instructions per wavefront / wavefronts per SIMD -- at one wavefront the ISSUE cost (a lone
// wavefront cannot issue back to back), at several the THROUGHPUT cost, which is what the roofline_valu entry of
// bench.py prices the tile kernel's SQ_INSTS_VALU with.  The last lines give the dependent-chain latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int UNROLL = 8, PER_TRIP = 64, REPS = 2048;

#define BODY8_(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)
#define BODY8(stmt) BODY8_(stmt) BODY8_(stmt) BODY8_(stmt) BODY8_(stmt) BODY8_(stmt) BODY8_(stmt) BODY8_(stmt) BODY8_(stmt)     /* 64 per trip: the loop's branch is noise */

template <int OP>
__global__ void __launch_bounds__(256) k_rate(double *out_d, uint64_t *out_t, double seed)
{
    extern __shared__ double s_ballast[];          // LDS ballast: exactly k workgroups fit a CU, i.e. k wavefronts a SIMD
    if (seed < 0) s_ballast[threadIdx.x] = seed;
    double d[UNROLL]; float f[UNROLL]; int n[UNROLL];
    for (int i = 0; i < UNROLL; ++i) { d[i] = seed + i + threadIdx.x * 1e-3; f[i] = (float)d[i]; n[i] = (int)threadIdx.x + i; }
    const double c1 = seed * 0.999, c2 = seed * 1e-3;
    const float g1 = (float)c1, g2 = (float)c2;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; ++r) {
        if (OP == 0) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(c1), "v"(c2));
            BODY8(S)
#undef S
        } else if (OP == 1) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c1));
            BODY8(S)
#undef S
        } else if (OP == 2) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(c2));
            BODY8(S)
#undef S
        } else if (OP == 3) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g1), "v"(g2));
            BODY8(S)
#undef S
        } else if (OP == 4) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
            BODY8(S)
#undef S
        } else if (OP == 5) {      // compare + select: the mask logic of coverage tests
#define S(i) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(n[i]) : "v"(d[i]), "v"(c1), "v"(n[(i + 1) & 7]) : "vcc");
            BODY8(S)
#undef S
        } else if (OP == 6) {
#define S(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            BODY8(S)
#undef S
        } else if (OP == 7) {
#define S(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
            BODY8(S)
#undef S
        } else if (OP == 8) {      // dependent chain: latency of v_fma_f64
#pragma unroll
            for (int u = 0; u < 8; ++u)
            asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\t"
                         "v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2"
                         : "+v"(d[0]) : "v"(c1), "v"(c2));
        } else if (OP == 9) {      // dependent chain: latency of v_fma_f32
#pragma unroll
            for (int u = 0; u < 8; ++u)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                         "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2"
                         : "+v"(f[0]) : "v"(g1), "v"(g2));
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    double acc = 0;
    for (int i = 0; i < UNROLL; ++i) acc += d[i] + f[i] + n[i];
    out_d[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) out_t[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

template <int OP>
int run(const char *name, int insts_per_iter, bool chain)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *d_out; uint64_t *d_t;
    const int max_blocks = cus * 8;
    CK(hipMalloc(&d_out, (size_t)max_blocks * 256 * sizeof(double)));
    CK(hipMalloc(&d_t, (size_t)max_blocks * 4 * sizeof(uint64_t)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-26s", name);
    for (int per_simd : { 1, 2, 4, 6, 8 }) {
        const int blocks = cus * per_simd;
        const size_t lds = (size_t)(160 * 1024 / per_simd) - 1024;      // with this much LDS each, a CU holds per_simd workgroups
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), lds, 0, d_out, d_t, 1.000001);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), lds, 0, d_out, d_t, 1.000001);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> t((size_t)blocks * 4);
        CK(hipMemcpy(t.data(), d_t, t.size() * 8, hipMemcpyDeviceToHost));
        std::sort(t.begin(), t.end());
        const double ticks = (double)t[t.size() / 2];                  // median wavefront, s_memtime ticks
        const double n_inst = (double)REPS * insts_per_iter;
        // SIMD time per wave-instruction: in s_memtime ticks (per wavefront / wavefronts per SIMD) and, from the launch's
        // wall clock (HIP events; launch overhead included, ~1 % at these lengths), in nanoseconds
        const double tick_cost = chain ? ticks / n_inst : ticks / n_inst / per_simd;
        const double ns_cost = chain ? ms * 1e6 / n_inst : ms * 1e6 / n_inst / per_simd;
        printf("  %d/SIMD %5.2f t %5.2f ns", per_simd, tick_cost, ns_cost);
    }
    printf("%s\n", chain ? "  (dependent chain: latency seen by one wavefront)" : "");
    CK(hipFree(d_out)); CK(hipFree(d_t));
    return 0;
}

int main()
{
    printf("SIMD time per wave-instruction (t = s_memtime ticks, ns = from the launch's HIP-event time), by wavefronts per SIMD\n");
    if (run<0>("v_fma_f64", PER_TRIP, false)) return 1;
    if (run<1>("v_mul_f64", PER_TRIP, false)) return 1;
    if (run<2>("v_add_f64", PER_TRIP, false)) return 1;
    if (run<3>("v_fma_f32", PER_TRIP, false)) return 1;
    if (run<4>("v_add_u32", PER_TRIP, false)) return 1;
    if (run<5>("v_cmp_gt_f64 + v_cndmask", 2 * PER_TRIP, false)) return 1;
    if (run<6>("v_cvt_f64_f32", PER_TRIP, false)) return 1;
    if (run<7>("v_rcp_f64", PER_TRIP, false)) return 1;
    if (run<8>("v_fma_f64 chain", 64, true)) return 1;
    if (run<9>("v_fma_f32 chain", 64, true)) return 1;
    return 0;
}
