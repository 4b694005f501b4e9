// div_check.hip -- is the shared-reciprocal quotient of rast_math.h's div3() bit-identical to
// IEEE-754 double division (the compiler's v_div_scale / v_rcp / v_div_fmas / v_div_fixup
// sequence) on MI355X?  Brute force over random operands of the kinds the renderer divides
// (barycentric x 1/w products over their sum), plus operands with random exponents inside the
// range div3() accepts, plus values at the range's borders.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../../py-numpy-renderer_amd/csrc -o div_check div_check.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "rast_math.h"

__device__ uint64_t rng(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ double unit(uint64_t &s) { return (double)(rng(s) >> 11) * (1.0 / 9007199254740992.0); }

__global__ void k_check(unsigned long long *mismatch, unsigned long long *fallback, int rounds)
{
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long bad = 0, fb = 0;
    for (int r = 0; r < rounds; ++r) {
        double a[3], b;
        const int kind = (int)(rng(s) % 4);
        if (kind == 0) {                // barycentrics x depths over their sum
            const float u = (float)unit(s), v = (float)(unit(s) * (1.0 - u)), w = 1.0f - u - v;
            const double d0 = 0.01 + unit(s) * 10, d1 = 0.01 + unit(s) * 10, d2 = 0.01 + unit(s) * 10;
            a[0] = (double)u * d0; a[1] = (double)v * d1; a[2] = (double)w * d2;
            b = mr::chain3((double)u, (double)v, (double)w, d0, d1, d2);
        } else if (kind == 1) {         // arbitrary mantissas, exponents inside the accepted range
            for (int i = 0; i < 3; ++i) {
                const uint64_t m = rng(s) & 0x800fffffffffffffull;
                const uint64_t e = 0x201 + rng(s) % (0x5fe - 0x201);
                a[i] = __longlong_as_double((long long)(m | (e << 52)));
            }
            const uint64_t m = rng(s) & 0x800fffffffffffffull;
            const uint64_t e = 0x201 + rng(s) % (0x5fe - 0x201);
            b = __longlong_as_double((long long)(m | (e << 52)));
        } else if (kind == 2) {         // anything at all: denormals, infinities, NaN, zeros
            for (int i = 0; i < 3; ++i) a[i] = __longlong_as_double((long long)rng(s));
            b = __longlong_as_double((long long)rng(s));
            if (rng(s) % 8 == 0) a[0] = 0.0;
            if (rng(s) % 16 == 0) b = 0.0;
        } else {                        // quotients next to a rounding boundary: (n + 1/2 ulp-ish) patterns
            const double q = 1.0 + unit(s);
            b = 1.0 + unit(s);
            a[0] = q * b; a[1] = __longlong_as_double(__double_as_longlong(a[0]) + 1);
            a[2] = __longlong_as_double(__double_as_longlong(a[0]) - 1);
        }
        double q[3];
        const bool fast = mr::div3(a[0], a[1], a[2], b, q);
        fb += fast ? 0 : 1;
        for (int i = 0; i < 3; ++i) {
            const double ref = a[i] / b;
            const bool same = __double_as_longlong(ref) == __double_as_longlong(q[i]) || (ref != ref && q[i] != q[i]);
            bad += same ? 0 : 1;
        }
    }
    if (bad) atomicAdd(mismatch, bad);
    atomicAdd(fallback, fb);
}

int main()
{
    unsigned long long *d, h[2] = { 0, 0 };
    (void)hipMalloc(&d, 16);
    (void)hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    const int blocks = 4096, threads = 256, rounds = 1000;
    hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, d, d + 1, rounds);
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double n = (double)blocks * threads * rounds;
    printf("div3 vs IEEE division: %.3g operand sets (x3 quotients), mismatches %llu, sets that took the IEEE fallback %llu (%.1f%%)\n",
           n, h[0], h[1], 100.0 * h[1] / n);
    return h[0] ? 1 : 0;
}
