// Evidence for DESIGN.md section 3 ("MFMA: not used"): does v_mfma_f64_16x16x4_f64 reproduce, bit for bit,
// the ascending fma chain  rn(a0*b0) -> fma(a1,b1,.) -> fma(a2,b2,.) -> fma(a3,b3,.)  that the vertex
// transform must follow (the reference's OpenBLAS order)?  And what would it save?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// one wave: 16 vertices x 16 output columns (MVP 4 | debug MVP 4 | 8 spare), K = 4
// layout of v_mfma_f64_16x16x4_f64: A lane l holds A[row = l % 16][k = l / 16]; B lane l holds B[k = l / 16][col = l % 16];
// D: 4 doubles per lane, D[row = 4 * i + l / 16][col = l % 16]
__global__ void k_mfma(const double *verts, const double *m16 /*4x16*/, double *out /*n x 16*/, int n)
{
    const int lane = threadIdx.x & 63;
    const int base = (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 16;
    if (base >= n) return;
    const int row = lane % 16, k = lane / 16;
    const double a = (base + row < n) ? verts[(size_t)(base + row) * 4 + k] : 0.0;
    const double b = m16[k * 16 + row];
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc = { 0.0, 0.0, 0.0, 0.0 };
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * i + (lane / 16);
        if (base + r < n) out[(size_t)(base + r) * 16 + (lane % 16)] = acc[i];
    }
}
__global__ void k_chain(const double *verts, const double *m16, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v0 = verts[i * 4], v1 = verts[i * 4 + 1], v2 = verts[i * 4 + 2], v3 = verts[i * 4 + 3];
    for (int j = 0; j < 8; ++j)
        out[(size_t)i * 16 + j] = fma(v3, m16[48 + j], fma(v2, m16[32 + j], fma(v1, m16[16 + j], v0 * m16[j])));
}

int main()
{
    const int n = 1 << 20;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> u(-2.0, 2.0);
    std::vector<double> hv((size_t)n * 4), hm(64, 0.0), a((size_t)n * 16), b((size_t)n * 16);
    for (int i = 0; i < n; ++i) { for (int k = 0; k < 3; ++k) hv[i * 4 + k] = (double)(float)u(rng); hv[i * 4 + 3] = 1.0; }
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 8; ++j) hm[k * 16 + j] = u(rng);
    double *dv, *dm, *da, *db;
    CK(hipMalloc(&dv, hv.size() * 8)); CK(hipMalloc(&dm, 64 * 8)); CK(hipMalloc(&da, a.size() * 8)); CK(hipMalloc(&db, b.size() * 8));
    CK(hipMemcpy(dv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dm, hm.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemset(da, 0, a.size() * 8)); CK(hipMemset(db, 0, b.size() * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float t_mfma = 1e9, t_chain = 1e9, ms;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mfma, dim3((n / 16 + 3) / 4), dim3(256), 0, 0, dv, dm, da, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t_mfma) t_mfma = ms;
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_chain, dim3((n + 255) / 256), dim3(256), 0, 0, dv, dm, db, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t_chain) t_chain = ms;
    }
    CK(hipMemcpy(a.data(), da, a.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), db, b.size() * 8, hipMemcpyDeviceToHost));
    size_t diff = 0, total = 0; double worst = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < 8; ++j) {
        const double x = a[(size_t)i * 16 + j], y = b[(size_t)i * 16 + j];
        ++total;
        if (memcmp(&x, &y, 8)) { ++diff; double r = (x - y) / y; if (r < 0) r = -r; if (r > worst) worst = r; }
    }
    printf("v_mfma_f64_16x16x4_f64 vs ascending fma chain: %zu of %zu outputs differ (%.3f %%), worst relative %.2e\n",
           diff, total, 100.0 * diff / total, worst);
    printf("time for %d vertices x 8 columns: mfma %.1f us, VALU chain %.1f us\n", n, t_mfma * 1e3, t_chain * 1e3);
    return 0;
}
