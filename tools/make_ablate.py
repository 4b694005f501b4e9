"""Copies csrc/ to tools/_ablate/csrc (git-ignored) with MR_ABLATE switches in k_tile: diagnostic builds for
tools/ablate_tile.sh (1 shading, 2 shadow quads, 3 small pairs, 4 big pairs, 5 the winners' sweep removed).  Never shipped."""
import os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "py-numpy-renderer_amd", "csrc"), os.path.join(ROOT, "tools", "_ablate", "csrc")
shutil.rmtree(os.path.dirname(dst), ignore_errors=True)
shutil.copytree(src, dst)
for name in os.listdir(dst):
    h = os.path.join(dst, name)
    text = open(h).read()
    open(h, "w").write(text.replace('"../../include/mi355rast.h"', '"../../../include/mi355rast.h"'))
p = os.path.join(dst, "kernels_tile.h")
s = open(p).read()
edits = [
    # 2 / 3 / 4: the tile behaves as if it listed no shadow quads / small pairs / big pairs; 5: no winners' sweep
    ("        n_small = min(n_small_raw, ta.cap[0]); n_big = min(n_big_raw, ta.cap[1]); n_quad = min(n_quad_raw, ta.cap[2]);\n",
     "        n_small = min(n_small_raw, ta.cap[0]); n_big = min(n_big_raw, ta.cap[1]); n_quad = min(n_quad_raw, ta.cap[2]);\n"
     "        if (MR_ABLATE == 2) n_quad = 0;\n        if (MR_ABLATE == 3) n_small = 0;\n        if (MR_ABLATE == 4) n_big = 0;\n"),
    ("            round = 0;\n            for (uint32_t i = my_pair; i < n_small; i += per_round, ++round) {\n                int first = 0;",
     "            round = 0;\n            for (uint32_t i = my_pair; i < (MR_ABLATE == 5 ? 0u : n_small); i += per_round, ++round) {\n                int first = 0;"),
    # 6: the winners' sweep answers from the cache only (what would a cache that always answers be worth?)
    ("                    if (!nodepth && walked <= SWEEP_CACHE_K) continue;     // the cache answered for every sample of this lane",
     "                    if (MR_ABLATE == 6 || (!nodepth && walked <= SWEEP_CACHE_K)) continue;"),
    # 1: no shading arithmetic (the records are still fetched)
    ("            shade_pixel(lc, t, sf, *mp, px, py, lit, rgb);",
     "            if (MR_ABLATE == 1) rgb[0] = (float)sf.world[0][0] + (float)mp->ns + t.d00 + (float)lc.light_pos[0]; else shade_pixel(lc, t, sf, *mp, px, py, lit, rgb);"),
    # 40 / 41: one / two EXTRA dependent scalar loads at the head of every listed tile's chain (what is a round trip worth?)
    ("        if (tid < TILE_STATS) s_cnt[tid] = 0;\n        s_gamma[tid] = sh.gamma_lut[tid];\n", "        if (MR_ABLATE == 40 || MR_ABLATE == 41) {\n            uint32_t x = ta.bin_count[(uint32_t)(tile * 7 + 1) % (uint32_t)(3 * n_tiles)];\n            if (MR_ABLATE == 41) x = ta.bin_count[(x + (uint32_t)tile * 13u) % (uint32_t)(3 * n_tiles)];\n            if (x == 0xdeadbeefu) return;\n        }\n        if (tid < TILE_STATS) s_cnt[tid] = 0;\n        s_gamma[tid] = sh.gamma_lut[tid];\n"),
    # MR_TILE_PAD_KB: LDS ballast, to hold fewer workgroups on a CU (how does the launch time follow occupancy?)
    ("    __shared__ float s_gamma[GAMMA_LUT_SIZE];\n", "    __shared__ float s_gamma[GAMMA_LUT_SIZE];\n#ifdef MR_TILE_PAD_KB\n    __shared__ uint32_t s_pad[MR_TILE_PAD_KB * 256];\n    if (kernargs<TileKernArgs>().fc.width < 0) { s_pad[threadIdx.x] = 1u; __syncthreads(); if (s_pad[threadIdx.x ^ 1] == 77u) return; }\n#endif\n"),
]
for a, b in edits:
    assert s.count(a) == 1, a
    s = s.replace(a, b)
s = s.replace("namespace mr {\n", "#ifndef MR_ABLATE\n#define MR_ABLATE 0\n#endif\nnamespace mr {\n", 1)
open(p, "w").write(s)
# k_setup: 5 no attribute gather / store, 6 no survivor walk, 7 no tile lists, 8 no TriRec store either
p = os.path.join(dst, "kernels_geometry.h")
s = open(p).read()
edits = [
    # 6 no survivor walk, 7 no tile lists, 8 no TriRec store, 9 no quad set-up, 10 no face workgroups, 11 no edge workgroups, 17 the sort alone
    ("    if (count_here) {\n        const int bw = bx1 - bx0;", "    if (count_here && MR_ABLATE != 6) {\n        const int bw = bx1 - bx0;"),
    ("    bin_triangles(fc, bins, (r & 1) != 0, (uint32_t)f, pb, clip);", "    if (MR_ABLATE != 7) bin_triangles(fc, bins, (r & 1) != 0, (uint32_t)f, pb, clip);"),
    ("    sa.tris[f] = t;        // (shading", "    if (MR_ABLATE != 8) sa.tris[f] = t;        // (shading"),
    ("    if (b < face_blocks) tri_setup_block<PRE_XFORM>(b);", "    if (b < face_blocks) { if (MR_ABLATE != 10 && MR_ABLATE != 17) tri_setup_block<PRE_XFORM>(b); }"),
    ("    else edge_block(b - face_blocks);", "    else if (MR_ABLATE != 11 && MR_ABLATE != 17) edge_block(b - face_blocks);"),
    # 60: time stamps along the chain of a wavefront that has silhouette edges (lane 0; sums, maxima and the count in the
    #     counters' padding words, printed by the host when MR_SETUP_TIMES is set): stage k = DBG_T(k)
    ("    unsigned long long todo0 = __ballot(sil[0]), todo1 = __ballot(sil[1]);\n", "    g_dbg_t0 = __builtin_amdgcn_s_memrealtime();\n    unsigned long long todo0 = __ballot(sil[0]), todo1 = __ballot(sil[1]);\n    if (todo0 | todo1) { DBG_T(0); }\n"),
    ("    // ---- clipping, one plane at a time\n", "    DBG_T(1);\n    // ---- clipping, one plane at a time\n"),
    ("    const bool alive = n >= 3;                           // obj/triangular.py:322-323\n", "    DBG_T(2);\n    const bool alive = n >= 3;                           // obj/triangular.py:322-323\n"),
    ("    double xs[2] = { lo_x, hi_x }, ys[2] = { lo_y, hi_y };\n", "    DBG_T(3);\n    double xs[2] = { lo_x, hi_x }, ys[2] = { lo_y, hi_y };\n"),
    ("    if (!boxed) return;\n    if (slot >= sa.quad_cap)", "    DBG_T(4);\n    if (!boxed) return;\n    if (slot >= sa.quad_cap)"),
    ("    const uint32_t base = (uint32_t)__shfl((int)base_raw, 0);\n", "    __builtin_amdgcn_s_waitcnt(0); DBG_T(5);\n    const uint32_t base = (uint32_t)__shfl((int)base_raw, 0);\n"),
]
for a, b in edits:
    assert s.count(a) == 1, a
    s = s.replace(a, b)
s = s.replace("namespace mr {\n", "#ifndef MR_ABLATE\n#define MR_ABLATE 0\n#endif\nnamespace mr {\n"
              "static __device__ unsigned long long g_dbg_t0_unused;\n"
              "#define g_dbg_t0 dbg_t0_local\n"
              "#define DBG_T(k) do { if (MR_ABLATE == 60 && (threadIdx.x & 63) == 0) { const SetupKernArgs &kd_ = kernargs<SetupKernArgs>(); "
              "const unsigned int gw_ = (blockIdx.x * blockDim.x + threadIdx.x) / 64u; "
              "unsigned int *slot_ = reinterpret_cast<unsigned int *>(kd_.sa.sil_edges) + (size_t)kd_.sa.quad_cap * 2 - (size_t)(gw_ % 8192u + 1u) * 8u; "
              "slot_[k] = (unsigned int)(__builtin_amdgcn_s_memrealtime() - dbg_t0_local) + 1u; if ((k) == 0) slot_[7] = (unsigned int)dbg_t0_local; } } while (0)\n", 1)
# the stamps need the wave's start time in scope of both functions: a per-thread variable declared at file scope is not
# possible on the device, so edge_block's start time is passed through a thread-local register variable
s = s.replace("__device__ __forceinline__ void quad_setup_group(bool have,", "__device__ __forceinline__ void quad_setup_group(unsigned long long dbg_t0_local, bool have,")
s = s.replace("        quad_setup_group(have, (int)(ls >> 2)", "        quad_setup_group(dbg_t0_local, have, (int)(ls >> 2)")
s = s.replace("    g_dbg_t0 = __builtin_amdgcn_s_memrealtime();\n", "    const unsigned long long dbg_t0_local = __builtin_amdgcn_s_memrealtime();\n")
s = s.replace("__device__ __forceinline__ void edge_block(uint32_t block)\n{\n", "__device__ __forceinline__ void edge_block(uint32_t block)\n{\n    const unsigned long long dbg_t_block = __builtin_amdgcn_s_memrealtime(); (void)dbg_t_block;\n")
open(p, "w").write(s)
# host: print the stamps
p = os.path.join(dst, "mi355rast.hip")
s = open(p).read()
a = "    mr::Sticky &st = *fs->h_sticky;\n"
assert s.count(a) == 1
s = s.replace(a, a + "    if (getenv(\"MR_SETUP_TIMES\") && fs->quad_cap >= 65536u) { std::vector<uint32_t> t(8192 * 8); (void)hipMemcpy(t.data(), fs->d_sil.as<uint32_t>() + (size_t)fs->quad_cap * 2 - t.size(), t.size() * 4, hipMemcpyDeviceToHost); double sum[6] = {}; uint32_t mx[6] = {}, n = 0, first = ~0u, last = 0; for (size_t w = 0; w < 8192; ++w) { const uint32_t *r = &t[w * 8]; if (!r[0] || !r[5]) continue; ++n; first = std::min(first, r[7]); for (int k = 0; k < 6; ++k) { sum[k] += r[k] - 1; mx[k] = std::max(mx[k], r[k] - 1); } last = std::max(last, r[7] + r[5]); } if (n) { fprintf(stderr, \"setup chain (%u wavefronts with silhouette edges; first test done to last store %.1f us), us since the test was done, mean / max:\", n, (last - first) * 0.01); for (int k = 0; k < 6; ++k) fprintf(stderr, \" [%d] %.1f / %.1f\", k, sum[k] / n * 0.01, mx[k] * 0.01); fprintf(stderr, \"\\n\"); } (void)hipMemset(fs->d_sil.as<uint32_t>() + (size_t)fs->quad_cap * 2 - t.size(), 0, t.size() * 4); }\n")
open(p, "w").write(s)
print("wrote", dst)
