"""Copies csrc/ to tools/_ablate/csrc (git-ignored) with MR_ABLATE switches in k_tile: diagnostic builds for
tools/ablate_tile.sh (1 shading, 2 shadow quads, 3 small pairs, 4 big pairs removed).  Never shipped."""
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
    # 40 / 41: one / two EXTRA dependent scalar loads at the head of every listed tile's chain (what is a round trip worth?)
    ("    if (tid < TILE_STATS) s_cnt[tid] = 0;\n    s_gamma[tid] = sh.gamma_lut[tid];\n", "    if (MR_ABLATE == 40 || MR_ABLATE == 41) {\n        uint32_t x = ta.bin_count[(uint32_t)(tile * 7 + 1) % (uint32_t)(3 * n_tiles)];\n        if (MR_ABLATE == 41) x = ta.bin_count[(x + (uint32_t)tile * 13u) % (uint32_t)(3 * n_tiles)];\n        if (x == 0xdeadbeefu) return;\n    }\n    if (tid < TILE_STATS) s_cnt[tid] = 0;\n    s_gamma[tid] = sh.gamma_lut[tid];\n"),
    # 30: extra time stamps in the (then unused) counter words of the tile record: [0] lists known, [1] big pairs done,
    #     [2] first sweep done, [3] shading's records loaded
    ("    big_pairs(false);\n", "    const unsigned long long t_lists = __builtin_amdgcn_s_memrealtime();\n    big_pairs(false);\n    const unsigned long long t_big = __builtin_amdgcn_s_memrealtime();\n    unsigned long long t_sweep0 = t_big;\n"),
    ("        if (sfrags) atomicAdd(&s_cnt[0], sfrags);\n        __syncthreads();\n", "        if (sfrags) atomicAdd(&s_cnt[0], sfrags);\n        __syncthreads();\n        t_sweep0 = __builtin_amdgcn_s_memrealtime();\n"),
    ("    if (tid < TILE_STATS) rec[tid] = s_cnt[tid];\n", "    if (tid < TILE_STATS) rec[tid] = s_cnt[tid];\n    if (MR_ABLATE == 30 && tid == 0) { rec[0] = (uint32_t)t_lists; rec[1] = (uint32_t)t_big; rec[2] = (uint32_t)t_sweep0; }\n"),
    ("        for (uint32_t base = 0; base < n_big; base += WAVE) {",
     "        for (uint32_t base = 0; base < (MR_ABLATE == 4 ? 0u : n_big); base += WAVE) {"),
    ("    if (n_small) {\n        // ---- 2. small pairs", "    if (n_small && MR_ABLATE != 3 && MR_ABLATE != 31) {\n        // ---- 2. small pairs"),
    # MR_TILE_PAD_KB: LDS ballast, to hold fewer workgroups on a CU (how does the launch time follow occupancy?)
    ("    __shared__ float s_gamma[GAMMA_LUT_SIZE];\n", "    __shared__ float s_gamma[GAMMA_LUT_SIZE];\n#ifdef MR_TILE_PAD_KB\n    __shared__ uint32_t s_pad[MR_TILE_PAD_KB * 256];\n    if (fc.width < 0) { s_pad[threadIdx.x] = (uint32_t)fc.height; __syncthreads(); if (s_pad[threadIdx.x ^ 1] == 77u) return; }\n#endif\n"),
    ("__global__ void __launch_bounds__(TILE_PX, 5)\nk_tile(", "#ifndef MR_TILE_OCC\n#define MR_TILE_OCC 5\n#endif\n__global__ void __launch_bounds__(TILE_PX, MR_TILE_OCC)\nk_tile("),
    ("    if (n_quad && (counters || __syncthreads_or(covered))) {",
     "    if (MR_ABLATE != 2 && MR_ABLATE != 31 && n_quad && (counters || __syncthreads_or(covered))) {"),
    ("            shade_pixel(fc, t, at, *mp, px, py, lit, rgb);",
     "            if (MR_ABLATE == 1) rgb[0] = (float)at.dp[0] + (float)mp->ns + t.d00; else shade_pixel(fc, t, at, *mp, px, py, lit, rgb);"),
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
    ("    sa.attrs[f] = at;\n", "    if (MR_ABLATE != 5 && MR_ABLATE != 8 && MR_ABLATE != 19 && MR_ABLATE != 21) sa.attrs[f] = at;\n"),
    ("    if (count_here) {\n        const int bw = bx1 - bx0;", "    if (count_here && MR_ABLATE != 6 && MR_ABLATE != 18) {\n        const int bw = bx1 - bx0;"),
    ("    bin_triangles(fc, bins, (r & 1) != 0, (uint32_t)f, pb, clip);", "    if (MR_ABLATE != 7 && MR_ABLATE != 20) bin_triangles(fc, bins, (r & 1) != 0, (uint32_t)f, pb, clip);"),
    ("    sa.tris[f] = t;\n", "    if (MR_ABLATE != 8 && MR_ABLATE != 21) sa.tris[f] = t;\n"),
    ("        quad_setup_group(fc, sa, bins, have, (int)(ls >> 2),", "        if (MR_ABLATE != 9) quad_setup_group(fc, sa, bins, have, (int)(ls >> 2),"),
    ("    if (b < face_blocks) tri_setup_block<PRE_XFORM>(fc, sa, bins, b);", "    if (b < face_blocks) { if (MR_ABLATE != 10) tri_setup_block<PRE_XFORM>(fc, sa, bins, b); }"),
    ("    else edge_block(fc, sa, bins, b - face_blocks);", "    else if (MR_ABLATE < 11 || MR_ABLATE > 21) edge_block(fc, sa, bins, b - face_blocks);"),
    # edges alone (no face workgroups, like 10), the quad set-up cut short: 50 after the extrusion, 51 after the clipping, 52 before the atomics
    ("    // ---- clipping, one plane at a time\n", "    if (MR_ABLATE == 50) { if (have && gl == 0 && v[0] > 1e300) sa.sil_edges[0] = 1; return; }\n    // ---- clipping, one plane at a time\n"),
    ("    const bool alive = n >= 3;                           // obj/triangular.py:322-323\n", "    if (MR_ABLATE == 51) { if (have && gl == 0 && v[0] > 1e300) sa.sil_edges[0] = n; return; }\n    const bool alive = n >= 3;                           // obj/triangular.py:322-323\n"),
    ("    // the quad's record slot and its work items of 64 tiles", "    if (MR_ABLATE == 52) { if (boxed && gl == 0 && bx0 > 100000) sa.sil_edges[0] = bx1; return; }\n    // the quad's record slot and its work items of 64 tiles"),
    # faces alone (no edge workgroups), cut short: 12 after the transform, 13 after the index row, 14 before the survivor walk
    ("    uint8_t *status = sa.status;\n", "    uint8_t *status = sa.status;\n    if (MR_ABLATE == 12) { status[f] = (uint8_t)(A.sx + B.sy + C.sz > 1e300); return 0; }\n"),
    ("    const double *wa = sa.verts + (size_t)ia.x * 4,", "    if (MR_ABLATE == 13) { sa.status[f] = (uint8_t)(ia.x + ib.y + ic.z + ff == -12345); return 0; }\n    const double *wa = sa.verts + (size_t)ia.x * 4,"),
    ("    status[f] = FACE_OK;\n", "    status[f] = FACE_OK;\n    if (MR_ABLATE == 14) { status[f] = (uint8_t)(t.inv_den > 1e30f); return 0; }\n"),
    # 12 and up: no tile-order sort either (15: the whole face path alone, 16: 15 without the workgroup epilogue, 17: the sort alone)
    ("    if (blockIdx.x == 0) { order_tiles_block(", "    if (blockIdx.x == 0) { if (MR_ABLATE < 12 || MR_ABLATE == 17 || (MR_ABLATE > 21 && MR_ABLATE < 50) || MR_ABLATE > 52) order_tiles_block("),
    ("    const unsigned long long bv = __ballot(r & 1), bc = __ballot(r & 2);\n", "    if (MR_ABLATE == 16) return;\n    const unsigned long long bv = __ballot(r & 1), bc = __ballot(r & 2);\n"),
    ("    if (b < face_blocks) { if (MR_ABLATE != 10) tri", "    if (b < face_blocks) { if (MR_ABLATE != 10 && MR_ABLATE != 17 && (MR_ABLATE < 50 || MR_ABLATE > 52)) tri"),
]
for a, b in edits:
    assert s.count(a) == 1, a
    s = s.replace(a, b)
s = s.replace("namespace mr {\n", "#ifndef MR_ABLATE\n#define MR_ABLATE 0\n#endif\nnamespace mr {\n", 1)
open(p, "w").write(s)
print("wrote", dst)
