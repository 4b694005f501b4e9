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
    ("        for (uint32_t base = 0; base < n_big; base += WAVE) {",
     "        for (uint32_t base = 0; base < (MR_ABLATE == 4 ? 0u : n_big); base += WAVE) {"),
    ("    if (n_small) {\n        s_key[lp]", "    if (n_small && MR_ABLATE != 3) {\n        s_key[lp]"),
    ("    if (n_quad && (counters || __syncthreads_or(covered))) {",
     "    if (MR_ABLATE != 2 && n_quad && (counters || __syncthreads_or(covered))) {"),
    ("            shade_pixel(fc, t, at, *mp, px, py, lit, rgb);",
     "            if (MR_ABLATE == 1) rgb[0] = (float)at.dp[0] + (float)mp->ns + t.d00; else shade_pixel(fc, t, at, *mp, px, py, lit, rgb);"),
]
for a, b in edits:
    assert s.count(a) == 1, a
    s = s.replace(a, b)
s = s.replace("namespace mr {\n", "#ifndef MR_ABLATE\n#define MR_ABLATE 0\n#endif\nnamespace mr {\n", 1)
open(p, "w").write(s)
print("wrote", dst)
