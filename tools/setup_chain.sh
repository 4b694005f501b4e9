#!/bin/bash
# GPU box: time stamps along the shadow-quad set-up chain of k_setup's edge wavefronts (diagnostic build, tools/make_ablate.py,
# MR_ABLATE=60): mean / max microseconds since the light-facing test, at [0] test done, [1] corners + extrusion, [2] clipping,
# [3] projection + box, [4] work items reserved, [5] everything stored.   usage: tools/setup_chain.sh [scene ...]
cd $GRAFT_REPO_ROOT
cp py-numpy-renderer_amd/libmi355rast.so /tmp/lib_orig.so
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DMR_ABLATE=60 -o py-numpy-renderer_amd/libmi355rast.so tools/_ablate/csrc/mi355rast.hip || exit 1
for scene in ${@:-c4_torus200k_1080p}; do
  echo "== $scene"; MR_SETUP_TIMES=1 python3 tools/render_loop.py $scene 6 frame-only 2>&1 | grep "setup chain" | tail -2
done
cp /tmp/lib_orig.so py-numpy-renderer_amd/libmi355rast.so
