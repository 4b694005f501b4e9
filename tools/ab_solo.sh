#!/bin/bash
# GPU box: rebuild with each set of flags and run tools/solo.py (one frame at a time; k_tile's span from its tile records)
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 $flags \
      -o py-numpy-renderer_amd/libmi355rast.so py-numpy-renderer_amd/csrc/mi355rast.hip || exit 1
  echo "== $flags"; python3 tools/solo.py ${SCENE:-c4_torus200k_1080p} 2>&1 | tail -1
done
