#!/bin/bash
# GPU box: rebuild the library with each set of -D flags and time the stages of one scene.
#   [TOOL=tools/diag_tiles.py] tools/try_variants.sh <scene> "<flags A>" "<flags B>" ...
scene=$1; shift
tool=${TOOL:-tools/time_config.py}
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 $flags \
      -o py-numpy-renderer_amd/libmi355rast.so py-numpy-renderer_amd/csrc/mi355rast.hip || exit 1
  echo "== $flags"
  timeout -k 10 120 python3 $tool $scene || exit 1
done
