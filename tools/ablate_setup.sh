#!/bin/bash
# GPU box: time of k_setup with one part removed (diagnostic builds from tools/_ablate; frames are wrong, only the clock counts)
cd $GRAFT_REPO_ROOT
cp py-numpy-renderer_amd/libmi355rast.so /tmp/lib_orig.so
for k in ${ABL:-0 5 6 7 8 9 10 11}; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DMR_ABLATE=$k -o py-numpy-renderer_amd/libmi355rast.so tools/_ablate/csrc/mi355rast.hip || exit 1
  echo "== ablate $k (0 none, 5 no attribute gather/store, 6 no survivor walk, 7 no tile lists, 8 no record stores at all, 9 no quad set-up, 10 no faces, 11 no edges)"
  bash tools/prof_kernels.sh abl$k 3 1 | grep "k_setup\|k_tile\|k_bin"
done
cp /tmp/lib_orig.so py-numpy-renderer_amd/libmi355rast.so
