#!/bin/bash
# GPU box: k_tile with LDS ballast that holds fewer workgroups on a CU: how the launch time follows occupancy
cd $GRAFT_REPO_ROOT
cp py-numpy-renderer_amd/libmi355rast.so /tmp/lib_orig.so
for pad in 1 16 24 36; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DMR_ABLATE=0 -DMR_TILE_PAD_KB=$pad -o py-numpy-renderer_amd/libmi355rast.so tools/_ablate/csrc/mi355rast.hip || exit 1
  echo "== k_tile with $pad KB of LDS ballast (17.9 KB of its own: 5 / 4 / 3 / 2 workgroups per CU at 1 / 16 / 24 / 36)"
  bash tools/prof_kernels.sh pad$pad 3 1 | grep "k_tile"
  timeout -k 10 300 python bench.py --config c4 --no-cpu-baseline | cut -c1-180
done
cp /tmp/lib_orig.so py-numpy-renderer_amd/libmi355rast.so
