#!/bin/bash
cd $GRAFT_REPO_ROOT
cp py-numpy-renderer_amd/libmi355rast.so /tmp/lib_orig.so
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DMR_ABLATE=30 -o py-numpy-renderer_amd/libmi355rast.so tools/_ablate/csrc/mi355rast.hip || exit 1
for s in "$@"; do python3 tools/diag_phases.py $s 2>&1 | grep -v amdgpu.ids; done
cp /tmp/lib_orig.so py-numpy-renderer_amd/libmi355rast.so
