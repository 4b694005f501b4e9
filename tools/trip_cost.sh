#!/bin/bash
# GPU box: what one more dependent memory round trip at the head of every listed tile's chain costs k_tile
cd $GRAFT_REPO_ROOT
cp py-numpy-renderer_amd/libmi355rast.so /tmp/lib_orig.so
for k in 0 40 41; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DMR_ABLATE=$k -o py-numpy-renderer_amd/libmi355rast.so tools/_ablate/csrc/mi355rast.hip || exit 1
  echo "== extra dependent scalar loads per listed tile: $k (0 none, 40 one, 41 two)"
  bash tools/prof_kernels.sh trip$k 3 1 | grep "k_tile"
  timeout -k 10 300 python bench.py --config c4 --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ms_per_frame', d['ms_per_frame'], 'single', d['latency_ms_single'])"
done
cp /tmp/lib_orig.so py-numpy-renderer_amd/libmi355rast.so
