#!/bin/bash
# GPU box: A/B of whole source trees on ONE box (boxes differ by 1-2 %): builds the library from each csrc directory given
# (e.g. a copy of the last commit's under tools/_ablate/base/csrc, made with
#   mkdir -p tools/_ablate/base && git archive HEAD py-numpy-renderer_amd/csrc | tar -x -C tools/_ablate/base --strip-components=1
# and the working tree's py-numpy-renderer_amd/csrc), then prints bench.py's key figures for the configs in $CFGS.
cd $GRAFT_REPO_ROOT
cp py-numpy-renderer_amd/libmi355rast.so /tmp/lib_orig.so
for rep in 1 ${REPS:-2}; do
for dir in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 $FLAGS \
      -o py-numpy-renderer_amd/libmi355rast.so $dir/mi355rast.hip || exit 1
  for c in ${CFGS:-c4}; do
    timeout -k 10 300 python3 bench.py --config $c --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | DIR="$dir" CFG=$c python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['gpu_ms_per_kernel_solo']
print('==', os.environ['DIR'], os.environ['CFG'], 'ms/frame', d['ms_per_frame'], 'solo', d['latency_ms_single'], 'solo kernels us', round(k['setup']*1e3,1), round(k['bin_work']*1e3,1), round(k['tile']*1e3,1), 'frac', d['frame_hbm_frac'])" || exit 1
  done
done
done
cp /tmp/lib_orig.so py-numpy-renderer_amd/libmi355rast.so
