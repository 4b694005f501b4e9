#!/bin/bash
# GPU box: rebuild the library with each set of -D flags and print bench.py's key figures (quoted regime and one
# frame at a time) for the configs in $CFGS (default "c4").   CFGS="c4 c5" tools/ab.sh "" "-DMR_TILE_WAVES=6" ...
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 $flags \
      -o py-numpy-renderer_amd/libmi355rast.so py-numpy-renderer_amd/csrc/mi355rast.hip || exit 1
  for c in ${CFGS:-c4}; do
    for rep in 1 ${REPS:-2}; do
    timeout -k 10 300 python3 bench.py --config $c --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | FL="$flags" CFG=$c python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['gpu_ms_per_kernel_solo']
print('==', repr(os.environ['FL']), os.environ['CFG'], 'ms/frame', d['ms_per_frame'], 'solo', d['latency_ms_single'], 'solo kernels us', round(k['setup']*1e3,1), round(k['bin_work']*1e3,1), round(k['tile']*1e3,1), 'frac', d['frame_hbm_frac'], 'host', d['scene_render_ms_host'], d['scene_render_ms_host_with_overlay'])" || exit 1
    done
  done
done
