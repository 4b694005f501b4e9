#!/bin/bash
# GPU box: rebuild the library with each set of -D flags and run bench.py (default and one frame at a time).
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 $flags \
      -o py-numpy-renderer_amd/libmi355rast.so py-numpy-renderer_amd/csrc/mi355rast.hip || exit 1
  for f in 4 1; do
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 300 --frames-in-flight $f | FL="$flags" python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('==', os.environ['FL'], 'fif', d['config']['frames_in_flight'], d['value'], d['ms_per_step'])" || exit 1
  done
done
