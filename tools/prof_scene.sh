#!/bin/bash
# GPU box: rocprofv3 kernel-trace of N mr_render calls of a named scene.  usage: tools/prof_scene.sh <scene> [n]
name=${1:-c3_diablo_floor_1080p}; n=${2:-20}
out=$GRAFT_REPO_ROOT/gpurun_out/prof/scene_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/render_loop.py $name $n > $out/run.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:52]:52s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
