#!/bin/bash
# GPU box: rocprofv3 kernel trace of Scene.render() with a moving camera, overlay on (tools/host_breakdown.py): per-kernel table
out=$GRAFT_REPO_ROOT/gpurun_out/prof/scene_render
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/host_breakdown.py > $out/run.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + '/*/*_stats.csv')):
    print(f.split('/')[-1])
    for r in csv.DictReader(open(f)):
        print(f"  {r['Name'][:60]:60s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:9.2f} max_us={float(r['MaxNs'])/1e3:9.2f}")
PY
