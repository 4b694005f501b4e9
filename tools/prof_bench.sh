#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of bench.py (the default regime) for one config; keeps the per-kernel
# stats CSV and bench.py's JSON line under gpurun_out/prof/<tag>/.   usage: tools/prof_bench.sh <tag> <config> [bench args]
tag=$1; cfg=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/prof/$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
python3 - "$out" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:9.2f} max_us={float(r['MaxNs'])/1e3:9.2f} total%={r['Percentage']}")
PY
