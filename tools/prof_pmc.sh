#!/bin/bash
# GPU box: arbitrary PMC counters per kernel on the c4 scene.  usage: tools/prof_pmc.sh <tag> <kernel-substring> COUNTER...
tag=$1; filt=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/render_loop.py c4_torus200k_1080p 6 frame-only > $out.log 2>&1
python3 - "$out" "$filt" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][-28:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
