#!/bin/bash
# GPU box: rocprofv3 kernel-trace of bench.py; prints the per-kernel table.
#   usage: tools/prof_kernels.sh <tag> [steps] [frames in flight: 1 = one frame at a time (default), 0 = bench.py's own default]
tag=${1:-x}; steps=${2:-100}; fif=${3:-1}
fifarg="--frames-in-flight $fif"; [ "$fif" = 0 ] && fifarg=""
out=$GRAFT_REPO_ROOT/gpurun_out/prof/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps $steps --warmup 10 --no-cpu-baseline $fifarg > $out/bench.log 2>&1
grep metric $out/bench.log | cut -c1-200
python3 - "$out" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv'))[-1]
tot = 0
for r in csv.DictReader(open(f)):
    if int(r['Calls']) > 50:
        tot += float(r['AverageNs']) / 1e3
    print(f"{r['Name'][:52]:52s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
print("sum of per-frame kernels (us):", round(tot, 1))
PY
