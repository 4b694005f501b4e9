#!/bin/bash
# GPU box: matrix-core counters of the optional stand-alone vertex kernel (MR_VERTEX_PATH=mfma), c4 scene.
out=$GRAFT_REPO_ROOT/gpurun_out/pmc/mfma
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MR_VERTEX_PATH=mfma
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/render_loop.py c4_torus200k_1080p 6 frame-only > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0][-28:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0][-28:]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
print({k: round(sum(v) / len(v), 1) for k, v in d.items()})
PY
