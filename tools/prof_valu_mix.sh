#!/bin/bash
# GPU box: the vector-instruction mix of the frame's kernels on one config (two --pmc passes), as JSON on stdout.
#   usage: tools/prof_valu_mix.sh [scene]     (the JSON is what bench.py's roofline_valu prices with tools/micro/valu_f64_rate)
scene=${1:-c4_torus200k_1080p}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc/mix
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $out/a -- python3 $GRAFT_REPO_ROOT/tools/render_loop.py $scene 6 frame-only > $out/a.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/b -- python3 $GRAFT_REPO_ROOT/tools/render_loop.py $scene 6 frame-only > $out/b.log 2>&1 || exit 1
python3 - "$out" "$scene" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in "ab":
    f = sorted(glob.glob(sys.argv[1] + "/" + sub + "/*/*_counter_collection.csv"))[-1]
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mr::", "")
        if name.startswith("k_setup") or name.startswith("k_bin_work") or name.startswith("k_tile"):
            agg[name.split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# render_loop renders 6 frames; the first is the scene's first (row-major order, cold): average the last four launches
print(json.dumps({"scene": sys.argv[2], "per_launch": {k: {c: round(sum(x[-4:]) / len(x[-4:])) for c, x in v.items()} for k, v in agg.items()}}, indent=1))
PY
