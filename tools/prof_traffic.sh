#!/bin/bash
# GPU box: HBM traffic counters per kernel, two separate --pmc passes (FETCH_SIZE uses 3 of the 4 TCC slots).
#   usage: tools/prof_traffic.sh <scene name, e.g. c4_torus200k_1080p> <tag, e.g. c4>
scene=${1:-c4_torus200k_1080p}; tag=${2:-c4}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc/traffic_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $GRAFT_REPO_ROOT/tools/render_loop.py $scene 8 frame-only > $out/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $GRAFT_REPO_ROOT/tools/render_loop.py $scene 8 frame-only > $out/write.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, json, collections
res = collections.defaultdict(dict)
for kind in ("fetch", "write"):
    f = glob.glob(f"{sys.argv[1]}/{kind}/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mr::", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v) // 2:]                      # steady state: the later launches (tile history primed)
        res[k][kind] = sum(v) / len(v)
    f = glob.glob(f"{sys.argv[1]}/{kind}/*/*_kernel_trace.csv")[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mr::", "")].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, v in dur.items():
        v = v[len(v) // 2:]
        res[k][kind + "_pass_us"] = sum(v) / len(v) / 1e3
for k, v in sorted(res.items()):
    print(f"{k:32s} FETCH_SIZE={v.get('fetch', 0):12.1f} KB WRITE_SIZE={v.get('write', 0):12.1f} KB  us={v.get('fetch_pass_us', 0):8.1f}")
json.dump(res, open(sys.argv[1] + "/traffic_raw.json", "w"), indent=1)
PY
