#!/bin/bash
# GPU box: FETCH_SIZE of tools/micro/fetch_calib's kernels against the bytes they are known to read.
cd $GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 -o tools/micro/fetch_calib tools/micro/fetch_calib.hip || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/pmc/calib
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -- $GRAFT_REPO_ROOT/tools/micro/fetch_calib > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, json, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
N, STRIDE = 1 << 21, 256
truth = {   # kernel -> (payload bytes, 64-byte sectors touched, 128-byte lines touched)
    "k_stream16": (N * STRIDE, N * STRIDE // 64, N * STRIDE // 128),
    "k_gather_dwords<3>": (N * 12, N, N),
    "k_gather_u4<7>": (N * 112, N * 2, N),
    "k_gather_u4<11>": (N * 176, N * 3, N * 2),
    "k_gather_u4<12>": (N * 192, N * 3, N * 2),
    "k_gather_shared<7, 4>": (N // 4 * 112, N // 4 * 2, N // 4),
    "k_gather_shared<11, 64>": (N // 64 * 176, N // 64 * 3, N // 64 * 2),
}
res = {}
for k, v in acc.items():
    kb = v[-1]                                   # second launch
    if k in truth:
        pay, s64, l128 = truth[k]
        res[k] = dict(fetch_size_kb=kb, payload_bytes=pay, sectors64_bytes=s64 * 64, lines128_bytes=l128 * 128,
                      fetch_over_payload=round(kb * 1024 / pay, 3), fetch_over_sectors64=round(kb * 1024 / (s64 * 64), 3),
                      fetch_over_lines128=round(kb * 1024 / (l128 * 128), 3))
        print(f"{k:26s} FETCH_SIZE {kb*1024/1e6:9.1f} MB | payload {pay/1e6:8.1f} MB (x{res[k]['fetch_over_payload']}) "
              f"| 64-B sectors {s64*64/1e6:8.1f} MB (x{res[k]['fetch_over_sectors64']}) | 128-B lines {l128*128/1e6:8.1f} MB (x{res[k]['fetch_over_lines128']})")
json.dump(res, open(sys.argv[1] + "/calib.json", "w"), indent=1)
PY
