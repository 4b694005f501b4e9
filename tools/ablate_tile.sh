#!/bin/bash
# GPU box: VALU instruction count, wave-cycles and launch time of k_tile (or $KERNEL) with one phase removed ($ABLATE: see tools/make_ablate.py) (diagnostic builds made by
# tools/make_ablate.py under tools/_ablate), one frame at a time under the profiler, then bench.py's quoted regime.
cd $GRAFT_REPO_ROOT
cp py-numpy-renderer_amd/libmi355rast.so /tmp/lib_orig.so
for k in ${ABLATE:-0 1 2 3 4 5}; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DMR_ABLATE=$k -o py-numpy-renderer_amd/libmi355rast.so tools/_ablate/csrc/mi355rast.hip || exit 1
  echo "== ablate $k (0 none, 1 shade, 2 quads, 3 small pairs, 4 big pairs, 5 winners sweep)"
  bash tools/prof_pmc.sh abl$k ${KERNEL:-k_tile} SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY
  KERNEL=${KERNEL:-k_tile} python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/pmc/abl*/*/*_kernel_trace.csv"), key=lambda p: __import__("os").path.getmtime(p))[-1]
d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if __import__("os").environ.get("KERNEL", "k_tile") in r["Kernel_Name"]]
print("kernel us:", [round(x / 1e3, 1) for x in d])
PY
  timeout -k 10 300 python3 bench.py --config ${CFG:-c4} --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['gpu_ms_per_kernel_solo']
print('bench ms/frame', d['ms_per_frame'], 'solo', d['latency_ms_single'], 'solo kernels us', round(k['setup']*1e3,1), round(k['bin_work']*1e3,1), round(k['tile']*1e3,1))"
done
cp /tmp/lib_orig.so py-numpy-renderer_amd/libmi355rast.so
rm -rf gpurun_out/pmc
