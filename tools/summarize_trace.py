"""rocprofv3 kernel trace of `python bench.py ...` -> per-kernel table split by phase.

bench.py launches every kernel once per frame: 2 counted frames (fragment counters), W warm-up
frames, K timed frames (several in flight), 40 frames one at a time with all stage marks (the
"solo" pass) and 1 counted check frame.  The timed-region rows are the ones bench.py's
roofline.avg_launch_ms must agree with; the solo rows the ones roofline.solo_launch_ms must.
    python tools/summarize_trace.py <kernel_trace.csv> <warmup W> <steps K>
"""
import collections, csv, sys
path, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
launches = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mr::", "")
    launches[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print(f"{'kernel':28s} {'calls':>6s} {'all avg us':>11s} {'timed avg us':>13s} {'solo avg us':>12s}")
for name, ls in sorted(launches.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    ls.sort()
    d = [x[1] / 1e3 for x in ls]
    if len(d) < 2 + warm + steps + 40:
        print(f"{name:28s} {len(d):6d} {sum(d)/len(d):11.1f}")
        continue
    timed, solo = d[2 + warm:2 + warm + steps], d[2 + warm + steps:2 + warm + steps + 40]
    print(f"{name:28s} {len(d):6d} {sum(d)/len(d):11.1f} {sum(timed)/len(timed):13.1f} {sum(solo)/len(solo):12.1f}")
