"""profiles/<round>_<tag>_pmc_fetch_write.txt (the table tools/prof_traffic.sh prints) -> profiles/<round>_traffic.json:
HBM bytes per launch of the frame's kernels, per config, as bench.py's roofline.traffic reads them.
    python tools/make_traffic_json.py r02 c4 c5"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tags = sys.argv[1], sys.argv[2:]
path = os.path.join(ROOT, "profiles", rnd + "_traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
for tag in tags:
    raw = {}
    for line in open(os.path.join(ROOT, "profiles", f"{rnd}_{tag}_pmc_fetch_write.txt")):
        m = re.match(r"(\S+)\s+FETCH_SIZE=\s*([\d.]+) KB WRITE_SIZE=\s*([\d.]+) KB\s+us=\s*([\d.]+)", line)
        if m:
            raw[m.group(1)] = {"fetch": float(m.group(2)), "write": float(m.group(3)), "fetch_pass_us": float(m.group(4))}
    raw = {k.split("<")[0]: v for k, v in raw.items()}           # k_tile<false> -> k_tile
    out[tag] = {k: int((2 * v.get("fetch", 0) + v.get("write", 0)) * 1024) for k, v in raw.items() if k.startswith("k_")}
    out[tag]["_raw_kb"] = {k: {m: round(x, 2) for m, x in v.items()} for k, v in raw.items() if k.startswith("k_")}
out["_note"] = ("HBM bytes per launch, frame-only mode as bench.py renders (tools/render_loop.py, steady state): "
                "(2*FETCH_SIZE + WRITE_SIZE)*1024 from two separate rocprofv3 --pmc passes (tools/prof_traffic.sh). "
                "FETCH_SIZE on gfx950 counts every 128-byte line fetched as 64 bytes, whatever the access shape "
                "(tools/micro/fetch_calib.hip, profiles/r02_fetch_calibration.json: streaming, 12-byte texel gathers, "
                "112/176/192-byte record gathers all read FETCH_SIZE = lines x 64 B), hence the factor 2: the figure is "
                "whole lines moved, not payload bytes")
json.dump(out, open(path, "w"), indent=1)
print({t: {k: v for k, v in out[t].items() if not k.startswith("_")} for t in tags})
