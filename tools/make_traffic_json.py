"""gpurun_out/pmc/traffic/traffic_raw.json (tools/prof_traffic.sh) -> profiles/<round>_traffic.json:
HBM bytes per launch of the three tile kernels, as bench.py's roofline.traffic reads them."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc", "traffic", "traffic_raw.json")))
out = {}
for kernel, key in (("k_shade", "shade"), ("k_tile_raster", "tile_raster"), ("k_tile_quads", "tile_quads")):
    out[key] = int((2 * raw[kernel]["fetch"] + raw[kernel]["write"]) * 1024)
out["_note"] = ("HBM bytes per launch on BASELINE config c4 (1920x1080, 200k tris), frame-only mode as bench.py renders: "
                "(2*FETCH_SIZE + WRITE_SIZE)*1024 from two separate rocprofv3 --pmc passes (tools/prof_traffic.sh); "
                "the factor 2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (checked on k_vertex: "
                "3.2 MB read -> FETCH_SIZE ~1585 KB)")
out["_raw_kb"] = raw
json.dump(out, open(os.path.join(ROOT, "profiles", sys.argv[1] + "_traffic.json"), "w"), indent=1)
print({k: v for k, v in out.items() if not k.startswith("_")})
